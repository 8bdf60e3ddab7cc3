// K4: every occurrence list sorted ascending (std::sort, benchmark/gapped-matching/include/index_sasearch.hpp:80), all lists of a
// batch at once, for 32-bit positions.
// Internal to search.hip's translation unit; included exactly once from there.
//
// The lists come out of locate grouped (list l = P[off[l], off[l+1])) but in suffix-array order inside.  Sorting 64-bit
// (list, position) keys with one device-wide radix sort moves 16 bytes per element and pass and spends its upper passes on list
// bits that are in order already (C3: 30 + 18 bits = 6 passes of 8 bits).  Here only the position bits are sorted, as 32-bit keys,
// INSIDE every list:
//   lists of up to kSortTile elements  one workgroup sorts the list in LDS (rocPRIM block_radix_sort), one read + one write;
//   longer lists                       LSD radix passes of <= 8 bits over tiles of kSortTile elements that never straddle two
//                                      lists: per tile a digit histogram, per list an exclusive scan of the histograms (by chunks
//                                      of kSortChunk tiles, so that a list of 2000 tiles is not one serial loop), then every tile
//                                      sorts its keys by the digit in LDS and writes them as runs behind its prefix.
// An even number of passes (2 or 4) brings the long lists back to the buffer they started in, where the short ones were sorted in
// place.  rocPRIM's block-level sort / load / store are the plumbing; the tiling, the per-list scans and the scatter are this file.
#pragma once
namespace {

constexpr uint32_t kSortTile = 4096;          // elements per tile: 256 threads x 16 (measured in round 3: a further class of lists up to 16384
                                              // elements sorted by one workgroup of 1024 threads in 64 KiB of LDS made the sort slower, 12.3 vs 11.4 ms)
constexpr uint32_t kSortChunk = 32;           // tiles per chunk of the histogram scan

// ---- short lists: whole list in one workgroup ------------------------------------------------------------------------------------
template <uint32_t kThreads, uint32_t kItems>
__global__ void __launch_bounds__(kThreads) list_sort_small_kernel(uint32_t* __restrict__ P, const uint64_t* __restrict__ off,
                                                                   const uint32_t* __restrict__ lists, uint32_t n_lists, uint32_t bits)
{
    using Load = rocprim::block_load<uint32_t, kThreads, kItems, rocprim::block_load_method::block_load_transpose>;
    using Store = rocprim::block_store<uint32_t, kThreads, kItems, rocprim::block_store_method::block_store_transpose>;
    using Sort = rocprim::block_radix_sort<uint32_t, kThreads, kItems>;
    __shared__ union { typename Load::storage_type load; typename Store::storage_type store; typename Sort::storage_type sort; } s;
    if (blockIdx.x >= n_lists) return;
    const uint32_t l = lists[blockIdx.x];
    const uint64_t begin = off[l];
    const uint32_t len = (uint32_t)(off[l + 1] - begin);
    uint32_t keys[kItems];
    Load().load(P + begin, keys, len, 0xFFFFFFFFu, s.load);       // positions are < 2^32 - 1: the padding sorts last
    __syncthreads();
    Sort().sort(keys, s.sort, 0, bits);
    __syncthreads();
    Store().store(P + begin, keys, len, s.store);
}

// ---- long lists ----------------------------------------------------------------------------------------------------------------------
struct SortList { uint64_t begin; uint32_t len, tile0, chunk0, pad; };      // a long list: elements, first tile, first chunk

// digit histogram of every tile: hist[tile][256]
__global__ void __launch_bounds__(256) list_sort_hist_kernel(const uint32_t* __restrict__ in, const SortList* __restrict__ lists,
                                                             const uint32_t* __restrict__ tile_list, uint32_t n_tiles, uint32_t shift,
                                                             uint32_t mask, uint16_t* __restrict__ hist)
{
    __shared__ uint32_t h[256];
    const uint32_t t = blockIdx.x;
    if (t >= n_tiles) return;
    const SortList L = lists[tile_list[t]];
    const uint32_t first = (t - L.tile0) * kSortTile;
    const uint32_t cnt = L.len - first < kSortTile ? L.len - first : kSortTile;
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t* src = in + L.begin + first;
#pragma unroll
    for (uint32_t i = 0; i < kSortTile / 256; ++i) {
        const uint32_t e = i * 256 + threadIdx.x;
        if (e < cnt) atomicAdd(&h[(src[e] >> shift) & mask], 1u);
    }
    __syncthreads();
    hist[(uint64_t)t * 256 + threadIdx.x] = (uint16_t)h[threadIdx.x];      // <= kSortTile = 2^12
}

// totals of every chunk of kSortChunk tiles: tot[chunk][256]
__global__ void __launch_bounds__(256) list_sort_chunk_kernel(const uint16_t* __restrict__ hist, const SortList* __restrict__ lists,
                                                              const uint32_t* __restrict__ chunk_list, uint32_t n_chunks,
                                                              uint32_t* __restrict__ tot)
{
    const uint32_t c = blockIdx.x;
    if (c >= n_chunks) return;
    const SortList L = lists[chunk_list[c]];
    const uint32_t tiles = (L.len + kSortTile - 1) / kSortTile;
    const uint32_t t0 = (c - L.chunk0) * kSortChunk, t1 = t0 + kSortChunk < tiles ? t0 + kSortChunk : tiles;
    uint32_t sum = 0;
    for (uint32_t t = t0; t < t1; ++t) sum += hist[(uint64_t)(L.tile0 + t) * 256 + threadIdx.x];
    tot[(uint64_t)c * 256 + threadIdx.x] = sum;
}

// per list: tot[chunk][d] becomes the number of keys of the list that go before the chunk's keys with digit d (smaller digits of the
// whole list + digit d of the chunks before it)
__global__ void __launch_bounds__(256) list_sort_scan_kernel(const SortList* __restrict__ lists, uint32_t n_long, uint32_t* __restrict__ tot)
{
    __shared__ uint32_t s[256];
    const uint32_t m = blockIdx.x;
    if (m >= n_long) return;
    const SortList L = lists[m];
    const uint32_t tiles = (L.len + kSortTile - 1) / kSortTile, chunks = (tiles + kSortChunk - 1) / kSortChunk;
    uint32_t run = 0;
    for (uint32_t c = 0; c < chunks; ++c) {
        uint32_t* p = tot + (uint64_t)(L.chunk0 + c) * 256 + threadIdx.x;
        const uint32_t v = *p;
        *p = run;
        run += v;
    }
    // exclusive scan of the digit totals over the 256 threads
    s[threadIdx.x] = run;
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t base = s[threadIdx.x] - run;
    for (uint32_t c = 0; c < chunks; ++c) tot[(uint64_t)(L.chunk0 + c) * 256 + threadIdx.x] += base;
}

// per chunk: pref[tile][d] = where the tile's keys with digit d start inside the list
__global__ void __launch_bounds__(256) list_sort_prefix_kernel(const uint16_t* __restrict__ hist, const SortList* __restrict__ lists,
                                                               const uint32_t* __restrict__ chunk_list, uint32_t n_chunks,
                                                               const uint32_t* __restrict__ tot, uint32_t* __restrict__ pref)
{
    const uint32_t c = blockIdx.x;
    if (c >= n_chunks) return;
    const SortList L = lists[chunk_list[c]];
    const uint32_t tiles = (L.len + kSortTile - 1) / kSortTile;
    const uint32_t t0 = (c - L.chunk0) * kSortChunk, t1 = t0 + kSortChunk < tiles ? t0 + kSortChunk : tiles;
    uint32_t run = tot[(uint64_t)c * 256 + threadIdx.x];
    for (uint32_t t = t0; t < t1; ++t) {
        const uint64_t at = (uint64_t)(L.tile0 + t) * 256 + threadIdx.x;
        pref[at] = run;
        run += hist[at];
    }
}

// every tile: the stable rank of every key by its digit (rocPRIM's warp-match ranking: keys in warp-striped order, i.e. loaded
// coalesced as they lie), keys placed by rank in LDS, then written as runs behind the tile's prefixes
__global__ void __launch_bounds__(256) list_sort_scatter_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                                const SortList* __restrict__ lists, const uint32_t* __restrict__ tile_list,
                                                                uint32_t n_tiles, uint32_t shift, uint32_t dbits,
                                                                const uint32_t* __restrict__ pref)
{
    constexpr uint32_t kItems = kSortTile / 256;
    using Rank = rocprim::block_radix_rank<256, 8, rocprim::block_radix_rank_algorithm::match>;
    static_assert(Rank::digits_per_thread == 1, "thread d holds digit d");
    // the passes rely on the ranks being STABLE in warp-striped key order (warp w, item i, lane l = element w*64*kItems + 64*i + l),
    // which is how rocPRIM's match algorithm numbers keys on a 64-wide wavefront; VLG_CHECK_SORT=1 (and the tests) verify the
    // outcome of every sort with lists_sorted_check_kernel, so a rocPRIM that numbers differently fails loudly, not silently
    static_assert(kSortTile == 256 * kItems && kItems * 64 * 4 == kSortTile, "four wavefronts of 64 lanes (gfx950), kItems keys per lane");
    __shared__ typename Rank::storage_type s_rank;
    __shared__ uint32_t s_keys[kSortTile];
    __shared__ uint32_t first[256];
    const uint32_t t = blockIdx.x;
    if (t >= n_tiles) return;
    const SortList L = lists[tile_list[t]];
    const uint32_t begin = (t - L.tile0) * kSortTile;
    const uint32_t cnt = L.len - begin < kSortTile ? L.len - begin : kSortTile;
    const uint32_t mask = (1u << dbits) - 1u;
    const uint32_t* src = in + L.begin + begin;
    const uint32_t wbase = (threadIdx.x >> 6) * (64 * kItems) + (threadIdx.x & 63);
    uint32_t keys[kItems];
    unsigned int ranks[kItems];
#pragma unroll
    for (uint32_t i = 0; i < kItems; ++i) {
        const uint32_t e = wbase + 64 * i;                                   // warp-striped: the order of the elements themselves
        keys[i] = e < cnt ? src[e] : 0xFFFFFFFFu;                            // the padding has the largest digit and stands last: it stays last
    }
    unsigned int prefix[1], counts[1];
    Rank().rank_keys(keys, ranks, s_rank, [shift, mask](const uint32_t& k) { return (k >> shift) & mask; }, prefix, counts);
    first[threadIdx.x] = prefix[0];                                          // where the tile's keys with digit threadIdx.x start
#pragma unroll
    for (uint32_t i = 0; i < kItems; ++i) s_keys[ranks[i]] = keys[i];
    __syncthreads();
    const uint32_t* pf = pref + (uint64_t)t * 256;
    uint32_t* dst = out + L.begin;
#pragma unroll
    for (uint32_t i = 0; i < kItems; ++i) {
        const uint32_t p = i * 256 + threadIdx.x;                            // neighbouring lanes write neighbouring keys of a run
        if (p < cnt) {
            const uint32_t k = s_keys[p], d = (k >> shift) & mask;
            dst[pf[d] + (p - first[d])] = k;
        }
    }
}

inline uint64_t list_sort_scratch_bytes(uint64_t n_long, uint64_t n_tiles, uint64_t n_chunks)
{
    return n_long * sizeof(SortList) + (n_tiles + n_chunks) * 4 + n_tiles * 256 * 6 + n_chunks * 256 * 4 + 8 * 256;
}

// The host side comes in two halves so that the tables are built and uploaded BEFORE the lists exist (while locate still runs):
// list_sort_prepare sizes and uploads everything (status VLG_E_WORKSPACE: no room in the arena -- nothing was carved or launched,
// the caller takes the device-wide sort), list_sort_enqueue launches the kernels.
// classes of short lists by length: one workgroup of (threads x items) >= length sorts the list; a workgroup's time is that of its
// padded size, so neighbouring classes are a factor 2 apart (VLG_SORT_CLASSES=3: the three classes 256 / 1024 / 4096 of before)
#ifndef VLG_SORT_CLASSES
#define VLG_SORT_CLASSES 5
#endif
constexpr uint32_t kSortClasses = VLG_SORT_CLASSES;
static_assert(kSortClasses == 3 || kSortClasses == 5, "three or five classes of short lists");
constexpr uint32_t kSortClassMax[5] = {256, kSortClasses == 5 ? 512u : 1024u, kSortClasses == 5 ? 1024u : kSortTile, 2048, kSortTile};
struct ListSortPlan {
    bool ready = false;
    uint32_t n_small[kSortClasses] = {};
    uint32_t* d_small[kSortClasses] = {};
    uint32_t n_long = 0, n_tiles = 0, n_chunks = 0;
    SortList* d_longs = nullptr;
    uint32_t *d_tile_list = nullptr, *d_chunk_list = nullptr, *d_tot = nullptr, *d_pref = nullptr;
    uint16_t* d_hist = nullptr;
};

inline vlg_status list_sort_prepare(const svec<uint64_t>& off64, uint32_t nd, Arena& A, hipStream_t st, ListSortPlan& lp)
{
    svec<uint32_t> small[kSortClasses];                            // <= kSortClassMax[c] elements
    svec<SortList> longs;
    svec<uint32_t> tile_list, chunk_list;
    for (uint32_t l = 0; l < nd; ++l) {
        const uint64_t len = off64[l + 1] - off64[l];
        if (len <= 1) continue;
        if (len <= kSortTile) {
            uint32_t c = 0;
            while (len > kSortClassMax[c]) ++c;
            small[c].push_back(l);
        } else {
            if (len > 0xFFFFFFFFull) return fail(VLG_E_INTERNAL, "list sort: list too long");
            const uint32_t tiles = (uint32_t)((len + kSortTile - 1) / kSortTile), chunks = (tiles + kSortChunk - 1) / kSortChunk;
            const uint32_t m = (uint32_t)longs.size();
            longs.push_back(SortList{off64[l], (uint32_t)len, (uint32_t)tile_list.size(), (uint32_t)chunk_list.size(), 0});
            tile_list.insert(tile_list.end(), tiles, m);
            chunk_list.insert(chunk_list.end(), chunks, m);
        }
    }
    {   // room for everything, or nothing is carved
        uint64_t need = 4096;
        for (uint32_t c = 0; c < kSortClasses; ++c) need += align_up(small[c].size() * 4, 256);
        if (!longs.empty()) need += list_sort_scratch_bytes(longs.size(), tile_list.size(), chunk_list.size());
        if (A.failed || A.size - A.used < need) return fail(VLG_E_WORKSPACE, "list sort: no room for its tables");
    }
    for (uint32_t c = 0; c < kSortClasses; ++c) {
        lp.n_small[c] = (uint32_t)small[c].size();
        if (small[c].empty()) continue;
        lp.d_small[c] = A.take<uint32_t>(small[c].size());
        VLG_HIP_TRY(hipMemcpyAsync(lp.d_small[c], small[c].data(), small[c].size() * 4, hipMemcpyHostToDevice, st));
    }
    lp.n_long = (uint32_t)longs.size(); lp.n_tiles = (uint32_t)tile_list.size(); lp.n_chunks = (uint32_t)chunk_list.size();
    if (lp.n_long) {
        lp.d_longs = A.take<SortList>(lp.n_long);
        lp.d_tile_list = A.take<uint32_t>(lp.n_tiles);
        lp.d_chunk_list = A.take<uint32_t>(lp.n_chunks);
        lp.d_hist = A.take<uint16_t>((uint64_t)lp.n_tiles * 256);
        lp.d_tot = A.take<uint32_t>((uint64_t)lp.n_chunks * 256);
        lp.d_pref = A.take<uint32_t>((uint64_t)lp.n_tiles * 256);
        if (A.failed) return fail(VLG_E_INTERNAL, "arena carve failed (list sort)");
        VLG_HIP_TRY(hipMemcpyAsync(lp.d_longs, longs.data(), lp.n_long * sizeof(SortList), hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(lp.d_tile_list, tile_list.data(), lp.n_tiles * 4, hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(lp.d_chunk_list, chunk_list.data(), lp.n_chunks * 4, hipMemcpyHostToDevice, st));
    }
    if (A.failed) return fail(VLG_E_INTERNAL, "arena carve failed (list sort)");
    lp.ready = true;
    return VLG_OK;
}

// P: the lists (in suffix-array order inside), `other`: a second buffer of the same size; the sorted lists end up in P.
inline vlg_status list_sort_enqueue(const ListSortPlan& lp, uint32_t* P, uint32_t* other, const uint64_t* d_off64, unsigned bits, hipStream_t st)
{
#define VLG_SMALL(C, T, I) do { static_assert(T * I == kSortClassMax[C], "class size"); \
        if (lp.n_small[C]) hipLaunchKernelGGL(HIP_KERNEL_NAME(list_sort_small_kernel<T, I>), dim3(lp.n_small[C]), dim3(T), 0, st, P, d_off64, lp.d_small[C], lp.n_small[C], bits); } while (0)
#if VLG_SORT_CLASSES == 5
    VLG_SMALL(0, 64, 4); VLG_SMALL(1, 128, 4); VLG_SMALL(2, 256, 4); VLG_SMALL(3, 256, 8); VLG_SMALL(4, 256, 16);
#else
    VLG_SMALL(0, 64, 4); VLG_SMALL(1, 256, 4); VLG_SMALL(2, 256, 16);
#endif
#undef VLG_SMALL
    VLG_HIP_TRY(hipGetLastError());
    if (!lp.n_long) return VLG_OK;
    const unsigned passes = bits <= 16 ? 2 : 4;                    // even: the lists come back to P
    const unsigned dbits = (bits + passes - 1) / passes;          // <= 8
    uint32_t* src = P;
    uint32_t* dst = other;
    for (unsigned p = 0; p < passes; ++p) {
        const uint32_t shift = p * dbits, mask = (1u << dbits) - 1u;
        hipLaunchKernelGGL(list_sort_hist_kernel, dim3(lp.n_tiles), dim3(256), 0, st, src, lp.d_longs, lp.d_tile_list, lp.n_tiles, shift, mask, lp.d_hist);
        hipLaunchKernelGGL(list_sort_chunk_kernel, dim3(lp.n_chunks), dim3(256), 0, st, lp.d_hist, lp.d_longs, lp.d_chunk_list, lp.n_chunks, lp.d_tot);
        hipLaunchKernelGGL(list_sort_scan_kernel, dim3(lp.n_long), dim3(256), 0, st, lp.d_longs, lp.n_long, lp.d_tot);
        hipLaunchKernelGGL(list_sort_prefix_kernel, dim3(lp.n_chunks), dim3(256), 0, st, lp.d_hist, lp.d_longs, lp.d_chunk_list, lp.n_chunks, lp.d_tot, lp.d_pref);
        hipLaunchKernelGGL(list_sort_scatter_kernel, dim3(lp.n_tiles), dim3(256), 0, st, src, dst, lp.d_longs, lp.d_tile_list, lp.n_tiles, shift, dbits, lp.d_pref);
        VLG_HIP_TRY(hipGetLastError());
        std::swap(src, dst);
    }
    return VLG_OK;
}

}  // namespace
