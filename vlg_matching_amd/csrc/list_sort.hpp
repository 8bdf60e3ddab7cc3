// K4: every occurrence list sorted ascending (std::sort, benchmark/gapped-matching/include/index_sasearch.hpp:80), all lists of a
// batch at once, for 32-bit positions.
// Internal to search.hip's translation unit; included exactly once from there.
//
// The lists come out of locate grouped (list l = P[off[l], off[l+1])) but in suffix-array order inside.  Sorting 64-bit
// (list, position) keys with one device-wide radix sort moves 16 bytes per element and pass and spends its upper passes on list
// bits that are in order already (C3: 30 + 18 bits = 6 passes of 8 bits).  Here only the position bits are sorted, as 32-bit keys,
// INSIDE every list:
//   lists of up to kSortTile elements  one workgroup sorts the list in LDS (a bucket sort by value ranges; rocPRIM block_radix_sort when the
//                                      positions are clustered), one read + one write;
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
// The positions of a pattern's occurrences are spread over the text, so a list of `len` keys dealt into as many equal value ranges
// ("bins", between the list's own minimum and maximum) leaves about one key per bin: an LDS counter per bin gives every key its bin's
// arrival number, a scan of the counters where the bins start, and a key's place is its bin's start + the keys of the same bin that
// are smaller (a loop over one or two keys).  One read and one write of the list and a few dozen instructions per key, where the
// block-wide radix sort ranks every key four times by 8-bit digits (the class VALU-bound at 95 %).  A list whose
// positions are clustered (some bin holds more than kBucketMaxOcc keys) takes the radix sort as before -- the branch is block-uniform.
#ifndef VLG_BUCKET_SORT
#define VLG_BUCKET_SORT 1
#endif
constexpr uint32_t kBucketMaxOcc = 24;
__device__ __forceinline__ uint32_t bin_at(uint32_t j) { return j + (j >> 5); }      // counters padded: thread t scans bins [t*k, t*k + k)

// kRankLoop: how the keys of a bin are put in order -- every key counts the smaller keys of its bin (a loop over one or two keys; what the
// short lists' kernel takes: C4 7.0 -> 5.8 ms for the lists of 2049..4096 keys) or one lane per multi-key bin sorts it in place (fewer
// registers: what the windows' kernel takes)
template <uint32_t kThreads, uint32_t kItems, bool kInlineFallback, bool kRankLoop = false>
struct BucketSort {
    using Load = rocprim::block_load<uint32_t, kThreads, kItems, rocprim::block_load_method::block_load_transpose>;
    using Store = rocprim::block_store<uint32_t, kThreads, kItems, rocprim::block_store_method::block_store_transpose>;
    using Sort = rocprim::block_radix_sort<uint32_t, kThreads, kItems>;
    static constexpr uint32_t kCap = kThreads * kItems, kPad = kCap + kCap / 32 + 1, kWaves = (kThreads + 63) / 64;
    static_assert(kCap < 0x10000u, "two 16-bit counts share one scan");
    struct Buckets { uint32_t cnt[kPad]; uint32_t tmp[kCap]; uint32_t work[kRankLoop ? 1 : kCap / 2 + 2]; };
    union Radix { typename Load::storage_type load; typename Store::storage_type store; typename Sort::storage_type sort; };
    union Both { Buckets b; Radix r; };
    using Storage = typename std::conditional<kInlineFallback, Both, Buckets>::type;
    struct Misc { uint32_t lo, hi, max, wave[kWaves]; };
    static __device__ __forceinline__ Buckets& buckets(Buckets& s) { return s; }
    static __device__ __forceinline__ Buckets& buckets(Both& s) { return s.b; }

    // the block-wide radix sort (clustered positions)
    static __device__ __forceinline__ void radix(const uint32_t* src, uint32_t* dst, uint32_t len, uint32_t bits, Radix& r)
    {
        uint32_t keys[kItems];
        Load().load(src, keys, len, 0xFFFFFFFFu, r.load);               // positions are < 2^32 - 1: the padding sorts last
        __syncthreads();
        Sort().sort(keys, r.sort, 0, bits);
        __syncthreads();
        Store().store(dst, keys, len, r.store);
    }

    // src[0, len) sorted ascending into dst[0, len) (the same array or another one), len <= kCap, by the whole workgroup.
    // false (block-uniform; only without the inline fallback): the keys are clustered, nothing was written
    static __device__ __forceinline__ bool run(const uint32_t* src, uint32_t* dst, uint32_t len, uint32_t bits, Storage& st, Misc& m_)
    {
        const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        Buckets& s = buckets(st);
        if (len == 0) return true;
        if (VLG_BUCKET_SORT) {
            uint32_t key[kItems], ba[kItems];                                 // bin << 8 | arrival number inside the bin (capped)
            if (tid == 0) { m_.lo = 0xFFFFFFFFu; m_.hi = 0; m_.max = 0; }
            for (uint32_t j = tid; j < kPad; j += kThreads) s.cnt[j] = 0;
            uint32_t lo = 0xFFFFFFFFu, hi = 0;
#pragma unroll
            for (uint32_t i = 0; i < kItems; ++i) {
                const uint32_t e = i * kThreads + tid;
                key[i] = e < len ? src[e] : 0u;
                if (e < len) { lo = key[i] < lo ? key[i] : lo; hi = key[i] > hi ? key[i] : hi; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t a = __shfl_xor(lo, o), c = __shfl_xor(hi, o);
                lo = a < lo ? a : lo; hi = c > hi ? c : hi;
            }
            __syncthreads();
            if (lane == 0) { atomicMin(&m_.lo, lo); atomicMax(&m_.hi, hi); }
            __syncthreads();
            lo = m_.lo; hi = m_.hi;
            // bin = (key - lo) * kCap / range, as a multiplication: m = floor(kCap * 2^32 / range), (key - lo) * m <= kCap * 2^32
            const uint64_t m = ((uint64_t)kCap << 32) / ((uint64_t)hi - lo + 1);
#pragma unroll
            for (uint32_t i = 0; i < kItems; ++i) {
                const uint32_t e = i * kThreads + tid;
                uint32_t bn = (uint32_t)(((uint64_t)(key[i] - lo) * m) >> 32);
                bn = bn < kCap ? bn : kCap - 1;
                const uint32_t arr = e < len ? atomicAdd(&s.cnt[bin_at(bn)], 1u) : 0u;
                ba[i] = (bn << 8) | (arr < 255u ? arr : 255u);
            }
            __syncthreads();
            // exclusive scan of the counters (thread t: bins [t * kItems, (t + 1) * kItems)), the fullest bin, and -- in the upper half of
            // the same scan -- the number of bins that hold more than one key: those go on a work list (start << 8 | keys)
            uint32_t sum = 0, mx = 0;
#pragma unroll
            for (uint32_t i = 0; i < kItems; ++i) {
                const uint32_t c = s.cnt[bin_at(tid * kItems + i)];
                sum += c + (c >= 2 ? 0x10000u : 0u);
                mx = c > mx ? c : mx;
            }
            uint32_t incl = sum;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += v; }
            for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_xor(mx, o); mx = v > mx ? v : mx; }
            if (lane == 63) m_.wave[wv] = incl;
            if (lane == 0) atomicMax(&m_.max, mx);
            __syncthreads();
            uint32_t run_ = incl - sum, all_ = 0;
            for (uint32_t w = 0; w < kWaves; ++w) { const uint32_t v = m_.wave[w]; if (w < wv) run_ += v; all_ += v; }
            const bool spread = m_.max <= kBucketMaxOcc;
            uint32_t at_work = run_ >> 16;
            run_ &= 0xFFFFu;
#pragma unroll
            for (uint32_t i = 0; i < kItems; ++i) {
                const uint32_t at = bin_at(tid * kItems + i), c = s.cnt[at];
                s.cnt[at] = run_;
                if (!kRankLoop && c >= 2 && spread) s.work[at_work++] = (run_ << 8) | c;
                run_ += c;
            }
            const uint32_t n_work = all_ >> 16;
            __syncthreads();
            if (spread) {
#pragma unroll
                for (uint32_t i = 0; i < kItems; ++i)
                    if (i * kThreads + tid < len) s.tmp[s.cnt[bin_at(ba[i] >> 8)] + (ba[i] & 255u)] = key[i];
                __syncthreads();
                if constexpr (kRankLoop) {
                    uint32_t fin[kItems];
#pragma unroll
                    for (uint32_t i = 0; i < kItems; ++i) {
                        const uint32_t bn = ba[i] >> 8;
                        const uint32_t s0 = s.cnt[bin_at(bn)], e0 = bn + 1 < kCap ? s.cnt[bin_at(bn + 1)] : len;
                        const uint32_t mine = s0 + (ba[i] & 255u);
                        uint32_t c = 0;
                        if (i * kThreads + tid < len)
                            for (uint32_t j = s0; j < e0; ++j) { const uint32_t v = s.tmp[j]; c += (v < key[i] || (v == key[i] && j < mine)) ? 1u : 0u; }
                        fin[i] = s0 + c;
                    }
                    __syncthreads();
                    uint32_t* outk = s.cnt;                                   // the counters are done with: the sorted keys line up here
#pragma unroll
                    for (uint32_t i = 0; i < kItems; ++i)
                        if (i * kThreads + tid < len) outk[fin[i]] = key[i];
                    __syncthreads();
#pragma unroll
                    for (uint32_t i = 0; i < kItems; ++i) {
                        const uint32_t e = i * kThreads + tid;
                        if (e < len) dst[e] = outk[e];
                    }
                    return true;
                }
#pragma unroll 1
                for (uint32_t w = tid; w < n_work; w += kThreads) {
                    const uint32_t e = s.work[w], s0 = e >> 8, n = e & 255u;
                    if (n == 2) {                                             // three quarters of them
                        const uint32_t a0 = s.tmp[s0], a1 = s.tmp[s0 + 1];
                        if (a0 > a1) { s.tmp[s0] = a1; s.tmp[s0 + 1] = a0; }
                    } else {
                        for (uint32_t j = s0 + 1; j < s0 + n; ++j) {          // insertion sort of a handful of keys
                            const uint32_t v = s.tmp[j];
                            uint32_t k = j;
                            while (k > s0) { const uint32_t u = s.tmp[k - 1]; if (u <= v) break; s.tmp[k] = u; --k; }
                            s.tmp[k] = v;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (uint32_t i = 0; i < kItems; ++i) {
                    const uint32_t e = i * kThreads + tid;
                    if (e < len) dst[e] = s.tmp[e];
                }
                return true;
            }
            if (!kInlineFallback) return false;
        }
        if constexpr (kInlineFallback) radix(src, dst, len, bits, st.r);
        return true;
    }
};

template <uint32_t kThreads, uint32_t kItems>
__global__ void __launch_bounds__(kThreads) list_sort_small_kernel(uint32_t* __restrict__ P, const uint64_t* __restrict__ off,
                                                                   const uint32_t* __restrict__ lists, uint32_t n_lists, uint32_t bits)
{
    using BS = BucketSort<kThreads, kItems, true, true>;
    __shared__ typename BS::Storage s;
    __shared__ typename BS::Misc misc;
    if (blockIdx.x >= n_lists) return;
    const uint32_t l = lists[blockIdx.x];
    const uint64_t begin = off[l];
    BS::run(P + begin, P + begin, (uint32_t)(off[l + 1] - begin), bits, s, misc);
}

// ---- long lists ----------------------------------------------------------------------------------------------------------------------
struct SortList { uint64_t begin; uint32_t len, tile0, chunk0, pad; };      // a long list: elements, first tile, first chunk

// digit histogram of every tile: hist[tile][256]
// (flagged_only, here and in the other pass kernels: a pass over the lists whose SortList::pad is set -- launched with a small grid that
//  strides over the tiles and leaves at once when *any_flagged is zero, which is the common case: no clustered list in the batch)
__global__ void __launch_bounds__(256) list_sort_hist_kernel(const uint32_t* __restrict__ in, const SortList* __restrict__ lists,
                                                             const uint32_t* __restrict__ tile_list, uint32_t n_tiles, uint32_t shift,
                                                             uint32_t mask, uint16_t* __restrict__ hist, bool flagged_only, const uint32_t* __restrict__ any_flagged,
                                                             const uint2* __restrict__ tile_desc /* {first key in the batch's array, keys} */)
{
    __shared__ uint32_t h[256];
    if (flagged_only && *any_flagged == 0) return;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        if (flagged_only && !lists[tile_list[t]].pad) continue;
        const uint2 d = tile_desc[t];                                 // one load in front of the keys, not tile -> list -> keys
        const uint32_t cnt = d.y;
        h[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t* src = in + d.x;
#pragma unroll
        for (uint32_t i = 0; i < kSortTile / 256; ++i) {
            const uint32_t e = i * 256 + threadIdx.x;
            if (e < cnt) atomicAdd(&h[(src[e] >> shift) & mask], 1u);
        }
        __syncthreads();
        hist[(uint64_t)t * 256 + threadIdx.x] = (uint16_t)h[threadIdx.x];      // <= kSortTile = 2^12
        __syncthreads();
    }
}

// totals of every chunk of kSortChunk tiles: tot[chunk][256]
__global__ void __launch_bounds__(256) list_sort_chunk_kernel(const uint16_t* __restrict__ hist, const SortList* __restrict__ lists,
                                                              const uint32_t* __restrict__ chunk_list, uint32_t n_chunks,
                                                              uint32_t* __restrict__ tot, bool flagged_only, const uint32_t* __restrict__ any_flagged)
{
    if (flagged_only && *any_flagged == 0) return;
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const SortList L = lists[chunk_list[c]];
        if (flagged_only && !L.pad) continue;
        const uint32_t tiles = (L.len + kSortTile - 1) / kSortTile;
        const uint32_t t0 = (c - L.chunk0) * kSortChunk, t1 = t0 + kSortChunk < tiles ? t0 + kSortChunk : tiles;
        uint32_t sum = 0;
        for (uint32_t t = t0; t < t1; ++t) sum += hist[(uint64_t)(L.tile0 + t) * 256 + threadIdx.x];
        tot[(uint64_t)c * 256 + threadIdx.x] = sum;
    }
}

// per list: tot[chunk][d] becomes the number of keys of the list that go before the chunk's keys with digit d (smaller digits of the
// whole list + digit d of the chunks before it)
__global__ void __launch_bounds__(256) list_sort_scan_kernel(const SortList* __restrict__ lists, uint32_t n_long, uint32_t* __restrict__ tot, bool flagged_only,
                                                             const uint32_t* __restrict__ any_flagged)
{
    __shared__ uint32_t s[256];
    const uint32_t m = blockIdx.x;
    if (m >= n_long) return;
    if (flagged_only && *any_flagged == 0) return;
    const SortList L = lists[m];
    if (flagged_only && !L.pad) return;
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
                                                               const uint32_t* __restrict__ tot, uint32_t* __restrict__ pref, bool flagged_only,
                                                               const uint32_t* __restrict__ any_flagged)
{
    if (flagged_only && *any_flagged == 0) return;
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const SortList L = lists[chunk_list[c]];
        if (flagged_only && !L.pad) continue;
        const uint32_t tiles = (L.len + kSortTile - 1) / kSortTile;
        const uint32_t t0 = (c - L.chunk0) * kSortChunk, t1 = t0 + kSortChunk < tiles ? t0 + kSortChunk : tiles;
        uint32_t run = tot[(uint64_t)c * 256 + threadIdx.x];
        for (uint32_t t = t0; t < t1; ++t) {
            const uint64_t at = (uint64_t)(L.tile0 + t) * 256 + threadIdx.x;
            pref[at] = run;
            run += hist[at];
        }
    }
}

// every tile: the stable rank of every key by its digit (rocPRIM's warp-match ranking: keys in warp-striped order, i.e. loaded
// coalesced as they lie), keys placed by rank in LDS, then written as runs behind the tile's prefixes
__global__ void __launch_bounds__(256) list_sort_scatter_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                                const SortList* __restrict__ lists, const uint32_t* __restrict__ tile_list,
                                                                uint32_t n_tiles, uint32_t shift, uint32_t dbits,
                                                                const uint32_t* __restrict__ pref, bool flagged_only, const uint32_t* __restrict__ any_flagged,
                                                                const uint2* __restrict__ tile_desc, const uint32_t* __restrict__ tile_base /* where the tile's list starts */)
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
    if (flagged_only && *any_flagged == 0) return;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        if (flagged_only && !lists[tile_list[t]].pad) continue;
        const uint2 d = tile_desc[t];
        const uint32_t cnt = d.y;
        const uint32_t mask = (1u << dbits) - 1u;
        const uint32_t* src = in + d.x;
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
        uint32_t* dst = out + tile_base[t];
#pragma unroll
        for (uint32_t i = 0; i < kItems; ++i) {
            const uint32_t p = i * 256 + threadIdx.x;                            // neighbouring lanes write neighbouring keys of a run
            if (p < cnt) {
                const uint32_t k = s_keys[p], d = (k >> shift) & mask;
                dst[pf[d] + (p - first[d])] = k;
            }
        }
        __syncthreads();
    }
}

// ---- long lists, the short way: two passes + windows ------------------------------------------------------------------------------
// After LSD passes over the TOP 16 bits only, a list is ordered by those bits and the keys that share them (a "group": len / 65536 keys
// of a list whose positions are spread over the text) stand together in any order.  The tiles of the list, their borders moved back to
// the start of the group they fall into, are windows of whole groups: a workgroup sorts its window in LDS (BucketSort: the window's
// keys span a narrow value range, dealt into bins again) and the list is sorted -- 2 ranking passes + 1 cheap pass, 32 instead of 48
// bytes per key.  A border is looked for at most kWinReach keys back; a list with a longer group (clustered positions) is flagged
// (SortList::pad) by the plan kernel, skipped by the window kernels, and sorted by the four full passes, which run for flagged lists only.
// A window whose keys are clustered inside it (BucketSort gives up) is sorted by the block-wide radix sort in a second kernel.
#ifndef VLG_WINDOW_SORT
#define VLG_WINDOW_SORT 1
#endif
#ifndef VLG_WINDOWS_PER_TILE
#define VLG_WINDOWS_PER_TILE 2
#endif
#ifndef VLG_WINDOW_RANK_LOOP
#define VLG_WINDOW_RANK_LOOP 0     /* measured on C3: work list 2.9 ms, rank loop 3.2 ms (96 registers, 4 spilled) */
#endif
#ifndef VLG_WINDOW_THREADS
#define VLG_WINDOW_THREADS 256
#endif
#ifndef VLG_WINDOW_ITEMS
#define VLG_WINDOW_ITEMS 12
#endif
constexpr uint32_t kWinPerTile = VLG_WINDOWS_PER_TILE, kWinSpan = kSortTile / kWinPerTile;       // a window: kWinSpan keys before its borders move
constexpr uint32_t kWinThreads = VLG_WINDOW_THREADS, kWinItems = VLG_WINDOW_ITEMS, kWinCap = kWinThreads * kWinItems, kWinReach = kWinCap - kWinSpan;
static_assert(kWinCap > kWinSpan && kSortTile % kWinPerTile == 0, "a window holds its span and the group it reaches back for");
constexpr uint32_t kWinMaxLen = 1u << 25;       // longer lists: groups of the top 16 bits would fill a window by themselves

// win[u]: where window u of the batch starts inside its list (u = kWinPerTile * tile + j; the list's length for a window behind its end)
__global__ void __launch_bounds__(256) list_sort_window_plan_kernel(const uint32_t* __restrict__ in, SortList* __restrict__ lists,
                                                                    const uint32_t* __restrict__ tile_list, uint32_t n_tiles, uint32_t shift,
                                                                    uint32_t* __restrict__ win, uint32_t* __restrict__ any /* [0]: lists flagged */)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_tiles * kWinPerTile) return;
    const uint32_t m = tile_list[u / kWinPerTile];
    const SortList L = lists[m];
    const uint32_t x = (u - L.tile0 * kWinPerTile) * kWinSpan;
    if (x == 0) { win[u] = 0; if (L.len > kWinMaxLen) { lists[m].pad = 1; any[0] = 1; } return; }
    if (x >= L.len) { win[u] = L.len; return; }
    const uint32_t* src = in + L.begin;
    const uint32_t h = src[x] >> shift;
    const uint32_t lo = x > kWinReach ? x - kWinReach : 0;
    uint32_t a = lo, b = x;                                                  // first key of the group of src[x], inside [lo, x]
    while (a < b) { const uint32_t mid = (a + b) >> 1; if ((src[mid] >> shift) < h) a = mid + 1; else b = mid; }
    if (a == lo && lo > 0 && (src[lo - 1] >> shift) == h) { lists[m].pad = 1; any[0] = 1; }   // the group starts further back than a window reaches
    win[u] = a;
}

// desc[u] = {where window u starts in the batch's array, its keys}: everything the window kernel needs in ONE load (it used to walk
// tile -> list -> borders -> keys, four dependent round trips in front of a few microseconds of work); no keys for a flagged list
__global__ void __launch_bounds__(256) list_sort_window_desc_kernel(const SortList* __restrict__ lists, const uint32_t* __restrict__ tile_list,
                                                                    uint32_t n_tiles, const uint32_t* __restrict__ win, uint2* __restrict__ desc)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_tiles * kWinPerTile) return;
    const SortList L = lists[tile_list[u / kWinPerTile]];
    const uint32_t k = u - L.tile0 * kWinPerTile;
    const uint32_t a = win[u], b = (uint64_t)(k + 1) * kWinSpan >= L.len ? L.len : win[u + 1];     // <= kWinCap keys (plan kernel)
    desc[u] = make_uint2((uint32_t)(L.begin + a), L.pad ? 0u : b - a);
}

// kRadix: the windows the bucket pass gave up on (win_flag), by the block-wide radix sort -- a kernel of its own so that the common
// pass does not carry its registers
// (measured on 2.7e8 keys: 4 workgroups per CU 1.87 ms, 5 -- what the register cap below buys -- 1.62 ms, 512 threads x 6 keys 1.83 ms:
//  the pass waits on its dependent loads and barriers, not on its instructions)
template <bool kRadix>
__global__ void __launch_bounds__(kWinThreads) __attribute__((amdgpu_waves_per_eu(5, 8))) list_sort_window_kernel(const uint32_t* in, uint32_t* out /* may be the same array */,
                                                               const uint2* __restrict__ desc, uint32_t n_windows, uint32_t bits,
                                                               uint8_t* __restrict__ win_flag, uint32_t* __restrict__ any /* [1]: windows flagged */)
{
    using BS = BucketSort<kWinThreads, kWinItems, false, VLG_WINDOW_RANK_LOOP != 0>;
    __shared__ typename std::conditional<kRadix, typename BS::Radix, typename BS::Storage>::type s;
    __shared__ typename BS::Misc misc;
    if (kRadix && any[1] == 0) return;
    for (uint32_t u = blockIdx.x; u < n_windows; u += gridDim.x) {
        if (kRadix && !win_flag[u]) continue;
        const uint2 d = desc[u];
        if constexpr (kRadix) {
            BS::radix(in + d.x, out + d.x, d.y, bits, s);
        } else {
            const bool done = BS::run(in + d.x, out + d.x, d.y, bits, s, misc);
            if (threadIdx.x == 0) { win_flag[u] = done ? 0 : 1; if (!done) any[1] = 1; }
        }
        __syncthreads();
    }
}

inline uint64_t list_sort_scratch_bytes(uint64_t n_long, uint64_t n_tiles, uint64_t n_chunks)
{
    return n_long * sizeof(SortList) + (n_tiles + n_chunks) * 4 + n_tiles * 256 * 6 + n_chunks * 256 * 4 + (n_tiles * kWinPerTile + 1) * 13 + n_tiles * 12 + 14 * 256;
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
    uint32_t *d_tile_list = nullptr, *d_chunk_list = nullptr, *d_tot = nullptr, *d_pref = nullptr, *d_win = nullptr;
    uint8_t* d_win_flag = nullptr;
    uint2* d_win_desc = nullptr;
    uint2* d_tile_desc = nullptr;
    uint32_t* d_tile_base = nullptr;
    uint32_t* d_any = nullptr;       // [0]: some list is flagged, [1]: some window is
    uint16_t* d_hist = nullptr;
};

inline vlg_status list_sort_prepare(const svec<uint64_t>& off64, uint32_t nd, Arena& A, hipStream_t st, ListSortPlan& lp)
{
    svec<uint32_t> small[kSortClasses];                            // <= kSortClassMax[c] elements
    svec<SortList> longs;
    svec<uint32_t> tile_list, chunk_list, tile_base;
    svec<uint2> tile_desc;
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
            if (off64[l] + len > 0xFFFFFFFFull) return fail(VLG_E_INTERNAL, "list sort: more than 2^32 keys");
            for (uint32_t t = 0; t < tiles; ++t) {
                const uint32_t first = t * kSortTile;
                tile_desc.push_back(make_uint2((uint32_t)(off64[l] + first), std::min<uint32_t>((uint32_t)len - first, kSortTile)));
            }
            tile_base.insert(tile_base.end(), tiles, (uint32_t)off64[l]);
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
        lp.d_tile_desc = A.take<uint2>(lp.n_tiles);
        lp.d_tile_base = A.take<uint32_t>(lp.n_tiles);
        lp.d_hist = A.take<uint16_t>((uint64_t)lp.n_tiles * 256);
        lp.d_tot = A.take<uint32_t>((uint64_t)lp.n_chunks * 256);
        lp.d_pref = A.take<uint32_t>((uint64_t)lp.n_tiles * 256);
        lp.d_win = A.take<uint32_t>((uint64_t)lp.n_tiles * kWinPerTile + 1);
        lp.d_win_flag = A.take<uint8_t>((uint64_t)lp.n_tiles * kWinPerTile + 1);
        lp.d_win_desc = A.take<uint2>((uint64_t)lp.n_tiles * kWinPerTile + 1);
        lp.d_any = A.take<uint32_t>(2);
        if (A.failed) return fail(VLG_E_INTERNAL, "arena carve failed (list sort)");
        VLG_HIP_TRY(hipMemcpyAsync(lp.d_longs, longs.data(), lp.n_long * sizeof(SortList), hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(lp.d_tile_list, tile_list.data(), lp.n_tiles * 4, hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(lp.d_chunk_list, chunk_list.data(), lp.n_chunks * 4, hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(lp.d_tile_desc, tile_desc.data(), lp.n_tiles * sizeof(uint2), hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemcpyAsync(lp.d_tile_base, tile_base.data(), lp.n_tiles * 4, hipMemcpyHostToDevice, st));
    }
    if (A.failed) return fail(VLG_E_INTERNAL, "arena carve failed (list sort)");
    lp.ready = true;
    return VLG_OK;
}

// P: the lists (in suffix-array order inside), `other`: a second buffer of the same size; the sorted lists end up in P.
inline vlg_status list_sort_enqueue(const ListSortPlan& lp, uint32_t* P, uint32_t* other, const uint64_t* d_off64, unsigned bits, hipStream_t st,
                                    bool allow_windows = true)
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
    uint32_t* src = P;
    uint32_t* dst = other;
    const uint32_t few = 1024;                                   // blocks of a pass that almost always has nothing to do
    auto pass = [&](uint32_t shift, uint32_t dbits, bool flagged_only) {
        const uint32_t mask = (1u << dbits) - 1u;
        const dim3 gt(flagged_only ? std::min(lp.n_tiles, few) : lp.n_tiles), gc(flagged_only ? std::min(lp.n_chunks, few) : lp.n_chunks);
        hipLaunchKernelGGL(list_sort_hist_kernel, gt, dim3(256), 0, st, src, lp.d_longs, lp.d_tile_list, lp.n_tiles, shift, mask, lp.d_hist, flagged_only, lp.d_any, lp.d_tile_desc);
        hipLaunchKernelGGL(list_sort_chunk_kernel, gc, dim3(256), 0, st, lp.d_hist, lp.d_longs, lp.d_chunk_list, lp.n_chunks, lp.d_tot, flagged_only, lp.d_any);
        hipLaunchKernelGGL(list_sort_scan_kernel, dim3(lp.n_long), dim3(256), 0, st, lp.d_longs, lp.n_long, lp.d_tot, flagged_only, lp.d_any);
        hipLaunchKernelGGL(list_sort_prefix_kernel, gc, dim3(256), 0, st, lp.d_hist, lp.d_longs, lp.d_chunk_list, lp.n_chunks, lp.d_tot, lp.d_pref, flagged_only, lp.d_any);
        hipLaunchKernelGGL(list_sort_scatter_kernel, gt, dim3(256), 0, st, src, dst, lp.d_longs, lp.d_tile_list, lp.n_tiles, shift, dbits, lp.d_pref, flagged_only, lp.d_any, lp.d_tile_desc, lp.d_tile_base);
        std::swap(src, dst);
    };
    static const bool windows = [] { const char* e = getenv("VLG_WINDOW_SORT"); return e ? e[0] != '0' : VLG_WINDOW_SORT != 0; }();
    const bool by_windows = windows && allow_windows && bits >= 17;
    VLG_HIP_TRY(hipMemsetAsync(lp.d_any, 0, 8, st));
    if (by_windows) {
        pass(bits - 16, 8, false);                                   // the top 16 bits, low digit first; the lists are back in P
        pass(bits - 8, 8, false);
        hipLaunchKernelGGL(list_sort_window_plan_kernel, dim3((lp.n_tiles * kWinPerTile + 255) / 256), dim3(256), 0, st, P, lp.d_longs, lp.d_tile_list, lp.n_tiles, bits - 16, lp.d_win, lp.d_any);
        const uint32_t n_windows = lp.n_tiles * kWinPerTile;
        hipLaunchKernelGGL(list_sort_window_desc_kernel, dim3((n_windows + 255) / 256), dim3(256), 0, st, lp.d_longs, lp.d_tile_list, lp.n_tiles, lp.d_win, lp.d_win_desc);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(list_sort_window_kernel<false>), dim3(n_windows), dim3(kWinThreads), 0, st, P, P, lp.d_win_desc, n_windows, bits, lp.d_win_flag, lp.d_any);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(list_sort_window_kernel<true>), dim3(std::min(n_windows, few)), dim3(kWinThreads), 0, st, P, P, lp.d_win_desc, n_windows, bits, lp.d_win_flag, lp.d_any);
        VLG_HIP_TRY(hipGetLastError());
    }
    const unsigned passes = bits <= 16 ? 2 : 4;                    // even: the lists come back to P
    const unsigned dbits = (bits + passes - 1) / passes;          // <= 8
    for (unsigned p = 0; p < passes; ++p) {                        // (by windows: only for the lists the plan kernel flagged)
        pass(p * dbits, dbits, by_windows);
        VLG_HIP_TRY(hipGetLastError());
    }
    return VLG_OK;
}

}  // namespace
