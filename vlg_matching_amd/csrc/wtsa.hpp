// The paper's index on the device: text + wavelet tree over the suffix array, searched lazily.
// Internal to search.hip's translation unit (results, workspace and timers live there); included exactly once, at its end.
//   vlg_index<alphabet_tag, wt_int<>>   include/sdsl/vlg_index.hpp:109-198, construct :375-392
//   wt_int                              include/sdsl/wt_int.hpp:215-255 (level-wise layout), :339-361 (operator[]), :824-939 (expand / ranges)
//   vlg_iterator                        include/sdsl/vlg_index.hpp:209-373 (relax / next / pull_forward)
//   forward_search                      include/sdsl/suffix_array_algorithm.hpp:48-112
//
// HBM layout: level l of the tree (l = 0: most significant bit of the suffix-array values) is ONE bit-vector of n values, cut into
// the same 256-bit super-blocks as K1 {7 x u32 data, u32 ones before the block inside the level}; the levels follow each other.
// A node is an interval [b, b + size) of its level; its children are the intervals [b, b + zeros) and [b + zeros, b + size) of the
// next level (wt_int.hpp:215-255).  Everything the search needs is two root-to-leaf walks with four rank reads per level:
//   count_less(l, len, x)   how many of SA[l, l + len) are smaller than x
//   quantile(l, len, q)     the q-th smallest of SA[l, l + len)
// so that "the first occurrence of sub-pattern i at or after position p" = quantile(count_less(p)): the sorted occurrence list of a
// sub-pattern is read by rank without ever being located or sorted.  The reference walks the same tree node by node with a cache of
// expanded nodes per sub-pattern (wt_range_walker); on the GPU a lane owns a query, and a batch of 10^5 queries keeps 10^5
// independent walks in flight.
#pragma once
#include "device_rank.hpp"

struct vlg_wtsa {
    uint64_t n_text = 0, n_vals = 0;       // symbols; suffix-array entries (n_text + 1)
    uint32_t sym_bytes = 1, levels = 0;
    uint64_t nb = 0;                       // super-blocks per level
    vlg::Block* d_blocks = nullptr;        // [levels][nb]
    void* d_text = nullptr;                // n_text symbols
};

namespace {

struct WtsaView {
    const Block* blocks;
    const void* text;
    uint64_t nb, n_vals, n_text;
    uint32_t levels, sym_bytes;
};

// ---- construction -------------------------------------------------------------------------------------------------------------
// An integer text is sorted as a byte text: every symbol becomes its five base-255 digits + 1, most significant first -- no zero
// byte, codes compare like the numbers, all codes equally long -- so the aligned suffixes of the byte text stand in the order of
// the integer text's suffixes.
__global__ void wtsa_expand_kernel(const uint32_t* __restrict__ syms, uint64_t n, uint8_t* __restrict__ bytes)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = syms[i];
        uint8_t d[5];
#pragma unroll
        for (int j = 4; j >= 0; --j) { d[j] = (uint8_t)(v % 255u + 1u); v /= 255u; }
#pragma unroll
        for (int j = 0; j < 5; ++j) bytes[5 * i + j] = d[j];
    }
}

// keep the suffixes that start on a symbol: flags, then (after a scan) the compaction
__global__ void wtsa_aligned_flags_kernel(const uint32_t* __restrict__ sa, uint64_t n, uint32_t* __restrict__ flag)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) flag[i] = sa[i] % 5u == 0 ? 1u : 0u;
}
__global__ void wtsa_aligned_compact_kernel(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ pos /* exclusive scan of the flags */,
                                            uint64_t n, uint32_t* __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (sa[i] % 5u == 0) out[pos[i]] = sa[i] / 5u;
}

// stable-sort keys of a level: the top `bits` bits of every value (the arrangement of level l is the suffix array stably sorted by
// the top l bits: wt_int.hpp:215-255)
__global__ void wtsa_prefix_keys_kernel(const uint32_t* __restrict__ vals, uint64_t n, uint32_t shift, uint32_t* __restrict__ keys)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) keys[i] = vals[i] >> shift;
}

// data words of one level from its arrangement: one thread per 32-bit word; pops[b] += the word's popcount (the counts come from a scan)
__global__ void wtsa_emit_kernel(const uint32_t* __restrict__ vals, uint64_t n, uint32_t bit, Block* __restrict__ blocks, uint64_t nb,
                                 uint32_t* __restrict__ pops)
{
    const uint64_t total = nb * 7;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t blk = g / 7;
        const uint32_t w = (uint32_t)(g % 7);
        const uint64_t first = blk * kBlockBits + 32ull * w;
        uint32_t word = 0;
        if (first < n) {
            const uint32_t m = n - first < 32 ? (uint32_t)(n - first) : 32u;
            for (uint32_t j = 0; j < m; ++j) word |= ((vals[first + j] >> bit) & 1u) << j;
        }
        blocks[blk].w[w] = word;
        if (word) atomicAdd(&pops[blk], (uint32_t)__popc(word));
    }
}
__global__ void wtsa_counts_kernel(Block* __restrict__ blocks, const uint32_t* __restrict__ before, uint64_t nb)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x) blocks[b].cnt = before[b];
}

// ---- the two walks ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t level_rank(const WtsaView& w, uint32_t base, uint64_t i) { return (uint32_t)node_rank1(w.blocks, base, i); }

// kQuantile false: number of values < xq among SA[l, l + len); true: the xq-th smallest of them (0-based, xq < len)
template <bool kQuantile>
__device__ __forceinline__ uint64_t wtsa_walk(const WtsaView& w, uint64_t l, uint64_t len, uint64_t xq)
{
    if (!kQuantile && (w.levels < 64 ? (xq >> w.levels) != 0 : false)) return len;    // beyond every value
    uint64_t b = 0, sz = w.n_vals, i = l, j = l + len, acc = 0;
    for (uint32_t lvl = 0; lvl < w.levels; ++lvl) {
        const uint32_t base = (uint32_t)(lvl * w.nb);
        const uint32_t rb = level_rank(w, base, b), ri = level_rank(w, base, b + i), rj = level_rank(w, base, b + j),
                       rn = level_rank(w, base, b + sz);
        const uint64_t ones_i = ri - rb, ones_j = rj - rb, ones_n = rn - rb;
        const uint64_t zi = i - ones_i, zj = j - ones_j, zeros_n = sz - ones_n;
        bool right;
        if (kQuantile) {
            const uint64_t z = zj - zi;
            right = xq >= z;
            if (right) { xq -= z; acc |= 1ull << (w.levels - 1 - lvl); }
        } else {
            right = (xq >> (w.levels - 1 - lvl)) & 1;
            if (right) acc += zj - zi;
        }
        if (right) { i = ones_i; j = ones_j; b += zeros_n; sz = ones_n; }
        else { i = zi; j = zj; sz = zeros_n; }
    }
    return acc;
}

__device__ __forceinline__ int64_t wtsa_symbol(const WtsaView& w, uint64_t p)
{
    if (p >= w.n_text) return -1;                                   // the sentinel is smaller than every symbol
    return w.sym_bytes == 1 ? (int64_t)reinterpret_cast<const uint8_t*>(w.text)[p] : (int64_t)reinterpret_cast<const uint32_t*>(w.text)[p];
}

__global__ void __launch_bounds__(256) wtsa_sa_kernel(WtsaView w, const uint64_t* __restrict__ idx, uint64_t* __restrict__ out, uint64_t count)
{
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (uint64_t)gridDim.x * blockDim.x)
        out[t] = idx[t] < w.n_vals ? wtsa_walk<true>(w, idx[t], 1, 0) : ~0ull;
}

// forward_search (suffix_array_algorithm.hpp:48-112): binary search on the suffix array, every probe one access + one comparison
// of the pattern with the text.  One lane per sub-pattern.
__global__ void __launch_bounds__(256) wtsa_ranges_kernel(WtsaView w, const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off,
                                                          uint64_t n_pat, uint64_t* __restrict__ sp, uint64_t* __restrict__ len)
{
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pat; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t m = (off[p + 1] - off[p]) / w.sym_bytes;
        const uint8_t* pat = blob + off[p];
        auto cmp = [&](uint64_t s) -> int {                         // suffix s against the pattern: -1 smaller, 0 has it as a prefix, +1 greater
            for (uint64_t t = 0; t < m; ++t) {
                const int64_t c = wtsa_symbol(w, s + t);
                const int64_t x = w.sym_bytes == 1 ? (int64_t)pat[t] : (int64_t)reinterpret_cast<const uint32_t*>(pat)[t];
                if (c != x) return c < x ? -1 : 1;
            }
            return 0;
        };
        uint64_t lo = 0, hi = w.n_vals;
        while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (cmp(wtsa_walk<true>(w, mid, 1, 0)) < 0) lo = mid + 1; else hi = mid; }
        const uint64_t first = lo;
        hi = w.n_vals;
        while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (cmp(wtsa_walk<true>(w, mid, 1, 0)) <= 0) lo = mid + 1; else hi = mid; }
        sp[p] = first;
        len[p] = lo - first;
    }
}

// ---- the lazy search ----------------------------------------------------------------------------------------------------------------
// Semantics: SURVEY.md Appendix C (what vlg_iterator yields, include/sdsl/vlg_index.hpp:227-291) -- the left-most, lazy,
// non-overlapping tuples.  One lane owns a query and keeps the reference's k monotone pointers, as RANKS into the sorted
// occurrence lists that exist only implicitly in the tree: `count_less` moves a pointer to the first element at or after a
// position, `quantile` reads the element under it.  Where the reference steps a pointer one element at a time (relax: next_right,
// vlg_index.hpp:233-249), the lane leaps: when the element of list i under the pointer lies beyond the window of the element of
// list i-1, every element of list i-1 before  v_i - hi_i  fails the same way, so pointer i-1 goes straight to the first element at
// or after that position.  A pointer that runs off its list ends the query ("stop entirely").  The work of a query is a few
// walks per match or failed alignment, whatever the lengths of its lists -- and it stops after max_matches matches.
struct WQuery { uint32_t k, sub0; uint64_t end_len, out_first, out_tuple; };

// kWave: ONE WAVEFRONT owns a query.  All 64 lanes run the same pointer machine (same walks, same addresses), and whenever the
// pointer of the query's SHORTEST list is set, the lanes look at the next 64 elements of that list at once -- one quantile walk each,
// then two count_less walks per neighbouring list -- and the pointer skips to the first element that has a partner inside the gap
// window on either side.  An element without one is in no tuple at all, so dropping it changes no least tuple and no restart
// position (the argument of the window filter); what it removes is the query's sequential walk over every
// fruitless alignment -- 3 k leaps per element of the shortest list, each two 30-level walks -- which is what a heavy query
// without early matches used to cost (seconds on C3).  kWave false: one lane per query, 64 queries per wavefront.
template <bool kEmit, bool kWave>
__global__ void __launch_bounds__(64) wtsa_search_kernel(WtsaView w, const uint64_t* __restrict__ sp, const uint64_t* __restrict__ len,
                                                         const uint64_t* __restrict__ lo, const uint64_t* __restrict__ hi,
                                                         const WQuery* __restrict__ qs, uint32_t nq, uint32_t kmax, uint64_t max_matches,
                                                         unsigned long long* __restrict__ counts, uint64_t* __restrict__ out_first,
                                                         uint64_t* __restrict__ out_tuples, unsigned long long* __restrict__ checksum)
{
    extern __shared__ uint32_t s_dyn[];                            // [2][kmax][64]: rank and value under every pointer, per lane
    const uint32_t lane = threadIdx.x;
    const uint64_t qi = kWave ? (uint64_t)blockIdx.x : (uint64_t)blockIdx.x * 64 + lane;
    const bool writer = !kWave || lane == 0;                      // kWave: every lane holds the same state, one of them reports it
    uint32_t* s_rank = s_dyn;
    uint32_t* s_val = s_dyn + (size_t)kmax * 64;
    WQuery Q{0, 0, 0, 0, 0};
    if (qi < nq) Q = qs[qi];
    const uint32_t k = Q.k;
    bool fin = qi >= nq || k == 0;
    for (uint32_t i = 0; i < k && !fin; ++i) fin = len[Q.sub0 + i] == 0;     // vlg_index.hpp:315-316: an empty range ends it at once
    uint32_t piv = 0;                                              // kWave: the shortest list of the query
    if (kWave && !fin && k >= 2) {
        uint64_t best = ~0ull;
        for (uint32_t i = 0; i < k; ++i) if (len[Q.sub0 + i] < best) { best = len[Q.sub0 + i]; piv = i; }
    }
    uint64_t emitted = 0;
    unsigned long long sum = 0;
    // pending walk: level `lv`; op 0 = count_less(key) -> rank, then always the quantile of that rank
    uint32_t lv = 0;
    bool need_count = false;                                       // false: the rank at lv is set, read its value
    uint64_t key = 0;
    uint32_t have = 0;                                             // levels (< 32 tracked; beyond: always recomputed) whose pointer holds a value
    if (!fin) s_rank[lane] = 0;
    while (__any(!fin)) {
        if (!fin) {
            const uint32_t s = Q.sub0 + lv;
            const uint64_t li = sp[s], ni = len[s];
            uint64_t r = s_rank[lv * 64 + lane];
            if (need_count) r = wtsa_walk<false>(w, li, ni, key);         // (every key lies beyond the element under the pointer: it only moves forward)
            if (kWave && k >= 2 && lv == piv) {
                // (wave-uniform from here to the ballot: every lane holds the same r)  64 elements of the shortest list at a time
                while (r < ni) {
                    const uint64_t cand = r + lane;
                    bool ok = false;
                    if (cand < ni) {
                        const uint64_t v = wtsa_walk<true>(w, li, ni, cand);
                        ok = true;
                        if (piv > 0) {                                      // an element u of the list before with lo <= v - u <= hi
                            const uint64_t l_ = lo[s], h_ = hi[s];
                            if (v < l_) ok = false;
                            else {
                                const uint64_t a = v > h_ ? v - h_ : 0, b = v - l_;
                                const uint64_t lp = sp[s - 1], np = len[s - 1];
                                ok = wtsa_walk<false>(w, lp, np, b + 1) > wtsa_walk<false>(w, lp, np, a);
                            }
                        }
                        if (ok && piv + 1 < k) {                            // an element x of the list behind with lo' <= x - v <= hi'
                            const uint64_t l_ = lo[s + 1], h_ = hi[s + 1];
                            const uint64_t a = v + l_ < v ? ~0ull : v + l_, b = v + h_ < v ? ~0ull : v + h_;
                            const uint64_t ln = sp[s + 1], nn = len[s + 1];
                            ok = a != ~0ull && (b == ~0ull ? nn : wtsa_walk<false>(w, ln, nn, b + 1)) > wtsa_walk<false>(w, ln, nn, a);
                        }
                    }
                    const unsigned long long m = __ballot(ok);
                    if (m) { r += (uint32_t)__ffsll((long long)m) - 1; break; }
                    r += 64;
                }
            }
            if (r >= ni) fin = true;                                        // the list has run out: nothing more for this query
            else {
                const uint64_t v = wtsa_walk<true>(w, li, ni, r);
                s_rank[lv * 64 + lane] = (uint32_t)r;
                s_val[lv * 64 + lane] = (uint32_t)v;
                if (lv < 32) have |= 1u << lv;
                // ---- decide the next walk: no memory is touched in here except the lane's own pointers ------------------------------
                for (;;) {
                    const uint64_t cur = s_val[lv * 64 + lane];
                    if (lv > 0) {
                        const uint64_t prev = s_val[(lv - 1) * 64 + lane];
                        const uint64_t h = hi[Q.sub0 + lv];
                        if (cur > (prev + h < prev ? ~0ull : prev + h)) {
                            // beyond the window of the element one list up: that pointer leaps to the first element that can reach cur
                            const uint64_t reach = cur - h;                 // (cur > prev + h >= h)
                            key = reach > prev + 1 ? reach : prev + 1;
                            --lv; need_count = true;
                            break;
                        }
                    }
                    if (lv + 1 == k) {                                      // a match: report it, then pull the first pointer behind it
                        const uint64_t first = s_val[lane];
                        if (writer) sum += first;
                        if (kEmit && writer) {
                            out_first[Q.out_first + emitted] = first;
                            if (out_tuples) {
                                uint64_t* tp = out_tuples + Q.out_tuple + emitted * k;
                                for (uint32_t i = 0; i < k; ++i) tp[i] = s_val[i * 64 + lane];
                            }
                        }
                        ++emitted;
                        if (max_matches && emitted >= max_matches) { fin = true; break; }
                        key = cur + Q.end_len;                              // vlg_index.hpp:254-266 (pull_forward)
                        lv = 0; need_count = true;
                        break;
                    }
                    // one list down: its pointer stays if it already stands at or behind the window's start
                    ++lv;
                    const uint64_t l = lo[Q.sub0 + lv];
                    const uint64_t from = cur + l < cur ? ~0ull : cur + l;
                    if (lv < 32 && ((have >> lv) & 1) && (uint64_t)s_val[lv * 64 + lane] >= from) continue;
                    key = from; need_count = true;
                    break;
                }
            }
        }
    }
    if (qi < nq && writer && counts) counts[qi] = emitted;
    if (kEmit && checksum) {
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
        if (lane == 0 && sum) atomicAdd(checksum, sum);
    }
}

// capped searches run once, every query writing into its own cap-sized stretch; this moves the matches together (query-major)
__global__ void wtsa_compact_kernel(const WQuery* __restrict__ from, const uint64_t* __restrict__ to_first, const uint64_t* __restrict__ to_tuple,
                                    const unsigned long long* __restrict__ counts, uint32_t nq, uint64_t cap, const uint64_t* __restrict__ t_first,
                                    const uint64_t* __restrict__ t_tuples, uint64_t* __restrict__ out_first, uint64_t* __restrict__ out_tuples)
{
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < (uint64_t)nq * cap; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t qi = t / cap, m = t - qi * cap;
        if (m >= counts[qi]) continue;
        const WQuery Q = from[qi];
        out_first[to_first[qi] + m] = t_first[Q.out_first + m];
        if (out_tuples) for (uint32_t i = 0; i < Q.k; ++i) out_tuples[to_tuple[qi] + m * Q.k + i] = t_tuples[Q.out_tuple + m * Q.k + i];
    }
}

WtsaView wtsa_view(const vlg_wtsa* x) { return WtsaView{x->d_blocks, x->d_text, x->nb, x->n_vals, x->n_text, x->levels, x->sym_bytes}; }

}  // namespace

extern "C" void vlg_wtsa_destroy(vlg_wtsa* x)
{
    if (!x) return;
    if (x->d_blocks) (void)hipFree(x->d_blocks);
    if (x->d_text) (void)hipFree(x->d_text);
    delete x;
}

extern "C" vlg_status vlg_wtsa_build(const void* h_text, uint64_t n_symbols, uint32_t symbol_bytes, vlg_wtsa** out)
{
    if (!out || (n_symbols && !h_text)) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    if (symbol_bytes != 1 && symbol_bytes != 4) return fail(VLG_E_INVALID, "symbol_bytes must be 1 or 4");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VLG_E_NO_DEVICE, "no HIP device available (the VLG library has no CPU fallback)");
    const uint64_t byte_len = n_symbols * (symbol_bytes == 1 ? 1 : 5);
    if (byte_len >= 0xFFFFFFF0ull) return fail(VLG_E_UNSUPPORTED, "text too long for the 32-bit suffix array of this index");
    release_cached_device_memory();
    vlg_wtsa* x = new vlg_wtsa();
    x->n_text = n_symbols; x->n_vals = n_symbols + 1; x->sym_bytes = symbol_bytes;
    x->levels = std::max(1u, bit_width64(n_symbols));               // values 0..n_symbols
    x->nb = x->n_vals / kBlockBits + 1;
    uint8_t* d_bytes = nullptr;
    uint32_t *d_sa = nullptr, *d_a = nullptr, *d_b = nullptr, *d_ka = nullptr, *d_kb = nullptr, *d_pops = nullptr;
    uint32_t *d_flag = nullptr, *d_pos = nullptr, *d_out = nullptr;        // integer texts: the aligned suffixes (freed below on every path)
    void *d_tmp = nullptr, *d_scan2 = nullptr;
    auto grid = [](uint64_t n) { return dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, 16384))); };
    auto run = [&]() -> vlg_status {
        VLG_HIP_TRY(hipMalloc(&x->d_text, std::max<uint64_t>(n_symbols * symbol_bytes, 16)));
        if (n_symbols) VLG_HIP_TRY(hipMemcpy(x->d_text, h_text, n_symbols * symbol_bytes, hipMemcpyHostToDevice));
        // ---- suffix array of text + sentinel (the sorter of the FM-index builder) -----------------------------------------------------
        VLG_HIP_TRY(hipMalloc((void**)&d_sa, (byte_len + 1) * 4));
        if (symbol_bytes == 1) {
            if (vlg_status st = vlg_suffix_array_device((const uint8_t*)x->d_text, n_symbols, d_sa, nullptr)) return st;
        } else {
            VLG_HIP_TRY(hipMalloc((void**)&d_bytes, std::max<uint64_t>(byte_len, 16)));
            hipLaunchKernelGGL(wtsa_expand_kernel, grid(n_symbols), dim3(256), 0, nullptr, (const uint32_t*)x->d_text, n_symbols, d_bytes);
            VLG_HIP_TRY(hipGetLastError());
            if (vlg_status st = vlg_suffix_array_device(d_bytes, byte_len, d_sa, nullptr)) return st;
            (void)hipFree(d_bytes); d_bytes = nullptr;
            // the suffixes that start on a symbol, in order
            const uint64_t nb5 = byte_len + 1;
            VLG_HIP_TRY(hipMalloc((void**)&d_flag, nb5 * 4));
            VLG_HIP_TRY(hipMalloc((void**)&d_pos, nb5 * 4));
            VLG_HIP_TRY(hipMalloc((void**)&d_out, x->n_vals * 4));
            hipLaunchKernelGGL(wtsa_aligned_flags_kernel, grid(nb5), dim3(256), 0, nullptr, d_sa, nb5, d_flag);
            size_t tb = 0;
            VLG_HIP_TRY(rocprim::exclusive_scan(nullptr, tb, d_flag, d_pos, 0u, nb5, rocprim::plus<uint32_t>(), nullptr));
            VLG_HIP_TRY(hipMalloc(&d_scan2, tb + 16));
            VLG_HIP_TRY(rocprim::exclusive_scan(d_scan2, tb, d_flag, d_pos, 0u, nb5, rocprim::plus<uint32_t>(), nullptr));
            hipLaunchKernelGGL(wtsa_aligned_compact_kernel, grid(nb5), dim3(256), 0, nullptr, d_sa, d_pos, nb5, d_out);
            VLG_HIP_TRY(hipGetLastError());
            VLG_HIP_TRY(hipDeviceSynchronize());
            (void)hipFree(d_scan2); d_scan2 = nullptr;
            (void)hipFree(d_flag); d_flag = nullptr;
            (void)hipFree(d_pos); d_pos = nullptr;
            (void)hipFree(d_sa);
            d_sa = d_out; d_out = nullptr;
        }
        // ---- the tree, one level at a time: emit the bits of the current arrangement, then sort stably by one more bit of prefix ---------
        const uint64_t n = x->n_vals;
        VLG_HIP_TRY(hipMalloc((void**)&x->d_blocks, (uint64_t)x->levels * x->nb * sizeof(Block)));
        VLG_HIP_TRY(hipMalloc((void**)&d_a, n * 4));
        VLG_HIP_TRY(hipMalloc((void**)&d_b, n * 4));
        VLG_HIP_TRY(hipMalloc((void**)&d_ka, n * 4));
        VLG_HIP_TRY(hipMalloc((void**)&d_kb, n * 4));
        VLG_HIP_TRY(hipMalloc((void**)&d_pops, (x->nb + 1) * 4));
        size_t sort_tb = 0, scan_tb = 0;
        VLG_HIP_TRY(rocprim::radix_sort_pairs(nullptr, sort_tb, d_ka, d_kb, d_a, d_b, n, 0, 32, nullptr));
        VLG_HIP_TRY(rocprim::exclusive_scan(nullptr, scan_tb, d_pops, d_pops, 0u, x->nb, rocprim::plus<uint32_t>(), nullptr));
        VLG_HIP_TRY(hipMalloc(&d_tmp, std::max(sort_tb, scan_tb) + 16));
        VLG_HIP_TRY(hipMemcpy(d_a, d_sa, n * 4, hipMemcpyDeviceToDevice));
        uint32_t* cur = d_a;
        uint32_t* other = d_b;
        for (uint32_t lvl = 0; lvl < x->levels; ++lvl) {
            Block* lb = x->d_blocks + (uint64_t)lvl * x->nb;
            VLG_HIP_TRY(hipMemsetAsync(d_pops, 0, (x->nb + 1) * 4, nullptr));
            hipLaunchKernelGGL(wtsa_emit_kernel, grid(x->nb * 7), dim3(256), 0, nullptr, cur, n, x->levels - 1 - lvl, lb, x->nb, d_pops);
            size_t tb = scan_tb;
            VLG_HIP_TRY(rocprim::exclusive_scan(d_tmp, tb, d_pops, d_pops, 0u, x->nb, rocprim::plus<uint32_t>(), nullptr));
            hipLaunchKernelGGL(wtsa_counts_kernel, grid(x->nb), dim3(256), 0, nullptr, lb, d_pops, x->nb);
            VLG_HIP_TRY(hipGetLastError());
            if (lvl + 1 < x->levels) {                              // arrangement of the next level: stable by the top lvl + 1 bits
                hipLaunchKernelGGL(wtsa_prefix_keys_kernel, grid(n), dim3(256), 0, nullptr, cur, n, x->levels - 1 - lvl, d_ka);
                tb = sort_tb;
                VLG_HIP_TRY(rocprim::radix_sort_pairs(d_tmp, tb, d_ka, d_kb, cur, other, n, 0, lvl + 1, nullptr));
                std::swap(cur, other);
            }
        }
        VLG_HIP_TRY(hipDeviceSynchronize());
        return VLG_OK;
    };
    vlg_status st = run();
    for (void* p : {(void*)d_bytes, (void*)d_sa, (void*)d_a, (void*)d_b, (void*)d_ka, (void*)d_kb, (void*)d_pops, d_tmp, (void*)d_flag, (void*)d_pos,
                    (void*)d_out, d_scan2})
        if (p) (void)hipFree(p);
    if (st) { vlg_wtsa_destroy(x); return st; }
    *out = x;
    return VLG_OK;
}

extern "C" vlg_status vlg_wtsa_get_info(const vlg_wtsa* x, vlg_wtsa_info* info)
{
    if (!x || !info) return fail(VLG_E_INVALID, "null argument");
    info->n = x->n_vals; info->symbol_bytes = x->sym_bytes; info->levels = x->levels; info->blocks_per_level = x->nb;
    info->hbm_bytes = x->n_text * x->sym_bytes + (uint64_t)x->levels * x->nb * sizeof(Block);
    return VLG_OK;
}

extern "C" vlg_status vlg_wtsa_sa_batch(const vlg_wtsa* x, const uint64_t* d_i, uint64_t* d_out, uint64_t count, void* stream)
{
    if (!x || (count && (!d_i || !d_out))) return fail(VLG_E_INVALID, "null argument");
    if (!count) return VLG_OK;
    hipLaunchKernelGGL(wtsa_sa_kernel, dim3(grid_for(count, 8192)), dim3(256), 0, (hipStream_t)stream, wtsa_view(x), d_i, d_out, count);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}

namespace {
// bits of one level as plain 64-bit words (bit i of the level = word[i >> 6] >> (i & 63)): what wt_int::tree holds for it
__global__ void wtsa_level_words_kernel(WtsaView w, uint32_t lvl, uint64_t* __restrict__ out, uint64_t n_words)
{
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_words; t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t word = 0;
        for (uint32_t half = 0; half < 2; ++half) {                 // 32 bits at a time: a 32-bit data word of a block, or parts of two
            const uint64_t bit = t * 64 + 32 * half;
            if (bit >= w.n_vals) break;
            const uint64_t blk = bit / kBlockBits;
            const uint32_t off = (uint32_t)(bit - blk * kBlockBits), wi = off >> 5, sh = off & 31;
            const Block* B = w.blocks + (uint64_t)lvl * w.nb + blk;
            uint64_t v = B->w[wi] >> sh;
            if (sh) {                                                // the rest comes from the next data word (maybe of the next block)
                const uint32_t nx = wi + 1 < 7 ? B->w[wi + 1] : (blk + 1 < w.nb ? B[1].w[0] : 0u);
                v |= (uint64_t)nx << (32 - sh);
            }
            word |= (v & 0xFFFFFFFFull) << (32 * half);
        }
        const uint64_t left = w.n_vals - t * 64;
        if (left < 64) word &= (1ull << left) - 1;
        out[t] = word;
    }
}
__global__ void __launch_bounds__(256) wtsa_range_walk_kernel(WtsaView w, const uint64_t* __restrict__ l, const uint64_t* __restrict__ len,
                                                              const uint64_t* __restrict__ x, int quantile, uint64_t* __restrict__ out, uint64_t count)
{
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = l[t], n = len[t];
        const bool ok = a <= w.n_vals && n <= w.n_vals - a && (!quantile || x[t] < n);
        out[t] = !ok ? ~0ull : (quantile ? wtsa_walk<true>(w, a, n, x[t]) : (n ? wtsa_walk<false>(w, a, n, x[t]) : 0));
    }
}
}  // namespace

extern "C" vlg_status vlg_wtsa_export_level(const vlg_wtsa* x, uint32_t level, uint64_t* h_words)
{
    if (!x || !h_words) return fail(VLG_E_INVALID, "null argument");
    if (level >= x->levels) return fail(VLG_E_INVALID, "no such level");
    const uint64_t nw = (x->n_vals + 63) / 64;
    uint64_t* d = nullptr;
    VLG_HIP_TRY(hipMalloc((void**)&d, nw * 8));
    hipLaunchKernelGGL(wtsa_level_words_kernel, dim3(grid_for(nw, 8192)), dim3(256), 0, nullptr, wtsa_view(x), level, d, nw);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(h_words, d, nw * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    VLG_HIP_TRY(e);
    return VLG_OK;
}

extern "C" vlg_status vlg_wtsa_range_walk_batch(const vlg_wtsa* x, const uint64_t* d_l, const uint64_t* d_len, const uint64_t* d_x, int quantile,
                                                uint64_t* d_out, uint64_t count, void* stream)
{
    if (!x || (count && (!d_l || !d_len || !d_x || !d_out))) return fail(VLG_E_INVALID, "null argument");
    if (!count) return VLG_OK;
    hipLaunchKernelGGL(wtsa_range_walk_kernel, dim3(grid_for(count, 8192)), dim3(256), 0, (hipStream_t)stream, wtsa_view(x), d_l, d_len, d_x,
                       quantile, d_out, count);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}

namespace {
vlg_status wtsa_ranges_device(const vlg_wtsa* x, const vlg_queries* q, uint64_t* d_sp, uint64_t* d_len, hipStream_t st)
{
    if (q->sym_bytes != x->sym_bytes) return fail(VLG_E_INVALID, "the query batch and the index have different alphabets");
    if (!q->nsub) return VLG_OK;
    hipLaunchKernelGGL(wtsa_ranges_kernel, dim3(grid_for(q->nsub, 4096)), dim3(256), 0, st, wtsa_view(x), q->d_blob, q->d_suboff, q->nsub, d_sp, d_len);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}
}  // namespace

extern "C" vlg_status vlg_wtsa_ranges(const vlg_wtsa* x, const vlg_queries* q, uint64_t* h_sp, uint64_t* h_ep, void* stream)
{
    if (!x || !q || (q->nsub && (!h_sp || !h_ep))) return fail(VLG_E_INVALID, "null argument");
    if (!q->nsub) return VLG_OK;
    uint64_t* d = nullptr;
    VLG_HIP_TRY(hipMalloc((void**)&d, 2 * q->nsub * 8));
    std::vector<uint64_t> h(2 * q->nsub);
    vlg_status s = wtsa_ranges_device(x, q, d, d + q->nsub, (hipStream_t)stream);
    hipError_t e = hipSuccess;
    if (!s) e = hipMemcpyAsync(h.data(), d, 2 * q->nsub * 8, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (!s && e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d);
    if (s) return s;
    VLG_HIP_TRY(e);
    for (uint64_t i = 0; i < q->nsub; ++i) { h_sp[i] = h[i]; h_ep[i] = h[i] + h[q->nsub + i] - 1; }    // sp = ep + 1 when there is no occurrence
    return VLG_OK;
}

extern "C" vlg_status vlg_wtsa_search_batch(const vlg_wtsa* x, const vlg_queries* q, uint64_t max_matches, vlg_workspace* ws, vlg_result** out)
{
    if (!x || !q || !ws || !out) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    hipStream_t st = ws->stream;
    const uint64_t nq = q->nq, nsub = q->nsub;
    if (nq > 0xFFFFFFF0ull) return fail(VLG_E_UNSUPPORTED, "too many queries in one batch");
    vlg_result* res = new vlg_result();
    memset(&res->sum, 0, sizeof res->sum);
    res->sum.n_queries = nq;
    res->counts.assign(nq, 0);
    res->k.resize(nq);
    for (uint64_t i = 0; i < nq; ++i) res->k[i] = (uint32_t)(q->qsub[i + 1] - q->qsub[i]);
    uint8_t* d_mem = nullptr;
    ResultPiece piece;
    piece.q0 = 0; piece.q1 = nq;
    auto run = [&]() -> vlg_status {
        if (!nq) { res->pieces.push_back(piece); return VLG_OK; }
        // device scratch: ranges, gap bounds, query table, counts, checksum
        const uint64_t bytes = (4 * (nsub + 1) + nq + 2) * 8 + (nq + 1) * sizeof(WQuery) + 1024;
        VLG_HIP_TRY(hipMalloc((void**)&d_mem, bytes));
        uint64_t* d_sp = (uint64_t*)d_mem;
        uint64_t* d_len = d_sp + nsub + 1;
        uint64_t* d_lo = d_len + nsub + 1;
        uint64_t* d_hi = d_lo + nsub + 1;
        unsigned long long* d_counts = (unsigned long long*)(d_hi + nsub + 1);
        unsigned long long* d_chk = d_counts + nq;
        WQuery* d_q = (WQuery*)(d_chk + 2);
        svec<WQuery> hq(nq);
        for (uint64_t i = 0; i < nq; ++i) hq[i] = WQuery{res->k[i], (uint32_t)q->qsub[i], q->end_len[i], 0, 0};
        svec<uint64_t> hlo(q->lo.begin(), q->lo.end()), hhi(q->hi.begin(), q->hi.end());
        if (nsub) {
            VLG_HIP_TRY(hipMemcpyAsync(d_lo, hlo.data(), nsub * 8, hipMemcpyHostToDevice, st));
            VLG_HIP_TRY(hipMemcpyAsync(d_hi, hhi.data(), nsub * 8, hipMemcpyHostToDevice, st));
        }
        VLG_HIP_TRY(hipMemcpyAsync(d_q, hq.data(), nq * sizeof(WQuery), hipMemcpyHostToDevice, st));
        VLG_HIP_TRY(hipMemsetAsync(d_chk, 0, 16, st));
        {
            Timed t(ws, KS_BSEARCH, 0);
            if (vlg_status s = wtsa_ranges_device(x, q, d_sp, d_len, st)) return s;
        }
        const uint32_t kmax = std::max<uint32_t>(q->kmax, 1);
        const size_t lds = (size_t)2 * kmax * 64 * 4;
        // one wavefront per query (its shortest list is looked at 64 elements at a time), or the older one lane per query
        const bool wave = [] { const char* e = getenv("VLG_WTSA_LANE_PER_QUERY"); return !(e && e[0] == '1'); }();
        const uint32_t wgs = wave ? (uint32_t)nq : (uint32_t)((nq + 63) / 64);
        const WtsaView w = wtsa_view(x);
        // A capped search whose matches fit a scratch buffer at cap per query runs ONCE (the uncapped one counts first, then emits)
        const uint64_t per_query = max_matches * (1 + (ws->tuples ? (uint64_t)kmax : 0)) * 8;
        if (wave && max_matches && max_matches <= (1u << 20) && nq * per_query <= (2ull << 30)) {
            uint64_t* t_first = nullptr;
            VLG_HIP_TRY(hipMalloc((void**)&t_first, nq * per_query + 2 * (nq + 1) * 8));
            struct Free { void* p; ~Free() { (void)hipFree(p); } } free_tmp{t_first};
            uint64_t* t_tuples = ws->tuples ? t_first + nq * max_matches : nullptr;
            uint64_t* d_to_first = t_first + nq * per_query / 8;
            uint64_t* d_to_tuple = d_to_first + nq + 1;
            for (uint64_t i = 0; i < nq; ++i) { hq[i].out_first = i * max_matches; hq[i].out_tuple = i * max_matches * kmax; }
            VLG_HIP_TRY(hipMemcpyAsync(d_q, hq.data(), nq * sizeof(WQuery), hipMemcpyHostToDevice, st));
            {
                Timed t(ws, KS_JOIN_CHAIN, 0);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(wtsa_search_kernel<true, true>), dim3(wgs), dim3(64), lds, st, w, d_sp, d_len, d_lo, d_hi, d_q,
                                   (uint32_t)nq, kmax, max_matches, d_counts, t_first, t_tuples, d_chk);
            }
            VLG_HIP_TRY(hipGetLastError());
            svec<unsigned long long> counts(nq);
            unsigned long long chk = 0;
            VLG_HIP_TRY(hipMemcpyAsync(counts.data(), d_counts, nq * 8, hipMemcpyDeviceToHost, st));
            VLG_HIP_TRY(hipMemcpyAsync(&chk, d_chk, 8, hipMemcpyDeviceToHost, st));
            VLG_HIP_TRY(hipStreamSynchronize(st));
            svec<uint64_t> to_first(nq + 1), to_tuple(nq + 1);
            uint64_t M = 0, TV = 0;
            for (uint64_t i = 0; i < nq; ++i) {
                to_first[i] = M; to_tuple[i] = TV;
                M += counts[i]; TV += ws->tuples ? counts[i] * hq[i].k : 0;
                res->counts[i] = counts[i];
            }
            piece.matches = M; piece.tuple_vals = TV;
            if (M) {
                VLG_HIP_TRY(result_alloc(&piece.d_first, M * 8, &piece.first_bytes));
                if (TV) VLG_HIP_TRY(result_alloc(&piece.d_tuples, TV * 8, &piece.tuple_bytes));
                VLG_HIP_TRY(hipMemcpyAsync(d_to_first, to_first.data(), nq * 8, hipMemcpyHostToDevice, st));
                VLG_HIP_TRY(hipMemcpyAsync(d_to_tuple, to_tuple.data(), nq * 8, hipMemcpyHostToDevice, st));
                Timed t(ws, KS_GATHER, 8ull * (M + TV));
                hipLaunchKernelGGL(wtsa_compact_kernel, dim3(grid_for(nq * max_matches, 8192)), dim3(256), 0, st, d_q, d_to_first, d_to_tuple, d_counts,
                                   (uint32_t)nq, max_matches, t_first, TV ? t_tuples : nullptr, static_cast<uint64_t*>(piece.d_first),
                                   static_cast<uint64_t*>(piece.d_tuples));
                VLG_HIP_TRY(hipGetLastError());
            }
            VLG_HIP_TRY(hipStreamSynchronize(st));
            res->pieces.push_back(piece);
            res->sum.n_matches = M;
            res->sum.n_tuple_values = TV;
            res->sum.checksum = chk;
            res->sum.n_chunks = 1;
            return VLG_OK;
        }
        {
            Timed t(ws, KS_JOIN_CHAIN, 0);
            if (wave)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(wtsa_search_kernel<false, true>), dim3(wgs), dim3(64), lds, st, w, d_sp, d_len, d_lo, d_hi, d_q,
                                   (uint32_t)nq, kmax, max_matches, d_counts, nullptr, nullptr, nullptr);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(wtsa_search_kernel<false, false>), dim3(wgs), dim3(64), lds, st, w, d_sp, d_len, d_lo, d_hi, d_q,
                                   (uint32_t)nq, kmax, max_matches, d_counts, nullptr, nullptr, nullptr);
        }
        VLG_HIP_TRY(hipGetLastError());
        svec<unsigned long long> counts(nq);
        VLG_HIP_TRY(hipMemcpyAsync(counts.data(), d_counts, nq * 8, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipStreamSynchronize(st));
        uint64_t M = 0, TV = 0;
        for (uint64_t i = 0; i < nq; ++i) {
            hq[i].out_first = M; hq[i].out_tuple = TV;
            M += counts[i]; TV += ws->tuples ? counts[i] * hq[i].k : 0;
            res->counts[i] = counts[i];
        }
        piece.matches = M; piece.tuple_vals = TV;
        if (M) {
            VLG_HIP_TRY(result_alloc(&piece.d_first, M * 8, &piece.first_bytes));
            if (TV) VLG_HIP_TRY(result_alloc(&piece.d_tuples, TV * 8, &piece.tuple_bytes));
            VLG_HIP_TRY(hipMemcpyAsync(d_q, hq.data(), nq * sizeof(WQuery), hipMemcpyHostToDevice, st));
            Timed t(ws, KS_GATHER, 8ull * (M + TV));
            if (wave)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(wtsa_search_kernel<true, true>), dim3(wgs), dim3(64), lds, st, w, d_sp, d_len, d_lo, d_hi, d_q,
                                   (uint32_t)nq, kmax, max_matches, d_counts, static_cast<uint64_t*>(piece.d_first), static_cast<uint64_t*>(piece.d_tuples), d_chk);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(wtsa_search_kernel<true, false>), dim3(wgs), dim3(64), lds, st, w, d_sp, d_len, d_lo, d_hi, d_q,
                                   (uint32_t)nq, kmax, max_matches, d_counts, static_cast<uint64_t*>(piece.d_first), static_cast<uint64_t*>(piece.d_tuples), d_chk);
            VLG_HIP_TRY(hipGetLastError());
        }
        unsigned long long chk = 0;
        VLG_HIP_TRY(hipMemcpyAsync(&chk, d_chk, 8, hipMemcpyDeviceToHost, st));
        VLG_HIP_TRY(hipStreamSynchronize(st));
        res->pieces.push_back(piece);
        res->sum.n_matches = M;
        res->sum.n_tuple_values = TV;
        res->sum.checksum = chk;
        res->sum.n_chunks = 1;
        return VLG_OK;
    };
    vlg_status stt;
    {
        HostPoolScope staging(&ws->host);
        try { stt = run(); }
        catch (const std::bad_alloc&) { stt = fail(VLG_E_OOM, "out of host memory (pinned staging)"); }
    }
    if (d_mem) (void)hipFree(d_mem);
    if (stt) { if (piece.d_first && res->pieces.empty()) result_cache().give(piece.d_first, piece.first_bytes);
               if (piece.d_tuples && res->pieces.empty()) result_cache().give(piece.d_tuples, piece.tuple_bytes);
               vlg_result_destroy(res); return stt; }
    *out = res;
    return VLG_OK;
}
