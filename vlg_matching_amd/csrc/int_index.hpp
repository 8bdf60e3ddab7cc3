// FM-index of an INTEGER text on the device: csa_wt<wt_int<>, dens, ., sa_order_sa_sampling, ., int_alphabet<>>.
// Internal to search.hip's translation unit; included exactly once, behind wtsa.hpp (whose builder kernels it shares).
//   int_alphabet (char2comp / comp2char / C)   include/sdsl/csa_alphabet_strategy.hpp:394-470, 496-536
//   wt_int::rank / inverse_select on the BWT   include/sdsl/wt_int.hpp:370-395, 405-430
//   LF, csa[i], backward_search, locate        suffix_array_helper.hpp:336-349, csa_wt.hpp:335-348, suffix_array_algorithm.hpp:250-326, 604-619
//
// HBM layout (ONE allocation, like the byte index; IntHeader in common.hpp):
//   levels  the BWT as a wavelet MATRIX over the COMPACT symbols (comp = rank of the symbol in the sorted alphabet, the sentinel 0
//           first): level l is one bit-vector of n bits -- bit l (from the top) of every symbol in the arrangement of that level, which is
//           the previous level's stably partitioned by its bit, zeros first -- cut into the 256-bit super-blocks of K1 {224 bits, ones
//           before}.  The reference keeps the raw symbols in a level-wise wt_int whose nodes are intervals: a rank there reads three
//           positions per level (node start, position, node end: wt_int.hpp:381-384).  In the matrix a position maps to the next level
//           by  bit ? Z[l] + rank1(p) : p - rank1(p)  (Z[l] = zeros of level l): ONE super-block read per level for rank and for
//           inverse_select alike, and the answers are the same numbers (divergence of layout only; tests compare rank, LF, csa[i],
//           intervals and tuples with the reference structure restated on the CPU).
//   Z       levels words; D[c] = C[c] - (first position of symbol c in the last arrangement), so that
//           C[c] + rank_c(i) = D[c] + walk(i, c)  and  LF(i) = D[c] + walk(i) with c read off the bits on the way;
//   C       sigma + 1 words; comp2char: sigma symbols ascending (queries are mapped by binary search); samples: SA[0], SA[d], ...
// Limits: symbols are uint32_t, none of them 0 (construct() refuses a 0 symbol: include/sdsl/construct.hpp:36-45); n <= 2^32 / 5
// (the suffix sorter sees five bytes per symbol).
#pragma once

namespace {

void bind_int_view(vlg_index* idx)
{
    const uint8_t* b = reinterpret_cast<const uint8_t*>(idx->d_blob);
    const IntHeader& h = idx->ihdr;
    IntView& v = idx->iview;
    v.bv_kind = (uint32_t)h.bv_kind; v.pad_ = 0;
    v.blocks = h.bv_kind == kBvPlain ? reinterpret_cast<const Block*>(b + h.off_levels) : nullptr;
    v.rrr_hdr = h.bv_kind == kBvRrr63 ? reinterpret_cast<const uint4*>(b + h.off_rrr_hdr) : nullptr;
    v.rrr_stream = h.bv_kind == kBvRrr63 ? reinterpret_cast<const uint64_t*>(b + h.off_rrr_stream) : nullptr;
    v.rrr_tables = h.bv_kind == kBvRrr63 ? reinterpret_cast<const RrrTables*>(b + h.off_binom) : nullptr;
    v.stride = h.bv_kind == kBvRrr63 ? h.n_sb : h.nb;
    v.Z = reinterpret_cast<const uint64_t*>(b + h.off_Z);
    v.D = reinterpret_cast<const uint64_t*>(b + h.off_D);
    v.C = reinterpret_cast<const uint64_t*>(b + h.off_C);
    v.comp2char = reinterpret_cast<const uint32_t*>(b + h.off_c2c);
    v.samples = reinterpret_cast<const uint32_t*>(b + h.off_samples);
    v.n = h.n; v.nb = h.nb; v.sigma = h.sigma; v.n_samples = h.n_samples; v.n_levels = h.levels; v.dens = h.dens;
    // what the generic entry points read from the byte header
    BlobHeader& g = idx->hdr;
    memset(&g, 0, sizeof g);
    g.magic = kIntBlobMagic; g.total_bytes = h.total_bytes; g.n = h.n; g.sigma = (uint32_t)std::min<uint64_t>(h.sigma, 0xFFFFFFFFull);
    g.dens = h.dens; g.n_samples = h.n_samples; g.sample_bytes = 4; g.bv_kind = h.bv_kind == kBvRrr63 ? VLG_BV_INT_MATRIX_RRR63 : VLG_BV_INT_MATRIX;
    g.n_blocks = (h.bv_kind == kBvRrr63 ? h.n_sb : h.nb) * h.levels; g.n_rrr_sb = h.n_sb * h.levels; g.rrr_stream_words = h.rrr_words;
    g.max_code_len = h.levels; g.wt_bits = h.n * h.levels;
    idx->is_int = true;
}

// int_alphabet::char2comp (csa_alphabet_strategy.hpp:421-437): 0 for a symbol that does not occur
__device__ __forceinline__ uint32_t int_char2comp(const IntView& v, uint32_t sym)
{
    uint64_t lo = 0, hi = v.sigma;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (v.comp2char[mid] < sym) lo = mid + 1; else hi = mid; }
    return lo < v.sigma && v.comp2char[lo] == sym ? (uint32_t)lo : 0u;
}

// what every workgroup that walks the matrix keeps in LDS: the zeros per level and whatever the bit-vector policy needs (BV: PlainBV or
// RrrBV of device_rank.hpp, reading the IntView as they read the byte index's IndexView; level l is "node" l * stride)
template <class BV>
struct IntLds {
    uint64_t Z[kMaxIntLevels];
    typename BV::Shared sh;
};
template <class BV>
__device__ __forceinline__ void stage_int(IntLds<BV>& s, const IntView& v)
{
    if (threadIdx.x < v.n_levels) s.Z[threadIdx.x] = v.Z[threadIdx.x];
    BV::stage(s.sh, v);
    __syncthreads();
}

// position of (the first i symbols' share of) symbol c in the last arrangement: D[c] + this = C[c] + rank_c(i)
template <class BV>
__device__ __forceinline__ uint64_t int_walk(const IntView& v, const IntLds<BV>& s, uint64_t p, uint32_t c, uint32_t& levels)
{
    for (uint32_t l = 0; l < v.n_levels; ++l) {
        const uint64_t r1 = BV::rank(v, s.sh, (uint32_t)(l * v.stride), p);
        ++levels;
        p = ((c >> (v.n_levels - 1 - l)) & 1) ? s.Z[l] + r1 : p - r1;
    }
    return p;
}

// backward_search (suffix_array_algorithm.hpp:250-278, 305-326), one lane per sub-pattern; symbols are the raw uint32_t of the query
template <class BV>
__global__ void __launch_bounds__(256) int_backward_search_kernel(IntView v, const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off, uint64_t n_pat,
                                                                  uint64_t* __restrict__ out_l, uint64_t* __restrict__ out_r,
                                                                  unsigned long long* __restrict__ stat_levels)
{
    __shared__ IntLds<BV> sZ;
    stage_int(sZ, v);
    uint32_t levels = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pat; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t* pat = reinterpret_cast<const uint32_t*>(blob + off[p]);
        uint64_t m = (off[p + 1] - off[p]) / 4;
        uint64_t l = 0, r = v.n - 1;
        while (m > 0 && r + 1 - l > 0) {
            const uint32_t c = pat[--m];
            const uint32_t cc = int_char2comp(v, c);
            if (cc == 0 && c > 0) { l = 1; r = 0; }                  // :263-265
            else if (l == 0 && r + 1 == v.n) { l = v.C[cc]; r = v.C[cc + 1] - 1; }   // :268-270
            else {
                const uint64_t d = v.D[cc];
                const uint64_t nl = d + int_walk(v, sZ, l, cc, levels);
                r = d + int_walk(v, sZ, r + 1, cc, levels) - 1;
                l = nl;
            }
        }
        out_l[p] = l;
        out_r[p] = r;
    }
    if (stat_levels) {
        unsigned long long t = levels;
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
        if ((threadIdx.x & 63) == 0 && t) atomicAdd(stat_levels, t);
    }
}

// csa[i] (csa_wt.hpp:335-348) for the SA indices in io[], in place; the lanes of a wave refill from the wave's slice like K3's
template <class BV>
__global__ void __launch_bounds__(256) int_locate_kernel(IntView v, uint32_t* __restrict__ io, uint64_t total, uint32_t per_wave,
                                                         unsigned long long* __restrict__ stats)
{
    __shared__ IntLds<BV> sZ;
    stage_int(sZ, v);
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t next = wave * per_wave;
    const uint64_t slice_end = next + per_wave < total ? next + per_wave : total;
    uint64_t t = 0, i = 0;
    uint32_t lvl = 0, c = 0, off = 0, n_lf = 0, n_lv = 0;
    bool active = false, need = true;
    for (;;) {
        const unsigned long long m = __ballot(need);
        if (m) {
            const uint32_t before = __popcll(m & ((1ull << lane) - 1ull));
            if (need) {
                const uint64_t cand = next + before;
                if (cand < slice_end) { t = cand; i = io[cand]; off = 0; lvl = 0; c = 0; active = true; }
                else active = false;
                need = false;
            }
            next += __popcll(m);
        }
        if (!__any(active)) break;
        if (active) {
            if (lvl == 0 && i % v.dens == 0) {                       // csa_sampling_strategy.hpp:102-111
                uint64_t r = (uint64_t)v.samples[i / v.dens] + off;
                if (r >= v.n) r -= v.n;
                io[t] = (uint32_t)r;
                need = true;
                active = false;
            } else if (v.n_levels == 0) {                            // only the sentinel exists
                i = 0; ++off;
            } else {
                uint32_t bit;
                uint64_t r1;
                BV::rank_bit(v, sZ.sh, (uint32_t)(lvl * v.stride), i, r1, bit);
                ++n_lv;
                i = bit ? sZ.Z[lvl] + r1 : i - r1;
                c = (c << 1) | bit;
                if (++lvl == v.n_levels) { i = v.D[c] + i; lvl = 0; c = 0; ++off; ++n_lf; }      // LF: suffix_array_helper.hpp:341-348
            }
        }
    }
    if (stats) {
        unsigned long long a = n_lf, b = n_lv;
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
        if (lane == 0) { if (a) atomicAdd(&stats[0], a); if (b) atomicAdd(&stats[1], b); }
    }
}

// ---- the sorted sweep on the wavelet matrix (round 4; kernels.hip: K3s explains the sweep, sweep_element the records) ---------------
// An LF step reads one super-block per matrix level and the symbol's D entry; the symbol read (its compact number, < sigma <= 65534) is
// the partition key.  Everything else -- rounds, partition, member bit-vector, records, resolution -- is run_locate_sweep's.
template <class BV>
__device__ __forceinline__ uint64_t int_lf(const IntView& v, const IntLds<BV>& sZ, uint64_t i, uint32_t& c, uint32_t& n_lv)
{
    uint64_t p = i;
    c = 0;
    for (uint32_t l = 0; l < v.n_levels; ++l) {
        uint32_t bit;
        uint64_t r1;
        BV::rank_bit(v, sZ.sh, (uint32_t)(l * v.stride), p, r1, bit);
        ++n_lv;
        p = bit ? sZ.Z[l] + r1 : p - r1;
        c = (c << 1) | bit;
    }
    return v.D[c] + p;                                               // LF: suffix_array_helper.hpp:341-348
}

__device__ __forceinline__ void int_counters_add(unsigned long long a, unsigned long long b, unsigned long long c, unsigned long long* __restrict__ stats,
                                                 unsigned long long* __restrict__ n_done)
{
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); c += __shfl_down(c, o); }
    if ((threadIdx.x & 63) == 0) { if (a) atomicAdd(&stats[0], a); if (b) atomicAdd(&stats[1], b); if (c && n_done) atomicAdd(n_done, c); }
}

template <class BV, bool kTrail, bool kFirst, bool kAhead>
__device__ __forceinline__ void int_sweep_element(const IntView& v, const IntLds<BV>& sZ, uint64_t e, uint64_t v64, uint64_t* __restrict__ val,
                                                  uint16_t* __restrict__ key, uint32_t step, uint32_t* __restrict__ out, const Block* __restrict__ member,
                                                  uint64_t* __restrict__ rec, uint64_t slot0, bool probed, uint32_t& n_lv, uint32_t& n_lf, uint32_t& n_fin)
{
    const uint64_t i = v64 & 0xFFFFFFFFull, slot = v64 >> 32;
    uint32_t owner = 0;
    if (i % v.dens == 0) {                                           // csa_sampling_strategy.hpp:102-111
        uint64_t r = (uint64_t)v.samples[i / v.dens] + step;
        if (r >= v.n) r -= v.n;                                      // csa_wt.hpp:343-347
        if (kTrail) rec[slot0 + slot] = r; else out[slot] = (uint32_t)r;
        key[e] = (uint16_t)v.sigma;
        ++n_fin;
    } else if (kTrail && !kFirst && !probed && member_probe(member, i, owner)) {
        const uint64_t delta = step;
        const uint64_t ro = rec[owner];
        uint64_t r;
        if (ro == ~0ull) r = (delta << 32) | owner;                  // still walking: follow it
        else if ((ro >> 32) == 0) r = ro + delta;                    // its position is known
        else r = ro + (delta << 32);                                 // it follows someone itself: follow that one
        rec[slot0 + slot] = r;
        key[e] = (uint16_t)v.sigma;
        ++n_fin;
    } else {
        if (kTrail && kFirst) rec[slot0 + slot] = ~0ull;
        uint32_t c;
        const uint64_t j = int_lf(v, sZ, i, c, n_lv);
        ++n_lf;
        if (kTrail && kAhead && member_probe(member, j, owner)) {
            rec[slot0 + slot] = ((uint64_t)(step + 1) << 32) | owner;
            key[e] = (uint16_t)v.sigma;
            ++n_fin;
        } else {
            val[e] = (v64 & ~0xFFFFFFFFull) | j;
            key[e] = (uint16_t)c;
        }
    }
}

template <class BV, bool kTrail>
__global__ void __launch_bounds__(256) int_sweep_step_kernel(IntView v, uint64_t* __restrict__ val, uint16_t* __restrict__ key, uint64_t count, uint32_t step,
                                                             uint32_t* __restrict__ out, unsigned long long* __restrict__ stats, unsigned long long* __restrict__ n_done,
                                                             const Block* __restrict__ member, uint64_t* __restrict__ rec, uint64_t slot0, bool probed)
{
    __shared__ IntLds<BV> sZ;
    stage_int(sZ, v);
    uint32_t n_lv = 0, n_lf = 0, n_fin = 0;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (uint64_t)gridDim.x * blockDim.x)
        int_sweep_element<BV, kTrail, false, false>(v, sZ, e, val[e], val, key, step, out, member, rec, slot0, probed, n_lv, n_lf, n_fin);
    int_counters_add(n_lf, n_lv, n_fin, stats, n_done);
}

template <class BV, bool kTrail, bool kAhead>
__global__ void __launch_bounds__(256) int_sweep_first_kernel(IntView v, const uint64_t* __restrict__ l, const uint64_t* __restrict__ out_off, uint64_t n_pat, uint64_t t0,
                                                              uint64_t total, uint64_t* __restrict__ val, uint16_t* __restrict__ key, uint32_t* __restrict__ out,
                                                              unsigned long long* __restrict__ stats, unsigned long long* __restrict__ n_done,
                                                              const Block* __restrict__ member, uint64_t* __restrict__ rec,
                                                              const uint32_t* __restrict__ chunk_list)
{
    constexpr uint32_t kPer = kSweepChunk / 256;
    __shared__ IntLds<BV> sZ;
    stage_int(sZ, v);
    uint32_t n_lv = 0, n_lf = 0, n_fin = 0;
    for (uint64_t base = t0 + (uint64_t)blockIdx.x * kSweepChunk; base < total; base += (uint64_t)gridDim.x * kSweepChunk) {
        uint64_t p = chunk_list[(base - t0) / kSweepChunk];         // the list of the chunk's first element (sweep_chunk_lists_kernel)
#pragma unroll 1
        for (uint32_t i = 0; i < kPer; ++i) {
            const uint64_t t = base + i * 256 + threadIdx.x;
            if (t < total) {
                while (out_off[p + 1] <= t) ++p;
                const uint64_t v64 = ((t - t0) << 32) | (l[p] + (t - out_off[p]));
                int_sweep_element<BV, kTrail, true, kAhead>(v, sZ, t - t0, v64, val, key, 0u, out, member, rec, t0, false, n_lv, n_lf, n_fin);
            }
        }
    }
    int_counters_add(n_lf, n_lv, n_fin, stats, n_done);
}

// the stragglers: int_locate_kernel's refilling lanes on the elements val[] = slot << 32 | SA index that have walked `step` steps
template <class BV>
__global__ void __launch_bounds__(256) int_sweep_tail_kernel(IntView v, uint32_t* __restrict__ out, uint64_t total, uint32_t per_wave, unsigned long long* __restrict__ stats,
                                                             const uint64_t* __restrict__ val, uint32_t step, uint64_t* __restrict__ rec, uint64_t slot0,
                                                             const Block* __restrict__ member)
{
    __shared__ IntLds<BV> sZ;
    stage_int(sZ, v);
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t next = wave * per_wave;
    const uint64_t slice_end = next + per_wave < total ? next + per_wave : total;
    uint64_t t = 0, i = 0;
    uint32_t off = 0, n_lf = 0, n_lv = 0;
    bool active = false, need = true;
    for (;;) {
        const unsigned long long m = __ballot(need);
        if (m) {
            const uint32_t before = __popcll(m & ((1ull << lane) - 1ull));
            if (need) {
                const uint64_t cand = next + before;
                if (cand < slice_end) { const uint64_t e = val[cand]; t = e >> 32; i = e & 0xFFFFFFFFull; off = step; active = true; }
                else active = false;
                need = false;
            }
            next += __popcll(m);
        }
        if (!__any(active)) break;
        if (active) {
            uint32_t owner = 0;
            if (i % v.dens == 0) {
                uint64_t r = (uint64_t)v.samples[i / v.dens] + off;
                if (r >= v.n) r -= v.n;
                if (rec) rec[slot0 + t] = r; else out[t] = (uint32_t)r;
                need = true;
                active = false;
            } else if (rec && member && off != 0 && member_probe(member, i, owner)) {
                const uint64_t delta = off;
                const uint64_t ro = rec[owner];
                uint64_t r;
                if (ro == ~0ull) r = (delta << 32) | owner;
                else if ((ro >> 32) == 0) r = ro + delta;
                else r = ro + (delta << 32);
                rec[slot0 + t] = r;
                need = true;
                active = false;
            } else {
                uint32_t c;
                i = int_lf(v, sZ, i, c, n_lv);
                ++off;
                ++n_lf;
            }
        }
    }
    int_counters_add(n_lf, n_lv, 0, stats, nullptr);
}

// wt_int::rank(i, c) on raw symbols (for the primitives test): out = #c in BWT[0, i)
template <class BV>
__global__ void __launch_bounds__(256) int_rank_kernel(IntView v, const uint64_t* __restrict__ pos, const uint32_t* __restrict__ sym, uint64_t* __restrict__ out,
                                                       uint64_t count)
{
    __shared__ IntLds<BV> sZ;
    stage_int(sZ, v);
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = sym[j];
        const uint32_t cc = int_char2comp(v, c);
        uint32_t lv = 0;
        out[j] = (cc == 0 && c > 0) ? 0 : v.D[cc] + int_walk(v, sZ, pos[j], cc, lv) - v.C[cc];
    }
}

// ---- construction -------------------------------------------------------------------------------------------------------------------
__global__ void int_heads_kernel(const uint32_t* __restrict__ sorted, uint64_t n, uint32_t* __restrict__ head)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) head[j] = (j == 0 || sorted[j] != sorted[j - 1]) ? 1u : 0u;
}
// gid = inclusive scan of the heads: symbol number gid - 1 starts at j (comp 0 is the sentinel, so text symbols get gid)
__global__ void int_alphabet_kernel(const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ gid, uint64_t n, uint32_t* __restrict__ comp2char,
                                    uint64_t* __restrict__ C)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x)
        if (j == 0 || sorted[j] != sorted[j - 1]) { comp2char[gid[j]] = sorted[j]; C[gid[j]] = j + 1; }      // one sentinel stands before every symbol
}
// BWT in compact symbols: comp(text[SA[i] - 1]), the sentinel (comp 0) where SA[i] = 0
__global__ void int_bwt_kernel(const uint32_t* __restrict__ text, const uint32_t* __restrict__ sa, uint64_t n, const uint32_t* __restrict__ comp2char,
                               uint64_t sigma, uint32_t* __restrict__ bwt)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = sa[i];
        uint32_t c = 0;
        if (s) {
            const uint32_t sym = text[s - 1];
            uint64_t lo = 1, hi = sigma;
            while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (comp2char[mid] < sym) lo = mid + 1; else hi = mid; }
            c = (uint32_t)lo;
        }
        bwt[i] = c;
    }
}
__global__ void int_bit_keys_kernel(const uint32_t* __restrict__ vals, uint64_t n, uint32_t bit, uint32_t* __restrict__ keys)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) keys[i] = (vals[i] >> bit) & 1u;
}
// D[c] = C[c] - first position of c in the last arrangement (the walk of position 0 along c's bits)
__global__ void int_D_kernel(IntView v, uint64_t* __restrict__ D)
{
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < v.sigma; c += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t p = 0;                                                // (the index under construction is a plain one)
        for (uint32_t l = 0; l < v.n_levels; ++l) {
            const uint64_t r1 = node_rank1(v.blocks, (uint32_t)(l * v.nb), p);
            p = ((c >> (v.n_levels - 1 - l)) & 1) ? v.Z[l] + r1 : p - r1;
        }
        D[c] = v.C[c] - p;
    }
}
__global__ void int_samples_kernel(const uint32_t* __restrict__ sa, uint64_t n_samples, uint32_t dens, uint32_t* __restrict__ samples)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_samples; j += (uint64_t)gridDim.x * blockDim.x) samples[j] = sa[j * dens];
}
__global__ void int_zero_check_kernel(const uint32_t* __restrict__ text, uint64_t n, uint32_t* __restrict__ flag)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) if (text[j] == 0) *flag = 1;
}

inline void layout_int(IntHeader& h)
{
    uint64_t off = align_up(sizeof(IntHeader), 256);
    h.off_levels = off;
    if (h.bv_kind == kBvRrr63) {
        h.off_rrr_hdr = off;    off = align_up(off + std::max<uint64_t>((uint64_t)h.levels * h.n_sb, 1) * 32, 256);
        h.off_rrr_stream = off; off = align_up(off + (h.rrr_words + 2) * 8, 256);
        h.off_binom = off;      off = align_up(off + 64 * 64 * 8, 256);
    } else {
        off = align_up(off + (uint64_t)h.levels * h.nb * sizeof(Block), 256);
    }
    h.off_Z = off;       off = align_up(off + (uint64_t)kMaxIntLevels * 8, 256);
    h.off_D = off;       off = align_up(off + h.sigma * 8, 256);
    h.off_C = off;       off = align_up(off + (h.sigma + 1) * 8, 256);
    h.off_c2c = off;     off = align_up(off + h.sigma * 4, 256);
    h.off_samples = off; off = align_up(off + h.n_samples * 4, 256);
    h.total_bytes = off;
}

}  // namespace

namespace vlg {
void layout_int_blob(IntHeader& h) { layout_int(h); }          // for vlg_index_compress (index.hip)

// a blob whose magic says "integer index": called by vlg_index_attach_blob (index.hip)
vlg_status attach_int_blob(const void* d_blob, uint64_t bytes, vlg_index* idx)
{
    VLG_HIP_TRY(hipMemcpy(&idx->ihdr, d_blob, sizeof(IntHeader), hipMemcpyDeviceToHost));
    const IntHeader& h = idx->ihdr;
    if (h.magic != kIntBlobMagic || h.total_bytes > bytes || h.levels > kMaxIntLevels || h.bv_kind > kBvRrr63)
        return fail(VLG_E_INVALID, "not a VLG integer-index blob");
    idx->d_blob = const_cast<void*>(d_blob);
    idx->owns_blob = false;
    bind_int_view(idx);
    return VLG_OK;
}
}  // namespace vlg

extern "C" vlg_status vlg_index_build_int(const uint32_t* h_text, uint64_t n_symbols, uint32_t dens, vlg_index** out)
{
    if (!out || (n_symbols && !h_text)) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VLG_E_NO_DEVICE, "no HIP device available (the VLG library has no CPU fallback)");
    if (!dens) dens = 32;
    const uint64_t byte_len = n_symbols * 5;
    if (byte_len >= 0xFFFFFFF0ull) return fail(VLG_E_UNSUPPORTED, "integer text too long for the 32-bit suffix array of this index");
    release_cached_device_memory();
    const uint64_t n = n_symbols + 1;
    vlg_index* idx = new vlg_index();
    uint32_t *d_text = nullptr, *d_sa5 = nullptr, *d_flag = nullptr, *d_pos = nullptr, *d_sa = nullptr, *d_a = nullptr, *d_b = nullptr, *d_ka = nullptr,
             *d_kb = nullptr, *d_pops = nullptr, *d_c2c = nullptr;
    uint8_t* d_bytes = nullptr;
    uint64_t* d_C = nullptr;
    void* d_tmp = nullptr;
    auto grid = [](uint64_t m) { return dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((m + 255) / 256, 16384))); };
    auto run = [&]() -> vlg_status {
        VLG_HIP_TRY(hipMalloc((void**)&d_text, std::max<uint64_t>(n_symbols * 4, 16)));
        if (n_symbols) VLG_HIP_TRY(hipMemcpy(d_text, h_text, n_symbols * 4, hipMemcpyHostToDevice));
        VLG_HIP_TRY(hipMalloc((void**)&d_flag, (byte_len + 2) * 4));
        VLG_HIP_TRY(hipMemset(d_flag, 0, 4));
        if (n_symbols) hipLaunchKernelGGL(int_zero_check_kernel, grid(n_symbols), dim3(256), 0, nullptr, d_text, n_symbols, d_flag);
        uint32_t has_zero = 0;
        VLG_HIP_TRY(hipMemcpy(&has_zero, d_flag, 4, hipMemcpyDeviceToHost));
        if (has_zero) return fail(VLG_E_ZERO_BYTE, "the integer text contains the symbol 0 (reserved for the sentinel: construct.hpp:36-45)");
        // ---- suffix array: the integer text as five base-255 digits + 1 per symbol, then the aligned suffixes (as vlg_wtsa_build) ------------
        VLG_HIP_TRY(hipMalloc((void**)&d_sa5, (byte_len + 1) * 4));
        VLG_HIP_TRY(hipMalloc((void**)&d_sa, n * 4));
        if (n_symbols) {
            VLG_HIP_TRY(hipMalloc((void**)&d_bytes, byte_len));
            hipLaunchKernelGGL(wtsa_expand_kernel, grid(n_symbols), dim3(256), 0, nullptr, d_text, n_symbols, d_bytes);
            VLG_HIP_TRY(hipGetLastError());
            if (vlg_status s = vlg_suffix_array_device(d_bytes, byte_len, d_sa5, nullptr)) return s;
            (void)hipFree(d_bytes); d_bytes = nullptr;
            const uint64_t nb5 = byte_len + 1;
            VLG_HIP_TRY(hipMalloc((void**)&d_pos, nb5 * 4));
            hipLaunchKernelGGL(wtsa_aligned_flags_kernel, grid(nb5), dim3(256), 0, nullptr, d_sa5, nb5, d_flag);
            size_t tb = 0;
            VLG_HIP_TRY(rocprim::exclusive_scan(nullptr, tb, d_flag, d_pos, 0u, nb5, rocprim::plus<uint32_t>(), nullptr));
            VLG_HIP_TRY(hipMalloc(&d_tmp, tb + 16));
            VLG_HIP_TRY(rocprim::exclusive_scan(d_tmp, tb, d_flag, d_pos, 0u, nb5, rocprim::plus<uint32_t>(), nullptr));
            hipLaunchKernelGGL(wtsa_aligned_compact_kernel, grid(nb5), dim3(256), 0, nullptr, d_sa5, d_pos, nb5, d_sa);
            VLG_HIP_TRY(hipGetLastError());
            VLG_HIP_TRY(hipDeviceSynchronize());
            (void)hipFree(d_tmp); d_tmp = nullptr;
            (void)hipFree(d_pos); d_pos = nullptr;
        } else {
            VLG_HIP_TRY(hipMemset(d_sa, 0, 4));
        }
        (void)hipFree(d_sa5); d_sa5 = nullptr;
        (void)hipFree(d_flag); d_flag = nullptr;
        // ---- int_alphabet: sorted distinct symbols and their cumulative counts (csa_alphabet_strategy.hpp:496-536) ----------------------
        VLG_HIP_TRY(hipMalloc((void**)&d_a, n * 4));
        VLG_HIP_TRY(hipMalloc((void**)&d_b, n * 4));
        VLG_HIP_TRY(hipMalloc((void**)&d_ka, n * 4));
        VLG_HIP_TRY(hipMalloc((void**)&d_kb, n * 4));
        size_t sort_tb = 0, scan_tb = 0, pair_tb = 0;
        VLG_HIP_TRY(rocprim::radix_sort_keys(nullptr, sort_tb, d_a, d_b, n, 0, 32, nullptr));
        VLG_HIP_TRY(rocprim::inclusive_scan(nullptr, scan_tb, d_ka, d_kb, n, rocprim::plus<uint32_t>(), nullptr));
        VLG_HIP_TRY(rocprim::radix_sort_pairs(nullptr, pair_tb, d_ka, d_kb, d_a, d_b, n, 0, 1, nullptr));
        VLG_HIP_TRY(hipMalloc(&d_tmp, std::max(std::max(sort_tb, scan_tb), pair_tb) + 16));
        uint64_t sigma = 1;
        if (n_symbols) {
            size_t tb = sort_tb;
            VLG_HIP_TRY(rocprim::radix_sort_keys(d_tmp, tb, d_text, d_b, n_symbols, 0, 32, nullptr));       // d_b = sorted text
            hipLaunchKernelGGL(int_heads_kernel, grid(n_symbols), dim3(256), 0, nullptr, d_b, n_symbols, d_ka);
            tb = scan_tb;
            VLG_HIP_TRY(rocprim::inclusive_scan(d_tmp, tb, d_ka, d_kb, n_symbols, rocprim::plus<uint32_t>(), nullptr));   // d_kb = symbol number (1-based)
            uint32_t distinct = 0;
            VLG_HIP_TRY(hipMemcpy(&distinct, d_kb + (n_symbols - 1), 4, hipMemcpyDeviceToHost));
            sigma = (uint64_t)distinct + 1;
        }
        IntHeader& h = idx->ihdr;
        memset(&h, 0, sizeof h);
        h.magic = kIntBlobMagic; h.n = n; h.sigma = sigma; h.dens = dens; h.n_samples = (n + dens - 1) / dens; h.nb = n / kBlockBits + 1;
        h.levels = sigma > 1 ? bit_width64(sigma - 1) : 0;
        if (h.levels > kMaxIntLevels || (uint64_t)h.levels * h.nb >= 0xFFFFFFF0ull) return fail(VLG_E_UNSUPPORTED, "integer index too large for 32-bit block numbers");
        layout_int(h);
        VLG_HIP_TRY(hipMalloc(&idx->d_blob, h.total_bytes));
        idx->owns_blob = true;
        uint8_t* b = reinterpret_cast<uint8_t*>(idx->d_blob);
        VLG_HIP_TRY(hipMemset(b, 0, h.off_levels));
        VLG_HIP_TRY(hipMemcpy(b, &h, sizeof h, hipMemcpyHostToDevice));
        bind_int_view(idx);
        d_c2c = reinterpret_cast<uint32_t*>(b + h.off_c2c);
        d_C = reinterpret_cast<uint64_t*>(b + h.off_C);
        VLG_HIP_TRY(hipMemset(d_c2c, 0, sigma * 4));                       // comp 0 = the sentinel, C[0] = 0
        VLG_HIP_TRY(hipMemset(d_C, 0, (sigma + 1) * 8));
        if (n_symbols) hipLaunchKernelGGL(int_alphabet_kernel, grid(n_symbols), dim3(256), 0, nullptr, d_b, d_kb, n_symbols, d_c2c, d_C);
        VLG_HIP_TRY(hipMemcpy(d_C + sigma, &n, 8, hipMemcpyHostToDevice));
        // ---- BWT in compact symbols, then the wavelet matrix level by level ----------------------------------------------------------------
        hipLaunchKernelGGL(int_bwt_kernel, grid(n), dim3(256), 0, nullptr, d_text, d_sa, n, d_c2c, sigma, d_a);
        VLG_HIP_TRY(hipGetLastError());
        VLG_HIP_TRY(hipMalloc((void**)&d_pops, (h.nb + 1) * 4));
        Block* lv = reinterpret_cast<Block*>(b + h.off_levels);
        uint64_t* d_Z = reinterpret_cast<uint64_t*>(b + h.off_Z);
        std::vector<uint64_t> Z(kMaxIntLevels, 0);
        uint32_t* cur = d_a;
        uint32_t* other = d_b;
        size_t pops_tb = 0;
        VLG_HIP_TRY(rocprim::exclusive_scan(nullptr, pops_tb, d_pops, d_pops, 0u, h.nb + 1, rocprim::plus<uint32_t>(), nullptr));
        void* d_tmp2 = nullptr;
        VLG_HIP_TRY(hipMalloc(&d_tmp2, pops_tb + 16));
        vlg_status lst = VLG_OK;
        for (uint32_t l = 0; l < h.levels && !lst; ++l) {
            const uint32_t bit = h.levels - 1 - l;
            Block* lb = lv + (uint64_t)l * h.nb;
            auto step = [&]() -> vlg_status {
                VLG_HIP_TRY(hipMemsetAsync(d_pops, 0, (h.nb + 1) * 4, nullptr));
                hipLaunchKernelGGL(wtsa_emit_kernel, grid(h.nb * 7), dim3(256), 0, nullptr, cur, n, bit, lb, h.nb, d_pops);
                size_t tb = pops_tb;
                VLG_HIP_TRY(rocprim::exclusive_scan(d_tmp2, tb, d_pops, d_pops, 0u, h.nb + 1, rocprim::plus<uint32_t>(), nullptr));
                hipLaunchKernelGGL(wtsa_counts_kernel, grid(h.nb), dim3(256), 0, nullptr, lb, d_pops, h.nb);
                uint32_t ones = 0;
                VLG_HIP_TRY(hipMemcpy(&ones, d_pops + h.nb, 4, hipMemcpyDeviceToHost));
                Z[l] = n - ones;
                if (l + 1 < h.levels) {                                // next arrangement: stable by this bit, zeros first
                    hipLaunchKernelGGL(int_bit_keys_kernel, grid(n), dim3(256), 0, nullptr, cur, n, bit, d_ka);
                    tb = pair_tb;
                    VLG_HIP_TRY(rocprim::radix_sort_pairs(d_tmp, tb, d_ka, d_kb, cur, other, n, 0, 1, nullptr));
                    std::swap(cur, other);
                }
                VLG_HIP_TRY(hipGetLastError());
                return VLG_OK;
            };
            lst = step();
        }
        (void)hipFree(d_tmp2);
        if (lst) return lst;
        VLG_HIP_TRY(hipMemcpy(d_Z, Z.data(), kMaxIntLevels * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(int_D_kernel, grid(sigma), dim3(256), 0, nullptr, idx->iview, reinterpret_cast<uint64_t*>(b + h.off_D));
        hipLaunchKernelGGL(int_samples_kernel, grid(h.n_samples), dim3(256), 0, nullptr, d_sa, h.n_samples, dens, reinterpret_cast<uint32_t*>(b + h.off_samples));
        VLG_HIP_TRY(hipGetLastError());
        VLG_HIP_TRY(hipDeviceSynchronize());
        return VLG_OK;
    };
    const vlg_status st = run();
    for (void* p : {(void*)d_text, (void*)d_sa5, (void*)d_flag, (void*)d_pos, (void*)d_sa, (void*)d_a, (void*)d_b, (void*)d_ka, (void*)d_kb, (void*)d_pops,
                    (void*)d_bytes, d_tmp})
        if (p) (void)hipFree(p);
    if (st) { vlg_index_destroy(idx); return st; }
    *out = idx;
    return VLG_OK;
}

// int_alphabet of the index (two-phase: null buffers give sigma): C[sigma + 1], comp2char[sigma] (comp 0 = the sentinel)
extern "C" vlg_status vlg_index_export_int_alphabet(const vlg_index* idx, uint64_t* sigma, uint64_t* h_C, uint64_t* h_comp2char)
{
    if (!idx || !sigma) return fail(VLG_E_INVALID, "null argument");
    if (!idx->is_int) return fail(VLG_E_INVALID, "not an integer-alphabet index");
    *sigma = idx->ihdr.sigma;
    if (h_C) VLG_HIP_TRY(hipMemcpy(h_C, idx->iview.C, (idx->ihdr.sigma + 1) * 8, hipMemcpyDeviceToHost));
    if (h_comp2char) {
        std::vector<uint32_t> t(idx->ihdr.sigma);
        VLG_HIP_TRY(hipMemcpy(t.data(), idx->iview.comp2char, idx->ihdr.sigma * 4, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < idx->ihdr.sigma; ++i) h_comp2char[i] = t[i];
    }
    return VLG_OK;
}

// wt_int::rank(i, c) on the BWT of an integer index: out[j] = #sym[j] in BWT[0, i[j])
extern "C" vlg_status vlg_int_rank_batch(const vlg_index* idx, const uint64_t* d_i, const uint32_t* d_sym, uint64_t* d_out, uint64_t count, void* stream)
{
    if (!idx || (count && (!d_i || !d_sym || !d_out))) return fail(VLG_E_INVALID, "null argument");
    if (!idx->is_int) return fail(VLG_E_INVALID, "not an integer-alphabet index");
    if (!count) return VLG_OK;
    if (idx->iview.bv_kind == kBvRrr63)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(int_rank_kernel<RrrBV>), dim3(grid_for(count, 8192)), dim3(256), 0, (hipStream_t)stream, idx->iview, d_i, d_sym, d_out, count);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(int_rank_kernel<PlainBV>), dim3(grid_for(count, 8192)), dim3(256), 0, (hipStream_t)stream, idx->iview, d_i, d_sym, d_out, count);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}

namespace vlg {

vlg_status launch_int_backward_search(const IntView& v, const uint8_t* d_blob, const uint64_t* d_off, uint64_t n_pat, uint64_t* d_l, uint64_t* d_r,
                                      unsigned long long* d_stat_levels, hipStream_t st)
{
    if (!n_pat) return VLG_OK;
    if (v.bv_kind == kBvRrr63)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(int_backward_search_kernel<RrrBV>), dim3(grid_for(n_pat, 4096)), dim3(256), 0, st, v, d_blob, d_off, n_pat, d_l, d_r, d_stat_levels);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(int_backward_search_kernel<PlainBV>), dim3(grid_for(n_pat, 4096)), dim3(256), 0, st, v, d_blob, d_off, n_pat, d_l, d_r, d_stat_levels);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}

// the integer index in the sorted sweep (sigma <= 65534: the partition key is 16 bits wide)
template <class BV>
static vlg_status launch_int_locate_sweep_bv(const IntView& v, const uint64_t* d_l, const uint64_t* d_out_off, uint64_t n_pat, uint64_t total, uint32_t* d_out,
                                   uint64_t* val_a, uint64_t* val_b, uint16_t* key_a, uint16_t* key_b, void* temp, size_t temp_bytes, unsigned long long* d_counter,
                                   unsigned long long* d_stats, uint64_t tail_threshold, hipStream_t stream, LaunchTimer* timer, Block* member,
                                   uint32_t n_member_lists, uint64_t* rec, const std::function<vlg_status()>* while_first_step)
{
    if (v.sigma >= 0xFFFFu || v.n_levels < 1 || v.n > (1ull << 32)) return fail(VLG_E_INTERNAL, "integer index: not for the sorted sweep");
    auto grid_of = [](uint64_t n, uint32_t cap) { return dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, cap))); };
    SweepKernels K;
    K.n = v.n;
    K.sigma = (uint32_t)v.sigma;
    K.first = [&](uint64_t t0, uint64_t t1, uint64_t* val, uint16_t* key, void* out, unsigned long long* counter, const Block* mem, uint64_t* rc, bool ahead,
                  uint32_t* chunk_list) {
        launch_sweep_chunk_lists(d_out_off, n_pat, t0, t1, chunk_list, stream);
        const dim3 g = grid_of((t1 - t0 + 7) / 8, 8192);
        uint32_t* o = static_cast<uint32_t*>(out);
        if (mem && ahead) hipLaunchKernelGGL(HIP_KERNEL_NAME(int_sweep_first_kernel<BV, true, true>), g, dim3(256), 0, stream, v, d_l, d_out_off, n_pat, t0, t1, val, key, o, d_stats, counter, mem, rc, chunk_list);
        else if (mem) hipLaunchKernelGGL(HIP_KERNEL_NAME(int_sweep_first_kernel<BV, true, false>), g, dim3(256), 0, stream, v, d_l, d_out_off, n_pat, t0, t1, val, key, o, d_stats, counter, mem, rc, chunk_list);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(int_sweep_first_kernel<BV, false, false>), g, dim3(256), 0, stream, v, d_l, d_out_off, n_pat, t0, t1, val, key, o, d_stats, counter, mem, rc, chunk_list);
    };
    K.step = [&](uint64_t* val, uint16_t* key, uint64_t alive, uint32_t step, void* out, unsigned long long* counter, const Block* mem, uint64_t* rc, uint64_t t0, bool probed) {
        const dim3 g = grid_of(alive, 4096);
        uint32_t* o = static_cast<uint32_t*>(out);
        if (mem) hipLaunchKernelGGL(HIP_KERNEL_NAME(int_sweep_step_kernel<BV, true>), g, dim3(256), 0, stream, v, val, key, alive, step, o, d_stats, counter, mem, rc, t0, probed);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(int_sweep_step_kernel<BV, false>), g, dim3(256), 0, stream, v, val, key, alive, step, o, d_stats, counter, mem, rc, t0, probed);
    };
    K.tail = [&](void* out, uint64_t alive, uint32_t per_wave, const uint64_t* val, uint32_t step, uint64_t* rc, uint64_t t0, const Block* mem, uint32_t blocks) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(int_sweep_tail_kernel<BV>), dim3(blocks), dim3(256), 0, stream, v, static_cast<uint32_t*>(out), alive, per_wave, d_stats, val, step, rc, t0, mem);
    };
    return run_locate_sweep<uint32_t, false>(K, d_l, d_out_off, n_pat, total, d_out, val_a, val_b, key_a, key_b, temp, temp_bytes, d_counter, tail_threshold, stream, timer,
                                             member, n_member_lists, rec, while_first_step);
}

vlg_status launch_int_locate_sweep(const IntView& v, const uint64_t* d_l, const uint64_t* d_out_off, uint64_t n_pat, uint64_t total, uint32_t* d_out,
                                   uint64_t* val_a, uint64_t* val_b, uint16_t* key_a, uint16_t* key_b, void* temp, size_t temp_bytes, unsigned long long* d_counter,
                                   unsigned long long* d_stats, uint64_t tail_threshold, hipStream_t stream, LaunchTimer* timer, Block* member,
                                   uint32_t n_member_lists, uint64_t* rec, const std::function<vlg_status()>* while_first_step)
{
    return v.bv_kind == kBvRrr63
               ? launch_int_locate_sweep_bv<RrrBV>(v, d_l, d_out_off, n_pat, total, d_out, val_a, val_b, key_a, key_b, temp, temp_bytes, d_counter, d_stats, tail_threshold,
                                                   stream, timer, member, n_member_lists, rec, while_first_step)
               : launch_int_locate_sweep_bv<PlainBV>(v, d_l, d_out_off, n_pat, total, d_out, val_a, val_b, key_a, key_b, temp, temp_bytes, d_counter, d_stats, tail_threshold,
                                                     stream, timer, member, n_member_lists, rec, while_first_step);
}

vlg_status launch_int_locate(const IntView& v, uint32_t* d_io, uint64_t total, unsigned long long* d_stats, hipStream_t st)
{
    if (!total) return VLG_OK;
    const uint64_t target_waves = 256ull * 32 * 4;
    uint64_t per_wave = (total + target_waves - 1) / target_waves;
    per_wave = std::min<uint64_t>(std::max<uint64_t>(per_wave, 64 * 16), 1u << 20);
    const uint64_t waves = (total + per_wave - 1) / per_wave;
    if (v.bv_kind == kBvRrr63)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(int_locate_kernel<RrrBV>), dim3((uint32_t)((waves + 3) / 4)), dim3(256), 0, st, v, d_io, total, (uint32_t)per_wave, d_stats);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(int_locate_kernel<PlainBV>), dim3((uint32_t)((waves + 3) / 4)), dim3(256), 0, st, v, d_io, total, (uint32_t)per_wave, d_stats);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}

}  // namespace vlg
