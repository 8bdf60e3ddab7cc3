// Device primitives: one 256-bit super-block read = one bit-rank (+ the bit itself).
// Replaces rank_support_v<1,1>::rank (include/sdsl/rank_support_v.hpp:114-124) together with the
// "- bv_pos_rank(v)" of wt_pc::rank / inverse_select (include/sdsl/wt_pc.hpp:362-368, 389-397).
#pragma once
#include "common.hpp"
#include "rrr_code.hpp"

namespace vlg {

struct BlockRegs {
    uint4 a, c;                // a = data words 0..3, c = data words 4..6 and cnt (c.w)
};

// 32 contiguous, 32-byte aligned bytes per lane: two global_load_dwordx4.
__device__ __forceinline__ BlockRegs load_block(const Block* blocks, uint32_t b)
{
    const uint4* p = reinterpret_cast<const uint4*>(blocks + b);
    return BlockRegs{p[0], p[1]};
}

// Ones in the node before position (block * 224 + off), off in [0, 224), and the bit at that position.
// The sweep is bound by vector instructions, not by memory (round 3: 6.4*10^8 LF steps of 4.6 levels in 9.9 ms, ~95 instructions per
// level), so this is written for instruction count: the popcounts of the whole words in front of `off` are a chain of seven
// accumulating v_bcnt (the block's own count included), the word that holds `off` and the count in front of it are picked by the three
// bits of off / 32 with conditional moves, one masked popcount finishes -- 30 instructions where four 64-bit words masked with clamped
// shifts took 60.
__device__ __forceinline__ uint32_t block_rank_bit(const BlockRegs& r, uint32_t off, uint32_t& bit)
{
    const uint32_t w = off >> 5, b = off & 31u;
    const uint32_t p1 = __popc(r.a.x) + r.c.w, p2 = __popc(r.a.y) + p1, p3 = __popc(r.a.z) + p2, p4 = __popc(r.a.w) + p3,
                   p5 = __popc(r.c.x) + p4, p6 = __popc(r.c.y) + p5;
    const bool b0 = (w & 1u) != 0, b1 = (w & 2u) != 0, b2 = (w & 4u) != 0;
    const uint32_t x01 = b0 ? r.a.y : r.a.x, x23 = b0 ? r.a.w : r.a.z, x45 = b0 ? r.c.y : r.c.x;
    const uint32_t x03 = b1 ? x23 : x01, x47 = b1 ? r.c.z : x45;
    const uint32_t word = b2 ? x47 : x03;
    const uint32_t q01 = b0 ? p1 : r.c.w, q23 = b0 ? p3 : p2, q45 = b0 ? p5 : p4;
    const uint32_t q03 = b1 ? q23 : q01, q47 = b1 ? p6 : q45;
    const uint32_t before = b2 ? q47 : q03;
    bit = (word >> b) & 1u;
    return before + __popc(word & ((1u << b) - 1u));
}
__device__ __forceinline__ uint32_t block_rank(const BlockRegs& r, uint32_t off)
{
    uint32_t bit;
    return block_rank_bit(r, off, bit);
}
__device__ __forceinline__ uint32_t block_bit(const BlockRegs& r, uint32_t off)
{
    uint32_t bit;
    (void)block_rank_bit(r, off, bit);
    return bit;
}

// i / 224 and i % 224 for i < 2^37
__device__ __forceinline__ void split224(uint64_t i, uint32_t& blk, uint32_t& off)
{
    uint32_t h = (uint32_t)(i >> 5);
    blk = h / 7u;
    off = (uint32_t)(i - (uint64_t)blk * kBlockBits);
}

// rank1 of the first i bits of node `base`
__device__ __forceinline__ uint64_t node_rank1(const Block* blocks, uint32_t base, uint64_t i)
{
    uint32_t blk, off;
    split224(i, blk, off);
    BlockRegs r = load_block(blocks, base + blk);
    return block_rank(r, off);
}

// the sweep's member bit-vector (kernels.hip: sweep_element): is SA index i an element of the batch, and which -- its slot is the
// number of member bits before it
__device__ __forceinline__ bool member_probe(const Block* __restrict__ member, uint64_t i, uint32_t& slot)
{
    uint32_t blk, off, bit;
    split224(i, blk, off);
    const BlockRegs r = load_block(member, blk);
    slot = block_rank_bit(r, off, bit);
    return bit != 0;
}

// ---- bit-vector policies: how one node-relative (rank1, bit) pair is obtained -----------------------------------
// (View: IndexView, or the integer index's IntView -- whatever carries `blocks` / `rrr_hdr`, `rrr_stream`, `rrr_tables`)
// Plain: one 256-bit super-block read (K1).
struct PlainBV {
    struct Shared {};
    template <class View> static __device__ __forceinline__ void stage(Shared&, const View&) {}
    template <class View> static __device__ __forceinline__ void rank_bit(const View& iv, const Shared&, uint32_t base, uint64_t i, uint64_t& r1, uint32_t& bit)
    {
        uint32_t blk, off;
        split224(i, blk, off);
        BlockRegs r = load_block(iv.blocks, base + blk);
        r1 = block_rank_bit(r, off, bit);
    }
    // node-relative positions below 2^32 (n <= 2^32): the same in 32-bit arithmetic
    template <class View> static __device__ __forceinline__ void rank_bit(const View& iv, const Shared&, uint32_t base, uint32_t i, uint32_t& r1, uint32_t& bit)
    {
        const uint32_t blk = (i >> 5) / 7u, off = i - blk * kBlockBits;
        const BlockRegs r = load_block(iv.blocks, base + blk);
        r1 = block_rank_bit(r, off, bit);
    }
    template <class View> static __device__ __forceinline__ uint64_t rank(const View& iv, const Shared&, uint32_t base, uint64_t i)
    {
        return node_rank1(iv.blocks, base, i);
    }
};

// rrr-63: 32-byte header {ones before, offset word position, 32 x 6-bit classes} per 32 blocks of 63 bits, offsets in a bit stream
// (each super-block's offsets start on a word) -- the sizes of rrr_vector<63> (include/sdsl/rrr_vector.hpp:145-237; rank:
// :444-480), the offsets numbered by halves so that a rank inside a block is a fixed, short computation (rrr_code.hpp).
struct RrrBV {
    struct Shared {
        RrrTables t;
    };
    template <class View> static __device__ __forceinline__ void stage(Shared& s, const View& iv)
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(iv.rrr_tables);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&s.t);
        for (uint32_t i = threadIdx.x; i < sizeof(RrrTables) / 4; i += blockDim.x) dst[i] = src[i];
    }
    // ones in the first `want` bits of the addressed block, plus bit number `want` when asked for
    template <bool kBit, class View>
    static __device__ __forceinline__ void decode(const View& iv, const Shared& s, uint32_t base, uint64_t i, uint64_t& r1, uint32_t& bit)
    {
        const uint64_t sb = i / kRrrSuperBits;
        const uint32_t r = (uint32_t)(i - sb * kRrrSuperBits);
        const uint32_t blk = r / kRrrBlockBits, off = r - blk * kRrrBlockBits;
        const uint4 h0 = iv.rrr_hdr[2 * ((uint64_t)base + sb)], h1 = iv.rrr_hdr[2 * ((uint64_t)base + sb) + 1];
        uint32_t rank = h0.x;
        uint64_t ptr = (uint64_t)h0.y << 6;
        uint64_t c0 = (uint64_t)h0.z | ((uint64_t)h0.w << 32), c1 = (uint64_t)h1.x | ((uint64_t)h1.y << 32),
                 c2 = (uint64_t)h1.z | ((uint64_t)h1.w << 32);
        // classes in front of the block (rrr_vector.hpp:463-467): the 192 bits of classes are shifted past one class per step
        // (measured in round 3: unrolling the sum over the 31 fixed places with a predicate is SLOWER -- C5 locate 33.3 vs 28.8 ms;
        //  so are two classes per step through a 4096-entry pair table in LDS, 8 KiB more per workgroup -- 34.4 ms)
        uint32_t bits = 0;
        for (uint32_t j = 0; j < blk; ++j) {
            const uint32_t k = (uint32_t)c0 & 63u;
            rank += k;
            bits += s.t.space[k];
            c0 = (c0 >> 6) | (c1 << 58);
            c1 = (c1 >> 6) | (c2 << 58);
            c2 >>= 6;
        }
        ptr += bits;
        bit = 0;
        if (kBit || off) {
            const uint32_t k = (uint32_t)c0 & 63u;
            const uint32_t len = s.t.space[k];
            uint64_t o = 0;
            if (len) {
                const uint64_t w = ptr >> 6, sh = ptr & 63;
                o = iv.rrr_stream[w] >> sh;
                if (sh + len > 64) o |= iv.rrr_stream[w + 1] << (64 - sh);
                o &= (len == 64) ? ~0ull : ((1ull << len) - 1);
            }
            rank += rrr_dec63(s.t, k, o, off, bit);
        }
        r1 = rank;
    }
    template <class View> static __device__ __forceinline__ void rank_bit(const View& iv, const Shared& s, uint32_t base, uint64_t i, uint64_t& r1, uint32_t& bit)
    {
        decode<true>(iv, s, base, i, r1, bit);
    }
    template <class View> static __device__ __forceinline__ void rank_bit(const View& iv, const Shared& s, uint32_t base, uint32_t i, uint32_t& r1, uint32_t& bit)
    {
        uint64_t r;
        decode<true>(iv, s, base, i, r, bit);
        r1 = (uint32_t)r;
    }
    template <class View> static __device__ __forceinline__ uint64_t rank(const View& iv, const Shared& s, uint32_t base, uint64_t i)
    {
        uint64_t r1; uint32_t bit;
        decode<false>(iv, s, base, i, r1, bit);
        return r1;
    }
};

// ---- SA sampling policies: is SA index i sampled, and if so what is SA[i] --------------------------------------------------------
// SaOrder: _sa_order_sampling (include/sdsl/csa_sampling_strategy.hpp:64-112): every dens-th SA index, samples[i / dens].
template <typename sample_t>
struct SaOrderSampling {
    uint32_t dens, dmask, dshift;
    bool pow2;
    const sample_t* samples;
    __device__ __forceinline__ explicit SaOrderSampling(const IndexView& iv)
        : dens(iv.dens), dmask(iv.dens - 1), dshift(31 - __clz(iv.dens)), pow2((iv.dens & (iv.dens - 1)) == 0),
          samples(reinterpret_cast<const sample_t*>(iv.samples)) {}
    __device__ __forceinline__ bool probe(uint64_t i, uint64_t& value) const
    {
        const bool sampled = pow2 ? ((i & dmask) == 0) : (i % dens == 0);
        if (sampled) value = (uint64_t)samples[pow2 ? (i >> dshift) : (i / dens)];
        return sampled;
    }
};
// TextOrder: _text_order_sampling (csa_sampling_strategy.hpp:127-246): is_sampled(i) = marked[i] (:185-188), value =
// samples[rank_marked(i)] * dens (:191-194).  The mark and the rank come out of the same 32-byte super-block.
// (sample_t: 4-byte condensed values for n <= 2^32, 8-byte ones for an index with wide SA indices -- the width the index keeps its
// samples in, csa_sampling_strategy.hpp:127-246 is width-agnostic)
template <typename sample_t>
struct TextOrderSampling {
    uint32_t dens;
    const Block* marked;
    const sample_t* samples;
    __device__ __forceinline__ explicit TextOrderSampling(const IndexView& iv)
        : dens(iv.dens), marked(iv.marked), samples(reinterpret_cast<const sample_t*>(iv.samples)) {}
    __device__ __forceinline__ bool probe(uint64_t i, uint64_t& value) const
    {
        uint32_t blk, off;
        split224(i, blk, off);
        const BlockRegs r = load_block(marked, blk);
        uint32_t bit;
        const uint32_t rank = block_rank_bit(r, off, bit);
        if (bit) value = (uint64_t)samples[rank] * dens;
        return bit != 0;
    }
};

// Node table + C (+ whatever the bit-vector policy needs) staged in LDS by every workgroup that walks the tree.
template <class BV>
struct WalkLds {
    DNode nodes[kMaxNodes];
    uint64_t C[257];
    typename BV::Shared sh;
};

template <class BV>
__device__ __forceinline__ void stage_walk(WalkLds<BV>& s, const IndexView& iv)
{
    for (uint32_t i = threadIdx.x; i < iv.n_nodes; i += blockDim.x) s.nodes[i] = iv.nodes[i];
    for (uint32_t i = threadIdx.x; i <= iv.sigma; i += blockDim.x) s.C[i] = iv.C[i];
    BV::stage(s.sh, iv);
    __syncthreads();
}

}  // namespace vlg
