// Device primitives: one 256-bit super-block read = one bit-rank (+ the bit itself).
// Replaces rank_support_v<1,1>::rank (include/sdsl/rank_support_v.hpp:114-124) together with the
// "- bv_pos_rank(v)" of wt_pc::rank / inverse_select (include/sdsl/wt_pc.hpp:362-368, 389-397).
#pragma once
#include "common.hpp"

namespace vlg {

struct BlockRegs {
    uint64_t q0, q1, q2, q3;   // q3 = data word 6 (low half) | cnt (high half)
};

// 32 contiguous, 32-byte aligned bytes per lane: two global_load_dwordx4.
__device__ __forceinline__ BlockRegs load_block(const Block* blocks, uint32_t b)
{
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(blocks + b);
    ulonglong2 a = p[0], c = p[1];
    return BlockRegs{a.x, a.y, c.x, c.y};
}

__device__ __forceinline__ uint64_t mask_lo(int k)   // k may be <= 0 or >= 64
{
    return k <= 0 ? 0ull : (k >= 64 ? ~0ull : ((1ull << k) - 1ull));
}

// ones in the node before position (block*224 + off), off in [0,224)
__device__ __forceinline__ uint32_t block_rank(const BlockRegs& r, uint32_t off)
{
    int o = (int)off;
    uint32_t c = (uint32_t)(r.q3 >> 32);
    c += __popcll(r.q0 & mask_lo(o));
    c += __popcll(r.q1 & mask_lo(o - 64));
    c += __popcll(r.q2 & mask_lo(o - 128));
    c += __popcll((r.q3 & 0xFFFFFFFFull) & mask_lo(o - 192));
    return c;
}

__device__ __forceinline__ uint32_t block_bit(const BlockRegs& r, uint32_t off)
{
    uint64_t w = off < 64 ? r.q0 : (off < 128 ? r.q1 : (off < 192 ? r.q2 : r.q3));
    return (uint32_t)(w >> (off & 63)) & 1u;
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

// Node table + C + paths staged in LDS by every workgroup that walks the tree.
struct TreeLds {
    DNode nodes[kMaxNodes];
    uint64_t C[257];
};

__device__ __forceinline__ void stage_tree(TreeLds& s, const IndexView& iv)
{
    for (uint32_t i = threadIdx.x; i < iv.n_nodes; i += blockDim.x) s.nodes[i] = iv.nodes[i];
    for (uint32_t i = threadIdx.x; i <= iv.sigma; i += blockDim.x) s.C[i] = iv.C[i];
    __syncthreads();
}

}  // namespace vlg
