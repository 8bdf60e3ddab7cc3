// K1 batched bit-rank, K2 wavelet-tree rank / backward search, K3 locate (LF iteration).
#include <algorithm>
#include <cstring>
#include <string.h>
#include <vector>
#include "common.hpp"
#include "device_rank.hpp"
#include "kernels.hpp"
#include <rocprim/rocprim.hpp>

using namespace vlg;

// =============================================================================================
// K1: rank_support_v<1,1>::rank on a plain bit-vector (include/sdsl/rank_support_v.hpp:114-124)
// =============================================================================================
struct vlg_bitvector {
    Block* d_blocks = nullptr;
    uint64_t nbits = 0;
    uint64_t n_blocks = 0;
};

namespace {

// host words -> blocks. One thread per block: 7 data words + running count comes from a scan.
__global__ void bv_pack_kernel(const uint64_t* __restrict__ src, uint64_t src_words, uint64_t nbits, Block* __restrict__ blocks,
                               uint64_t n_blocks, uint32_t* __restrict__ pops)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t pc = 0;
        Block B;
#pragma unroll
        for (uint32_t w = 0; w < 7; ++w) {
            uint64_t bit = b * kBlockBits + 32u * w;
            uint32_t val = 0;
            if (bit < nbits) {
                uint64_t wi = bit >> 6;
                uint32_t s = (uint32_t)(bit & 63);
                uint64_t lo = src[wi] >> s;
                if (s > 32 && wi + 1 < src_words) lo |= src[wi + 1] << (64 - s);
                val = (uint32_t)lo;
                uint64_t left = nbits - bit;
                if (left < 32) val &= (1u << left) - 1u;
            }
            B.w[w] = val;
            pc += __popc(val);
        }
        B.cnt = 0;
        blocks[b] = B;
        pops[b] = pc;
    }
}

// single-workgroup-per-tile scan is plenty for a creation-time pass: serial over tiles of 1024 blocks
__global__ void bv_count_kernel(Block* __restrict__ blocks, const uint32_t* __restrict__ pops, uint64_t n_blocks)
{
    // one workgroup, chunked inclusive scan with a running carry
    __shared__ uint32_t s[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n_blocks; base += 1024) {
        uint64_t i = base + threadIdx.x;
        uint32_t v = i < n_blocks ? pops[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            uint32_t t = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_blocks) blocks[i].cnt = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) bitrank_kernel(const Block* __restrict__ blocks, const uint64_t* __restrict__ idx,
                                                       uint64_t* __restrict__ out, uint64_t count)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x)
        out[j] = node_rank1(blocks, 0, idx[j]);
}

}  // namespace

extern "C" vlg_status vlg_bitvector_create(const uint64_t* h_words, uint64_t nbits, vlg_bitvector** out)
{
    if (!out || (nbits && !h_words)) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VLG_E_NO_DEVICE, "no HIP device available");
    if (nbits >= (1ull << 32)) return fail(VLG_E_UNSUPPORTED, "stand-alone bit-vectors are limited to 2^32-1 bits (32-bit block counts)");
    vlg_bitvector* bv = new vlg_bitvector();
    bv->nbits = nbits;
    bv->n_blocks = nbits / kBlockBits + 1;
    uint64_t words = (nbits + 63) / 64;
    uint64_t* d_src = nullptr;
    uint32_t* d_pops = nullptr;
    auto run = [&]() -> vlg_status {
        VLG_HIP_TRY(hipMalloc((void**)&bv->d_blocks, bv->n_blocks * sizeof(Block)));
        VLG_HIP_TRY(hipMalloc((void**)&d_src, words * 8 + 8));
        VLG_HIP_TRY(hipMalloc((void**)&d_pops, bv->n_blocks * 4));
        if (words) VLG_HIP_TRY(hipMemcpy(d_src, h_words, words * 8, hipMemcpyHostToDevice));
        uint32_t grid = (uint32_t)std::min<uint64_t>((bv->n_blocks + 255) / 256, 8192);
        hipLaunchKernelGGL(bv_pack_kernel, dim3(grid), dim3(256), 0, nullptr, d_src, words, nbits, bv->d_blocks, bv->n_blocks, d_pops);
        hipLaunchKernelGGL(bv_count_kernel, dim3(1), dim3(1024), 0, nullptr, bv->d_blocks, d_pops, bv->n_blocks);
        VLG_HIP_TRY(hipGetLastError());
        VLG_HIP_TRY(hipDeviceSynchronize());
        return VLG_OK;
    };
    vlg_status st = run();
    if (d_src) (void)hipFree(d_src);
    if (d_pops) (void)hipFree(d_pops);
    if (st) { vlg_bitvector_destroy(bv); return st; }
    *out = bv;
    return VLG_OK;
}

extern "C" vlg_status vlg_bitvector_rank_batch(const vlg_bitvector* bv, const uint64_t* d_idx, uint64_t* d_out, uint64_t count, void* stream)
{
    if (!bv || (count && (!d_idx || !d_out))) return fail(VLG_E_INVALID, "null argument");
    if (!count) return VLG_OK;
    uint32_t grid = (uint32_t)std::min<uint64_t>((count + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(bitrank_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, bv->d_blocks, d_idx, d_out, count);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}

extern "C" uint64_t vlg_bitvector_hbm_bytes(const vlg_bitvector* bv) { return bv ? bv->n_blocks * sizeof(Block) : 0; }

extern "C" void vlg_bitvector_destroy(vlg_bitvector* bv)
{
    if (!bv) return;
    if (bv->d_blocks) (void)hipFree(bv->d_blocks);
    delete bv;
}

// =============================================================================================
// K6: rank on an H0-compressed bit-vector -- rrr_vector<63> / rank_support_rrr<1,63>
//     (include/sdsl/rrr_vector.hpp:444-480; coding include/sdsl/rrr_helper.hpp:304-320, 411-460).
// Blocks of 63 bits are stored as (class = popcount, offset = rank of the block among all blocks of that class in
// the combinatorial number system); 32 blocks form a super-block.  HBM layout, one aligned 32-byte header per
// super-block: { u32 ones before it, u32 bit position of its first offset, 32 x 6-bit classes }, offsets in a
// separate bit stream.  One rank = the header read + one read of <= 61 offset bits; the block is decoded on the fly
// against the binomial table C(n,k), n < 64, staged in LDS (32 KiB).
// =============================================================================================
struct vlg_rrr_bitvector {
    uint64_t nbits = 0, n_sb = 0, stream_words = 0;
    uint4* d_hdr = nullptr;          // 2 x uint4 per super-block
    uint64_t* d_stream = nullptr;
    uint64_t* d_binom = nullptr;     // [64][64]
};

namespace {

constexpr uint32_t kRrrBlock = 63, kRrrSuper = 32;

struct RrrLds {
    uint64_t binom[64][64];
    uint8_t space[64];
};

__device__ __forceinline__ uint32_t rrr_class(uint64_t c0, uint64_t c1, uint64_t c2, uint32_t j)
{
    uint32_t bit = 6u * j;                       // classes are packed little-endian into 192 bits
    uint32_t w = bit >> 6, o = bit & 63;
    uint64_t lo = w == 0 ? c0 : (w == 1 ? c1 : c2);
    uint64_t hi = w == 0 ? c1 : c2;
    uint64_t v = lo >> o;
    if (o > 58) v |= hi << (64 - o);
    return (uint32_t)v & 63u;
}

__global__ void __launch_bounds__(256) rrr_rank_kernel(const uint4* __restrict__ hdr, const uint64_t* __restrict__ stream,
                                                       const uint64_t* __restrict__ binom, const uint64_t* __restrict__ idx,
                                                       uint64_t* __restrict__ out, uint64_t count)
{
    __shared__ RrrLds s;
    for (uint32_t i = threadIdx.x; i < 64 * 64; i += blockDim.x) (&s.binom[0][0])[i] = binom[i];
    __syncthreads();
    if (threadIdx.x < 64) {                      // space_for_bt: bits of C(63,k), 0 for the two uniform classes
        uint64_t c = s.binom[63][threadIdx.x];
        s.space[threadIdx.x] = (c == 1) ? 0 : (uint8_t)(64 - __clzll((long long)c));
    }
    __syncthreads();
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = idx[q];
        const uint64_t sb = i / (kRrrBlock * kRrrSuper);
        const uint32_t r = (uint32_t)(i - sb * (kRrrBlock * kRrrSuper));
        const uint32_t blk = r / kRrrBlock, off = r - blk * kRrrBlock;
        const uint4 h0 = hdr[2 * sb], h1 = hdr[2 * sb + 1];
        uint64_t rank = h0.x;
        uint64_t ptr = h0.y;
        const uint64_t c0 = (uint64_t)h0.z | ((uint64_t)h0.w << 32), c1 = (uint64_t)h1.x | ((uint64_t)h1.y << 32),
                       c2 = (uint64_t)h1.z | ((uint64_t)h1.w << 32);
        for (uint32_t j = 0; j < blk; ++j) {     // rrr_vector.hpp:463-467
            uint32_t k = rrr_class(c0, c1, c2, j);
            rank += k;
            ptr += s.space[k];
        }
        if (off) {
            uint32_t k = rrr_class(c0, c1, c2, blk);
            const uint32_t len = s.space[k];
            uint64_t nr = 0;
            if (len) {
                const uint64_t w = ptr >> 6, o = ptr & 63;
                nr = stream[w] >> o;
                if (o + len > 64) nr |= stream[w + 1] << (64 - o);
                nr &= (len == 64) ? ~0ull : ((1ull << len) - 1);
            }
            // decode_popcount (rrr_helper.hpp:411-460): walk the block from bit 0, C(nn-1,k) decides each bit
            uint32_t ones = 0;
            if (k == kRrrBlock) ones = off;
            else if (k) {
                uint32_t nn = kRrrBlock;
                for (uint32_t b = 0; b < off && k; ++b, --nn) {
                    const uint64_t c = s.binom[nn - 1][k];
                    if (nr >= c) { nr -= c; --k; ++ones; }
                }
            }
            rank += ones;
        }
        out[q] = rank;
    }
}

}  // namespace

extern "C" vlg_status vlg_rrr_bitvector_create(const uint64_t* h_words, uint64_t nbits, vlg_rrr_bitvector** out)
{
    if (!out || (nbits && !h_words)) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VLG_E_NO_DEVICE, "no HIP device available");
    if (nbits >= (1ull << 32)) return fail(VLG_E_UNSUPPORTED, "stand-alone bit-vectors are limited to 2^32-1 bits");
    // binomial table (rrr_helper.hpp:173-207)
    std::vector<uint64_t> binom(64 * 64, 0);
    for (int nn = 0; nn < 64; ++nn) binom[nn * 64] = 1;
    for (int nn = 1; nn < 64; ++nn)
        for (int k = 1; k < 64; ++k) binom[nn * 64 + k] = (k == nn) ? 1 : (k > nn ? 0 : binom[(nn - 1) * 64 + k - 1] + binom[(nn - 1) * 64 + k]);
    auto space = [&](uint32_t k) -> uint32_t { uint64_t c = binom[63 * 64 + k]; return c == 1 ? 0 : 64 - (uint32_t)__builtin_clzll(c); };
    auto get = [&](uint64_t pos, uint32_t len) -> uint64_t {      // bits [pos, pos+len) of the input, zero beyond nbits
        uint64_t v = 0;
        for (uint32_t b = 0; b < len; ) {
            uint64_t p = pos + b;
            if (p >= nbits) break;
            uint64_t w = p >> 6, o = p & 63;
            uint32_t take = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(64 - o, len - b), nbits - p);
            uint64_t chunk = (h_words[w] >> o) & (take == 64 ? ~0ull : ((1ull << take) - 1));
            v |= chunk << b;
            b += take;
        }
        return v;
    };
    const uint64_t n_blocks = nbits / kRrrBlock + 1;
    const uint64_t n_sb = (n_blocks + kRrrSuper - 1) / kRrrSuper;
    std::vector<uint32_t> hdr(n_sb * 8, 0);
    std::vector<uint64_t> stream(1, 0);
    uint64_t sbits = 0, ones = 0;
    for (uint64_t sb = 0; sb < n_sb; ++sb) {
        uint32_t* H = &hdr[sb * 8];
        H[0] = (uint32_t)ones;
        H[1] = (uint32_t)sbits;
        uint64_t cls[3] = {0, 0, 0};
        for (uint32_t j = 0; j < kRrrSuper; ++j) {
            uint64_t bin = get((sb * kRrrSuper + j) * kRrrBlock, kRrrBlock);
            uint32_t k = (uint32_t)__builtin_popcountll(bin);
            uint32_t bit = 6 * j, w = bit >> 6, o = bit & 63;
            cls[w] |= (uint64_t)k << o;
            if (o > 58) cls[w + 1] |= (uint64_t)k >> (64 - o);
            ones += k;
            uint32_t len = space(k);
            if (len) {                                               // bin_to_nr: rrr_helper.hpp:304-320
                uint64_t nr = 0, b = bin;
                uint32_t kk = k, nn = kRrrBlock;
                while (b) { if (b & 1) { nr += binom[(nn - 1) * 64 + kk]; --kk; } b >>= 1; --nn; }
                if ((sbits + len + 127) / 64 >= stream.size()) stream.resize(stream.size() * 2 + 4, 0);
                uint64_t w2 = sbits >> 6, o2 = sbits & 63;
                stream[w2] |= nr << o2;
                if (o2 + len > 64) stream[w2 + 1] |= nr >> (64 - o2);
                sbits += len;
            }
        }
        H[2] = (uint32_t)cls[0]; H[3] = (uint32_t)(cls[0] >> 32);
        H[4] = (uint32_t)cls[1]; H[5] = (uint32_t)(cls[1] >> 32);
        H[6] = (uint32_t)cls[2]; H[7] = (uint32_t)(cls[2] >> 32);
    }
    if (sbits >= (1ull << 32)) return fail(VLG_E_UNSUPPORTED, "offset stream longer than 2^32 bits");
    vlg_rrr_bitvector* bv = new vlg_rrr_bitvector();
    bv->nbits = nbits; bv->n_sb = n_sb; bv->stream_words = sbits / 64 + 2;
    stream.resize(bv->stream_words, 0);
    auto run = [&]() -> vlg_status {
        VLG_HIP_TRY(hipMalloc((void**)&bv->d_hdr, n_sb * 32));
        VLG_HIP_TRY(hipMalloc((void**)&bv->d_stream, bv->stream_words * 8));
        VLG_HIP_TRY(hipMalloc((void**)&bv->d_binom, 64 * 64 * 8));
        VLG_HIP_TRY(hipMemcpy(bv->d_hdr, hdr.data(), n_sb * 32, hipMemcpyHostToDevice));
        VLG_HIP_TRY(hipMemcpy(bv->d_stream, stream.data(), bv->stream_words * 8, hipMemcpyHostToDevice));
        VLG_HIP_TRY(hipMemcpy(bv->d_binom, binom.data(), 64 * 64 * 8, hipMemcpyHostToDevice));
        return VLG_OK;
    };
    vlg_status st = run();
    if (st) { vlg_rrr_bitvector_destroy(bv); return st; }
    *out = bv;
    return VLG_OK;
}

extern "C" vlg_status vlg_rrr_bitvector_rank_batch(const vlg_rrr_bitvector* bv, const uint64_t* d_idx, uint64_t* d_out, uint64_t count,
                                                   void* stream)
{
    if (!bv || (count && (!d_idx || !d_out))) return fail(VLG_E_INVALID, "null argument");
    if (!count) return VLG_OK;
    uint32_t grid = (uint32_t)std::min<uint64_t>((count + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(rrr_rank_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, bv->d_hdr, bv->d_stream, bv->d_binom, d_idx, d_out,
                       count);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}

extern "C" uint64_t vlg_rrr_bitvector_hbm_bytes(const vlg_rrr_bitvector* bv) { return bv ? bv->n_sb * 32 + bv->stream_words * 8 : 0; }

extern "C" void vlg_rrr_bitvector_destroy(vlg_rrr_bitvector* bv)
{
    if (!bv) return;
    if (bv->d_hdr) (void)hipFree(bv->d_hdr);
    if (bv->d_stream) (void)hipFree(bv->d_stream);
    if (bv->d_binom) (void)hipFree(bv->d_binom);
    delete bv;
}

// =============================================================================================
// K2: wt_pc::rank (include/sdsl/wt_pc.hpp:350-373) and backward_search
//     (include/sdsl/suffix_array_algorithm.hpp:250-278, 305-326)
// =============================================================================================
namespace vlg {

// #c in BWT[0,i): walk the code of c from the root; one super-block read per level.
template <class BV>
__device__ __forceinline__ uint64_t wt_rank_dev(const IndexView& iv, const WalkLds<BV>& s, uint64_t path, uint64_t i, uint32_t& levels)
{
    uint32_t len = (uint32_t)(path >> 56);
    if (len == 0) return (iv.sigma == 1) ? i : 0;       // sigma==1: wt_pc.hpp:355-357 (the only symbol has an empty code)
    uint64_t res = i;
    uint32_t v = 0;
    for (uint32_t l = 0; l < len && res; ++l, path >>= 1) {       // "and result": wt_pc.hpp:361
        uint32_t bit = (uint32_t)(path & 1);
        uint64_t r1 = BV::rank(iv, s.sh, s.nodes[v].base, res);
        ++levels;
        res = bit ? r1 : res - r1;
        { const DNode nd = s.nodes[v]; v = (bit ? nd.child[1] : nd.child[0]) & ~kLeafFlag; }
    }
    return res;
}

}  // namespace vlg

namespace {

template <class BV>
__global__ void __launch_bounds__(256) wt_rank_kernel(IndexView iv, const uint64_t* __restrict__ pos, const uint8_t* __restrict__ sym,
                                                      uint64_t* __restrict__ out, uint64_t count)
{
    __shared__ WalkLds<BV> s;
    stage_walk(s, iv);
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x) {
        uint8_t c = sym[j];
        uint64_t path = iv.paths[c];
        uint32_t lv = 0;
        bool present = (c == 0) || iv.char2comp[c] != 0;          // c_to_leaf valid  (wt_pc.hpp:352-354)
        out[j] = present ? wt_rank_dev(iv, s, path, pos[j], lv) : 0;
    }
}

// one lane per pattern; the two ranks of a step are independent loads
template <class BV>
__global__ void __launch_bounds__(256) backward_search_kernel(IndexView iv, const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off,
                                                              uint64_t n_pat, uint64_t* __restrict__ out_l, uint64_t* __restrict__ out_r,
                                                              unsigned long long* __restrict__ stat_levels)
{
    __shared__ WalkLds<BV> s;
    stage_walk(s, iv);
    uint32_t levels = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pat; p += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t b = off[p], e = off[p + 1];
        uint64_t l = 0, r = iv.n - 1;
        while (b < e && r + 1 - l > 0) {                          // suffix_array_algorithm.hpp:319
            --e;
            uint8_t c = blob[e];
            uint32_t cc = iv.char2comp[c];
            if (cc == 0 && c > 0) { l = 1; r = 0; }               // :263-265
            else {
                uint64_t c_begin = s.C[cc];
                if (l == 0 && r + 1 == iv.n) { l = c_begin; r = s.C[cc + 1] - 1; }      // :268-270
                else {
                    uint64_t path = iv.paths[c];
                    uint64_t nl = c_begin + wt_rank_dev(iv, s, path, l, levels);          // :272
                    uint64_t nr = c_begin + wt_rank_dev(iv, s, path, r + 1, levels) - 1;  // :273
                    l = nl; r = nr;
                }
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

// Statistics counters: a wave-level sum, then ONE atomic per workgroup and counter -- a single word takes ~90 atomics per
// microsecond, so one per wave (16 k waves a launch) would cost every launch of the sweep a fifth of a millisecond.
template <int N>
__device__ __forceinline__ void block_add(unsigned long long (&v)[N], unsigned long long* const (&dst)[N])
{
    __shared__ unsigned long long s_acc[N];
    if (threadIdx.x < N) s_acc[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        unsigned long long x = v[k];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&s_acc[k], x);
    }
    __syncthreads();
    if (threadIdx.x < N && s_acc[threadIdx.x] && dst[threadIdx.x]) atomicAdd(dst[threadIdx.x], s_acc[threadIdx.x]);
}

// ---- member bit-vector: which SA indices are elements of the batch, and which (sweep_element explains what for) ----------------
// bit i = one of the lists [l[d], l[d] + off[d + 1] - off[d]), d < n_lists, holds SA index i; the lists are pairwise disjoint and
// ascend, and list d owns the slots [off[d], off[d + 1]) -- so the number of set bits before i IS the slot of index i.  Same 256-bit
// super-blocks as the wavelet tree (224 bits + the count before them), built block by block: no clearing pass, no atomics.
__global__ void __launch_bounds__(256) member_build_kernel(const uint64_t* __restrict__ l, const uint64_t* __restrict__ off, uint32_t n_lists,
                                                           uint64_t n_blocks, Block* __restrict__ out)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t B0 = b * kBlockBits, B1 = B0 + kBlockBits;
        uint32_t lo = 0, hi = n_lists;                       // first list that ends behind B0
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (l[mid] + (off[mid + 1] - off[mid]) <= B0) lo = mid + 1; else hi = mid;
        }
        uint32_t d = lo;
        Block B;
#pragma unroll
        for (uint32_t w = 0; w < 7; ++w) B.w[w] = 0;
        B.cnt = (uint32_t)(d < n_lists ? off[d] + (B0 > l[d] ? B0 - l[d] : 0) : off[n_lists]);
        for (; d < n_lists && l[d] < B1; ++d) {
            const uint64_t end = l[d] + (off[d + 1] - off[d]);
            const uint32_t a = (uint32_t)((l[d] > B0 ? l[d] : B0) - B0), e = (uint32_t)((end < B1 ? end : B1) - B0);
#pragma unroll
            for (uint32_t w = 0; w < 7; ++w) {
                const uint32_t x = a > 32 * w ? a : 32 * w, y = e < 32 * w + 32 ? e : 32 * w + 32;
                if (x < y) B.w[w] |= (y - x == 32 ? ~0u : ((1u << (y - x)) - 1u)) << (x - 32 * w);
            }
        }
        out[b] = B;
    }
}

// =============================================================================================
// K3: csa[i] = LF iteration to the next sampled SA index (include/sdsl/csa_wt.hpp:335-348,
//     LF = C[c] + inverse_select(i): suffix_array_helper.hpp:336-349, wt_pc.hpp:385-402).
//
// io[t] holds the SA index on entry and the text position on exit (in place).
// Work is dealt to lanes, not to waves: a wave owns a contiguous slice of io[] and every lane that
// finishes an occurrence immediately pulls the next one of the slice (ballot + prefix popcount), so
// all 64 lanes issue one 32-byte super-block read per iteration whatever the (geometric) number
// of LF steps and whatever the code lengths.
// =============================================================================================
// kTail: the same walk for the stragglers of the sorted sweep (K3s below): the elements are val[] = slot << kShift | SA index, they
// have walked `step` steps already, positions go to out[slot] -- or to rec[slot0 + slot] when LF steps are shared, and then a walk
// also ends on the first index that is an element of the batch itself (sweep_element explains the records and `member`).
// kWide: SA indices need 33 bits and the samples are 64-bit words (n > 2^32, or VLG_FORCE_POS64); the positions written may still be
// 32-bit (pos_t) when the text has at most 2^32 characters -- only the tail mode can split the two, the in-place mode keeps the SA
// index in io[] itself.
template <typename pos_t, class BV, bool kTail = false, bool kWide = (sizeof(pos_t) == 8), bool kTextOrder = false>
__global__ void __launch_bounds__(256) locate_kernel(IndexView iv, pos_t* __restrict__ io, uint64_t total, uint32_t per_wave,
                                                     unsigned long long* __restrict__ stats /* [2]: lf steps, levels */,
                                                     const uint64_t* __restrict__ val = nullptr, uint32_t step = 0,
                                                     uint64_t* __restrict__ rec = nullptr, uint64_t slot0 = 0,
                                                     const Block* __restrict__ member = nullptr)
{
    static_assert(kTail || kWide == (sizeof(pos_t) == 8), "in place, io[] holds the SA index: its width is the index width");
    using sample_t = typename std::conditional<kWide, uint64_t, uint32_t>::type;
    constexpr uint32_t kShift = kWide ? 33 : 32;
    constexpr uint64_t kPosMask = (1ull << kShift) - 1;
    __shared__ WalkLds<BV> s;
    stage_walk(s, iv);
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t next = wave * per_wave;                       // wave-uniform cursor into the slice
    const uint64_t slice_end = next + per_wave < total ? next + per_wave : total;
    using Sampling = typename std::conditional<kTextOrder, TextOrderSampling<sample_t>, SaOrderSampling<sample_t>>::type;
    const Sampling sampling(iv);

    uint64_t t = 0;          // slot being worked on
    uint64_t i = 0;          // SA index at the root, node-relative index below it
    uint32_t v = 0, off = 0;
    bool active = false, need = true;
    uint32_t n_lf = 0, n_lv = 0;
    for (;;) {
        // ---- refill ---------------------------------------------------------------------------
        unsigned long long m = __ballot(need);
        if (m) {
            uint32_t before = __popcll(m & ((1ull << lane) - 1ull));
            if (need) {
                uint64_t cand = next + before;
                if (cand < slice_end) {
                    if (kTail) { const uint64_t e = val[cand]; t = e >> kShift; i = e & kPosMask; off = step; }
                    else { t = cand; i = io[cand]; off = 0; }
                    v = 0;
                    active = true;
                }
                else active = false;
                need = false;
            }
            next += __popcll(m);
        }
        if (!__any(active)) break;
        if (active) {
            uint64_t sv = 0;
            uint32_t owner = 0;
            if (v == 0 && sampling.probe(i, sv)) {         // csa_sampling_strategy.hpp:102-111 / :185-194
                uint64_t r = sv + off;
                if (r >= iv.n) r -= iv.n;                  // csa_wt.hpp:343-347
                if (kTail && rec) rec[slot0 + t] = r;
                else io[t] = (pos_t)r;
                need = true;
                active = false;
            } else if (kTail && member && v == 0 && off != 0 && member_probe(member, i, owner)) {
                // this index is where element `owner` started: the rest of the walk is that element's (sweep_element)
                const uint64_t delta = off;
                const uint64_t ro = rec[owner];
                uint64_t r;
                if (ro == ~0ull) r = (delta << kShift) | owner;               // still walking: follow it
                else if ((ro >> kShift) == 0) r = ro + delta;                  // its position is known
                else r = ro + (delta << kShift);                               // it follows someone itself: follow that one
                rec[slot0 + t] = r;
                need = true;
                active = false;
            } else if (iv.sigma == 1) {                    // degenerate: only the sentinel exists
                i = 0; ++off;
            } else {
                // one level of inverse_select: the bit and the rank come from the same block
                const DNode nd = s.nodes[v];
                uint32_t bit;
                uint64_t r1;
                BV::rank_bit(iv, s.sh, nd.base, i, r1, bit);
                ++n_lv;
                uint64_t ni = bit ? r1 : i - r1;
                uint32_t ch = bit ? nd.child[1] : nd.child[0];      // (a select, not an indexed read: the node stays in registers)
                if (ch & kLeafFlag) {                      // reached the symbol: LF = C[c] + rank
                    i = s.C[ch & ~kLeafFlag] + ni;
                    v = 0;
                    ++off;
                    ++n_lf;
                } else {
                    i = ni;
                    v = ch;
                }
            }
        }
    }
    if (stats) {
        unsigned long long v[2] = {n_lf, n_lv};
        unsigned long long* const dst[2] = {&stats[0], &stats[1]};
        block_add<2>(v, dst);
    }
}

// io[off[p] + j] = l[p] + j : the SA indices of every occurrence (input of locate_kernel), plus seg[]
template <typename pos_t>
__global__ void expand_kernel(const uint64_t* __restrict__ l, const uint64_t* __restrict__ out_off, uint64_t n_pat, uint64_t total,
                              pos_t* __restrict__ io, uint32_t* __restrict__ seg)
{
    // tile of 256 consecutive slots per workgroup iteration; first lane finds the segment by binary search
    __shared__ uint64_t s_first;
    for (uint64_t base = (uint64_t)blockIdx.x * 256; base < total; base += (uint64_t)gridDim.x * 256) {
        if (threadIdx.x == 0) {
            uint64_t lo = 0, hi = n_pat;                   // last p with out_off[p] <= base
            while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (out_off[mid] <= base) lo = mid; else hi = mid; }
            s_first = lo;
        }
        __syncthreads();
        uint64_t t = base + threadIdx.x;
        if (t < total) {
            uint64_t p = s_first;
            while (out_off[p + 1] <= t) ++p;               // empty segments are skipped too
            io[t] = (pos_t)(l[p] + (t - out_off[p]));
            if (seg) seg[t] = (uint32_t)p;
        }
        __syncthreads();
    }
}

// =============================================================================================
// K3s: locate as a synchronous SORTED SWEEP (n <= 2^32).
//
// All occurrences advance one LF step per round.  The round's elements are kept in ascending SA-index order:
// LF restricted to one symbol is monotone (LF(i) = C[c] + rank_c(i)), so after a round a STABLE partition of the
// elements by the symbol they read restores the order -- no comparison sort.  With ascending positions the 64
// lanes of a wave read the same or neighbouring super-blocks at every level of the tree (coalesced loads instead
// of 64 unrelated 64-byte requests), which is what lifts the kernel off the random-access wall of HBM
// (tools/k1_bench.py: ~50 G random ranks/s vs ~280 G sorted ranks/s).
// An element leaves the sweep when it reaches a sampled SA index (csa_sampling_strategy.hpp:102-111).
// val = slot << 32 | position;  key = comp of the symbol read, or sigma for "finished".
// =============================================================================================
// val = slot << kShift | position: 32 + 32 bits for n <= 2^32, 31 + 33 bits beyond (then a sweep covers at most 2^31 occurrences
// [t0, t1) at a time and slots are relative to t0).
template <uint32_t kShift>
__global__ void sweep_init_kernel(const uint64_t* __restrict__ l, const uint64_t* __restrict__ out_off, uint64_t n_pat, uint64_t t0, uint64_t total,
                                  uint64_t* __restrict__ val)
{
    constexpr uint32_t kPer = 8;                           // elements per thread: one list lookup per 2048 elements
    __shared__ uint64_t s_first;
    for (uint64_t base = t0 + (uint64_t)blockIdx.x * 256 * kPer; base < total; base += (uint64_t)gridDim.x * 256 * kPer) {
        if (threadIdx.x == 0) {
            uint64_t lo = 0, hi = n_pat;
            while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (out_off[mid] <= base) lo = mid; else hi = mid; }
            s_first = lo;
        }
        __syncthreads();
        uint64_t p = s_first;
#pragma unroll
        for (uint32_t i = 0; i < kPer; ++i) {
            const uint64_t t = base + i * 256 + threadIdx.x;
            if (t < total) {
                while (out_off[p + 1] <= t) ++p;
                val[t - t0] = ((t - t0) << kShift) | (l[p] + (t - out_off[p]));
            }
        }
        __syncthreads();
    }
}

// The lists that hold the elements [base, end) of a workgroup's turn, staged in LDS: looking an element's list up (whose interval it
// belongs to, where that starts) is a chain of dependent reads in front of everything else the element does, and the next element's
// chain starts where this one's ended -- out of LDS it costs tens of cycles instead of L2 round trips.  A turn whose elements spread
// over more than kListStage lists (lists of a few elements each) walks the global arrays as before.  VLG_STAGE_LISTS=0: never staged.
#ifndef VLG_STAGE_LISTS
#define VLG_STAGE_LISTS 1
#endif
constexpr bool kStageLists = VLG_STAGE_LISTS != 0;
constexpr uint32_t kListStage = 256;
struct ListStage { uint64_t off[kListStage + 1]; uint64_t l[kListStage]; };
__device__ __forceinline__ bool stage_lists(ListStage& ls, const uint64_t* __restrict__ out_off, const uint64_t* __restrict__ l, uint64_t n_pat,
                                            uint64_t first, uint64_t end)
{
    for (uint32_t j = threadIdx.x; j <= kListStage; j += blockDim.x) {
        const uint64_t p = first + j;
        ls.off[j] = out_off[p < n_pat ? p : n_pat];
        if (j < kListStage) ls.l[j] = l[p < n_pat ? p : n_pat - 1];
    }
    __syncthreads();
    return ls.off[kListStage] >= end;                      // (the same word in every thread: the branch on it is uniform)
}

// The suffix array itself resident in HBM (SA-order sampling with density 1: csa_wt<wt_huff<>, 1, .>, 4 B x n -- 4.3 GB for a 1 GiB text,
// what 288 GB of HBM afford and the reference's CPU index does not): locate is a copy of the SA intervals, csa[i] = sample[i]
// (csa_wt.hpp:335-348 with zero LF steps).  Same list lookup as sweep_init_kernel; reads and writes are coalesced inside a list.
template <typename pos_t, typename sample_t>
__global__ void __launch_bounds__(256) sa_dense_copy_kernel(const sample_t* __restrict__ sa, const uint64_t* __restrict__ l, const uint64_t* __restrict__ out_off,
                                                            uint64_t n_pat, uint64_t total, pos_t* __restrict__ out)
{
    constexpr uint32_t kPer = 8;
    __shared__ uint64_t s_first;
    __shared__ ListStage s_lists;
    for (uint64_t base = (uint64_t)blockIdx.x * 256 * kPer; base < total; base += (uint64_t)gridDim.x * 256 * kPer) {
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t lo = 0, hi = n_pat;
            while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (out_off[mid] <= base) lo = mid; else hi = mid; }
            s_first = lo;
        }
        __syncthreads();
        uint64_t p = s_first;
        const uint64_t end = base + 256 * kPer < total ? base + 256 * kPer : total;
        if (kStageLists && stage_lists(s_lists, out_off, l, n_pat, p, end)) {
            uint32_t q = 0;
#pragma unroll
            for (uint32_t i = 0; i < kPer; ++i) {
                const uint64_t t = base + i * 256 + threadIdx.x;
                if (t < total) {
                    while (s_lists.off[q + 1] <= t) ++q;
                    out[t] = (pos_t)sa[s_lists.l[q] + (t - s_lists.off[q])];
                }
            }
            continue;
        }
#pragma unroll
        for (uint32_t i = 0; i < kPer; ++i) {
            const uint64_t t = base + i * 256 + threadIdx.x;
            if (t < total) {
                while (out_off[p + 1] <= t) ++p;
                out[t] = (pos_t)sa[l[p] + (t - out_off[p])];
            }
        }
    }
}

// kTrail: LF steps are shared inside the batch.  An LF walk from SA index i visits the indices of the text positions SA[i] - 1,
// SA[i] - 2, ...; when it stands on an index that is ITSELF an element of the batch (the start of another occurrence's walk: text
// position SA[i] - k is an occurrence too) the rest of the walk is that element's walk, so it stops there and records
// (that element, k): csa[i] = csa[LF^k(i)] + k (csa_wt.hpp:335-348 applied to a value another lane computes).  Which indices are
// elements is known before the sweep starts -- the batch's lists are SA intervals -- and kept as a rank-enabled bit-vector over
// the SA indices in the usual 256-bit super-blocks (`member`, member_build_kernel): one 32-byte read says whether index i is an
// element AND which one (its slot = the number of member indices before it: the lists lie in SA order in the slot space).
// This replaces the table of round 2 / 3 (8 bytes per text position, written and read at random by every step, told apart by
// generation stamps) with n / 7 bytes that are only read; a walk also stops wherever it can, not only where another one has passed
// EARLIER, so every non-member index is visited by at most one walk.
// rec[slot]: a position (high bits 0), or delta << kShift | slot of the element it follows; ~0 while the element is still walking.
// slot0 = first slot of the sweep.
// one element of one round: v64 = its word (slot << kShift | SA index), e = its place in val / key
// kAhead (round 0): the index an element steps ONTO is looked up at once -- six in ten elements of a dense batch stand next to
// another occurrence in the text -- so that they leave the sweep before its largest partition instead of after it; `probed` tells
// round 1 that its elements have been looked up already.
template <class BV, typename pos_t, bool kTrail, bool kWide, bool kFirst = false, bool kAhead = false, class Sampling>
__device__ __forceinline__ void sweep_element(const IndexView& iv, const WalkLds<BV>& s, const Sampling& sampling, uint64_t e, uint64_t v64,
                                              uint64_t* __restrict__ val, uint16_t* __restrict__ key, uint32_t step, pos_t* __restrict__ out,
                                              const Block* __restrict__ member, uint64_t* __restrict__ rec, uint64_t slot0,
                                              uint32_t& n_lv, uint32_t& n_lf, uint32_t& n_fin, bool probed = false, uint8_t* __restrict__ front = nullptr)
{
    constexpr uint32_t kShift = kWide ? 33 : 32;
    constexpr uint64_t kPosMask = (1ull << kShift) - 1;
    uint64_t i = v64 & kPosMask;
    uint8_t in_front = 0xFF;                             // (kFirst: the symbol in front of an element that stops on its first step)
    uint64_t sv = 0;
    uint32_t owner = 0;
    if (sampling.probe(i, sv)) {
        uint64_t r = sv + step;
        if (r >= iv.n) r -= iv.n;                        // csa_wt.hpp:343-347
        if (kTrail) rec[slot0 + (v64 >> kShift)] = r;
        else out[v64 >> kShift] = (pos_t)r;
        key[e] = (uint16_t)iv.sigma;
        ++n_fin;
    } else if (kTrail && !kFirst && !probed && member_probe(member, i, owner)) {
        // (round 0: every element stands on its own index.)  Index i is where element `owner` started: same text trail, `step`
        // positions further left
        const uint64_t delta = step;
        const uint64_t ro = rec[owner];
        uint64_t r;
        if (ro == ~0ull) r = (delta << kShift) | owner;                       // still walking: follow it
        else if ((ro >> kShift) == 0) r = ro + delta;                          // its position is known
        else r = ro + (delta << kShift);                                       // it follows someone itself: follow that one
        rec[slot0 + (v64 >> kShift)] = r;
        key[e] = (uint16_t)iv.sigma;
        ++n_fin;
    } else {
        if (kTrail && kFirst) rec[slot0 + (v64 >> kShift)] = ~0ull;               // still walking (no pass clears the records beforehand)
        uint32_t v = 0, c;
        using walk_t = typename std::conditional<kWide, uint64_t, uint32_t>::type;      // node-relative positions: < n
        walk_t pos = (walk_t)i;
        for (;;) {                                       // inverse_select: wt_pc.hpp:385-402
            const DNode nd = s.nodes[v];
            uint32_t bit;
            walk_t r1;
            BV::rank_bit(iv, s.sh, nd.base, pos, r1, bit);
            ++n_lv;
            pos = bit ? r1 : pos - r1;
            uint32_t ch = bit ? nd.child[1] : nd.child[0];      // (a select, not an indexed read: the node stays in registers)
            if (ch & kLeafFlag) { c = ch & ~kLeafFlag; break; }
            v = ch;
        }
        ++n_lf;
        const uint64_t j = s.C[c] + pos;                                      // LF: suffix_array_helper.hpp:341-348
        if (kTrail && kAhead && member_probe(member, j, owner)) {
            // (the owner is in its own round 0 right now: its record reads "still walking" or is not written yet -- either way this
            // element follows it)
            rec[slot0 + (v64 >> kShift)] = ((uint64_t)(step + 1) << kShift) | owner;
            key[e] = (uint16_t)iv.sigma;
            ++n_fin;
            in_front = (uint8_t)c;
        } else {
            val[e] = (v64 & ~kPosMask) | j;
            key[e] = (uint16_t)c;
        }
    }
    if (kTrail && kFirst && front) front[slot0 + (v64 >> kShift)] = in_front;
}

#ifndef VLG_SWEEP_PAIRS
#define VLG_SWEEP_PAIRS 1
#endif
constexpr bool kSweepPairs = VLG_SWEEP_PAIRS != 0;
template <class BV, typename pos_t, bool kTrail, bool kWide, bool kTextOrder>
__global__ void __launch_bounds__(256) sweep_step_kernel(IndexView iv, uint64_t* __restrict__ val, uint16_t* __restrict__ key, uint64_t count,
                                                         uint32_t step, pos_t* __restrict__ out,
                                                         unsigned long long* __restrict__ stats /* lf, levels */,
                                                         unsigned long long* __restrict__ n_done, const Block* __restrict__ member,
                                                         uint64_t* __restrict__ rec, uint64_t slot0, bool probed)
{
    __shared__ WalkLds<BV> s;
    stage_walk(s, iv);
    using sample_t = typename std::conditional<kWide, uint64_t, uint32_t>::type;
    using Sampling = typename std::conditional<kTextOrder, TextOrderSampling<sample_t>, SaOrderSampling<sample_t>>::type;
    const Sampling sampling(iv);
    uint32_t n_lv = 0, n_lf = 0, n_fin = 0;
    // (pairs of elements as in round 0 -- sweep_first_pair -- were measured here too, on C4: nothing; the later rounds' elements are sparse)
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (uint64_t)gridDim.x * blockDim.x)
        sweep_element<BV, pos_t, kTrail, kWide>(iv, s, sampling, e, val[e], val, key, step, out, member, rec, slot0, n_lv, n_lf, n_fin, probed);
    unsigned long long v[3] = {n_lf, n_lv, n_fin};
    unsigned long long* const dst[3] = {&stats[0], &stats[1], n_done};
    block_add<3>(v, dst);
}

// Two elements of round 0 side by side (plain bit-vectors): every tree level and the look-ahead probe of both are loaded before either
// is used, so a lane has two dependent chains in flight instead of one (the kernel runs at full occupancy on 46 registers and waits
// ~1 us per wave-wide dependent load: more waves cannot come, more loads per wave can).  Same outcome as two sweep_element calls.
template <typename pos_t, bool kTrail, bool kWide, bool kAhead, class Sampling>
__device__ __forceinline__ void sweep_first_pair(const IndexView& iv, const WalkLds<PlainBV>& s, const Sampling& sampling, bool onA, uint64_t eA, uint64_t wA,
                                                 bool onB, uint64_t eB, uint64_t wB, uint64_t* __restrict__ val, uint16_t* __restrict__ key,
                                                 pos_t* __restrict__ out, const Block* __restrict__ member, uint64_t* __restrict__ rec, uint64_t slot0,
                                                 uint32_t& n_lv, uint32_t& n_lf, uint32_t& n_fin, uint8_t* __restrict__ front)
{
    constexpr uint32_t kShift = kWide ? 33 : 32;
    constexpr uint64_t kPosMask = (1ull << kShift) - 1;
    using walk_t = typename std::conditional<kWide, uint64_t, uint32_t>::type;
    const uint64_t iA = wA & kPosMask, iB = wB & kPosMask;
    const bool hadA = onA, hadB = onB;
    uint64_t sv = 0;
    if (onA && sampling.probe(iA, sv)) {                                     // csa_wt.hpp:343-347 (round 0: no steps yet)
        if (kTrail) rec[slot0 + (wA >> kShift)] = sv; else out[wA >> kShift] = (pos_t)sv;
        key[eA] = (uint16_t)iv.sigma; ++n_fin; onA = false;
    }
    if (onB && sampling.probe(iB, sv)) {
        if (kTrail) rec[slot0 + (wB >> kShift)] = sv; else out[wB >> kShift] = (pos_t)sv;
        key[eB] = (uint16_t)iv.sigma; ++n_fin; onB = false;
    }
    if (kTrail) {                                                            // still walking (no pass clears the records beforehand)
        if (onA) rec[slot0 + (wA >> kShift)] = ~0ull;
        if (onB) rec[slot0 + (wB >> kShift)] = ~0ull;
    }
    // inverse_select of both (wt_pc.hpp:385-402), level by level
    uint32_t vA = 0, vB = 0, cA = 0, cB = 0;
    walk_t pA = (walk_t)iA, pB = (walk_t)iB;
    bool a = onA, b = onB;
    while (a || b) {
        const DNode ndA = s.nodes[vA], ndB = s.nodes[vB];
        uint32_t blkA, offA, blkB, offB;
        split224((uint64_t)pA, blkA, offA);
        split224((uint64_t)pB, blkB, offB);
        BlockRegs rA, rB;
        if (a) rA = load_block(iv.blocks, ndA.base + blkA);
        if (b) rB = load_block(iv.blocks, ndB.base + blkB);
        if (a) {
            uint32_t bit;
            const walk_t r1 = (walk_t)block_rank_bit(rA, offA, bit);
            ++n_lv;
            pA = bit ? r1 : pA - r1;
            const uint32_t ch = bit ? ndA.child[1] : ndA.child[0];
            if (ch & kLeafFlag) { cA = ch & ~kLeafFlag; a = false; } else vA = ch;
        }
        if (b) {
            uint32_t bit;
            const walk_t r1 = (walk_t)block_rank_bit(rB, offB, bit);
            ++n_lv;
            pB = bit ? r1 : pB - r1;
            const uint32_t ch = bit ? ndB.child[1] : ndB.child[0];
            if (ch & kLeafFlag) { cB = ch & ~kLeafFlag; b = false; } else vB = ch;
        }
    }
    const uint64_t jA = s.C[cA] + (uint64_t)pA, jB = s.C[cB] + (uint64_t)pB;  // LF: suffix_array_helper.hpp:341-348
    n_lf += (onA ? 1u : 0u) + (onB ? 1u : 0u);
    bool stopA = false, stopB = false;
    uint32_t ownA = 0, ownB = 0;
    if (kTrail && kAhead) {                                                  // the look-ahead probes of both, their blocks in flight together
        uint32_t blkA, offA, blkB, offB, bit;
        split224(jA, blkA, offA);
        split224(jB, blkB, offB);
        BlockRegs rA, rB;
        if (onA) rA = load_block(member, blkA);
        if (onB) rB = load_block(member, blkB);
        if (onA) { ownA = block_rank_bit(rA, offA, bit); stopA = bit != 0; }
        if (onB) { ownB = block_rank_bit(rB, offB, bit); stopB = bit != 0; }
    }
    if (onA) {
        if (stopA) { rec[slot0 + (wA >> kShift)] = (1ull << kShift) | ownA; key[eA] = (uint16_t)iv.sigma; ++n_fin; }
        else { val[eA] = (wA & ~kPosMask) | jA; key[eA] = (uint16_t)cA; }
    }
    if (onB) {
        if (stopB) { rec[slot0 + (wB >> kShift)] = (1ull << kShift) | ownB; key[eB] = (uint16_t)iv.sigma; ++n_fin; }
        else { val[eB] = (wB & ~kPosMask) | jB; key[eB] = (uint16_t)cB; }
    }
    if (kTrail && front) {                                                   // the symbol in front of an element that stopped on its first step
        if (hadA) front[slot0 + (wA >> kShift)] = (onA && stopA) ? (uint8_t)cA : (uint8_t)0xFF;
        if (hadB) front[slot0 + (wB >> kShift)] = (onB && stopB) ? (uint8_t)cB : (uint8_t)0xFF;
    }
}

__global__ void __launch_bounds__(256) sweep_chunk_lists_kernel(const uint64_t* __restrict__ out_off, uint64_t n_pat, uint64_t t0, uint64_t t1,
                                                                uint32_t* __restrict__ chunk_list)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t base = t0 + c * kSweepChunk;
    if (base >= t1) return;
    uint64_t lo = 0, hi = n_pat;                                                 // last list with out_off[p] <= base
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (out_off[mid] <= base) lo = mid; else hi = mid; }
    chunk_list[c] = (uint32_t)lo;
}

// Round 0 without the pass that would write the elements' words first and the read that would fetch them again: an element's word
// follows from its place -- slot t - t0, SA index l[list] + (t - first slot of the list) -- so a workgroup looks its list up once per
// 2048 consecutive elements (as sweep_init_kernel does) and walks them at once.
template <class BV, typename pos_t, bool kTrail, bool kWide, bool kTextOrder, bool kAhead>
__global__ void __launch_bounds__(256) sweep_first_kernel(IndexView iv, const uint64_t* __restrict__ l, const uint64_t* __restrict__ out_off, uint64_t n_pat,
                                                          uint64_t t0, uint64_t total, uint64_t* __restrict__ val, uint16_t* __restrict__ key,
                                                          pos_t* __restrict__ out, unsigned long long* __restrict__ stats,
                                                          unsigned long long* __restrict__ n_done, const Block* __restrict__ member,
                                                          uint64_t* __restrict__ rec, const uint32_t* __restrict__ chunk_list, uint8_t* __restrict__ front)
{
    __shared__ WalkLds<BV> s;
    __shared__ ListStage s_lists;
    stage_walk(s, iv);
    using sample_t = typename std::conditional<kWide, uint64_t, uint32_t>::type;
    using Sampling = typename std::conditional<kTextOrder, TextOrderSampling<sample_t>, SaOrderSampling<sample_t>>::type;
    const Sampling sampling(iv);
    constexpr uint32_t kShift = kWide ? 33 : 32;
    constexpr uint32_t kPer = kSweepChunk / 256;
    uint32_t n_lv = 0, n_lf = 0, n_fin = 0;
    for (uint64_t base = t0 + (uint64_t)blockIdx.x * kSweepChunk; base < total; base += (uint64_t)gridDim.x * kSweepChunk) {
        __syncthreads();                                                         // (the lists staged for the previous turn have been read)
        uint64_t p = chunk_list[(base - t0) / kSweepChunk];                      // the list of the chunk's first element (sweep_chunk_lists_kernel)
        const uint64_t end = base + 256 * kPer < total ? base + 256 * kPer : total;
        const bool staged = kStageLists && stage_lists(s_lists, out_off, l, n_pat, p, end);
        uint32_t q = 0;
        auto word_of = [&](uint64_t t) -> uint64_t {                            // slot << kShift | SA index of element t (t ascends from call to call)
            uint64_t sai;
            if (staged) {
                while (s_lists.off[q + 1] <= t) ++q;
                sai = s_lists.l[q] + (t - s_lists.off[q]);
            } else {
                while (out_off[p + 1] <= t) ++p;
                sai = l[p] + (t - out_off[p]);
            }
            return ((t - t0) << kShift) | sai;
        };
        // (measured, round 4: C4 -- 33-bit indices, a deeper tree -- locate 98 -> 93 ms; C3 16.9 -> 17.4 ms: there the kernel has no issue slots
        //  to spare and loses a wave per SIMD to the registers: pairs for wide indices only)
        if constexpr (std::is_same<BV, PlainBV>::value && kSweepPairs && kWide) {
            static_assert(kPer % 2 == 0, "elements are taken in pairs");
#pragma unroll 1
            for (uint32_t i = 0; i < kPer; i += 2) {
                const uint64_t tA = base + i * 256 + threadIdx.x, tB = tA + 256;
                const bool onA = tA < total, onB = tB < total;
                const uint64_t wA = onA ? word_of(tA) : 0, wB = onB ? word_of(tB) : 0;
                sweep_first_pair<pos_t, kTrail, kWide, kAhead>(iv, s, sampling, onA, tA - t0, wA, onB, tB - t0, wB, val, key, out, member, rec, t0, n_lv, n_lf, n_fin, front);
            }
        } else {
#pragma unroll 1
            for (uint32_t i = 0; i < kPer; ++i) {
                const uint64_t t = base + i * 256 + threadIdx.x;
                if (t < total)
                    sweep_element<BV, pos_t, kTrail, kWide, true, kAhead>(iv, s, sampling, t - t0, word_of(t), val, key, 0u, out, member, rec, t0, n_lv, n_lf, n_fin, false, front);
            }
        }
    }
    unsigned long long v[3] = {n_lf, n_lv, n_fin};
    unsigned long long* const dst[3] = {&stats[0], &stats[1], n_done};
    block_add<3>(v, dst);
}

// stragglers: the elements still alive after the sweep are finished without sorting them any more, by locate_kernel<.., kTail>:
// how long an element still walks is geometrically distributed (one SA index in `dens` is sampled), so a lane that kept one
// element to its end would idle most of the time behind the longest walk of its wave; there a lane that has finished takes the
// next element of its wave's slice.

// Records of a trail-sharing sweep -> positions: every element follows its chain of records (element it follows, steps apart) to
// the end, at most kResolveHops hops per round, and replaces its record by what it found -- a position, or a shorter pointer for
// the next round.  Records are single 64-bit words, so a reader sees a valid state of the element it follows whatever that
// one's own thread is doing; chains collapse as the elements ahead finish (measured on C3: 2 + 2 hops per round over four
// rounds 24.5 ms, to the end in one round + one checking round 18.6 ms).
// Round 3 measured three ways around that wall, none of which paid (C3, per batch): (1) not looking at the owner's record when an
// element stops in the first rounds, where the owner is all but certainly still walking: the sweep gains 1 ms, this kernel loses 2.5
// -- the looks are nearly free inside the sweep, which is not request-bound; (2) taking the hops in the order the sweep left the
// stopped elements in (the per-round tails of its buffers, last round first, then first to last): 35.5 instead of 17.7 ms --
// neighbours in SA order are not neighbours in slot order; (3) following the chains on a second stream beside the sweep's late rounds:
// this kernel 18.0 -> 13.3 ms, but the rounds and their partitions slow down by 7 ms: the requests are conserved, not hidden.
constexpr uint32_t kResolveHops = 4096;       // (64 left 2·10⁵ of C3's 6·10⁸ pointers open -- runs of more than 64 occurrences side by side in the text -- and cost a second pass over all records: 1 ms)
// A workgroup takes kResolveChunk CONSECUTIVE records per turn (not every gridDim-th group of 256): consecutive elements of a list
// that read the same symbol in front of them follow consecutive elements of another list, so the lines a wave fetches for its hops
// are the lines the next waves of the same chunk need -- through the CU's own L1 when they belong to one workgroup.
#ifndef VLG_RESOLVE_CHUNK
#define VLG_RESOLVE_CHUNK 4096
#endif
constexpr uint32_t kResolveChunk = VLG_RESOLVE_CHUNK;
// The first pass, regrouped.  The 64 consecutive elements a wave holds have ~20 different symbols in front of them, so the records they
// follow lie in ~20 other lists, three side by side in each: ~40 lines per wave-wide hop (VLG_RESOLVE_STATS).  Elements with the SAME
// symbol in front follow CONSECUTIVE records (LF restricted to a symbol is monotone), so a workgroup sorts its kGroupChunk records by that
// symbol (front[]: written in round 0 for the elements that stopped on their first step -- six in ten on C3, nine in ten on C4; a counting
// sort in LDS, the kernel's VALU is idle) and takes the hops in that order: the first hop of a wave reads 8 lines.  Results are staged in
// LDS and written back in record order, coalesced.  (Measured, round 4: chunks of 1024 / 2048 / 4096 records -- C3 16.7 / 16.9 / 22.9 ms, C4
// 118 / 109 / 145 ms: the kernel needs its waves; a second, stable regrouping by the symbol in front of the OWNER for the second hop -- rocPRIM's
// match ranking twice per chunk -- 17.4 vs 17.2 ms: not kept.)
#ifndef VLG_GROUP_CHUNK
#define VLG_GROUP_CHUNK 2048
#endif
constexpr uint32_t kGroupChunk = VLG_GROUP_CHUNK;
template <typename pos_t, bool kWide>
__global__ void __launch_bounds__(256) trail_resolve_grouped_kernel(uint64_t* __restrict__ rec, uint64_t count, pos_t* __restrict__ out,
                                                                    const uint8_t* __restrict__ front, unsigned long long* __restrict__ n_open)
{
    constexpr uint32_t kShift = kWide ? 33 : 32, kPer = kGroupChunk / 256;
    constexpr uint64_t kLow = (1ull << kShift) - 1;
    __shared__ uint64_t s_rec[kGroupChunk];
    __shared__ uint16_t s_idx[kGroupChunk];
    __shared__ uint32_t s_bin[258];
    uint32_t open = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * kGroupChunk; base < count; base += (uint64_t)gridDim.x * kGroupChunk) {
        const uint32_t n = (uint32_t)(count - base < kGroupChunk ? count - base : kGroupChunk);
        for (uint32_t j = threadIdx.x; j < 258; j += 256) s_bin[j] = 0;
        __syncthreads();
        uint32_t grp[kPer], arr[kPer];
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            const uint32_t e = threadIdx.x + 256 * k;
            grp[k] = 257; arr[k] = 0;
            if (e < n) {
                const uint64_t r = rec[base + e];
                s_rec[e] = r;
                grp[k] = (r >> kShift) ? (uint32_t)front[base + e] : 256u;       // positions (nothing to follow) stand last
                arr[k] = atomicAdd(&s_bin[grp[k]], 1u);
            }
        }
        __syncthreads();
        {   // exclusive scan of the 257 counters (one wave does it: 5 per lane)
            if (threadIdx.x < 64) {
                uint32_t v[5], sum = 0;
#pragma unroll
                for (uint32_t i = 0; i < 5; ++i) { const uint32_t b = threadIdx.x * 5 + i; v[i] = b < 257 ? s_bin[b] : 0; sum += v[i]; }
                uint32_t incl = sum;
                for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o); if ((int)threadIdx.x >= o) incl += u; }
                uint32_t run = incl - sum;
#pragma unroll
                for (uint32_t i = 0; i < 5; ++i) { const uint32_t b = threadIdx.x * 5 + i; if (b < 257) s_bin[b] = run; run += v[i]; }
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            const uint32_t e = threadIdx.x + 256 * k;
            if (e < n) s_idx[s_bin[grp[k]] + arr[k]] = (uint16_t)e;
        }
        __syncthreads();
#pragma unroll 1
        for (uint32_t k = 0; k < kPer; ++k) {
            const uint32_t p = threadIdx.x + 256 * k;
            if (p < n) {
                const uint32_t e = s_idx[p];
                uint64_t r = s_rec[e];
                if (r >> kShift) {
                    for (uint32_t h = 0; h < kResolveHops && (r >> kShift); ++h) {
                        const uint64_t ro = rec[r & kLow];
                        const uint64_t delta = r >> kShift;
                        r = (ro >> kShift) == 0 ? ro + delta : ro + (delta << kShift);
                    }
                    s_rec[e] = r;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            const uint32_t e = threadIdx.x + 256 * k;
            if (e < n) {
                const uint64_t r = s_rec[e];
                rec[base + e] = r;
                if ((r >> kShift) == 0) out[base + e] = (pos_t)r;
                else ++open;
            }
        }
        __syncthreads();
    }
    unsigned long long v[1] = {open};
    unsigned long long* const dst[1] = {n_open};
    block_add<1>(v, dst);
}

template <typename pos_t, bool kWide>
__global__ void __launch_bounds__(256) trail_resolve_kernel(uint64_t* __restrict__ rec, uint64_t count, pos_t* __restrict__ out,
                                                            unsigned long long* __restrict__ n_open, uint32_t round, bool diag)
{
    constexpr uint32_t kShift = kWide ? 33 : 32;
    constexpr uint64_t kLow = (1ull << kShift) - 1;
    uint32_t open = 0;
    unsigned long long n_hops = 0, n_lines = 0, n_ptr = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * kResolveChunk; base < count; base += (uint64_t)gridDim.x * kResolveChunk) {
        const uint64_t end = base + kResolveChunk < count ? base + kResolveChunk : count;
        for (uint64_t e = base + threadIdx.x; e < end; e += 256) {
            uint64_t r = rec[e];
            if (r >> kShift) {
                ++n_ptr;
                for (uint32_t h = 0; h < kResolveHops && (r >> kShift); ++h) {
                    if (diag) {                                               // hops, and distinct 64-byte lines per wave-wide hop
                        ++n_hops;
                        const uint64_t line = (r & kLow) >> 3;
                        bool first = true;
                        const unsigned long long act = __ballot(true);
                        for (int o = 1; o < 64; ++o) {
                            const int src = (int)((threadIdx.x & 63) - o);
                            const uint64_t other = __shfl(line, src & 63);
                            if (src >= 0 && ((act >> src) & 1) && other == line) first = false;
                        }
                        if (first) ++n_lines;
                    }
                    const uint64_t ro = rec[r & kLow];
                    const uint64_t delta = r >> kShift;
                    r = (ro >> kShift) == 0 ? ro + delta : ro + (delta << kShift);
                }
                rec[e] = r;
                if ((r >> kShift) == 0) out[e] = (pos_t)r;
                else ++open;
            } else if (round == 0) {
                out[e] = (pos_t)r;                                                // elements that never followed anyone
            }
        }
    }
    if (diag) {
        unsigned long long v[4] = {open, n_hops, n_lines, n_ptr};
        unsigned long long* const dst[4] = {n_open, n_open + 1, n_open + 2, n_open + 3};
        block_add<4>(v, dst);
        return;
    }
    unsigned long long v[1] = {open};
    unsigned long long* const dst[1] = {n_open};
    block_add<1>(v, dst);
}

// =============================================================================================
// K3u: locate by UNSAMPLING the suffix array (dense batches).
//
// csa[i] walks LF from i to the next sampled index (csa_wt.hpp:335-348).  Trail sharing (above) already lets the walks of one batch
// share their steps; when a batch asks for a large part of all text positions (BASELINE configs 3 and 4: 60 % of them) the limit of
// that idea is cheaper still: start ONE walker at every SA sample (i = d j, SA[i] = samples[j]) and let it walk LF until it stands on
// the next sampled index, writing SA[LF(i)] = SA[i] - 1 on every step (suffix_array_helper.hpp:336-349).  The walks are disjoint
// and cover every SA index exactly once: n LF steps whatever the batch holds, no trail table (8 B x n), no records, no pointer
// jumping afterwards -- the occurrence lists are then plain copies of SA intervals (sa_dense_copy_kernel).  Everything is recomputed
// from the index's samples for every batch; nothing survives the call.  The walkers are kept in ascending SA-index order exactly as
// in the sorted sweep (stable partition by the symbol read), so neighbouring lanes read neighbouring super-blocks.
// Element words: narrow  val = SA value << 32 | SA index,  key = symbol;
//                wide    val = (SA value & 2^31 - 1) << 33 | SA index (33 bits),  key = symbol | (bit 31 of the SA value) << 15
// (a wide index has n <= 2^32 + 1 here, so every value a walker WRITES is a text position < 2^32: only the sentinel suffix's own
// sample, SA[0] = n - 1, can be 2^32, and it is never written through a walker nor part of a pattern's interval).
// =============================================================================================
constexpr uint16_t kUnsampleDead = 0x7FFFu;                   // key of a walker that has arrived (sorts behind every symbol)

template <bool kWide> struct WalkerWord;
template <> struct WalkerWord<false> {
    static __device__ __forceinline__ void unpack(uint64_t w, uint16_t, uint64_t& i, uint32_t& v) { i = w & 0xFFFFFFFFull; v = (uint32_t)(w >> 32); }
    static __device__ __forceinline__ uint64_t word(uint64_t i, uint32_t v) { return ((uint64_t)v << 32) | i; }
    static __device__ __forceinline__ uint16_t key(uint32_t c, uint32_t) { return (uint16_t)c; }
};
template <> struct WalkerWord<true> {
    static __device__ __forceinline__ void unpack(uint64_t w, uint16_t k, uint64_t& i, uint32_t& v)
    {
        i = w & ((1ull << 33) - 1);
        v = (uint32_t)(w >> 33) | ((uint32_t)(k >> 15) << 31);
    }
    static __device__ __forceinline__ uint64_t word(uint64_t i, uint32_t v) { return ((uint64_t)(v & 0x7FFFFFFFu) << 33) | i; }
    static __device__ __forceinline__ uint16_t key(uint32_t c, uint32_t v) { return (uint16_t)(c | ((v >> 31) << 15)); }
};

// one LF step from SA index i: the symbol read (compact) and LF(i)
template <class BV, bool kWide>
__device__ __forceinline__ uint64_t lf_step(const IndexView& iv, const WalkLds<BV>& s, uint64_t i, uint32_t& c, uint32_t& n_lv)
{
    using walk_t = typename std::conditional<kWide, uint64_t, uint32_t>::type;      // node-relative positions: < n
    uint32_t v = 0;
    walk_t pos = (walk_t)i;
    for (;;) {                                               // inverse_select: wt_pc.hpp:385-402
        const DNode nd = s.nodes[v];
        uint32_t bit;
        walk_t r1;
        BV::rank_bit(iv, s.sh, nd.base, pos, r1, bit);
        ++n_lv;
        pos = bit ? r1 : pos - r1;
        const uint32_t ch = bit ? nd.child[1] : nd.child[0];
        if (ch & kLeafFlag) { c = ch & ~kLeafFlag; break; }
        v = ch;
    }
    return s.C[c] + (uint64_t)pos;                           // LF: suffix_array_helper.hpp:341-348
}

// kFirst: round 0 -- element e is sample e (no words read); otherwise the walkers [0, count) of val / key, dead ones skipped (the
// host may pass a count from a few rounds ago: the dead are at the end, the partition keeps them there).
template <class BV, bool kWide, bool kFirst>
__global__ void __launch_bounds__(256) unsample_step_kernel(IndexView iv, uint64_t* __restrict__ val, uint16_t* __restrict__ key, uint64_t count,
                                                            uint32_t* __restrict__ sa, unsigned long long* __restrict__ stats /* lf, levels */,
                                                            unsigned long long* __restrict__ n_done)
{
    __shared__ WalkLds<BV> s;
    stage_walk(s, iv);
    using sample_t = typename std::conditional<kWide, uint64_t, uint32_t>::type;
    const sample_t* __restrict__ samples = reinterpret_cast<const sample_t*>(iv.samples);
    const uint32_t dens = iv.dens, dmask = dens - 1;
    const bool pow2 = (dens & dmask) == 0;
    uint32_t n_lv = 0, n_lf = 0, n_fin = 0;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t i;
        uint32_t v;
        if (kFirst) {
            i = e * dens;
            const uint64_t sv = (uint64_t)samples[e];
            v = (uint32_t)sv;                                // (2^32 for the sentinel suffix of a wide index: v - 1 below is still right)
            sa[i] = v;
            if (iv.sigma == 1) { key[e] = kUnsampleDead; ++n_fin; continue; }      // degenerate: only the sentinel exists
        } else {
            const uint16_t k = key[e];
            if ((k & 0x7FFFu) == kUnsampleDead) continue;
            WalkerWord<kWide>::unpack(val[e], k, i, v);
        }
        uint32_t c;
        const uint64_t i2 = lf_step<BV, kWide>(iv, s, i, c, n_lv);
        ++n_lf;
        const uint32_t v2 = (!kWide && v == 0) ? (uint32_t)(iv.n - 1) : v - 1;      // SA[LF(i)] = SA[i] - 1 (mod n): csa_wt.hpp:343-347
        const bool arrived = pow2 ? ((i2 & dmask) == 0) : (i2 % dens == 0);
        if (arrived) { key[e] = kUnsampleDead; ++n_fin; }
        else {
            sa[i2] = v2;
            val[e] = WalkerWord<kWide>::word(i2, v2);
            key[e] = WalkerWord<kWide>::key(c, v2);
        }
    }
    unsigned long long v3[3] = {n_lf, n_lv, n_fin};
    unsigned long long* const dst[3] = {&stats[0], &stats[1], n_done};
    block_add<3>(v3, dst);
}

// The last walkers (how long a walk is, is geometrically distributed: a few are still on their way after 100 rounds) finish without
// being sorted any more, lane by lane with refill as in locate_kernel: a wave owns a slice of the walkers, a lane that arrives pulls
// the next one.
template <class BV, bool kWide>
__global__ void __launch_bounds__(256) unsample_tail_kernel(IndexView iv, const uint64_t* __restrict__ val, const uint16_t* __restrict__ key, uint64_t total,
                                                            uint32_t per_wave, uint32_t* __restrict__ sa, unsigned long long* __restrict__ stats)
{
    __shared__ WalkLds<BV> s;
    stage_walk(s, iv);
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t next = wave * per_wave;
    const uint64_t slice_end = next + per_wave < total ? next + per_wave : total;
    const uint32_t dens = iv.dens, dmask = dens - 1;
    const bool pow2 = (dens & dmask) == 0;
    uint64_t i = 0;
    uint32_t v = 0, node = 0;
    bool active = false, need = true;
    uint32_t n_lf = 0, n_lv = 0;
    for (;;) {
        const unsigned long long m = __ballot(need);
        if (m) {
            const uint32_t before = __popcll(m & ((1ull << lane) - 1ull));
            if (need) {
                const uint64_t cand = next + before;
                active = false;
                if (cand < slice_end) {
                    const uint16_t k = key[cand];
                    if ((k & 0x7FFFu) != kUnsampleDead) { WalkerWord<kWide>::unpack(val[cand], k, i, v); node = 0; active = true; }
                }
                need = !active && cand < slice_end;          // a dead walker: take another one next turn
            }
            next += __popcll(m);
        }
        if (!__any(active) && !__any(need)) break;
        if (active) {
            const DNode nd = s.nodes[node];
            uint32_t bit;
            uint64_t r1;
            BV::rank_bit(iv, s.sh, nd.base, i, r1, bit);
            ++n_lv;
            const uint64_t ni = bit ? r1 : i - r1;
            const uint32_t ch = bit ? nd.child[1] : nd.child[0];
            if (ch & kLeafFlag) {
                i = s.C[ch & ~kLeafFlag] + ni;
                v = (!kWide && v == 0) ? (uint32_t)(iv.n - 1) : v - 1;
                node = 0;
                ++n_lf;
                const bool arrived = pow2 ? ((i & dmask) == 0) : (i % dens == 0);
                if (arrived) { need = true; active = false; }
                else sa[i] = v;
            } else {
                i = ni;
                node = ch;
            }
        }
    }
    if (stats) {
        unsigned long long v2[2] = {n_lf, n_lv};
        unsigned long long* const dst[2] = {&stats[0], &stats[1]};
        block_add<2>(v2, dst);
    }
}

// ISA samples (csa_sampling_strategy.hpp:626-642: isa_sample[SA[i] / d'] = i for every i with SA[i] % d' == 0), computed from the
// index alone: a lane starts at one SA sample (i, SA[i]) and walks LF -- (LF(i), SA[i] - 1) -- until the next sampled
// index, so every SA index is visited exactly once and every text position passes by with its SA index.
template <class BV, typename pos_t>
__global__ void __launch_bounds__(256) isa_samples_kernel(IndexView iv, uint32_t inv_dens, uint64_t* __restrict__ out)
{
    __shared__ WalkLds<BV> s;
    stage_walk(s, iv);
    const pos_t* samples = reinterpret_cast<const pos_t*>(iv.samples);
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < iv.n_samples; j += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t i = j * iv.dens, v = samples[j];
        do {
            if (v % inv_dens == 0) out[v / inv_dens] = i;
            uint32_t node = 0, c;
            uint64_t pos = i;
            for (;;) {                                       // inverse_select: wt_pc.hpp:385-402
                const DNode nd = s.nodes[node];
                uint32_t bit;
                uint64_t r1;
                BV::rank_bit(iv, s.sh, nd.base, pos, r1, bit);
                pos = bit ? r1 : pos - r1;
                const uint32_t ch = bit ? nd.child[1] : nd.child[0];      // (a select, not an indexed read: the node stays in registers)
                if (ch & kLeafFlag) { c = ch & ~kLeafFlag; break; }
                node = ch;
            }
            i = s.C[c] + pos;                                // LF
            v = v ? v - 1 : iv.n - 1;
        } while (i % iv.dens);
    }
}

template <typename pos_t>
__global__ void widen_kernel(const pos_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t count)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x) out[j] = in[j];
}
template <typename pos_t>
__global__ void narrow_kernel(const uint64_t* __restrict__ in, pos_t* __restrict__ out, uint64_t count)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x) out[j] = (pos_t)in[j];
}

inline uint32_t grid_for(uint64_t n, uint32_t cap = 16384) { return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, cap)); }

}  // namespace

namespace vlg {

vlg_status launch_backward_search(const IndexView& iv, const uint8_t* d_blob, const uint64_t* d_off, uint64_t n_pat, uint64_t* d_l,
                                  uint64_t* d_r, unsigned long long* d_stat_levels, hipStream_t stream)
{
    if (!n_pat) return VLG_OK;
    if (iv.bv_kind == kBvRrr63)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(backward_search_kernel<RrrBV>), dim3(grid_for(n_pat, 4096)), dim3(256), 0, stream, iv, d_blob, d_off,
                           n_pat, d_l, d_r, d_stat_levels);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(backward_search_kernel<PlainBV>), dim3(grid_for(n_pat, 4096)), dim3(256), 0, stream, iv, d_blob, d_off,
                           n_pat, d_l, d_r, d_stat_levels);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}

template <typename pos_t>
vlg_status launch_expand(const uint64_t* d_l, const uint64_t* d_out_off, uint64_t n_pat, uint64_t total, pos_t* d_io, uint32_t* d_seg,
                         hipStream_t stream)
{
    if (!total) return VLG_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(expand_kernel<pos_t>), dim3(grid_for(total, 32768)), dim3(256), 0, stream, d_l, d_out_off, n_pat,
                       total, d_io, d_seg);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}
template vlg_status launch_expand<uint32_t>(const uint64_t*, const uint64_t*, uint64_t, uint64_t, uint32_t*, uint32_t*, hipStream_t);
template vlg_status launch_expand<uint64_t>(const uint64_t*, const uint64_t*, uint64_t, uint64_t, uint64_t*, uint32_t*, hipStream_t);

// Slices: enough waves to fill 256 CUs x 32 waves several times over, but slices long enough that the
// drain tail (the slowest occurrence of a slice) stays a small fraction of the slice.
template <typename pos_t>
vlg_status launch_locate(const IndexView& iv, pos_t* d_io, uint64_t total, unsigned long long* d_stats, hipStream_t stream)
{
    if (!total) return VLG_OK;
    const uint64_t target_waves = 256ull * 32 * 4;
    uint64_t per_wave = (total + target_waves - 1) / target_waves;
    per_wave = std::max<uint64_t>(per_wave, 64 * 16);
    per_wave = std::min<uint64_t>(per_wave, 1u << 20);
    uint64_t waves = (total + per_wave - 1) / per_wave;
    uint64_t wgs = (waves + 3) / 4;
    const bool rrr = iv.bv_kind == kBvRrr63, text_order = iv.sampling == kSamplingTextOrder;
    if (iv.sample_bytes != sizeof(pos_t)) return fail(VLG_E_INTERNAL, "locate: sample width does not match the instantiation");
    const dim3 grid((uint32_t)wgs);
#define VLG_LOCATE(BV, TO) hipLaunchKernelGGL(HIP_KERNEL_NAME(locate_kernel<pos_t, BV, false, (sizeof(pos_t) == 8), TO>), grid, dim3(256), 0, stream, iv, d_io, total, (uint32_t)per_wave, d_stats)
    if constexpr (sizeof(pos_t) == 4) {
        if (text_order) { if (rrr) VLG_LOCATE(RrrBV, true); else VLG_LOCATE(PlainBV, true); }
        else { if (rrr) VLG_LOCATE(RrrBV, false); else VLG_LOCATE(PlainBV, false); }
    } else {
        if (text_order) { if (rrr) VLG_LOCATE(RrrBV, true); else VLG_LOCATE(PlainBV, true); }
        else { if (rrr) VLG_LOCATE(RrrBV, false); else VLG_LOCATE(PlainBV, false); }
    }
#undef VLG_LOCATE
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}
template vlg_status launch_locate<uint32_t>(const IndexView&, uint32_t*, uint64_t, unsigned long long*, hipStream_t);
template vlg_status launch_locate<uint64_t>(const IndexView&, uint64_t*, uint64_t, unsigned long long*, hipStream_t);


size_t sweep_temp_bytes(uint64_t total, uint32_t sigma, hipStream_t stream)
{
    size_t tb = 0;
    rocprim::double_buffer<uint16_t> k(nullptr, nullptr);
    rocprim::double_buffer<uint64_t> v(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, tb, k, v, total, 0, bit_width64(sigma), stream);   // the sweep's own two buffers alternate
    return tb;
}

// d_l / d_out_off: SA interval starts and output offsets of the n_pat lists; d_out receives SA values (unsorted, SA order).
// Scratch (caller-provided): val_a, val_b (u64 each), key_a, key_b (u16 each) for min(total, sweep_batch_max<pos_t>()) elements,
// temp (sweep_temp_bytes), counter (8 B, zeroed here).
// The sweep's driver: rounds, partitions, records and the stragglers' slices for ANY index that can launch the three kernels of
// SweepKernels (kernels.hpp) -- the byte index below, the integer-alphabet index in int_index.hpp.
void launch_sweep_chunk_lists(const uint64_t* d_out_off, uint64_t n_pat, uint64_t t0, uint64_t t1, uint32_t* chunk_list, hipStream_t stream)
{
    const uint64_t chunks = (t1 - t0 + kSweepChunk - 1) / kSweepChunk;
    if (!chunks) return;
    hipLaunchKernelGGL(sweep_chunk_lists_kernel, dim3((uint32_t)((chunks + 255) / 256)), dim3(256), 0, stream, d_out_off, n_pat, t0, t1, chunk_list);
}

template <typename pos_t, bool kWide>
vlg_status run_locate_sweep(const SweepKernels& K, const uint64_t* d_l, const uint64_t* d_out_off, uint64_t n_pat, uint64_t total,
                            pos_t* d_out, uint64_t* val_a, uint64_t* val_b, uint16_t* key_a, uint16_t* key_b, void* temp,
                            size_t temp_bytes, unsigned long long* d_counter, uint64_t tail_threshold,
                            hipStream_t stream, LaunchTimer* timer, Block* member /* member_blocks(n) super-blocks, or null: no LF step is shared */,
                            uint32_t n_member_lists /* the first so many lists are pairwise disjoint and ascend: they make up `member` */,
                            uint64_t* rec /* total words */,
                            const std::function<vlg_status()>* while_first_step /* host work to do while the first step runs, or null */)
{
    bool hook_due = while_first_step != nullptr;
    constexpr uint32_t kShift = kWide ? 33 : 32;
    if (K.n > (1ull << kShift)) return fail(VLG_E_UNSUPPORTED, "sorted sweep: text too long for the packed position");
    if (sizeof(pos_t) == 4 && K.n > (1ull << 32) + 1) return fail(VLG_E_INTERNAL, "sorted sweep: positions do not fit 32 bits");
    if (K.sigma >= 0xFFFFu) return fail(VLG_E_INTERNAL, "sorted sweep: alphabet too large for the 16-bit partition key");
    const unsigned bits = bit_width64(K.sigma);             // keys 0..sigma (sigma = finished, sorts last)
    const uint64_t batch_max = sweep_batch_max<kWide>();
    if (member) {
        if (total > 0xFFFFFF00ull) return fail(VLG_E_INTERNAL, "member bit-vector: slots need 32 bits");
        if (timer) timer->begin(0);
        hipLaunchKernelGGL(member_build_kernel, dim3(grid_for(member_blocks(K.n), 16384)), dim3(256), 0, stream, d_l, d_out_off, n_member_lists,
                           member_blocks(K.n), member);
        if (timer) timer->end(0);
        VLG_HIP_TRY(hipGetLastError());
        // more than one sweep: an element may stop at an element of a LATER sweep, whose record must read "still walking" until then
        if (total > batch_max) VLG_HIP_TRY(hipMemsetAsync(rec, 0xFF, total * 8, stream));
    }
    bool front_valid = K.front != nullptr && [] { const char* e = getenv("VLG_RESOLVE_GROUPED"); return !(e && e[0] == '0'); }();
    for (uint64_t t0 = 0; t0 < total; t0 += batch_max) {
        const uint64_t t1 = std::min(total, t0 + batch_max);
        // (a sweep too short for a single round hands its elements to the stragglers' kernel, which reads their words)
        const bool fused_first = t1 - t0 > tail_threshold && [] { const char* e = getenv("VLG_NO_FUSED_FIRST_ROUND"); return !(e && e[0] == '1'); }();
        if (!fused_first) front_valid = false;                   // (front[] is written by the fused first round only)
        static const bool ahead = [] { const char* e = getenv("VLG_SWEEP_LOOKAHEAD"); return !(e && e[0] == '0'); }();
        if (!fused_first) {
            if (member) VLG_HIP_TRY(hipMemsetAsync(rec + t0, 0xFF, (t1 - t0) * 8, stream));     // "still walking" (the first-round kernel writes it itself)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(sweep_init_kernel<kShift>), dim3(grid_for((t1 - t0 + 7) / 8, 32768)), dim3(256), 0, stream, d_l, d_out_off, n_pat,
                               t0, t1, val_a);
        }
        VLG_HIP_TRY(hipGetLastError());
        uint64_t alive = t1 - t0;
        uint32_t step = 0;
        pos_t* out = d_out + t0;
        while (alive > tail_threshold && step < 0xFFFFFFu) {
            VLG_HIP_TRY(hipMemsetAsync(d_counter, 0, 8, stream));
            if (timer) timer->begin(0);
            if (fused_first && step == 0) K.first(t0, t1, val_a, key_a, out, d_counter, member, rec, ahead, reinterpret_cast<uint32_t*>(val_b));      // round 0 makes the elements' words itself
            else K.step(val_a, key_a, alive, step, out, d_counter, member, rec, t0, ahead && fused_first && step == 1);
            if (timer) timer->end(0);
            VLG_HIP_TRY(hipGetLastError());
            size_t tb = temp_bytes;
            if (timer) timer->begin(1, 20ull * alive);           // key + element read once and written once
            rocprim::double_buffer<uint16_t> dk(key_a, key_b);
            rocprim::double_buffer<uint64_t> dv(val_a, val_b);
            hipError_t se = rocprim::radix_sort_pairs(temp, tb, dk, dv, alive, 0, bits, stream);
            if (timer) timer->end(1);
            VLG_HIP_TRY(se);
            unsigned long long done = 0;
            VLG_HIP_TRY(hipMemcpyAsync(&done, d_counter, 8, hipMemcpyDeviceToHost, stream));
            if (hook_due) { hook_due = false; if (vlg_status hs = (*while_first_step)()) return hs; }
            VLG_HIP_TRY(hipStreamSynchronize(stream));
            key_a = dk.current(); key_b = dk.alternate();
            val_a = dv.current(); val_b = dv.alternate();
            alive -= done;
            ++step;
            if (step > 1u << 20) return fail(VLG_E_INTERNAL, "locate sweep did not converge");
        }
        if (alive) {
            // slices as launch_locate cuts them: enough waves to fill the chip several times over, long enough that a slice's slowest
            // element is a small part of it
            const uint64_t target_waves = 256ull * 32 * 4;
            uint64_t per_wave = (alive + target_waves - 1) / target_waves;
            per_wave = std::min<uint64_t>(std::max<uint64_t>(per_wave, 64 * 16), 1u << 20);
            const uint64_t waves = (alive + per_wave - 1) / per_wave;
            if (timer) timer->begin(0);
            K.tail(out, alive, (uint32_t)per_wave, val_a, step, member ? rec : nullptr, t0, member, (uint32_t)((waves + 3) / 4));
            if (timer) timer->end(0);
            VLG_HIP_TRY(hipGetLastError());
        }
    }
    if (hook_due) { hook_due = false; if (vlg_status hs = (*while_first_step)()) return hs; }
    if (member) {
        // every element has a record now; jump pointers until all of them are positions
        static const bool diag = [] { const char* e = getenv("VLG_RESOLVE_STATS"); return e && e[0] == '1'; }();
        for (uint32_t round = 0;; ++round) {
            VLG_HIP_TRY(hipMemsetAsync(d_counter, 0, diag ? 32 : 8, stream));
            if (timer) timer->begin(2, round == 0 ? total * (8ull + sizeof(pos_t)) : 0);     // every record read, every position written
            if (round == 0 && front_valid && !diag)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(trail_resolve_grouped_kernel<pos_t, kWide>), dim3((uint32_t)std::min<uint64_t>((total + kGroupChunk - 1) / kGroupChunk, 1u << 20)),
                                   dim3(256), 0, stream, rec, total, d_out, K.front, d_counter);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(trail_resolve_kernel<pos_t, kWide>), dim3((uint32_t)std::min<uint64_t>((total + kResolveChunk - 1) / kResolveChunk, 65536)),
                                   dim3(256), 0, stream, rec, total, d_out, d_counter, round, diag);
            if (timer) timer->end(2);
            VLG_HIP_TRY(hipGetLastError());
            unsigned long long open = 0;
            VLG_HIP_TRY(hipMemcpyAsync(&open, d_counter, 8, hipMemcpyDeviceToHost, stream));
            VLG_HIP_TRY(hipStreamSynchronize(stream));
            if (diag) {
                unsigned long long c[4] = {0, 0, 0, 0};
                VLG_HIP_TRY(hipMemcpy(c, d_counter, 32, hipMemcpyDeviceToHost));
                fprintf(stderr, "[vlg resolve] round %u: %llu records, %llu pointers, %llu hops, %llu distinct lines (per wave-wide hop), %llu still open\n", round,
                        (unsigned long long)total, c[3], c[1], c[2], c[0]);
            }
            if (!open) break;
            if (round > 64) return fail(VLG_E_INTERNAL, "trail records did not resolve");
        }
    }
    return VLG_OK;
}
template vlg_status run_locate_sweep<uint32_t, false>(const SweepKernels&, const uint64_t*, const uint64_t*, uint64_t, uint64_t, uint32_t*, uint64_t*, uint64_t*,
                                                      uint16_t*, uint16_t*, void*, size_t, unsigned long long*, uint64_t, hipStream_t, LaunchTimer*, Block*,
                                                      uint32_t, uint64_t*, const std::function<vlg_status()>*);

// the byte index (IndexView: Huffman-shaped tree, plain or rrr bit-vectors, either sampling) in the sweep
template <typename pos_t, bool kWide>
vlg_status launch_locate_sweep(const IndexView& iv, const uint64_t* d_l, const uint64_t* d_out_off, uint64_t n_pat, uint64_t total,
                               pos_t* d_out, uint64_t* val_a, uint64_t* val_b, uint16_t* key_a, uint16_t* key_b, void* temp,
                               size_t temp_bytes, unsigned long long* d_counter, unsigned long long* d_stats, uint64_t tail_threshold,
                               hipStream_t stream, LaunchTimer* timer, Block* member, uint32_t n_member_lists, uint64_t* rec,
                               const std::function<vlg_status()>* while_first_step, uint8_t* front)
{
    if (iv.sample_bytes != (kWide ? 8u : 4u)) return fail(VLG_E_INTERNAL, "sorted sweep: sample width does not match the instantiation");
    const bool rrr = iv.bv_kind == kBvRrr63, text_order = iv.sampling == kSamplingTextOrder;
    if (iv.dens == 1 && !text_order && total) {                  // every SA index is sampled: no walk, no trails, no records
        using sample_t = typename std::conditional<kWide, uint64_t, uint32_t>::type;
        if (timer) timer->begin(0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sa_dense_copy_kernel<pos_t, sample_t>), dim3(grid_for((total + 7) / 8, 32768)), dim3(256), 0, stream,
                           reinterpret_cast<const sample_t*>(iv.samples), d_l, d_out_off, n_pat, total, d_out);
        if (timer) timer->end(0);
        VLG_HIP_TRY(hipGetLastError());
        if (while_first_step) if (vlg_status hs = (*while_first_step)()) return hs;
        return VLG_OK;
    }
    SweepKernels K;
    K.n = iv.n;
    K.sigma = iv.sigma;
    K.front = rec ? front : nullptr;
    uint8_t* const fr = K.front;
    K.first = [&](uint64_t t0, uint64_t t1, uint64_t* val, uint16_t* key, void* out_, unsigned long long* counter, const Block* mem, uint64_t* rc, bool ahead,
                  uint32_t* chunk_list) {
        pos_t* out = static_cast<pos_t*>(out_);
        launch_sweep_chunk_lists(d_out_off, n_pat, t0, t1, chunk_list, stream);
        const dim3 grid_first(grid_for((t1 - t0 + 7) / 8, 8192));
#define VLG_FIRST(BV, TR, TO) do { if (ahead) hipLaunchKernelGGL(HIP_KERNEL_NAME(sweep_first_kernel<BV, pos_t, TR, kWide, TO, TR>), grid_first, dim3(256), 0, stream, iv, d_l, d_out_off, n_pat, t0, t1, val, key, out, d_stats, counter, mem, rc, chunk_list, fr); \
                                    else hipLaunchKernelGGL(HIP_KERNEL_NAME(sweep_first_kernel<BV, pos_t, TR, kWide, TO, false>), grid_first, dim3(256), 0, stream, iv, d_l, d_out_off, n_pat, t0, t1, val, key, out, d_stats, counter, mem, rc, chunk_list, fr); } while (0)
#define VLG_FIRST_BV(TR, TO) do { if (rrr) VLG_FIRST(RrrBV, TR, TO); else VLG_FIRST(PlainBV, TR, TO); } while (0)
        if (text_order) { if (mem) VLG_FIRST_BV(true, true); else VLG_FIRST_BV(false, true); }
        else { if (mem) VLG_FIRST_BV(true, false); else VLG_FIRST_BV(false, false); }
#undef VLG_FIRST_BV
#undef VLG_FIRST
    };
    K.step = [&](uint64_t* val, uint16_t* key, uint64_t alive, uint32_t step, void* out_, unsigned long long* counter, const Block* mem, uint64_t* rc, uint64_t t0, bool probed) {
        pos_t* out = static_cast<pos_t*>(out_);
        const dim3 grid(grid_for(alive, 4096));
#define VLG_STEP(BV, TR, TO) hipLaunchKernelGGL(HIP_KERNEL_NAME(sweep_step_kernel<BV, pos_t, TR, kWide, TO>), grid, dim3(256), 0, stream, iv, val, key, alive, step, out, d_stats, counter, mem, rc, t0, probed)
#define VLG_STEP_BV(TR, TO) do { if (rrr) VLG_STEP(RrrBV, TR, TO); else VLG_STEP(PlainBV, TR, TO); } while (0)
        if (text_order) { if (mem) VLG_STEP_BV(true, true); else VLG_STEP_BV(false, true); }
        else { if (mem) VLG_STEP_BV(true, false); else VLG_STEP_BV(false, false); }
#undef VLG_STEP_BV
#undef VLG_STEP
    };
    K.tail = [&](void* out_, uint64_t alive, uint32_t per_wave, const uint64_t* val, uint32_t step, uint64_t* rc, uint64_t t0, const Block* mem, uint32_t blocks) {
        pos_t* out = static_cast<pos_t*>(out_);
        const dim3 grid(blocks);
#define VLG_TAIL(BV, TO) hipLaunchKernelGGL(HIP_KERNEL_NAME(locate_kernel<pos_t, BV, true, kWide, TO>), grid, dim3(256), 0, stream, iv, out, alive, per_wave, d_stats, val, step, rc, t0, mem)
        if (text_order) { if (rrr) VLG_TAIL(RrrBV, true); else VLG_TAIL(PlainBV, true); }
        else { if (rrr) VLG_TAIL(RrrBV, false); else VLG_TAIL(PlainBV, false); }
#undef VLG_TAIL
    };
    return run_locate_sweep<pos_t, kWide>(K, d_l, d_out_off, n_pat, total, d_out, val_a, val_b, key_a, key_b, temp, temp_bytes, d_counter, tail_threshold, stream, timer,
                                          member, n_member_lists, rec, while_first_step);
}
// K3u (above): the whole suffix array into sa_full (n words of 32 bits), then the SA intervals of the lists into d_out.
// val / key buffers for n_samples walkers; temp as for the sweep.  Rounds are enqueued kUnsampleSync at a time: the count of walkers
// still on their way is read back only then (the kernels skip the ones that have arrived, which the partition keeps at the end).
constexpr uint32_t kUnsampleSync = 4;
template <bool kWide>
vlg_status launch_unsample(const IndexView& iv, const uint64_t* d_l, const uint64_t* d_out_off, uint64_t n_pat, uint64_t total, uint32_t* d_out,
                           uint32_t* sa_full, uint64_t* val_a, uint64_t* val_b, uint16_t* key_a, uint16_t* key_b, void* temp, size_t temp_bytes,
                           unsigned long long* d_counter, unsigned long long* h_done /* 8 pinned bytes */, unsigned long long* d_stats, uint64_t tail_threshold,
                           hipStream_t stream, LaunchTimer* timer, const std::function<vlg_status()>* while_first_step)
{
    bool hook_due = while_first_step != nullptr;
    if (iv.sampling != kSamplingSaOrder || iv.dens < 2) return fail(VLG_E_INTERNAL, "unsampling needs SA-order samples of density >= 2");
    if (iv.sample_bytes != (kWide ? 8u : 4u)) return fail(VLG_E_INTERNAL, "unsampling: sample width does not match the instantiation");
    if (iv.n > (1ull << 32) + 1 || (!kWide && iv.n > (1ull << 32))) return fail(VLG_E_UNSUPPORTED, "unsampling: text too long for 32-bit positions");
    if (iv.sigma >= 0x7FFFu) return fail(VLG_E_INTERNAL, "unsampling: alphabet too large for the key");
    const bool rrr = iv.bv_kind == kBvRrr63;
    const unsigned bits = bit_width64(iv.sigma);            // keys 0 .. sigma - 1, dead = all ones in these bits too (sorts last)
    uint64_t alive = iv.n_samples, done_total = 0;
    VLG_HIP_TRY(hipMemsetAsync(d_counter, 0, 8, stream));
    uint32_t round = 0;
    while (alive > tail_threshold) {
        for (uint32_t r = 0; r < kUnsampleSync; ++r, ++round) {
            if (timer) timer->begin(0);
            const dim3 grid(grid_for(alive, 8192));
#define VLG_UNS(BV) do { if (round == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(unsample_step_kernel<BV, kWide, true>), grid, dim3(256), 0, stream, iv, val_a, key_a, alive, sa_full, d_stats, d_counter); \
                         else hipLaunchKernelGGL(HIP_KERNEL_NAME(unsample_step_kernel<BV, kWide, false>), grid, dim3(256), 0, stream, iv, val_a, key_a, alive, sa_full, d_stats, d_counter); } while (0)
            if (rrr) VLG_UNS(RrrBV); else VLG_UNS(PlainBV);
#undef VLG_UNS
            if (timer) timer->end(0);
            VLG_HIP_TRY(hipGetLastError());
            size_t tb = temp_bytes;
            if (timer) timer->begin(1, 20ull * alive);
            rocprim::double_buffer<uint16_t> dk(key_a, key_b);
            rocprim::double_buffer<uint64_t> dv(val_a, val_b);
            const hipError_t se = rocprim::radix_sort_pairs(temp, tb, dk, dv, alive, 0, bits, stream);
            if (timer) timer->end(1);
            VLG_HIP_TRY(se);
            key_a = dk.current(); key_b = dk.alternate();
            val_a = dv.current(); val_b = dv.alternate();
            if (hook_due) { hook_due = false; if (vlg_status hs = (*while_first_step)()) return hs; }
        }
        VLG_HIP_TRY(hipMemcpyAsync(h_done, d_counter, 8, hipMemcpyDeviceToHost, stream));
        VLG_HIP_TRY(hipStreamSynchronize(stream));
        done_total = *h_done;
        if (done_total > iv.n_samples) return fail(VLG_E_INTERNAL, "unsampling: more walkers arrived than were started");
        alive = iv.n_samples - done_total;
        if (round > (1u << 20)) return fail(VLG_E_INTERNAL, "unsampling did not converge");
    }
    if (round == 0 && iv.n_samples) {
        // too few walkers for a single sorted round: round 0 still makes their words (and writes the samples themselves)
        if (timer) timer->begin(0);
        const dim3 grid(grid_for(alive, 8192));
        if (rrr) hipLaunchKernelGGL(HIP_KERNEL_NAME(unsample_step_kernel<RrrBV, kWide, true>), grid, dim3(256), 0, stream, iv, val_a, key_a, alive, sa_full, d_stats, d_counter);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(unsample_step_kernel<PlainBV, kWide, true>), grid, dim3(256), 0, stream, iv, val_a, key_a, alive, sa_full, d_stats, d_counter);
        if (timer) timer->end(0);
        VLG_HIP_TRY(hipGetLastError());
        ++round;
    }
    if (hook_due) { hook_due = false; if (vlg_status hs = (*while_first_step)()) return hs; }
    if (alive || round == 1) {
        // (after an unsorted round 0 the dead are anywhere: the tail looks at all of them)
        const uint64_t span = round == 1 ? iv.n_samples : alive;
        const uint64_t target_waves = 256ull * 32 * 4;
        uint64_t per_wave = (span + target_waves - 1) / target_waves;
        per_wave = std::min<uint64_t>(std::max<uint64_t>(per_wave, 64 * 4), 1u << 20);
        const uint64_t waves = (span + per_wave - 1) / per_wave;
        const dim3 grid((uint32_t)((waves + 3) / 4));
        if (timer) timer->begin(0);
        if (rrr) hipLaunchKernelGGL(HIP_KERNEL_NAME(unsample_tail_kernel<RrrBV, kWide>), grid, dim3(256), 0, stream, iv, val_a, key_a, span, (uint32_t)per_wave, sa_full, d_stats);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(unsample_tail_kernel<PlainBV, kWide>), grid, dim3(256), 0, stream, iv, val_a, key_a, span, (uint32_t)per_wave, sa_full, d_stats);
        if (timer) timer->end(0);
        VLG_HIP_TRY(hipGetLastError());
    }
    if (total) {
        if (timer) timer->begin(0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sa_dense_copy_kernel<uint32_t, uint32_t>), dim3(grid_for((total + 7) / 8, 32768)), dim3(256), 0, stream,
                           sa_full, d_l, d_out_off, n_pat, total, d_out);
        if (timer) timer->end(0);
        VLG_HIP_TRY(hipGetLastError());
    }
    return VLG_OK;
}
template vlg_status launch_unsample<false>(const IndexView&, const uint64_t*, const uint64_t*, uint64_t, uint64_t, uint32_t*, uint32_t*, uint64_t*, uint64_t*,
                                           uint16_t*, uint16_t*, void*, size_t, unsigned long long*, unsigned long long*, unsigned long long*, uint64_t, hipStream_t,
                                           LaunchTimer*, const std::function<vlg_status()>*);
template vlg_status launch_unsample<true>(const IndexView&, const uint64_t*, const uint64_t*, uint64_t, uint64_t, uint32_t*, uint32_t*, uint64_t*, uint64_t*,
                                          uint16_t*, uint16_t*, void*, size_t, unsigned long long*, unsigned long long*, unsigned long long*, uint64_t, hipStream_t,
                                          LaunchTimer*, const std::function<vlg_status()>*);

#define VLG_SWEEP_INST(P, W)                                                                                                          \
    template vlg_status launch_locate_sweep<P, W>(const IndexView&, const uint64_t*, const uint64_t*, uint64_t, uint64_t, P*, uint64_t*, \
                                                  uint64_t*, uint16_t*, uint16_t*, void*, size_t, unsigned long long*, unsigned long long*, \
                                                  uint64_t, hipStream_t, LaunchTimer*, Block*, uint32_t, uint64_t*,                       \
                                                  const std::function<vlg_status()>*, uint8_t*);
VLG_SWEEP_INST(uint32_t, false)
VLG_SWEEP_INST(uint32_t, true)        // n = 2^32 + 1 (BASELINE config 4): 33-bit SA indices, 32-bit text positions
VLG_SWEEP_INST(uint64_t, true)
#undef VLG_SWEEP_INST

template <typename T>
vlg_status launch_narrow(const uint64_t* d_in, T* d_out, uint64_t count, hipStream_t stream)
{
    if (!count) return VLG_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(narrow_kernel<T>), dim3(grid_for(count)), dim3(256), 0, stream, d_in, d_out, count);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}
template vlg_status launch_narrow<uint32_t>(const uint64_t*, uint32_t*, uint64_t, hipStream_t);
template <typename T>
vlg_status launch_widen(const T* d_in, uint64_t* d_out, uint64_t count, hipStream_t stream)
{
    if (!count) return VLG_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(widen_kernel<T>), dim3(grid_for(count)), dim3(256), 0, stream, d_in, d_out, count);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}
template vlg_status launch_widen<uint32_t>(const uint32_t*, uint64_t*, uint64_t, hipStream_t);

}  // namespace vlg

extern "C" vlg_status vlg_wt_rank_batch(const vlg_index* idx, const uint64_t* d_i, const uint8_t* d_c, uint64_t* d_out, uint64_t count,
                                        void* stream)
{
    if (!idx || (count && (!d_i || !d_c || !d_out))) return fail(VLG_E_INVALID, "null argument");
    if (idx->is_int) return fail(VLG_E_INVALID, "integer-alphabet index: use vlg_int_rank_batch");
    if (!count) return VLG_OK;
    if (idx->view.bv_kind == kBvRrr63)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(wt_rank_kernel<RrrBV>), dim3(grid_for(count, 8192)), dim3(256), 0, (hipStream_t)stream, idx->view, d_i, d_c,
                           d_out, count);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(wt_rank_kernel<PlainBV>), dim3(grid_for(count, 8192)), dim3(256), 0, (hipStream_t)stream, idx->view, d_i,
                           d_c, d_out, count);
    VLG_HIP_TRY(hipGetLastError());
    return VLG_OK;
}

extern "C" vlg_status vlg_backward_search_batch(const vlg_index* idx, const uint8_t* d_blob, const uint64_t* d_off, uint64_t n_patterns,
                                                uint64_t* d_l, uint64_t* d_r, void* stream)
{
    if (!idx || (n_patterns && (!d_off || !d_l || !d_r))) return fail(VLG_E_INVALID, "null argument");
    // (an integer-alphabet index takes patterns of little-endian uint32_t symbols, offsets in bytes)
    if (idx->is_int) return launch_int_backward_search(idx->iview, d_blob, d_off, n_patterns, d_l, d_r, nullptr, (hipStream_t)stream);
    return launch_backward_search(idx->view, d_blob, d_off, n_patterns, d_l, d_r, nullptr, (hipStream_t)stream);
}

extern "C" vlg_status vlg_sa_batch(const vlg_index* idx, const uint64_t* d_i, uint64_t* d_out, uint64_t count, void* stream)
{
    if (!idx || (count && (!d_i || !d_out))) return fail(VLG_E_INVALID, "null argument");
    if (!count) return VLG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (idx->hdr.sample_bytes == 8) {
        if (d_out != d_i) VLG_HIP_TRY(hipMemcpyAsync(d_out, d_i, count * 8, hipMemcpyDeviceToDevice, st));
        return launch_locate<uint64_t>(idx->view, d_out, count, nullptr, st);
    }
    uint32_t* tmp = nullptr;
    VLG_HIP_TRY(hipMalloc((void**)&tmp, count * 4));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(narrow_kernel<uint32_t>), dim3(grid_for(count)), dim3(256), 0, st, d_i, tmp, count);
    vlg_status s2 = idx->is_int ? launch_int_locate(idx->iview, tmp, count, nullptr, st) : launch_locate<uint32_t>(idx->view, tmp, count, nullptr, st);
    if (!s2) hipLaunchKernelGGL(HIP_KERNEL_NAME(widen_kernel<uint32_t>), dim3(grid_for(count)), dim3(256), 0, st, tmp, d_out, count);
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (s2) return s2;
    VLG_HIP_TRY(e);
    return VLG_OK;
}

extern "C" vlg_status vlg_locate_batch(const vlg_index* idx, const uint64_t* d_l, const uint64_t* d_r, const uint64_t* d_out_off,
                                       uint64_t n_patterns, uint64_t total, uint64_t* d_out, void* stream)
{
    (void)d_r;
    if (!idx || (n_patterns && (!d_l || !d_out_off)) || (total && !d_out)) return fail(VLG_E_INVALID, "null argument");
    if (!total) return VLG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (idx->hdr.sample_bytes == 8) {
        if (vlg_status s = launch_expand<uint64_t>(d_l, d_out_off, n_patterns, total, d_out, nullptr, st)) return s;
        return launch_locate<uint64_t>(idx->view, d_out, total, nullptr, st);
    }
    uint32_t* tmp = nullptr;
    VLG_HIP_TRY(hipMalloc((void**)&tmp, total * 4));
    vlg_status s2 = launch_expand<uint32_t>(d_l, d_out_off, n_patterns, total, tmp, nullptr, st);
    if (!s2) s2 = idx->is_int ? launch_int_locate(idx->iview, tmp, total, nullptr, st) : launch_locate<uint32_t>(idx->view, tmp, total, nullptr, st);
    if (!s2) hipLaunchKernelGGL(HIP_KERNEL_NAME(widen_kernel<uint32_t>), dim3(grid_for(total)), dim3(256), 0, st, tmp, d_out, total);
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (s2) return s2;
    VLG_HIP_TRY(e);
    return VLG_OK;
}

extern "C" vlg_status vlg_index_isa_samples(const vlg_index* idx, uint32_t inv_dens, uint64_t* h_out, uint64_t count)
{
    if (!idx || !h_out || !inv_dens) return fail(VLG_E_INVALID, "null argument");
    const uint64_t n = idx->hdr.n;
    if (count != (n - 1) / inv_dens + 1) return fail(VLG_E_INVALID, "ISA sample count must be (n-1)/inv_dens + 1");
    if (idx->hdr.sampling != kSamplingSaOrder) return fail(VLG_E_UNSUPPORTED, "ISA samples are computed from an SA-order index");
    if (idx->is_int) return fail(VLG_E_UNSUPPORTED, "ISA samples of an integer-alphabet index are not built");
    uint64_t* d_out = nullptr;
    VLG_HIP_TRY(hipMalloc((void**)&d_out, count * 8));
    VLG_HIP_TRY(hipMemset(d_out, 0, count * 8));
    const dim3 grid(grid_for(idx->view.n_samples, 8192));
    const bool rrr = idx->view.bv_kind == kBvRrr63, wide = idx->hdr.sample_bytes == 8;
    if (rrr && wide) hipLaunchKernelGGL(HIP_KERNEL_NAME(isa_samples_kernel<RrrBV, uint64_t>), grid, dim3(256), 0, nullptr, idx->view, inv_dens, d_out);
    else if (rrr) hipLaunchKernelGGL(HIP_KERNEL_NAME(isa_samples_kernel<RrrBV, uint32_t>), grid, dim3(256), 0, nullptr, idx->view, inv_dens, d_out);
    else if (wide) hipLaunchKernelGGL(HIP_KERNEL_NAME(isa_samples_kernel<PlainBV, uint64_t>), grid, dim3(256), 0, nullptr, idx->view, inv_dens, d_out);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(isa_samples_kernel<PlainBV, uint32_t>), grid, dim3(256), 0, nullptr, idx->view, inv_dens, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(h_out, d_out, count * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    VLG_HIP_TRY(e);
    return VLG_OK;
}
