// Index objects in HBM: creation from reference-layout parts, export, blob replication, and the
// on-device builder (suffix sort by prefix doubling -> BWT -> wavelet-tree super-blocks -> SA samples).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string.h>
#include "common.hpp"
#include <functional>
#include "device_rank.hpp"
#include "rrr_code.hpp"
#include <rocprim/rocprim.hpp>

using namespace vlg;

namespace {

struct InnerTable {          // inner nodes sorted by first block (= BFS order), for block->node lookup
    uint32_t count;
    uint32_t base[256];
    uint64_t bv_pos[256];
    uint64_t size[256];
    uint32_t depth[256];
    uint32_t node[256];
};

__device__ __forceinline__ uint32_t find_inner_by_block(const InnerTable& t, uint32_t blk)
{
    uint32_t lo = 0, hi = t.count;            // last k with base[k] <= blk
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (t.base[mid] <= blk) lo = mid; else hi = mid;
    }
    return lo;
}

// sdsl bit-vector words -> data words of the 256-bit super-blocks (pull: one thread per u32 word)
__global__ void repack_kernel(const uint64_t* __restrict__ src, uint64_t src_words, Block* __restrict__ blocks,
                              uint64_t n_blocks, const InnerTable* __restrict__ tab)
{
    __shared__ InnerTable t;
    for (uint32_t i = threadIdx.x; i < sizeof(InnerTable) / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(&t)[i] = reinterpret_cast<const uint32_t*>(tab)[i];
    __syncthreads();
    uint64_t total = n_blocks * 8;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t blk = (uint32_t)(g >> 3), w = (uint32_t)(g & 7);
        if (w == 7) continue;                                    // cnt word: second pass
        uint32_t k = find_inner_by_block(t, blk);
        uint64_t rel = (uint64_t)(blk - t.base[k]) * kBlockBits + 32u * w;   // node-relative bit
        uint32_t val = 0;
        if (rel < t.size[k]) {
            uint64_t bit = t.bv_pos[k] + rel;
            uint64_t wi = bit >> 6;
            uint32_t s = (uint32_t)(bit & 63);
            uint64_t lo = src[wi] >> s;
            if (s > 32 && wi + 1 < src_words) lo |= src[wi + 1] << (64 - s);
            val = (uint32_t)lo;
            uint64_t left = t.size[k] - rel;
            if (left < 32) val &= (1u << left) - 1u;
        }
        blocks[blk].w[w] = val;
    }
}

struct BlockPop {
    const Block* blocks;
    __device__ uint64_t operator()(uint64_t b) const
    {
        const Block& B = blocks[b];
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) c += __popc(B.w[i]);
        return c;
    }
};

// cnt = ones of the node before the block = global exclusive scan - scan at the node's first block
__global__ void fill_counts_kernel(Block* __restrict__ blocks, const uint64_t* __restrict__ scan, uint64_t n_blocks,
                                   const InnerTable* __restrict__ tab)
{
    __shared__ InnerTable t;
    for (uint32_t i = threadIdx.x; i < sizeof(InnerTable) / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(&t)[i] = reinterpret_cast<const uint32_t*>(tab)[i];
    __syncthreads();
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t k = find_inner_by_block(t, (uint32_t)b);
        blocks[b].cnt = (uint32_t)(scan[b] - scan[t.base[k]]);
    }
}

// super-blocks -> sdsl bit-vector words (one thread per u64 output word; export only)
__global__ void unpack_kernel(const Block* __restrict__ blocks, uint64_t* __restrict__ dst, uint64_t dst_words,
                              uint64_t bv_bits, const InnerTable* __restrict__ tab)
{
    __shared__ InnerTable t;
    for (uint32_t i = threadIdx.x; i < sizeof(InnerTable) / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(&t)[i] = reinterpret_cast<const uint32_t*>(tab)[i];
    __syncthreads();
    for (uint64_t W = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; W < dst_words; W += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t out = 0;
        for (uint32_t j = 0; j < 64; ++j) {
            uint64_t bit = W * 64 + j;
            if (bit >= bv_bits) break;
            uint32_t lo = 0, hi = t.count;                     // last k with bv_pos[k] <= bit
            while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (t.bv_pos[mid] <= bit) lo = mid; else hi = mid; }
            uint64_t rel = bit - t.bv_pos[lo];
            uint64_t blk = t.base[lo] + rel / kBlockBits;
            uint32_t off = (uint32_t)(rel % kBlockBits);
            out |= (uint64_t)((blocks[blk].w[off >> 5] >> (off & 31)) & 1u) << j;
        }
        dst[W] = out;
    }
}

InnerTable make_inner_table(const HostTree& t)
{
    InnerTable tab;
    memset(&tab, 0, sizeof tab);
    for (uint32_t v = 0; v < t.n_nodes; ++v)
        if (t.nodes[v].child[0] != 0xFFFF) {
            uint32_t k = tab.count++;
            tab.base[k] = t.dnodes[v].base;
            tab.bv_pos[k] = t.nodes[v].bv_pos;
            tab.size[k] = t.node_size[v];
            tab.depth[k] = t.node_depth[v];
            tab.node[k] = v;
        }
    return tab;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 256); }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

void bind_view(vlg_index* idx)
{
    uint8_t* b = reinterpret_cast<uint8_t*>(idx->d_blob);
    const BlobHeader& h = idx->hdr;
    idx->view.blocks = reinterpret_cast<const Block*>(b + h.off_blocks);
    idx->view.nodes = reinterpret_cast<const DNode*>(b + h.off_nodes);
    idx->view.C = reinterpret_cast<const uint64_t*>(b + h.off_C);
    idx->view.paths = reinterpret_cast<const uint64_t*>(b + h.off_paths);
    idx->view.char2comp = b + h.off_c2c;
    idx->view.samples = b + h.off_samples;
    idx->view.n = h.n;
    idx->view.n_samples = h.n_samples;
    idx->view.n_nodes = h.n_nodes;
    idx->view.sigma = h.sigma;
    idx->view.dens = h.dens;
    idx->view.sample_bytes = h.sample_bytes;
    idx->view.bv_kind = (uint32_t)h.bv_kind;
    idx->view.sampling = h.sampling;
    idx->view.marked = h.sampling == kSamplingTextOrder ? reinterpret_cast<const Block*>(b + h.off_marked) : nullptr;
    idx->view.rrr_hdr = reinterpret_cast<const uint4*>(b + h.off_rrr_hdr);
    idx->view.rrr_stream = reinterpret_cast<const uint64_t*>(b + h.off_rrr_stream);
    idx->view.rrr_tables = reinterpret_cast<const RrrTables*>(b + h.off_binom);
}

// Section offsets and the total size from the counts in the header (sections in a fixed order, 256-byte aligned).
struct BlobSection { uint64_t BlobHeader::*off; uint64_t bytes; };
inline void blob_sections(const BlobHeader& h, BlobSection (&sec)[11])
{
    const BlobSection all[11] = {
        {&BlobHeader::off_blocks, h.n_blocks * sizeof(Block)},
        {&BlobHeader::off_nodes, (uint64_t)kMaxNodes * sizeof(DNode)},
        {&BlobHeader::off_C, 257 * 8},
        {&BlobHeader::off_paths, 256 * 8},
        {&BlobHeader::off_c2c, 256},
        {&BlobHeader::off_samples, h.n_samples * h.sample_bytes},
        {&BlobHeader::off_refnodes, (uint64_t)kMaxNodes * sizeof(vlg_wt_node)},
        {&BlobHeader::off_rrr_hdr, h.n_rrr_sb * 32},
        {&BlobHeader::off_rrr_stream, (h.rrr_stream_words + 2) * 8},
        {&BlobHeader::off_binom, h.n_rrr_sb ? 64ull * 64 * 8 : 0ull},
        {&BlobHeader::off_marked, h.sampling == kSamplingTextOrder ? (h.n / kBlockBits + 1) * sizeof(Block) : 0ull},
    };
    for (int i = 0; i < 11; ++i) sec[i] = all[i];
}
inline void layout_blob(BlobHeader& h)
{
    BlobSection sec[11];
    blob_sections(h, sec);
    uint64_t off = align_up(sizeof(BlobHeader), 256);
    for (const BlobSection& s : sec) { h.*(s.off) = off; off = align_up(off + s.bytes, 256); }
    h.total_bytes = off;
}

// Plan the blob from the host tree, allocate it, upload the small tables. Blocks + samples stay to be filled.
vlg_status alloc_blob(vlg_index* idx, uint64_t n, uint32_t dens, hipStream_t stream, uint64_t n_rrr_sb = 0, uint64_t rrr_words = 0)
{
    HostTree& t = idx->tree;
    BlobHeader& h = idx->hdr;
    memset(&h, 0, sizeof h);
    h.magic = kBlobMagic;
    h.n = n;
    h.wt_bits = t.wt_bits;
    h.n_blocks = n_rrr_sb ? 0 : t.n_blocks;
    h.bv_kind = n_rrr_sb ? kBvRrr63 : kBvPlain;
    h.n_rrr_sb = n_rrr_sb;
    h.rrr_stream_words = rrr_words;
    h.sigma = t.sigma;
    h.dens = dens;
    h.n_nodes = t.n_nodes;
    h.max_code_len = t.max_code_len;
    h.n_samples = (n + dens - 1) / dens;
    // SA samples and SA indices are 32-bit up to n = 2^32 and 64-bit words beyond; VLG_FORCE_POS64=1 or 2 selects the wide form on any
    // text (what a text of 4 GiB and more runs on; tests use the switch to exercise it on small inputs: 1 = 64-bit positions
    // everywhere, 2 = wide SA indices with 32-bit text positions, the mix BASELINE config 4 runs on -- see vlg_search_batch)
    const char* f64 = getenv("VLG_FORCE_POS64");
    h.sample_bytes = (n <= 0x100000000ull && !(f64 && (f64[0] == '1' || f64[0] == '2'))) ? 4 : 8;
    layout_blob(h);
    VLG_HIP_TRY(hipMalloc(&idx->d_blob, h.total_bytes));
    idx->owns_blob = true;
    uint8_t* b = reinterpret_cast<uint8_t*>(idx->d_blob);
    VLG_HIP_TRY(hipMemcpyAsync(b, &h, sizeof h, hipMemcpyHostToDevice, stream));
    VLG_HIP_TRY(hipMemsetAsync(b + h.off_nodes, 0, kMaxNodes * sizeof(DNode), stream));
    if (t.n_nodes)
        VLG_HIP_TRY(hipMemcpyAsync(b + h.off_nodes, t.dnodes.data(), t.n_nodes * sizeof(DNode), hipMemcpyHostToDevice, stream));
    VLG_HIP_TRY(hipMemsetAsync(b + h.off_C, 0, 257 * 8, stream));
    VLG_HIP_TRY(hipMemcpyAsync(b + h.off_C, t.C.data(), t.C.size() * 8, hipMemcpyHostToDevice, stream));
    VLG_HIP_TRY(hipMemcpyAsync(b + h.off_paths, t.paths.data(), 256 * 8, hipMemcpyHostToDevice, stream));
    VLG_HIP_TRY(hipMemcpyAsync(b + h.off_c2c, t.char2comp, 256, hipMemcpyHostToDevice, stream));
    VLG_HIP_TRY(hipMemsetAsync(b + h.off_refnodes, 0, kMaxNodes * sizeof(vlg_wt_node), stream));
    if (t.n_nodes)
        VLG_HIP_TRY(hipMemcpyAsync(b + h.off_refnodes, t.nodes.data(), t.n_nodes * sizeof(vlg_wt_node), hipMemcpyHostToDevice, stream));
    VLG_HIP_TRY(hipStreamSynchronize(stream));   // host sources above may go out of scope
    bind_view(idx);
    return VLG_OK;
}

// Second pass shared by from_parts and the builder: popcount scan -> node-relative cnt words.
vlg_status fill_block_counts(vlg_index* idx, const InnerTable* d_tab, hipStream_t stream)
{
    uint64_t nb = idx->hdr.n_blocks;
    if (!nb) return VLG_OK;
    Block* blocks = const_cast<Block*>(idx->view.blocks);
    DevBuf scan, temp;
    VLG_HIP_TRY(scan.alloc(nb * 8));
    auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), BlockPop{blocks});
    size_t tb = 0;
    VLG_HIP_TRY(rocprim::exclusive_scan(nullptr, tb, in, scan.as<uint64_t>(), (uint64_t)0, nb, rocprim::plus<uint64_t>(), stream));
    VLG_HIP_TRY(temp.alloc(tb));
    VLG_HIP_TRY(rocprim::exclusive_scan(temp.p, tb, in, scan.as<uint64_t>(), (uint64_t)0, nb, rocprim::plus<uint64_t>(), stream));
    uint32_t grid = (uint32_t)std::min<uint64_t>((nb + 255) / 256, 8192);
    hipLaunchKernelGGL(fill_counts_kernel, dim3(grid), dim3(256), 0, stream, blocks, scan.as<uint64_t>(), nb, d_tab);
    VLG_HIP_TRY(hipGetLastError());
    VLG_HIP_TRY(hipStreamSynchronize(stream));
    return VLG_OK;
}

vlg_status check_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(VLG_E_NO_DEVICE, "no HIP device available (the VLG library has no CPU fallback)");
    return VLG_OK;
}

}  // namespace

extern "C" vlg_status vlg_device_count(int* n)
{
    if (!n) return fail(VLG_E_INVALID, "null argument");
    *n = 0;
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess) { *n = 0; return fail(VLG_E_NO_DEVICE, hipGetErrorString(e)); }
    return VLG_OK;
}

extern "C" vlg_status vlg_set_device(int ordinal)
{
    VLG_HIP_TRY(hipSetDevice(ordinal));
    return VLG_OK;
}

extern "C" vlg_status vlg_index_from_parts(const vlg_index_parts* p, vlg_index** out)
{
    release_cached_device_memory();          // an index wants its memory now; parked result buffers can be allocated again

    if (!p || !out) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    if (vlg_status st = check_device()) return st;
    if (p->n == 0 || !p->char2comp || !p->C || !p->nodes || (p->bv_bits && !p->bv_words) || !p->sa_samples)
        return fail(VLG_E_INVALID, "incomplete index parts");
    if (p->sa_sample_dens == 0) return fail(VLG_E_INVALID, "sa_sample_dens must be > 0");
    if (p->n > (1ull << 36)) return fail(VLG_E_UNSUPPORTED, "text longer than 2^36");
    if (p->n_samples != (p->n + p->sa_sample_dens - 1) / p->sa_sample_dens)
        return fail(VLG_E_INVALID, "n_samples must be ceil(n/dens)");
    vlg_index* idx = new vlg_index();
    vlg_status st = tree_from_nodes(p->nodes, p->n_nodes, p->bv_bits, p->char2comp, p->C, p->sigma, idx->tree);
    if (st) { delete idx; return st; }
    hipStream_t stream = nullptr;
    st = alloc_blob(idx, p->n, p->sa_sample_dens, stream);
    if (st) { vlg_index_destroy(idx); return st; }
    auto run = [&]() -> vlg_status {
        const BlobHeader& h = idx->hdr;
        uint8_t* b = reinterpret_cast<uint8_t*>(idx->d_blob);
        // samples
        if (h.sample_bytes == 4) {
            std::vector<uint32_t> s32(h.n_samples);
            for (uint64_t i = 0; i < h.n_samples; ++i) s32[i] = (uint32_t)p->sa_samples[i];
            VLG_HIP_TRY(hipMemcpy(b + h.off_samples, s32.data(), h.n_samples * 4, hipMemcpyHostToDevice));
        } else {
            VLG_HIP_TRY(hipMemcpy(b + h.off_samples, p->sa_samples, h.n_samples * 8, hipMemcpyHostToDevice));
        }
        if (h.n_blocks) {
            InnerTable tab = make_inner_table(idx->tree);
            DevBuf d_tab, d_src;
            VLG_HIP_TRY(d_tab.alloc(sizeof tab));
            VLG_HIP_TRY(hipMemcpy(d_tab.p, &tab, sizeof tab, hipMemcpyHostToDevice));
            uint64_t words = (p->bv_bits + 63) / 64;
            VLG_HIP_TRY(d_src.alloc(words * 8 + 8));
            VLG_HIP_TRY(hipMemcpy(d_src.p, p->bv_words, words * 8, hipMemcpyHostToDevice));
            Block* blocks = const_cast<Block*>(idx->view.blocks);
            uint32_t grid = (uint32_t)std::min<uint64_t>((h.n_blocks * 8 + 255) / 256, 16384);
            hipLaunchKernelGGL(repack_kernel, dim3(grid), dim3(256), 0, stream, d_src.as<uint64_t>(), words, blocks,
                               h.n_blocks, d_tab.as<InnerTable>());
            VLG_HIP_TRY(hipGetLastError());
            if (vlg_status s2 = fill_block_counts(idx, d_tab.as<InnerTable>(), stream)) return s2;
        }
        VLG_HIP_TRY(hipDeviceSynchronize());
        return VLG_OK;
    };
    st = run();
    if (st) { vlg_index_destroy(idx); return st; }
    *out = idx;
    return VLG_OK;
}

extern "C" vlg_status vlg_index_export_parts(const vlg_index* idx, vlg_index_parts* sizes, vlg_index_parts_out* out)
{
    if (!idx || !sizes) return fail(VLG_E_INVALID, "null argument");
    if (idx->is_int) return fail(VLG_E_UNSUPPORTED, "an integer-alphabet index has no byte-alphabet parts (vlg_index_export_int_alphabet gives its alphabet)");
    const BlobHeader& h = idx->hdr;
    if (h.bv_kind != kBvPlain && out && (out->bv_words || out->nodes))
        return fail(VLG_E_UNSUPPORTED, "exporting the bit-vector of an rrr-compressed index is not supported; export the plain index");
    memset(sizes, 0, sizeof *sizes);
    sizes->n = h.n;
    sizes->sigma = h.sigma;
    sizes->sa_sample_dens = h.dens;
    sizes->bv_bits = h.wt_bits;
    sizes->n_nodes = h.n_nodes;
    sizes->n_samples = h.n_samples;
    if (!out) return VLG_OK;
    const HostTree& t = idx->tree;
    if (out->char2comp) memcpy(out->char2comp, t.char2comp, 256);
    if (out->C) { memset(out->C, 0, 257 * 8); memcpy(out->C, t.C.data(), t.C.size() * 8); }
    const uint8_t* b = reinterpret_cast<const uint8_t*>(idx->d_blob);
    uint64_t words = (h.wt_bits + 63) / 64;
    std::vector<uint64_t> bv(words);
    if (words && (out->bv_words || out->nodes)) {
        InnerTable tab = make_inner_table(t);
        DevBuf d_tab, d_dst;
        VLG_HIP_TRY(d_tab.alloc(sizeof tab));
        VLG_HIP_TRY(hipMemcpy(d_tab.p, &tab, sizeof tab, hipMemcpyHostToDevice));
        VLG_HIP_TRY(d_dst.alloc(words * 8));
        uint32_t grid = (uint32_t)std::min<uint64_t>((words + 255) / 256, 16384);
        hipLaunchKernelGGL(unpack_kernel, dim3(grid), dim3(256), 0, nullptr, idx->view.blocks, d_dst.as<uint64_t>(), words,
                           h.wt_bits, d_tab.as<InnerTable>());
        VLG_HIP_TRY(hipGetLastError());
        VLG_HIP_TRY(hipMemcpy(bv.data(), d_dst.p, words * 8, hipMemcpyDeviceToHost));
        if (out->bv_words) memcpy(out->bv_words, bv.data(), words * 8);
    }
    if (out->nodes) {
        // bv_pos_rank of inner nodes = rank1(bv_pos) over the concatenated bit-vector (wt_helper.hpp:243-250)
        std::vector<vlg_wt_node> nodes = t.nodes;
        std::vector<uint32_t> order;
        for (uint32_t v = 0; v < t.n_nodes; ++v) if (nodes[v].child[0] != 0xFFFF) order.push_back(v);
        uint64_t cum = 0, pos = 0;
        for (uint32_t v : order) {                       // inner nodes are in increasing bv_pos order
            uint64_t target = nodes[v].bv_pos;
            while (pos < target) {
                uint64_t w = pos >> 6, o = pos & 63;
                uint64_t take = std::min<uint64_t>(64 - o, target - pos);
                uint64_t m = take == 64 ? ~0ull : (((1ull << take) - 1) << o);
                cum += (uint64_t)__builtin_popcountll(bv[w] & m);
                pos += take;
            }
            nodes[v].bv_pos_rank = cum;
        }
        memcpy(out->nodes, nodes.data(), nodes.size() * sizeof(vlg_wt_node));
    }
    if (out->sa_samples) {
        if (h.sample_bytes == 4) {
            std::vector<uint32_t> s32(h.n_samples);
            VLG_HIP_TRY(hipMemcpy(s32.data(), b + h.off_samples, h.n_samples * 4, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < h.n_samples; ++i) out->sa_samples[i] = s32[i];
        } else {
            VLG_HIP_TRY(hipMemcpy(out->sa_samples, b + h.off_samples, h.n_samples * 8, hipMemcpyDeviceToHost));
        }
    }
    return VLG_OK;
}

extern "C" vlg_status vlg_index_get_info(const vlg_index* idx, vlg_index_info* info)
{
    if (!idx || !info) return fail(VLG_E_INVALID, "null argument");
    const BlobHeader& h = idx->hdr;
    info->n = h.n;
    info->sigma = h.sigma;
    info->sa_sample_dens = h.dens;
    info->n_nodes = h.n_nodes;
    info->max_code_len = h.max_code_len;
    info->wt_bits = h.wt_bits;
    info->n_blocks = h.n_blocks;
    info->n_samples = h.n_samples;
    info->hbm_bytes = h.total_bytes;
    info->pos_bytes = h.sample_bytes;
    info->bv_kind = (uint32_t)h.bv_kind;
    info->sampling = h.sampling;
    info->reserved = 0;
    return VLG_OK;
}

extern "C" void vlg_index_destroy(vlg_index* idx)
{
    if (!idx) return;
    if (idx->d_blob && idx->owns_blob) (void)hipFree(idx->d_blob);
    delete idx;
}

extern "C" vlg_status vlg_index_blob_bytes(const vlg_index* idx, uint64_t* bytes)
{
    if (!idx || !bytes) return fail(VLG_E_INVALID, "null argument");
    *bytes = idx->hdr.total_bytes;
    return VLG_OK;
}

extern "C" vlg_status vlg_index_blob_export(const vlg_index* idx, void* d_blob, uint64_t bytes, void* stream)
{
    if (!idx || !d_blob) return fail(VLG_E_INVALID, "null argument");
    if (bytes < idx->hdr.total_bytes) return fail(VLG_E_INVALID, "blob buffer too small");
    VLG_HIP_TRY(hipMemcpyAsync(d_blob, idx->d_blob, idx->hdr.total_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    VLG_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return VLG_OK;
}

// One process driving several GPUs (the C++ host's -g N): a second copy of the read-only index on another device of the node,
// moved by a peer copy (xGMI between the GPUs of one node).  One process per GPU uses export / RCCL broadcast / attach instead.
extern "C" vlg_status vlg_index_replicate(const vlg_index* src, int device, vlg_index** out)
{
    if (!src || !out) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    int ndev = 0, cur = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VLG_E_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(VLG_E_INVALID, "no such device");
    VLG_HIP_TRY(hipGetDevice(&cur));
    hipPointerAttribute_t at;
    VLG_HIP_TRY(hipPointerGetAttributes(&at, src->d_blob));
    void* d = nullptr;
    vlg_index* idx = nullptr;
    auto run = [&]() -> vlg_status {
        VLG_HIP_TRY(hipSetDevice(device));
        VLG_HIP_TRY(hipMalloc(&d, src->hdr.total_bytes));
        VLG_HIP_TRY(hipMemcpyPeer(d, device, src->d_blob, at.device, src->hdr.total_bytes));
        VLG_HIP_TRY(hipDeviceSynchronize());
        if (vlg_status st = vlg_index_attach_blob(d, src->hdr.total_bytes, &idx)) return st;
        idx->owns_blob = true;
        return VLG_OK;
    };
    vlg_status st = run();
    if (st && d && !idx) (void)hipFree(d);
    (void)hipSetDevice(cur);
    if (st) return st;
    *out = idx;
    return VLG_OK;
}

extern "C" vlg_status vlg_index_attach_blob(const void* d_blob, uint64_t bytes, vlg_index** out)
{
    if (!d_blob || !out) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    if (bytes < sizeof(BlobHeader)) return fail(VLG_E_INVALID, "blob too small");
    vlg_index* idx = new vlg_index();
    auto run = [&]() -> vlg_status {
        VLG_HIP_TRY(hipMemcpy(&idx->hdr, d_blob, sizeof(BlobHeader), hipMemcpyDeviceToHost));
        if (idx->hdr.magic == kIntBlobMagic) return attach_int_blob(d_blob, bytes, idx);          // integer-alphabet index (int_index.hpp)
        const BlobHeader& h = idx->hdr;
        if (h.magic != kBlobMagic || h.total_bytes > bytes || h.n_nodes > kMaxNodes || h.sigma > 256)
            return fail(VLG_E_INVALID, "not a VLG index blob");
        idx->d_blob = const_cast<void*>(d_blob);
        idx->owns_blob = false;
        bind_view(idx);
        // rebuild the host-side tree copy from the tables stored in the blob
        const uint8_t* b = reinterpret_cast<const uint8_t*>(d_blob);
        std::vector<uint64_t> Cc(h.sigma + 1, 0);
        std::vector<vlg_wt_node> nodes(h.n_nodes ? h.n_nodes : 1);
        uint8_t c2c[256];
        VLG_HIP_TRY(hipMemcpy(Cc.data(), b + h.off_C, (h.sigma + 1) * 8, hipMemcpyDeviceToHost));
        VLG_HIP_TRY(hipMemcpy(c2c, b + h.off_c2c, 256, hipMemcpyDeviceToHost));
        if (h.n_nodes) VLG_HIP_TRY(hipMemcpy(nodes.data(), b + h.off_refnodes, h.n_nodes * sizeof(vlg_wt_node), hipMemcpyDeviceToHost));
        HostTree t2;
        vlg_status st = tree_from_nodes(nodes.data(), h.n_nodes, h.wt_bits, c2c, Cc.data(), h.sigma, t2);
        if (st) return st;
        idx->tree = t2;
        return VLG_OK;
    };
    vlg_status st = run();
    if (st) { delete idx; return st; }
    *out = idx;
    return VLG_OK;
}

// =============================================================================================
// rrr-63 variant of an index (BASELINE config 5: csa_wt<wt_huff<rrr_vector<63>>>): every node's bit-vector is re-encoded
// on the device as 32-byte headers + an offset stream (layout of K6, node-relative counts, each super-block's offsets
// word-aligned).  The tree, C, samples are shared with the plain index; search results are identical.
// =============================================================================================
namespace {

struct RrrTable {            // inner nodes, as InnerTable plus the first rrr super-block of each node
    uint32_t count;
    uint32_t pbase[256];     // first plain block
    uint32_t rbase[256];     // first rrr super-block
    uint64_t size[256];
};

// 63 bits of a node starting at node-relative position pos (zero beyond the node)
__device__ __forceinline__ uint64_t plain_bits63(const Block* __restrict__ blocks, uint32_t pbase, uint64_t pos, uint64_t size)
{
    uint64_t v = 0;
    if (pos >= size) return 0;
    uint32_t len = (uint32_t)(size - pos < 63 ? size - pos : 63);
    uint32_t got = 0;
    while (got < len) {
        uint64_t p = pos + got;
        uint64_t blk = p / kBlockBits;
        uint32_t off = (uint32_t)(p - blk * kBlockBits);
        uint32_t w = off >> 5, o = off & 31;
        uint32_t take = 32 - o;
        if (take > len - got) take = len - got;
        uint32_t word = blocks[pbase + blk].w[w] >> o;
        if (take < 32) word &= (1u << take) - 1u;
        v |= (uint64_t)word << got;
        got += take;
    }
    return v;
}

__device__ __forceinline__ uint32_t rrr_find(const RrrTable& t, uint32_t sb)
{
    uint32_t lo = 0, hi = t.count;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (t.rbase[mid] <= sb) lo = mid; else hi = mid; }
    return lo;
}

// pass 1: ones and offset words of every super-block;  pass 2 (hdr != nullptr): headers + offsets
__global__ void __launch_bounds__(256) rrr_encode_kernel(const Block* __restrict__ blocks, const RrrTable* __restrict__ tab, uint64_t n_sb,
                                                         const RrrTables* __restrict__ code, uint64_t* __restrict__ ones_out,
                                                         uint64_t* __restrict__ words_out, const uint64_t* __restrict__ ones_scan,
                                                         const uint64_t* __restrict__ words_scan, uint4* __restrict__ hdr,
                                                         uint64_t* __restrict__ stream)
{
    __shared__ RrrTable t;
    __shared__ RrrTables ct;
    for (uint32_t i = threadIdx.x; i < sizeof(RrrTable) / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(&t)[i] = reinterpret_cast<const uint32_t*>(tab)[i];
    for (uint32_t i = threadIdx.x; i < sizeof(RrrTables) / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(&ct)[i] = reinterpret_cast<const uint32_t*>(code)[i];
    __syncthreads();
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_sb; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k_node = rrr_find(t, (uint32_t)g);
        const uint64_t sb = g - t.rbase[k_node];
        uint64_t ones = 0, bits = 0;
        uint64_t cls[3] = {0, 0, 0};
        uint64_t wpos = hdr ? words_scan[g] * 64 : 0;          // bit position in the stream
        for (uint32_t j = 0; j < kRrrBlocksPerSuper; ++j) {
            const uint64_t bin = plain_bits63(blocks, t.pbase[k_node], sb * kRrrSuperBits + (uint64_t)j * kRrrBlockBits, t.size[k_node]);
            const uint32_t k = (uint32_t)__popcll(bin);
            const uint32_t len = ct.space[k];
            ones += k;
            if (hdr) {
                const uint32_t b = 6 * j, w = b >> 6, o = b & 63;
                cls[w] |= (uint64_t)k << o;
                if (o > 58) cls[w + 1] |= (uint64_t)k >> (64 - o);
                if (len) {
                    uint32_t kk;
                    const uint64_t nr = rrr_enc63(ct, bin, kk);  // the block's number inside its class (rrr_code.hpp)
                    const uint64_t p = wpos + bits, w2 = p >> 6, o2 = p & 63;
                    stream[w2] |= nr << o2;                      // the region of a super-block is word-aligned and private
                    if (o2 + len > 64) stream[w2 + 1] |= nr >> (64 - o2);
                }
            }
            bits += len;
        }
        if (!hdr) { ones_out[g] = ones; words_out[g] = (bits + 63) / 64; }
        else {
            const uint32_t rel = (uint32_t)(ones_scan[g] - ones_scan[t.rbase[k_node]]);
            hdr[2 * g] = make_uint4(rel, (uint32_t)words_scan[g], (uint32_t)cls[0], (uint32_t)(cls[0] >> 32));
            hdr[2 * g + 1] = make_uint4((uint32_t)cls[1], (uint32_t)(cls[1] >> 32), (uint32_t)cls[2], (uint32_t)(cls[2] >> 32));
        }
    }
}

}  // namespace

namespace {

// The two passes of the re-encoding around whatever allocates the target: pass 1 counts ones and offset words per super-block and scans
// them, `place` allocates and returns where headers / offsets / the block code's tables go, pass 2 writes them.
struct RrrTarget { uint4* hdr; uint64_t* stream; void* tables; };
vlg_status rrr_encode(const Block* blocks, const RrrTable& tab, uint64_t n_sb, hipStream_t stream,
                      const std::function<vlg_status(uint64_t total_words, RrrTarget& where)>& place)
{
    std::vector<uint8_t> code(64 * 64 * 8, 0);                // the block code's tables (the blob keeps 32 KiB for them)
    build_rrr_tables(*reinterpret_cast<RrrTables*>(code.data()));
    uint64_t total_words = 0;
    DevBuf d_tab, d_code, d_ones, d_words, d_tmp;
    VLG_HIP_TRY(d_tab.alloc(sizeof tab));
    VLG_HIP_TRY(d_code.alloc(64 * 64 * 8));
    VLG_HIP_TRY(hipMemcpy(d_tab.p, &tab, sizeof tab, hipMemcpyHostToDevice));
    VLG_HIP_TRY(hipMemcpy(d_code.p, code.data(), 64 * 64 * 8, hipMemcpyHostToDevice));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n_sb + 255) / 256, 4096);
    if (n_sb) {
        VLG_HIP_TRY(d_ones.alloc((n_sb + 1) * 8));
        VLG_HIP_TRY(d_words.alloc((n_sb + 1) * 8));
        hipLaunchKernelGGL(rrr_encode_kernel, dim3(grid), dim3(256), 0, stream, blocks, d_tab.as<RrrTable>(), n_sb, d_code.as<RrrTables>(),
                           d_ones.as<uint64_t>(), d_words.as<uint64_t>(), nullptr, nullptr, nullptr, nullptr);
        VLG_HIP_TRY(hipGetLastError());
        uint64_t last_words = 0;
        VLG_HIP_TRY(hipMemcpyAsync(&last_words, d_words.as<uint64_t>() + (n_sb - 1), 8, hipMemcpyDeviceToHost, stream));
        size_t tb = 0;
        VLG_HIP_TRY(rocprim::exclusive_scan(nullptr, tb, d_ones.as<uint64_t>(), d_ones.as<uint64_t>(), (uint64_t)0, n_sb, rocprim::plus<uint64_t>(), stream));
        VLG_HIP_TRY(d_tmp.alloc(tb));
        VLG_HIP_TRY(rocprim::exclusive_scan(d_tmp.p, tb, d_ones.as<uint64_t>(), d_ones.as<uint64_t>(), (uint64_t)0, n_sb, rocprim::plus<uint64_t>(), stream));
        VLG_HIP_TRY(rocprim::exclusive_scan(d_tmp.p, tb, d_words.as<uint64_t>(), d_words.as<uint64_t>(), (uint64_t)0, n_sb, rocprim::plus<uint64_t>(), stream));
        uint64_t last_off = 0;
        VLG_HIP_TRY(hipMemcpyAsync(&last_off, d_words.as<uint64_t>() + (n_sb - 1), 8, hipMemcpyDeviceToHost, stream));
        VLG_HIP_TRY(hipStreamSynchronize(stream));
        total_words = last_off + last_words;
        if (total_words > 0xFFFFFFF0ull) return fail(VLG_E_UNSUPPORTED, "rrr offset stream too long");
    }
    RrrTarget w{nullptr, nullptr, nullptr};
    if (vlg_status st = place(total_words, w)) return st;
    VLG_HIP_TRY(hipMemcpyAsync(w.tables, d_code.p, 64 * 64 * 8, hipMemcpyDeviceToDevice, stream));
    VLG_HIP_TRY(hipMemsetAsync(w.stream, 0, (total_words + 2) * 8, stream));
    VLG_HIP_TRY(hipMemsetAsync(w.hdr, 0, std::max<uint64_t>(n_sb, 1) * 32, stream));
    if (n_sb) {
        hipLaunchKernelGGL(rrr_encode_kernel, dim3(grid), dim3(256), 0, stream, blocks, d_tab.as<RrrTable>(), n_sb, d_code.as<RrrTables>(),
                           nullptr, nullptr, d_ones.as<uint64_t>(), d_words.as<uint64_t>(), w.hdr, w.stream);
        VLG_HIP_TRY(hipGetLastError());
    }
    VLG_HIP_TRY(hipStreamSynchronize(stream));
    return VLG_OK;
}

// The integer index (int_index.hpp): every level of the wavelet matrix is one bit-vector of n bits = one "node" of the table; the
// alphabet, Z, D and the samples are copied.  The reference's counterpart: csa_wt<wt_int<rrr_vector<63>>, ., ., ., ., int_alphabet<>>
// (test/csa_int_test.cpp:32).
vlg_status compress_int(const vlg_index* src, vlg_index** out)
{
    const IntHeader& sh = src->ihdr;
    if (sh.bv_kind != kBvPlain) return fail(VLG_E_INVALID, "source index must use plain bit-vectors");
    IntHeader h = sh;
    h.bv_kind = kBvRrr63;
    h.n_sb = sh.n / kRrrSuperBits + 1;
    const uint64_t n_sb = h.n_sb * sh.levels;
    if (n_sb > 0xFFFFFFF0ull) return fail(VLG_E_UNSUPPORTED, "too many rrr super-blocks");
    static_assert(kMaxIntLevels <= 256, "one table entry per level");
    RrrTable tab;
    memset(&tab, 0, sizeof tab);
    tab.count = sh.levels;
    for (uint32_t l = 0; l < sh.levels; ++l) {
        tab.pbase[l] = (uint32_t)(l * sh.nb);
        tab.rbase[l] = (uint32_t)(l * h.n_sb);
        tab.size[l] = sh.n;
    }
    void* d_blob = nullptr;
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(src->d_blob);
    vlg_status st = rrr_encode(src->iview.blocks, tab, n_sb, nullptr, [&](uint64_t total_words, RrrTarget& w) -> vlg_status {
        h.rrr_words = total_words;
        layout_int_blob(h);
        VLG_HIP_TRY(hipMalloc(&d_blob, h.total_bytes));
        uint8_t* b = reinterpret_cast<uint8_t*>(d_blob);
        VLG_HIP_TRY(hipMemset(b, 0, h.off_levels));
        VLG_HIP_TRY(hipMemcpy(b, &h, sizeof h, hipMemcpyHostToDevice));
        VLG_HIP_TRY(hipMemcpy(b + h.off_Z, sb + sh.off_Z, kMaxIntLevels * 8, hipMemcpyDeviceToDevice));
        VLG_HIP_TRY(hipMemcpy(b + h.off_D, sb + sh.off_D, sh.sigma * 8, hipMemcpyDeviceToDevice));
        VLG_HIP_TRY(hipMemcpy(b + h.off_C, sb + sh.off_C, (sh.sigma + 1) * 8, hipMemcpyDeviceToDevice));
        VLG_HIP_TRY(hipMemcpy(b + h.off_c2c, sb + sh.off_c2c, sh.sigma * 4, hipMemcpyDeviceToDevice));
        VLG_HIP_TRY(hipMemcpy(b + h.off_samples, sb + sh.off_samples, sh.n_samples * 4, hipMemcpyDeviceToDevice));
        w.hdr = reinterpret_cast<uint4*>(b + h.off_rrr_hdr);
        w.stream = reinterpret_cast<uint64_t*>(b + h.off_rrr_stream);
        w.tables = b + h.off_binom;
        return VLG_OK;
    });
    if (st) { if (d_blob) (void)hipFree(d_blob); return st; }
    vlg_index* idx = new vlg_index();
    st = attach_int_blob(d_blob, h.total_bytes, idx);
    if (st) { (void)hipFree(d_blob); delete idx; return st; }
    idx->owns_blob = true;
    *out = idx;
    return VLG_OK;
}

}  // namespace

extern "C" vlg_status vlg_index_compress(const vlg_index* src, int kind, vlg_index** out)
{
    release_cached_device_memory();          // an index wants its memory now; parked result buffers can be allocated again

    if (!src || !out) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    if (kind != VLG_BV_RRR63) return fail(VLG_E_INVALID, "unknown bit-vector kind");
    if (src->is_int) return compress_int(src, out);
    if (src->hdr.bv_kind != kBvPlain) return fail(VLG_E_INVALID, "source index must use plain bit-vectors");
    if (src->hdr.sampling != kSamplingSaOrder) return fail(VLG_E_INVALID, "compress the SA-order index first, then resample it (vlg_index_resample)");
    vlg_index* idx = new vlg_index();
    idx->tree = src->tree;
    HostTree& t = idx->tree;
    RrrTable tab;
    memset(&tab, 0, sizeof tab);
    uint64_t n_sb = 0;
    for (uint32_t v = 0; v < t.n_nodes; ++v)
        if (t.nodes[v].child[0] != 0xFFFF) {
            uint32_t k = tab.count++;
            tab.pbase[k] = src->tree.dnodes[v].base;
            tab.rbase[k] = (uint32_t)n_sb;
            tab.size[k] = t.node_size[v];
            t.dnodes[v].base = (uint32_t)n_sb;
            n_sb += t.node_size[v] / kRrrSuperBits + 1;
        }
    if (n_sb > 0xFFFFFFF0ull) { delete idx; return fail(VLG_E_UNSUPPORTED, "too many rrr super-blocks"); }
    hipStream_t stream = nullptr;
    vlg_status st = rrr_encode(src->view.blocks, tab, n_sb, stream, [&](uint64_t total_words, RrrTarget& w) -> vlg_status {
        if (vlg_status s2 = alloc_blob(idx, src->hdr.n, src->hdr.dens, stream, std::max<uint64_t>(n_sb, 1), total_words)) return s2;
        const BlobHeader& h = idx->hdr;
        uint8_t* b = reinterpret_cast<uint8_t*>(idx->d_blob);
        const uint8_t* sb = reinterpret_cast<const uint8_t*>(src->d_blob);
        VLG_HIP_TRY(hipMemcpyAsync(b + h.off_samples, sb + src->hdr.off_samples, h.n_samples * h.sample_bytes, hipMemcpyDeviceToDevice, stream));
        w.hdr = reinterpret_cast<uint4*>(b + h.off_rrr_hdr);
        w.stream = reinterpret_cast<uint64_t*>(b + h.off_rrr_stream);
        w.tables = b + h.off_binom;
        return VLG_OK;
    });
    if (st) { vlg_index_destroy(idx); return st; }
    *out = idx;
    return VLG_OK;
}

// =============================================================================================
// SA sampling strategies: vlg_index_resample makes a second index over the same BWT with another sample density
// and / or text_order_sa_sampling (include/sdsl/csa_sampling_strategy.hpp:127-246: the SA values that are multiples of dens, found
// through a marked bit-vector over the SA indices) instead of sa_order_sa_sampling (:64-112: every dens-th SA index).
// =============================================================================================
namespace {

// every SA value from the SA-order samples: a lane starts at one sample (i, SA[i]) and walks LF -- (LF(i), SA[i] - 1) -- up to the next
// sampled index, so that every SA index is visited exactly once with its value
// (sample_t = sa_t: 32-bit values for n <= 2^32, 64-bit ones for an index with wide SA indices -- SA[0] = n - 1 can be 2^32)
template <class BV, typename sample_t>
__global__ void __launch_bounds__(256) sa_expand_kernel(IndexView iv, sample_t* __restrict__ sa)
{
    __shared__ WalkLds<BV> s;
    stage_walk(s, iv);
    const sample_t* samples = reinterpret_cast<const sample_t*>(iv.samples);
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < iv.n_samples; j += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t i = j * iv.dens, v = samples[j];
        do {
            sa[i] = (sample_t)v;
            uint32_t node = 0, c = 0;
            uint64_t pos = i;
            if (iv.sigma > 1) {
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
            } else i = 0;
            v = v ? v - 1 : iv.n - 1;
        } while (i % iv.dens);
    }
}

// marks of one super-block: bit = (SA[i] % dens == 0); pops[b] = its ones
template <typename sa_t>
__global__ void marked_pack_kernel(const sa_t* __restrict__ sa, uint64_t n, uint32_t dens, Block* __restrict__ blocks, uint64_t nb,
                                   uint32_t* __restrict__ pops)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x) {
        Block B;
        uint32_t pc = 0;
        for (uint32_t w = 0; w < 7; ++w) {
            uint32_t word = 0;
            const uint64_t first = b * kBlockBits + 32ull * w;
            for (uint32_t j = 0; j < 32 && first + j < n; ++j) word |= (sa[first + j] % dens == 0 ? 1u : 0u) << j;
            B.w[w] = word;
            pc += __popc(word);
        }
        B.cnt = 0;
        blocks[b] = B;
        pops[b] = pc;
    }
}
__global__ void marked_counts_kernel(Block* __restrict__ blocks, const uint32_t* __restrict__ before, uint64_t nb)
{
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x) blocks[b].cnt = before[b];
}
// samples[rank_marked(i)] = SA[i] / dens for the marked i (csa_sampling_strategy.hpp:158-164)
template <typename sa_t>
__global__ void text_order_samples_kernel(const sa_t* __restrict__ sa, uint64_t n, uint32_t dens, const Block* __restrict__ blocks,
                                          sa_t* __restrict__ samples)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const sa_t v = sa[i];
        if (v % dens) continue;
        const uint64_t blk = i / kBlockBits;
        const uint32_t off = (uint32_t)(i - blk * kBlockBits);
        const Block& B = blocks[blk];
        uint32_t r = B.cnt;
        for (uint32_t w = 0; w < (off >> 5); ++w) r += __popc(B.w[w]);
        if (off & 31) r += __popc(B.w[off >> 5] & ((1u << (off & 31)) - 1u));
        samples[r] = v / dens;
    }
}
template <typename sa_t>
__global__ void sa_order_samples_kernel(const sa_t* __restrict__ sa, uint64_t n_samples, uint32_t dens, sa_t* __restrict__ samples)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_samples; j += (uint64_t)gridDim.x * blockDim.x) samples[j] = sa[j * dens];
}
// marks as plain words (bit i = word[i >> 6] >> (i & 63)), for export
__global__ void marked_words_kernel(const Block* __restrict__ blocks, uint64_t n, uint64_t* __restrict__ out, uint64_t n_words)
{
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_words; t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t word = 0;
        for (uint32_t j = 0; j < 64; ++j) {
            const uint64_t i = t * 64 + j;
            if (i >= n) break;
            const uint64_t blk = i / kBlockBits;
            const uint32_t off = (uint32_t)(i - blk * kBlockBits);
            word |= (uint64_t)((blocks[blk].w[off >> 5] >> (off & 31)) & 1u) << j;
        }
        out[t] = word;
    }
}

}  // namespace

namespace {
// sa_t = the width the source keeps its samples in (and the new index keeps its own in): uint32_t for n <= 2^32, uint64_t for wide SA indices
template <typename sa_t>
vlg_status resample_run(const vlg_index* src, int sampling, uint32_t dens, vlg_index* idx, hipStream_t stream)
{
    const uint64_t n = src->hdr.n;
    DevBuf d_sa, d_pops, d_tmp;
    VLG_HIP_TRY(d_sa.alloc(n * sizeof(sa_t)));
    const dim3 grid((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((src->view.n_samples + 255) / 256, 8192)));
    if (src->view.bv_kind == kBvRrr63) hipLaunchKernelGGL(HIP_KERNEL_NAME(sa_expand_kernel<RrrBV, sa_t>), grid, dim3(256), 0, stream, src->view, d_sa.as<sa_t>());
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(sa_expand_kernel<PlainBV, sa_t>), grid, dim3(256), 0, stream, src->view, d_sa.as<sa_t>());
    VLG_HIP_TRY(hipGetLastError());
    BlobHeader& h = idx->hdr;
    h = src->hdr;
    h.dens = dens;
    h.sampling = (uint32_t)sampling;
    h.n_samples = (n + dens - 1) / dens;
    layout_blob(h);
    VLG_HIP_TRY(hipMalloc(&idx->d_blob, h.total_bytes));
    idx->owns_blob = true;
    uint8_t* b = reinterpret_cast<uint8_t*>(idx->d_blob);
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(src->d_blob);
    VLG_HIP_TRY(hipMemcpyAsync(b, &h, sizeof h, hipMemcpyHostToDevice, stream));
    BlobSection sec[11], old[11];
    blob_sections(h, sec);
    blob_sections(src->hdr, old);
    for (int i = 0; i < 11; ++i) {
        if (sec[i].off == &BlobHeader::off_samples || sec[i].off == &BlobHeader::off_marked || !sec[i].bytes) continue;
        if (old[i].bytes != sec[i].bytes) return fail(VLG_E_INTERNAL, "blob sections changed size");
        VLG_HIP_TRY(hipMemcpyAsync(b + h.*(sec[i].off), sb + src->hdr.*(old[i].off), sec[i].bytes, hipMemcpyDeviceToDevice, stream));
    }
    sa_t* samples = reinterpret_cast<sa_t*>(b + h.off_samples);
    if (sampling == VLG_SAMPLING_SA_ORDER) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sa_order_samples_kernel<sa_t>), dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((h.n_samples + 255) / 256, 8192))), dim3(256), 0, stream,
                           d_sa.as<sa_t>(), h.n_samples, dens, samples);
    } else {
        const uint64_t nb = n / kBlockBits + 1;
        if (h.n_samples > 0xFFFFFFF0ull) return fail(VLG_E_UNSUPPORTED, "text-order sampling: more than 2^32 marked indices (choose a larger density)");
        Block* mk = reinterpret_cast<Block*>(b + h.off_marked);
        VLG_HIP_TRY(d_pops.alloc((nb + 1) * 4));
        const dim3 gb((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((nb + 255) / 256, 8192)));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(marked_pack_kernel<sa_t>), gb, dim3(256), 0, stream, d_sa.as<sa_t>(), n, dens, mk, nb, d_pops.as<uint32_t>());
        size_t tb = 0;
        VLG_HIP_TRY(rocprim::exclusive_scan(nullptr, tb, d_pops.as<uint32_t>(), d_pops.as<uint32_t>(), 0u, nb, rocprim::plus<uint32_t>(), stream));
        VLG_HIP_TRY(d_tmp.alloc(tb + 16));
        VLG_HIP_TRY(rocprim::exclusive_scan(d_tmp.p, tb, d_pops.as<uint32_t>(), d_pops.as<uint32_t>(), 0u, nb, rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(marked_counts_kernel, gb, dim3(256), 0, stream, mk, d_pops.as<uint32_t>(), nb);
        VLG_HIP_TRY(hipMemsetAsync(samples, 0, h.n_samples * sizeof(sa_t), stream));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(text_order_samples_kernel<sa_t>), dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, 16384))), dim3(256), 0, stream,
                           d_sa.as<sa_t>(), n, dens, mk, samples);
    }
    VLG_HIP_TRY(hipGetLastError());
    VLG_HIP_TRY(hipStreamSynchronize(stream));
    bind_view(idx);
    return VLG_OK;
}
}  // namespace

extern "C" vlg_status vlg_index_resample(const vlg_index* src, int sampling, uint32_t dens, vlg_index** out)
{
    release_cached_device_memory();
    if (!src || !out) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    if (sampling != VLG_SAMPLING_SA_ORDER && sampling != VLG_SAMPLING_TEXT_ORDER) return fail(VLG_E_INVALID, "unknown sampling strategy");
    if (!dens) dens = 32;
    if (src->is_int) return fail(VLG_E_UNSUPPORTED, "resampling is built for byte-alphabet indexes");
    if (src->hdr.sampling != kSamplingSaOrder) return fail(VLG_E_INVALID, "the source index must be sampled in SA order");
    if (src->hdr.sample_bytes != 4 && src->hdr.sample_bytes != 8) return fail(VLG_E_INTERNAL, "unknown sample width");
    vlg_index* idx = new vlg_index();
    idx->tree = src->tree;
    hipStream_t stream = nullptr;
    // (the new index keeps the source's sample width: 8 bytes when SA indices are wide -- n > 2^32, or VLG_FORCE_POS64)
    const vlg_status st = src->hdr.sample_bytes == 8 ? resample_run<uint64_t>(src, sampling, dens, idx, stream) : resample_run<uint32_t>(src, sampling, dens, idx, stream);
    if (st) { vlg_index_destroy(idx); return st; }
    *out = idx;
    return VLG_OK;
}

extern "C" vlg_status vlg_index_export_marked(const vlg_index* idx, uint64_t* h_words)
{
    if (!idx || !h_words) return fail(VLG_E_INVALID, "null argument");
    if (idx->hdr.sampling != kSamplingTextOrder) return fail(VLG_E_INVALID, "the index is not sampled in text order");
    const uint64_t n = idx->hdr.n, nw = (n + 63) / 64;
    DevBuf d;
    VLG_HIP_TRY(d.alloc(nw * 8));
    hipLaunchKernelGGL(marked_words_kernel, dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((nw + 255) / 256, 8192))), dim3(256), 0, nullptr, idx->view.marked, n,
                       d.as<uint64_t>(), nw);
    VLG_HIP_TRY(hipGetLastError());
    VLG_HIP_TRY(hipMemcpy(h_words, d.p, nw * 8, hipMemcpyDeviceToHost));
    return VLG_OK;
}

// =============================================================================================
// On-device builder.
// =============================================================================================
namespace {

__global__ void count_zero_kernel(const uint8_t* __restrict__ text, uint64_t n_text, unsigned long long* __restrict__ zeros)
{
    unsigned long long local = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_text; i += (uint64_t)gridDim.x * blockDim.x)
        local += text[i] == 0;
    if (local) atomicAdd(zeros, local);
}

// T' = text + 0 sentinel (construct.hpp:47-52); key = first 8 bytes of suffix i, big-endian.
__global__ void sa_init_keys(const uint8_t* __restrict__ text, uint64_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t k = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t p = i + j;
            uint64_t c = (p < n) ? text[p] : 0;          // the sentinel and everything behind it
            k = (k << 8) | c;
        }
        keys[i] = k;
        vals[i] = (uint32_t)i;
    }
}

// head[j] = 1 if sorted key j starts a new group else 0;  *n_groups += #group heads.  The running sum of the flags minus one
// is the dense rank of the group, which stays below 2^32 - 1 while any group still has two members.
__global__ void sa_group_heads(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ head,
                               unsigned long long* __restrict__ n_groups)
{
    unsigned long long local = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        bool h = (j == 0) || keys[j] != keys[j - 1];
        head[j] = h ? 1u : 0u;
        local += h;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(n_groups, local);
}

__global__ void sa_scatter_rank(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ head, uint64_t n,
                                uint32_t* __restrict__ rank_of)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x)
        rank_of[sa[j]] = head[j] - 1u;          // head[] holds the inclusive running count of group heads
}

// Second key: rank of the suffix h further on, +1, or 0 when that is the sentinel suffix or beyond.  Ranks are dense
// group numbers, at most n_groups - 1 <= ns - 2 while the loop runs, so both halves of the key fit 32 bits up to ns = 2^32.
__global__ void sa_next_keys(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ rank_of, uint64_t n, uint64_t h,
                             uint64_t* __restrict__ keys)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t s = sa[j];
        uint64_t r1 = rank_of[s];
        uint64_t r2 = (s + h < n) ? (uint64_t)rank_of[s + h] + 1 : 0;
        keys[j] = (r1 << 32) | r2;
    }
}

// bwt[j] = T'[SA[j]-1] (construct_bwt.hpp:71-75); also the symbol histogram
__global__ void bwt_kernel(const uint8_t* __restrict__ text, const uint32_t* __restrict__ sa, uint64_t n,
                           uint8_t* __restrict__ bwt, unsigned long long* __restrict__ hist)
{
    __shared__ unsigned int h[256];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        // SA[0] = n-1 (the sentinel suffix), SA[j] = sa[j-1] otherwise
        uint8_t c;
        if (j == 0) c = n > 1 ? text[n - 2] : 0;
        else { uint32_t s = sa[j - 1]; c = s ? text[s - 1] : 0; }
        bwt[j] = c;
        atomicAdd(&h[c], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

__global__ void sample_kernel(const uint32_t* __restrict__ sa, uint64_t n_text, uint64_t n_samples, uint32_t dens, void* __restrict__ out,
                              uint32_t sample_bytes)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_samples; j += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v = j ? (uint64_t)sa[j * dens - 1] : n_text;   // csa_sampling_strategy.hpp:92-98; SA[0] = n-1
        if (sample_bytes == 4) reinterpret_cast<uint32_t*>(out)[j] = (uint32_t)v;
        else reinterpret_cast<uint64_t*>(out)[j] = v;
    }
}

struct LevelTables {                 // per-symbol lookup for one depth of the wavelet tree
    uint16_t key_next[256];          // node id of the ancestor at depth d+1 if the code is longer, else 511 (dead)
    uint8_t bit[256];                // code bit at depth d
    uint32_t inner_of[256];          // index into InnerTable of the depth-d ancestor (valid when alive at d)
};

// keys for the stable partition that produces the arrangement of depth d+1 from depth d
__global__ void wt_keys_kernel(const uint8_t* __restrict__ syms, uint64_t count, const LevelTables* __restrict__ lt,
                               uint16_t* __restrict__ keys)
{
    __shared__ uint16_t kn[256];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) kn[i] = lt->key_next[i];
    __syncthreads();
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x)
        keys[j] = kn[syms[j]];
}

// Emit the data words of every block of every inner node at depth d from the depth-d arrangement
// (symbols grouped by node in BFS order, BWT order inside a node).  One thread per u32 data word.
__global__ void wt_emit_kernel(const uint8_t* __restrict__ syms, Block* __restrict__ blocks, const InnerTable* __restrict__ tab,
                               const LevelTables* __restrict__ lt, uint32_t first_inner, uint32_t n_inner_at_depth,
                               uint64_t first_block, uint64_t n_blocks_at_depth, const uint64_t* __restrict__ arr_start)
{
    __shared__ InnerTable t;
    __shared__ uint8_t bit[256];
    for (uint32_t i = threadIdx.x; i < sizeof(InnerTable) / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(&t)[i] = reinterpret_cast<const uint32_t*>(tab)[i];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) bit[i] = lt->bit[i];
    __syncthreads();
    (void)n_inner_at_depth;
    uint64_t total = n_blocks_at_depth * 8;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t blk = (uint32_t)(first_block + (g >> 3)), w = (uint32_t)(g & 7);
        if (w == 7) continue;
        uint32_t k = find_inner_by_block(t, blk);
        uint64_t rel = (uint64_t)(blk - t.base[k]) * kBlockBits + 32u * w;
        uint32_t val = 0;
        if (rel < t.size[k]) {
            uint64_t left = t.size[k] - rel;
            uint32_t m = left < 32 ? (uint32_t)left : 32u;
            const uint8_t* s = syms + arr_start[k - first_inner] + rel;
            for (uint32_t j = 0; j < m; ++j) val |= (uint32_t)bit[s[j]] << j;
        }
        blocks[blk].w[w] = val;
    }
}

template <class K, class V>
vlg_status sort_pairs(DevBuf& temp, size_t& temp_cap, K* kin, K* kout, V* vin, V* vout, uint64_t n, unsigned b0, unsigned b1,
                      hipStream_t stream)
{
    size_t tb = 0;
    VLG_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, n, b0, b1, stream));
    if (tb > temp_cap) {
        if (temp.p) { (void)hipFree(temp.p); temp.p = nullptr; }
        VLG_HIP_TRY(temp.alloc(tb));
        temp_cap = tb;
    }
    VLG_HIP_TRY(rocprim::radix_sort_pairs(temp.p, tb, kin, kout, vin, vout, n, b0, b1, stream));
    return VLG_OK;
}

inline uint32_t grid_for(uint64_t n) { return (uint32_t)std::min<uint64_t>((n + 255) / 256, 16384); }

__global__ void sa_export_kernel(const uint32_t* __restrict__ sa, uint64_t n_text, uint32_t* __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_text; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = i ? sa[i - 1] : (uint32_t)n_text;          // the sentinel suffix sorts first
}

// d_sa_out (optional): the suffix array of text + sentinel, n_text + 1 entries (needs n_text < 2^32); sa_only: stop there.
vlg_status build_on_device(const uint8_t* d_text, uint64_t n_text, uint32_t dens, hipStream_t stream, vlg_index** out,
                           uint32_t* d_sa_out = nullptr, bool sa_only = false)
{
    const uint64_t n = n_text + 1;
    // The sentinel suffix is the smallest one by definition (SA[0] = n-1), so only the n_text proper suffixes are sorted:
    // their ids and ranks fit 32 bits for texts up to 2^32 bytes (BASELINE config 4).
    const uint64_t ns = n_text;
    if (n_text > 0x100000000ull) return fail(VLG_E_UNSUPPORTED, "device builder handles texts of up to 2^32 bytes");
    vlg_index* idx = new vlg_index();
    auto run = [&]() -> vlg_status {
        DevBuf keys_a, keys_b, sa_a, sa_b, rank_of, head, temp, counter, bwt;
        size_t temp_cap = 0;
        VLG_HIP_TRY(keys_a.alloc(ns * 8));
        VLG_HIP_TRY(keys_b.alloc(ns * 8));
        VLG_HIP_TRY(sa_a.alloc(ns * 4));
        VLG_HIP_TRY(sa_b.alloc(ns * 4));
        VLG_HIP_TRY(rank_of.alloc(ns * 4));
        VLG_HIP_TRY(head.alloc(ns * 4));
        VLG_HIP_TRY(counter.alloc(8 + 256 * 8));
        unsigned long long* d_groups = counter.as<unsigned long long>();
        unsigned long long* d_hist = d_groups + 1;
        const uint32_t g = grid_for(n);
        {   // a zero byte in the text is a std::logic_error in the reference (construct.hpp:36-45)
            VLG_HIP_TRY(hipMemsetAsync(d_groups, 0, 8, stream));
            hipLaunchKernelGGL(count_zero_kernel, dim3(std::min<uint32_t>(g, 4096)), dim3(256), 0, stream, d_text, n_text, d_groups);
            unsigned long long zeros = 0;
            VLG_HIP_TRY(hipMemcpyAsync(&zeros, d_groups, 8, hipMemcpyDeviceToHost, stream));
            VLG_HIP_TRY(hipStreamSynchronize(stream));
            if (zeros) return fail(VLG_E_ZERO_BYTE, "text contains a zero byte (sdsl::construct throws std::logic_error)");
        }
        // --- suffix array by prefix doubling (8 characters first, then h = 8, 16, ...) ------------
        uint32_t* sa_cur = sa_b.as<uint32_t>();
        if (ns) {
            hipLaunchKernelGGL(sa_init_keys, dim3(g), dim3(256), 0, stream, d_text, ns, keys_a.as<uint64_t>(), sa_a.as<uint32_t>());
            VLG_HIP_TRY(hipGetLastError());
            if (vlg_status st = sort_pairs(temp, temp_cap, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), sa_a.as<uint32_t>(),
                                           sa_b.as<uint32_t>(), ns, 0, 64, stream)) return st;
            uint64_t* keys_sorted = keys_b.as<uint64_t>();
            uint64_t* keys_other = keys_a.as<uint64_t>();
            uint32_t* sa_other = sa_a.as<uint32_t>();
            for (uint64_t h = 8;; h <<= 1) {
                VLG_HIP_TRY(hipMemsetAsync(d_groups, 0, 8, stream));
                hipLaunchKernelGGL(sa_group_heads, dim3(g), dim3(256), 0, stream, keys_sorted, ns, head.as<uint32_t>(), d_groups);
                VLG_HIP_TRY(hipGetLastError());
                unsigned long long groups = 0;
                VLG_HIP_TRY(hipMemcpyAsync(&groups, d_groups, 8, hipMemcpyDeviceToHost, stream));
                VLG_HIP_TRY(hipStreamSynchronize(stream));
                if (groups == ns) break;
                if (h > 2 * n) return fail(VLG_E_INTERNAL, "suffix sort did not converge");
                size_t tb = 0;
                VLG_HIP_TRY(rocprim::inclusive_scan(nullptr, tb, head.as<uint32_t>(), head.as<uint32_t>(), ns, rocprim::plus<uint32_t>(), stream));
                if (tb > temp_cap) { if (temp.p) { (void)hipFree(temp.p); temp.p = nullptr; } VLG_HIP_TRY(temp.alloc(tb)); temp_cap = tb; }
                VLG_HIP_TRY(rocprim::inclusive_scan(temp.p, tb, head.as<uint32_t>(), head.as<uint32_t>(), ns, rocprim::plus<uint32_t>(), stream));
                hipLaunchKernelGGL(sa_scatter_rank, dim3(g), dim3(256), 0, stream, sa_cur, head.as<uint32_t>(), ns, rank_of.as<uint32_t>());
                hipLaunchKernelGGL(sa_next_keys, dim3(g), dim3(256), 0, stream, sa_cur, rank_of.as<uint32_t>(), ns, h, keys_other);
                VLG_HIP_TRY(hipGetLastError());
                if (vlg_status st = sort_pairs(temp, temp_cap, keys_other, keys_sorted, sa_cur, sa_other, ns, 0, 64, stream)) return st;
                std::swap(sa_cur, sa_other);                   // keys_sorted now holds the sorted keys again
            }
        }
        if (d_sa_out) {
            hipLaunchKernelGGL(sa_export_kernel, dim3(g), dim3(256), 0, stream, sa_cur, n_text, d_sa_out);
            VLG_HIP_TRY(hipGetLastError());
            VLG_HIP_TRY(hipStreamSynchronize(stream));
        }
        if (sa_only) return VLG_OK;
        // free the big sort buffers we no longer need
        (void)hipFree(keys_a.p); keys_a.p = nullptr;
        (void)hipFree(keys_b.p); keys_b.p = nullptr;
        (void)hipFree(rank_of.p); rank_of.p = nullptr;
        (void)hipFree(head.p); head.p = nullptr;
        // --- BWT + alphabet ------------------------------------------------------------------------
        VLG_HIP_TRY(bwt.alloc(n));
        VLG_HIP_TRY(hipMemsetAsync(d_hist, 0, 256 * 8, stream));
        hipLaunchKernelGGL(bwt_kernel, dim3(std::min<uint32_t>(g, 4096)), dim3(256), 0, stream, d_text, sa_cur, n, bwt.as<uint8_t>(), d_hist);
        VLG_HIP_TRY(hipGetLastError());
        uint64_t counts[256];
        VLG_HIP_TRY(hipMemcpyAsync(counts, d_hist, sizeof counts, hipMemcpyDeviceToHost, stream));
        VLG_HIP_TRY(hipStreamSynchronize(stream));
        if (counts[0] != 1) return fail(VLG_E_ZERO_BYTE, "text contains a zero byte (sdsl::construct throws std::logic_error)");
        if (vlg_status st = tree_from_counts(counts, idx->tree)) return st;
        if (vlg_status st = alloc_blob(idx, n, dens, stream)) return st;
        const BlobHeader& hd = idx->hdr;
        uint8_t* blob = reinterpret_cast<uint8_t*>(idx->d_blob);
        // --- SA samples ----------------------------------------------------------------------------
        hipLaunchKernelGGL(sample_kernel, dim3(grid_for(hd.n_samples)), dim3(256), 0, stream, sa_cur, n_text, hd.n_samples, dens,
                           blob + hd.off_samples, hd.sample_bytes);
        VLG_HIP_TRY(hipGetLastError());
        VLG_HIP_TRY(hipStreamSynchronize(stream));
        (void)hipFree(sa_a.p); sa_a.p = nullptr;
        (void)hipFree(sa_b.p); sa_b.p = nullptr;
        // --- wavelet tree, one depth at a time -------------------------------------------------------
        const HostTree& t = idx->tree;
        if (hd.n_blocks) {
            InnerTable tab = make_inner_table(t);
            DevBuf d_tab, d_lt, d_arr, syms_b, keys16_a, keys16_b;
            VLG_HIP_TRY(d_tab.alloc(sizeof tab));
            VLG_HIP_TRY(hipMemcpy(d_tab.p, &tab, sizeof tab, hipMemcpyHostToDevice));
            VLG_HIP_TRY(d_lt.alloc(sizeof(LevelTables)));
            VLG_HIP_TRY(d_arr.alloc(256 * 8));
            VLG_HIP_TRY(syms_b.alloc(n));
            VLG_HIP_TRY(keys16_a.alloc(n * 2));
            VLG_HIP_TRY(keys16_b.alloc(n * 2));
            uint8_t* cur = bwt.as<uint8_t>();
            uint8_t* other = syms_b.as<uint8_t>();
            uint64_t alive = n;                              // symbols whose code is longer than d
            Block* blocks = const_cast<Block*>(idx->view.blocks);
            for (uint32_t d = 0; d < t.max_code_len; ++d) {
                // inner nodes of this depth are consecutive in BFS order
                uint32_t first_inner = 0, n_inner = 0;
                for (uint32_t k = 0; k < tab.count; ++k)
                    if (tab.depth[k] == d) { if (!n_inner) first_inner = k; ++n_inner; }
                if (!n_inner) break;
                LevelTables lt;
                memset(&lt, 0, sizeof lt);
                uint64_t alive_next = 0;
                for (uint32_t c = 0; c < 256; ++c) {
                    lt.key_next[c] = 511;
                    if (t.c_to_leaf[c] == 0xFFFF) continue;
                    uint32_t len = (uint32_t)(t.paths[c] >> 56);
                    if (len <= d) continue;
                    lt.bit[c] = (uint8_t)((t.paths[c] >> d) & 1);
                    if (len > d + 1) {
                        uint32_t v = 0;
                        for (uint32_t l = 0; l <= d; ++l) v = t.nodes[v].child[(t.paths[c] >> l) & 1];
                        lt.key_next[c] = (uint16_t)v;
                        uint32_t cc = t.char2comp[c];
                        alive_next += t.C[cc + 1] - t.C[cc];
                    }
                }
                std::vector<uint64_t> arr_start(n_inner);
                uint64_t acc = 0, first_block = tab.base[first_inner], last_block = 0;
                for (uint32_t k = 0; k < n_inner; ++k) {
                    arr_start[k] = acc;
                    acc += tab.size[first_inner + k];
                    last_block = (uint64_t)tab.base[first_inner + k] + tab.size[first_inner + k] / kBlockBits + 1;
                }
                if (acc != alive) return fail(VLG_E_INTERNAL, "wavelet tree level size mismatch");
                VLG_HIP_TRY(hipMemcpyAsync(d_lt.p, &lt, sizeof lt, hipMemcpyHostToDevice, stream));
                VLG_HIP_TRY(hipMemcpyAsync(d_arr.p, arr_start.data(), n_inner * 8, hipMemcpyHostToDevice, stream));
                uint64_t nb = last_block - first_block;
                hipLaunchKernelGGL(wt_emit_kernel, dim3(grid_for(nb * 8)), dim3(256), 0, stream, cur, blocks, d_tab.as<InnerTable>(),
                                   d_lt.as<LevelTables>(), first_inner, n_inner, first_block, nb, d_arr.as<uint64_t>());
                VLG_HIP_TRY(hipGetLastError());
                if (alive_next) {
                    hipLaunchKernelGGL(wt_keys_kernel, dim3(grid_for(alive)), dim3(256), 0, stream, cur, alive, d_lt.as<LevelTables>(),
                                       keys16_a.as<uint16_t>());
                    VLG_HIP_TRY(hipGetLastError());
                    if (vlg_status st = sort_pairs(temp, temp_cap, keys16_a.as<uint16_t>(), keys16_b.as<uint16_t>(), cur, other,
                                                   alive, 0, 9, stream)) return st;
                    std::swap(cur, other);
                }
                VLG_HIP_TRY(hipStreamSynchronize(stream));   // lt / arr_start are reused next iteration
                alive = alive_next;
                if (!alive) break;
            }
            if (vlg_status st = fill_block_counts(idx, d_tab.as<InnerTable>(), stream)) return st;
        }
        VLG_HIP_TRY(hipStreamSynchronize(stream));
        return VLG_OK;
    };
    vlg_status st = run();
    if (st || sa_only) { vlg_index_destroy(idx); return st; }
    *out = idx;
    return VLG_OK;
}

}  // namespace

extern "C" vlg_status vlg_suffix_array_device(const uint8_t* d_text, uint64_t n_text, uint32_t* d_sa, void* stream)
{
    if (!d_sa || (n_text && !d_text)) return fail(VLG_E_INVALID, "null argument");
    if (n_text >= 0xFFFFFFFFull) return fail(VLG_E_UNSUPPORTED, "32-bit suffix array: text too long");
    if (vlg_status st = check_device()) return st;
    release_cached_device_memory();
    vlg_index* none = nullptr;
    return build_on_device(d_text, n_text, 32, (hipStream_t)stream, &none, d_sa, true);
}

extern "C" vlg_status vlg_index_build_device(const uint8_t* d_text, uint64_t n_text, uint32_t dens, void* stream, vlg_index** out)
{
    release_cached_device_memory();          // an index wants its memory now; parked result buffers can be allocated again

    if (!out || (n_text && !d_text)) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    if (vlg_status st = check_device()) return st;
    if (dens == 0) dens = 32;
    return build_on_device(d_text, n_text, dens, (hipStream_t)stream, out);
}

extern "C" vlg_status vlg_index_build(const uint8_t* h_text, uint64_t n_text, uint32_t dens, vlg_index** out)
{
    if (!out || (n_text && !h_text)) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    if (vlg_status st = check_device()) return st;
    DevBuf d;
    VLG_HIP_TRY(d.alloc(n_text + 16));
    if (n_text) VLG_HIP_TRY(hipMemcpy(d.p, h_text, n_text, hipMemcpyHostToDevice));
    return vlg_index_build_device(d.as<uint8_t>(), n_text, dens, nullptr, out);
}
