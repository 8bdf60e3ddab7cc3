// Shared declarations of the MI355X VLG library (device layout, error plumbing).
//
// HBM layout of the index ("blob": ONE contiguous device allocation, 256-byte aligned sections):
//
//   [BlobHeader][blocks][nodes][C][paths][char2comp][samples][reference-layout nodes]
//
// blocks : the Huffman-shaped wavelet tree of the BWT, level-contiguous like the reference
//          (BFS node order, include/sdsl/wt_helper.hpp:170-206) but re-cut into 256-bit
//          super-blocks  { 7 x u32 data = 224 bits, u32 cnt }.  Every inner node starts on a
//          block boundary and owns size/224 + 1 blocks; cnt = number of 1s of THIS NODE before the
//          block (node-relative, so bv_pos_rank(v) of wt_pc.hpp:364 is folded away and the count
//          fits 32 bits for n <= 2^32).  One rank = one aligned 32-byte read.
// nodes  : DNode per tree node {first block, child[0], child[1]}; a child word with bit 31 set is a
//          leaf and carries the alphabet rank (comp) of its symbol in the low bits.
// C      : u64[sigma+1] cumulative counts (lib/csa_alphabet_strategy.cpp:25-55)
// paths  : u64[256] m_path of wt_helper.hpp:219-240 (code bits, first edge in bit 0; length in 56..63)
// samples: SA[0], SA[d], SA[2d], ... as u32 (n <= 2^32) or u64 (sa_order_sa_sampling, the default); or, for
//          text_order_sa_sampling (csa_sampling_strategy.hpp:127-246), SA[i] / d of the SA indices i with SA[i] % d == 0 in
//          ascending i, plus [marked]: super-blocks {224 marks, ones before} over the SA indices -- is_sampled(i) and the rank
//          that addresses the sample come out of ONE 32-byte read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <new>
#include <string>
#include <vector>
#include "../../include/vlg_hip.h"

namespace vlg {

constexpr uint32_t kBlockBits = 224;
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint64_t kBlobMagic = 0x32424C4756ULL;  // "VGLB2" (2: rrr offsets numbered by halves, rrr_code.hpp)
constexpr uint32_t kMaxNodes = 512;

struct __attribute__((aligned(16))) DNode {
    uint32_t base;      // first 32-byte block of the node (inner nodes only)
    uint32_t child[2];  // node id, or kLeafFlag | comp
    uint32_t size_lo;   // node size in bits (low 32 bits; informational)
};

struct __attribute__((aligned(32))) Block {
    uint32_t w[7];
    uint32_t cnt;
};

struct BlobHeader {
    uint64_t magic;
    uint64_t total_bytes;
    uint64_t n;
    uint64_t wt_bits;
    uint64_t n_blocks;
    uint64_t n_samples;
    uint32_t sigma;
    uint32_t dens;
    uint32_t n_nodes;
    uint32_t max_code_len;
    uint32_t sample_bytes;
    uint32_t sampling;            // 0 = SA order (SA[0], SA[d], ...: _sa_order_sampling), 1 = text order (_text_order_sampling: a marked
                                  // bit-vector over SA indices + SA / d of the marked ones)
    uint64_t off_blocks, off_nodes, off_C, off_paths, off_c2c, off_samples, off_refnodes;
    uint64_t bv_kind;             // 0 = plain 256-bit super-blocks, 1 = rrr-63 (headers + offset stream as K6, block code of rrr_code.hpp)
    uint64_t off_rrr_hdr, off_rrr_stream, off_binom, n_rrr_sb, rrr_stream_words;
    uint64_t off_marked;          // text-order sampling: n / 224 + 1 super-blocks {224 marks, ones before} over the SA indices
};

// What kernels receive (by value).
struct IndexView {
    const Block* blocks;
    const DNode* nodes;
    const uint64_t* C;
    const uint64_t* paths;
    const uint8_t* char2comp;
    const void* samples;
    uint64_t n;
    uint64_t n_samples;
    uint32_t n_nodes;
    uint32_t sigma;
    uint32_t dens;
    uint32_t sample_bytes;
    // rrr-63 variant of the wavelet-tree bit-vectors (bv_kind == 1): DNode.base counts 32-byte headers
    uint32_t bv_kind;
    uint32_t sampling;                    // kSamplingSaOrder / kSamplingTextOrder
    const Block* marked;                  // text-order sampling: marks over the SA indices (samples = SA / dens of the marked ones)
    const uint4* rrr_hdr;
    const uint64_t* rrr_stream;
    const struct RrrTables* rrr_tables;   // rrr_code.hpp
};

// ---- FM-index of an integer text (int_index.hpp): its own blob, recognised by its magic -------------------------------------
constexpr uint64_t kIntBlobMagic = 0x3149474C56ULL;   // "VLGI1"
constexpr uint32_t kMaxIntLevels = 32;
struct IntHeader {
    uint64_t magic, total_bytes;                      // (the first two words as in BlobHeader: what export / attach look at)
    uint64_t n, sigma, n_samples, nb;                 // nb = super-blocks per level
    uint32_t levels, dens;
    uint64_t off_levels, off_Z, off_D, off_C, off_c2c, off_samples;
    // rrr-63 levels (vlg_index_compress of an integer index; all zero in a plain one -- the header is followed by zeros up to 256 bytes,
    // so blobs written before these fields existed read as plain): the layout of the byte index's rrr variant, one "node" per level
    uint64_t bv_kind, n_sb /* super-blocks of 32 x 63 bits per level */, rrr_words, off_rrr_hdr, off_rrr_stream, off_binom;
};
struct IntView {
    const Block* blocks;                              // plain: [n_levels][nb] wavelet matrix of the BWT over compact symbols
    const uint4* rrr_hdr;                             // rrr-63: [n_levels][n_sb] headers, the offset stream and the block code's tables
    const uint64_t* rrr_stream;                       //         (the members the bit-vector policies of device_rank.hpp read)
    const struct RrrTables* rrr_tables;
    uint64_t stride;                                  // super-blocks per level: nb (plain) or n_sb (rrr)
    uint32_t bv_kind, pad_;
    const uint64_t* Z;                                // zeros per level
    const uint64_t* D;                                // C[c] - first position of c in the last arrangement
    const uint64_t* C;                                // [sigma + 1]
    const uint32_t* comp2char;                        // [sigma] ascending
    const uint32_t* samples;                          // SA[0], SA[dens], ...
    uint64_t n, nb, sigma, n_samples;
    uint32_t n_levels, dens;
};

constexpr uint32_t kBvPlain = 0, kBvRrr63 = 1;
constexpr uint32_t kSamplingSaOrder = 0, kSamplingTextOrder = 1;
constexpr uint32_t kRrrBlockBits = 63, kRrrBlocksPerSuper = 32, kRrrSuperBits = 63 * 32;

void set_error(const std::string& msg);
vlg_status fail(vlg_status st, const std::string& msg);
void release_cached_device_memory();      // result buffers parked for reuse (search.hip) go back to the driver

#define VLG_HIP_TRY(expr)                                                                        \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return ::vlg::fail(_e == hipErrorOutOfMemory ? VLG_E_OOM : VLG_E_NO_DEVICE,          \
                               std::string(#expr) + ": " + hipGetErrorString(_e));               \
    } while (0)

// Host memory the batch path copies from / into.  Pageable vectors there are pinned by the runtime on the fly, and
// freeing them afterwards (munmap) makes the driver evict and restore the process's GPU queues: tens of ms in which the
// next kernel does not start.  These vectors live in pinned blocks owned by the workspace instead: handed out bump-style
// while a batch runs, recycled at the next batch, never unmapped in between.
struct HostPool {
    struct Blk { uint8_t* p; size_t cap; bool pinned; };
    std::vector<Blk> blocks;
    size_t cur = 0, off = 0;
    void* take(size_t bytes);              // 64-byte aligned; pageable blocks (kept just as long) when pinned memory cannot be had
    void reset() { cur = 0; off = 0; }
    void release();                        // back to the driver (workspace destruction)
};
HostPool*& host_pool_slot();               // pool of the batch this thread is running (nullptr outside a batch)
struct HostPoolScope {
    HostPool* prev;
    explicit HostPoolScope(HostPool* p) : prev(host_pool_slot()) { host_pool_slot() = p; p->reset(); }
    ~HostPoolScope() { host_pool_slot() = prev; }
};
template <class T>
struct StageAlloc {
    using value_type = T;
    StageAlloc() = default;
    template <class U> StageAlloc(const StageAlloc<U>&) {}
    T* allocate(size_t n)
    {
        HostPool* hp = host_pool_slot();
        void* p = hp ? hp->take(n * sizeof(T)) : nullptr;
        if (!p) throw std::bad_alloc();
        return (T*)p;
    }
    void deallocate(T*, size_t) noexcept {}
    template <class U> bool operator==(const StageAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const StageAlloc<U>&) const { return false; }
};
template <class T> using svec = std::vector<T, StageAlloc<T>>;   // a vector whose storage is staged for device copies
// ... and one whose resize(n) leaves the new elements as they are (arrays that a device copy fills completely)
template <class T>
struct StageAllocRaw : StageAlloc<T> {
    using value_type = T;
    StageAllocRaw() = default;
    template <class U> StageAllocRaw(const StageAllocRaw<U>&) {}
    template <class U> struct rebind { using other = StageAllocRaw<U>; };
    template <class U> void construct(U*) noexcept {}
    template <class U, class... Args> void construct(U* p, Args&&... args) { ::new ((void*)p) U(std::forward<Args>(args)...); }
};
template <class T> using rvec = std::vector<T, StageAllocRaw<T>>;

inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
inline uint32_t bit_width64(uint64_t x) { return x ? 64u - (uint32_t)__builtin_clzll(x) : 0u; }

// Host-side description of the tree, shared by from_parts and the device builder.
struct HostTree {
    uint32_t n_nodes = 0;
    uint32_t sigma = 0;
    uint32_t max_code_len = 0;
    std::vector<vlg_wt_node> nodes;     // reference layout
    std::vector<uint64_t> node_size;    // bits per inner node (0 for leaves)
    std::vector<uint32_t> node_depth;
    std::vector<uint64_t> paths;        // [256]
    std::vector<uint16_t> c_to_leaf;    // [256]
    std::vector<DNode> dnodes;          // device layout
    uint64_t wt_bits = 0;
    uint64_t n_blocks = 0;
    uint8_t char2comp[256];
    std::vector<uint64_t> C;            // [sigma+1]
};

// Huffman shape + BFS layout from symbol counts (restates wt_huff.hpp:91-117, wt_helper.hpp:170-241).
vlg_status tree_from_counts(const uint64_t counts[256], HostTree& t);
// Adopt reference nodes (from a loaded index).
vlg_status tree_from_nodes(const vlg_wt_node* nodes, uint32_t n_nodes, uint64_t bv_bits, const uint8_t* char2comp,
                           const uint64_t* C, uint32_t sigma, HostTree& t);

}  // namespace vlg

struct vlg_index {
    void* d_blob = nullptr;
    bool owns_blob = true;
    vlg::BlobHeader hdr;
    vlg::IndexView view;
    vlg::HostTree tree;   // host copy (node table, C, ...) for export and planning
    bool is_int = false;  // integer-alphabet FM-index (int_index.hpp): ihdr / iview are valid, hdr carries the generic fields only
    vlg::IntHeader ihdr;
    vlg::IntView iview;
};

namespace vlg {
vlg_status attach_int_blob(const void* d_blob, uint64_t bytes, vlg_index* idx);   // int_index.hpp
void layout_int_blob(IntHeader& h);                                               // int_index.hpp: offsets and total_bytes from the sizes
}
