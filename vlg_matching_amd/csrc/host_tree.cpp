// Host-side logic of the library: error text, the prefix-code tree, HBM block layout planning.
#include "common.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <queue>

namespace vlg {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
vlg_status fail(vlg_status st, const std::string& msg)
{
    g_err = msg;
    return st;
}
const char* last_error_cstr() { return g_err.c_str(); }

// ---- pinned staging memory of a workspace (common.hpp) ----------------------------------------------------------
HostPool*& host_pool_slot()
{
    static thread_local HostPool* slot = nullptr;
    return slot;
}
void* HostPool::take(size_t bytes)
{
    bytes = (bytes + 63) & ~(size_t)63;
    if (!bytes) bytes = 64;
    while (cur < blocks.size()) {
        if (off + bytes <= blocks[cur].cap) { void* p = blocks[cur].p + off; off += bytes; return p; }
        ++cur; off = 0;                                 // the rest of the block is left unused until the next batch
    }
    Blk b{nullptr, std::max<size_t>(bytes, (size_t)64 << 20), true};
    if (hipHostMalloc((void**)&b.p, b.cap, hipHostMallocDefault) != hipSuccess) {
        // no pinned memory to be had: a pageable block that also stays until the workspace goes (what must not happen is an
        // unmap between batches)
        (void)hipGetLastError();
        b.pinned = false;
        b.p = (uint8_t*)malloc(b.cap);
        if (!b.p) return nullptr;
    }
    blocks.push_back(b);
    cur = blocks.size() - 1;
    off = bytes;
    return b.p;
}
void HostPool::release()
{
    for (Blk& b : blocks) { if (b.pinned) (void)hipHostFree(b.p); else free(b.p); }
    blocks.clear();
    cur = off = 0;
}

// Fill everything of HostTree that follows from t.nodes (reference layout) + char2comp.
static vlg_status finish_tree(HostTree& t)
{
    const uint32_t nn = t.n_nodes;
    if (nn > kMaxNodes) return fail(VLG_E_INVALID, "wavelet tree has more than 511 nodes");
    t.node_size.assign(nn, 0);
    t.node_depth.assign(nn, 0);
    t.paths.assign(256, 0);
    t.c_to_leaf.assign(256, 0xFFFF);
    t.dnodes.assign(nn ? nn : 1, DNode{0, {0, 0}, 0});
    t.max_code_len = 0;
    // inner node v owns bits [bv_pos(v), bv_pos(next inner node)): wt_helper.hpp:338-342
    uint64_t next_pos = t.wt_bits;
    for (int64_t v = (int64_t)nn - 1; v >= 0; --v) {
        const vlg_wt_node& nd = t.nodes[v];
        if (nd.child[0] != 0xFFFF) {
            if (nd.bv_pos > next_pos) return fail(VLG_E_INVALID, "node bv_pos not monotone");
            t.node_size[v] = next_pos - nd.bv_pos;
            next_pos = nd.bv_pos;
        }
    }
    for (uint32_t v = 0; v < nn; ++v) {
        const vlg_wt_node& nd = t.nodes[v];
        if (v) {
            if (nd.parent >= v) return fail(VLG_E_INVALID, "node table is not in BFS order");
            t.node_depth[v] = t.node_depth[nd.parent] + 1;
        }
        if (nd.child[0] == 0xFFFF) {
            t.c_to_leaf[(uint8_t)nd.bv_pos_rank] = (uint16_t)v;
            t.max_code_len = std::max(t.max_code_len, t.node_depth[v]);
        } else {
            // children are appended behind their parent in BFS order (wt_helper.hpp:170-206); anything else (a cycle in a damaged
            // file) would make the device tree walks spin
            for (int k = 0; k < 2; ++k) {
                const uint32_t ch = nd.child[k];
                if (ch >= nn || ch <= v || t.nodes[ch].parent != v) return fail(VLG_E_INVALID, "child index out of range or not in BFS order");
            }
        }
    }
    // Super-block counts and block_rank() are 32-bit: the 1-bits of an inner node (= the size of its right child) must stay
    // below 2^32.  Holds for every text of up to 2^32 - 1 bytes; beyond that it depends on the symbol statistics.
    for (uint32_t v = 0; v < nn; ++v) {
        const vlg_wt_node& nd = t.nodes[v];
        if (nd.child[0] == 0xFFFF) continue;
        const uint32_t c1 = nd.child[1];
        uint64_t ones;
        if (t.nodes[c1].child[0] != 0xFFFF) ones = t.node_size[c1];
        else {
            const uint32_t comp = t.char2comp[(uint8_t)t.nodes[c1].bv_pos_rank];
            ones = comp + 1 < t.C.size() ? t.C[comp + 1] - t.C[comp] : 0;
        }
        if (ones > 0xFFFFFFFFull || t.node_size[v] > (1ull << 36))
            return fail(VLG_E_UNSUPPORTED, "a wavelet-tree node holds 2^32 or more 1-bits: the 32-bit node-relative block counts cannot index it");
    }
    // m_path: wt_helper.hpp:219-240
    uint64_t prev_c = 0;
    for (uint32_t c = 0; c < 256; ++c) {
        if (t.c_to_leaf[c] != 0xFFFF) {
            uint32_t v = t.c_to_leaf[c];
            uint64_t pw = 0, pl = 0;
            while (v != 0) {
                uint32_t p = t.nodes[v].parent;
                pw <<= 1;
                if (t.nodes[p].child[1] == v) pw |= 1ull;
                ++pl;
                v = p;
            }
            if (pl > 56) return fail(VLG_E_UNSUPPORTED, "code depth greater than 56");
            t.paths[c] = pw | (pl << 56);
            prev_c = c;
        } else {
            t.paths[c] = prev_c;
        }
    }
    // HBM layout: every inner node gets size/224 + 1 blocks, in BFS (= level) order
    uint64_t blk = 0;
    for (uint32_t v = 0; v < nn; ++v) {
        const vlg_wt_node& nd = t.nodes[v];
        DNode& d = t.dnodes[v];
        if (nd.child[0] == 0xFFFF) continue;
        if (blk > 0xFFFFFFFFull) return fail(VLG_E_UNSUPPORTED, "more than 2^32 super-blocks");
        d.base = (uint32_t)blk;
        d.size_lo = (uint32_t)t.node_size[v];
        blk += t.node_size[v] / kBlockBits + 1;
        for (int k = 0; k < 2; ++k) {
            uint32_t ch = nd.child[k];
            if (t.nodes[ch].child[0] == 0xFFFF)
                d.child[k] = kLeafFlag | t.char2comp[(uint8_t)t.nodes[ch].bv_pos_rank];
            else
                d.child[k] = ch;
        }
    }
    if (blk > 0xFFFFFFFFull) return fail(VLG_E_UNSUPPORTED, "more than 2^32 super-blocks");
    t.n_blocks = blk;
    return VLG_OK;
}

vlg_status tree_from_counts(const uint64_t counts[256], HostTree& t)
{
    // byte_alphabet: lib/csa_alphabet_strategy.cpp:25-55
    t.sigma = 0;
    t.C.assign(1, 0);
    memset(t.char2comp, 0, 256);
    for (int c = 0; c < 256; ++c)
        if (counts[c]) {
            t.char2comp[c] = (uint8_t)t.sigma++;
            t.C.push_back(t.C.back() + counts[c]);
        }
    // Huffman shape: wt_huff.hpp:91-117 -- min-heap on (frequency, node id), ids in creation order
    struct Tmp { uint64_t freq, sym; int64_t parent, child[2]; };
    std::vector<Tmp> tmp;
    typedef std::pair<uint64_t, uint64_t> P;
    std::priority_queue<P, std::vector<P>, std::greater<P>> pq;
    for (int c = 0; c < 256; ++c)
        if (counts[c]) {
            pq.push(P(counts[c], tmp.size()));
            tmp.push_back(Tmp{counts[c], (uint64_t)c, -1, {-1, -1}});
        }
    while (pq.size() > 1) {
        P a = pq.top(); pq.pop();
        P b = pq.top(); pq.pop();
        tmp[a.second].parent = tmp[b.second].parent = (int64_t)tmp.size();
        pq.push(P(a.first + b.first, tmp.size()));
        tmp.push_back(Tmp{a.first + b.first, 0, -1, {(int64_t)a.second, (int64_t)b.second}});
    }
    // BFS numbering: wt_helper.hpp:170-206
    t.n_nodes = (uint32_t)tmp.size();
    t.nodes.assign(t.n_nodes, vlg_wt_node{0, 0, 0xFFFF, {0xFFFF, 0xFFFF}});
    t.wt_bits = 0;
    if (t.n_nodes) {
        std::vector<int64_t> src(t.n_nodes);   // BFS id -> tmp id
        src[0] = (int64_t)tmp.size() - 1;
        uint32_t next = 1;
        for (uint32_t v = 0; v < t.n_nodes; ++v) {
            const Tmp& s = tmp[src[v]];
            vlg_wt_node& nd = t.nodes[v];
            nd.bv_pos = t.wt_bits;
            if (s.child[0] >= 0) {
                t.wt_bits += s.freq;
                for (int k = 0; k < 2; ++k) {
                    src[next] = s.child[k];
                    t.nodes[next].parent = (uint16_t)v;
                    nd.child[k] = (uint16_t)next++;
                }
                nd.bv_pos_rank = 0;   // filled after the bits exist (wt_helper.hpp:243-250)
            } else {
                nd.bv_pos_rank = s.sym;
            }
        }
    }
    return finish_tree(t);
}

vlg_status tree_from_nodes(const vlg_wt_node* nodes, uint32_t n_nodes, uint64_t bv_bits, const uint8_t* char2comp,
                           const uint64_t* C, uint32_t sigma, HostTree& t)
{
    if (sigma == 0 || sigma > 256 || n_nodes != 2 * sigma - 1)
        return fail(VLG_E_INVALID, "n_nodes must be 2*sigma-1");
    t.sigma = sigma;
    t.n_nodes = n_nodes;
    t.nodes.assign(nodes, nodes + n_nodes);
    t.wt_bits = bv_bits;
    memcpy(t.char2comp, char2comp, 256);
    t.C.assign(C, C + sigma + 1);
    return finish_tree(t);
}

}  // namespace vlg

extern "C" const char* vlg_last_error(void) { return vlg::last_error_cstr(); }
extern "C" const char* vlg_version(void) { return "vlg-mi355x 0.1 (gfx950)"; }
