// Reader for the reference's on-disk format of csa_wt<wt_huff<>, t_dens, t_inv_dens, sa_order_sa_sampling<>, isa_sampling<>,
// byte_alphabet>, so that an index built and stored by stock sdsl loads straight into HBM.
//   csa_wt::serialize      include/sdsl/csa_wt.hpp:374-393        wavelet tree, SA samples, ISA samples, alphabet
//   wt_pc::serialize       include/sdsl/wt_pc.hpp:638-652         size, sigma, bv, bv_rank, bv_select1, bv_select0, tree
//   int_vector<w>          include/sdsl/int_vector.hpp:584-600,1507-1557   u64 size in bits [, u8 width if w == 0], ceil(bits/64) words
//   rank_support_v         include/sdsl/rank_support_v.hpp:134-148  one int_vector<64>
//   select_support_mcl     include/sdsl/select_support_mcl.hpp:424-494
//   _byte_tree / _node     include/sdsl/wt_helper.hpp:112-131,275-301      22 bytes per node
//   byte_alphabet          lib/csa_alphabet_strategy.cpp:103-121
// Host-only: nothing here touches the GPU.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "common.hpp"

struct vlg_sdsl_file {
    std::vector<uint8_t> raw;
    uint64_t n = 0, wt_sigma = 0, bv_bits = 0, n_samples = 0;
    uint32_t sigma = 0, dens = 0;
    std::vector<uint64_t> bv, C, samples;
    std::vector<vlg_wt_node> nodes;
    uint8_t char2comp[256];
};

namespace {

struct Cursor {
    const uint8_t* p; const uint8_t* e; bool ok = true;
    template <class T> T get() { T v{}; if ((size_t)(e - p) < sizeof(T)) { ok = false; return v; } memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
    const uint8_t* take(uint64_t bytes) { if ((uint64_t)(e - p) < bytes) { ok = false; return nullptr; } const uint8_t* r = p; p += bytes; return r; }
    // int_vector<w>: returns pointer to the words; width read from the stream when fixed_width == 0
    const uint8_t* int_vector(uint8_t fixed_width, uint64_t& bits, uint8_t& width)
    {
        bits = get<uint64_t>();
        width = fixed_width ? fixed_width : get<uint8_t>();
        return take(((bits + 63) / 64) * 8);
    }
    void skip_int_vector(uint8_t fixed_width) { uint64_t b; uint8_t w; (void)int_vector(fixed_width, b, w); }
    void skip_select_mcl()
    {
        uint64_t arg_cnt = get<uint64_t>();
        uint64_t sb = (arg_cnt + 4095) >> 12;
        if (!arg_cnt) return;
        skip_int_vector(0);                                  // m_superblock
        uint64_t bits; uint8_t w;
        const uint8_t* mol = int_vector(1, bits, w);         // mini_or_long (possibly empty)
        for (uint64_t i = 0; i < sb && ok; ++i) skip_int_vector(0);   // either a long super-block or a mini-block vector
        (void)mol;
    }
};

uint64_t read_packed(const uint8_t* words, uint64_t idx, uint8_t width)
{
    uint64_t bit = idx * width, w = bit >> 6, o = bit & 63;
    uint64_t lo, hi = 0;
    memcpy(&lo, words + 8 * w, 8);
    uint64_t v = lo >> o;
    if (o + width > 64) { memcpy(&hi, words + 8 * (w + 1), 8); v |= hi << (64 - o); }
    return width == 64 ? v : (v & ((1ull << width) - 1));
}

}  // namespace

extern "C" vlg_status vlg_sdsl_file_open(const char* path, uint32_t sa_sample_dens, vlg_sdsl_file** out)
{
    using namespace vlg;
    if (!path || !out) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(VLG_E_INVALID, std::string("cannot open ") + path);
    vlg_sdsl_file* f = new vlg_sdsl_file();
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    f->raw.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = f->raw.empty() ? 0 : fread(f->raw.data(), 1, f->raw.size(), fp);
    fclose(fp);
    auto bad = [&](const char* what) { delete f; return fail(VLG_E_INVALID, std::string("not a csa_wt<wt_huff<>> file (") + what + ")"); };
    if (got != f->raw.size()) return bad("short read");
    Cursor c{f->raw.data(), f->raw.data() + f->raw.size()};
    // ---- wavelet tree (wt_pc.hpp:638-652) -----------------------------------------------------------
    f->n = c.get<uint64_t>();
    f->wt_sigma = c.get<uint64_t>();
    uint8_t w;
    const uint8_t* bvw = c.int_vector(1, f->bv_bits, w);
    if (!c.ok || f->wt_sigma == 0 || f->wt_sigma > 256) return bad("wavelet tree header");
    f->bv.assign((f->bv_bits + 63) / 64 + 1, 0);
    if (f->bv_bits) memcpy(f->bv.data(), bvw, ((f->bv_bits + 63) / 64) * 8);
    c.skip_int_vector(64);                                    // rank_support_v basic blocks (rebuilt on the device as counts)
    c.skip_select_mcl();                                      // select_1
    c.skip_select_mcl();                                      // select_0
    uint64_t n_nodes = c.get<uint64_t>();
    if (!c.ok || n_nodes != 2 * f->wt_sigma - 1) return bad("tree size");
    f->nodes.resize(n_nodes);
    for (uint64_t v = 0; v < n_nodes; ++v) {                  // _node::serialize: u64 bv_pos, u64 bv_pos_rank, u16 parent, u16 child[2]
        vlg_wt_node& nd = f->nodes[v];
        nd.bv_pos = c.get<uint64_t>();
        nd.bv_pos_rank = c.get<uint64_t>();
        nd.parent = c.get<uint16_t>();
        nd.child[0] = c.get<uint16_t>();
        nd.child[1] = c.get<uint16_t>();
    }
    c.take(256 * 2);                                          // m_c_to_leaf (derived from the nodes)
    c.take(256 * 8);                                          // m_path
    if (!c.ok) return bad("tree");
    // ---- SA samples: _sa_order_sampling is an int_vector<0> (csa_sampling_strategy.hpp:64-112) ------------
    uint64_t sbits; uint8_t sw;
    const uint8_t* sw_words = c.int_vector(0, sbits, sw);
    if (!c.ok || sw == 0) return bad("SA samples");
    f->n_samples = sbits / sw;
    f->samples.resize(f->n_samples);
    for (uint64_t i = 0; i < f->n_samples; ++i) f->samples[i] = read_packed(sw_words, i, sw);
    // ---- ISA samples: int_vector<0>, not used by count/locate ---------------------------------------------------
    c.skip_int_vector(0);
    // ---- byte_alphabet (lib/csa_alphabet_strategy.cpp:103-121) -----------------------------------------------------
    uint64_t b; uint8_t ww;
    const uint8_t* c2c = c.int_vector(8, b, ww);
    if (!c.ok || b != 256 * 8) return bad("char2comp");
    memcpy(f->char2comp, c2c, 256);
    c.skip_int_vector(8);                                     // comp2char
    const uint8_t* Cw = c.int_vector(64, b, ww);
    uint16_t sigma = c.get<uint16_t>();
    if (!c.ok || b != (uint64_t)(sigma + 1) * 64 || sigma != f->wt_sigma) return bad("alphabet");
    f->sigma = sigma;
    f->C.resize(sigma + 1);
    memcpy(f->C.data(), Cw, (sigma + 1) * 8);
    if (f->C[sigma] != f->n) return bad("C[sigma] != size");
    // density: t_dens is a template parameter, not stored; n_samples = ceil(n / dens) must hold for the given value
    f->dens = sa_sample_dens ? sa_sample_dens : 32;
    if (f->n_samples != (f->n + f->dens - 1) / f->dens) return bad("SA sample density does not match the file");
    *out = f;
    return VLG_OK;
}

extern "C" vlg_status vlg_sdsl_file_parts(const vlg_sdsl_file* f, vlg_index_parts* p)
{
    using namespace vlg;
    if (!f || !p) return fail(VLG_E_INVALID, "null argument");
    p->n = f->n; p->sigma = f->sigma; p->sa_sample_dens = f->dens;
    p->char2comp = f->char2comp; p->C = f->C.data();
    p->bv_words = f->bv.data(); p->bv_bits = f->bv_bits;
    p->nodes = f->nodes.data(); p->n_nodes = (uint32_t)f->nodes.size();
    p->sa_samples = f->samples.data(); p->n_samples = f->n_samples;
    return VLG_OK;
}

extern "C" void vlg_sdsl_file_close(vlg_sdsl_file* f) { delete f; }

extern "C" vlg_status vlg_index_load_sdsl(const char* path, uint32_t sa_sample_dens, vlg_index** out)
{
    if (!out) return vlg::fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    vlg_sdsl_file* f = nullptr;
    if (vlg_status st = vlg_sdsl_file_open(path, sa_sample_dens, &f)) return st;
    vlg_index_parts p;
    vlg_sdsl_file_parts(f, &p);
    vlg_status st = vlg_index_from_parts(&p, out);
    vlg_sdsl_file_close(f);
    return st;
}
