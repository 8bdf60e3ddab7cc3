// Reader for the reference's on-disk format of csa_wt<wt_huff<>, t_dens, t_inv_dens, sa_order_sa_sampling<>, isa_sampling<>,
// byte_alphabet>, so that an index built and stored by stock sdsl loads straight into HBM.
//   csa_wt::serialize      include/sdsl/csa_wt.hpp:374-393        wavelet tree, SA samples, ISA samples, alphabet
//   wt_pc::serialize       include/sdsl/wt_pc.hpp:638-652         size, sigma, bv, bv_rank, bv_select1, bv_select0, tree
//   int_vector<w>          include/sdsl/int_vector.hpp:584-600,1507-1557   u64 size in bits [, u8 width if w == 0], ceil(bits/64) words
//   rank_support_v         include/sdsl/rank_support_v.hpp:134-148  one int_vector<64>
//   select_support_mcl     include/sdsl/select_support_mcl.hpp:424-494
//   _byte_tree / _node     include/sdsl/wt_helper.hpp:112-131,275-301      22 bytes per node
//   byte_alphabet          lib/csa_alphabet_strategy.cpp:103-121
//   rrr_vector<63>         include/sdsl/rrr_vector.hpp:349-372 (csa_wt<wt_huff<rrr_vector<63>>>, BASELINE config 5: the stock image is
//                          decoded block by block on the host and re-coded on the device, vlg_index_load_sdsl_kind)
// Host-only: nothing here touches the GPU.
#include <cstdio>
#include <cstring>
#include <memory>
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
        if (bits > 8 * (uint64_t)(e - p) || width > 64) { ok = false; return nullptr; }      // also keeps (bits + 63) from wrapping
        return take(((bits + 63) / 64) * 8);
    }
    void skip_int_vector(uint8_t fixed_width) { uint64_t b; uint8_t w; (void)int_vector(fixed_width, b, w); }
    void skip_select_mcl()
    {
        uint64_t arg_cnt = get<uint64_t>();
        if (!arg_cnt) return;
        if (arg_cnt > 8 * (uint64_t)(e - p) * 4096) { ok = false; return; }                  // every super-block takes at least a header
        uint64_t sb = (arg_cnt + 4095) >> 12;
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

// rrr_vector<63> (include/sdsl/rrr_vector.hpp:349-372: size, bt, btnr, btnrp, rank samples, invert) back to plain bits.  Block i
// holds k ones, k = the stored class, or 63 - it where its super-block of 32 blocks is inverted (:186-201, 469-471); its offset of
// space_for_bt(k) bits (rrr_helper.hpp:282-284) numbers it among the blocks of that class (bin_to_nr, :304-320): bit b is set iff
// nr >= C(62 - b, k) for the k still to place (decode_bit, :323-375, walked over the whole block).
static bool rrr63_to_plain(Cursor& c, uint64_t& bits_out, std::vector<uint64_t>& plain)
{
    const uint64_t size = c.get<uint64_t>();
    uint64_t bt_bits, btnr_bits, b3;
    uint8_t bt_w, w1, w3;
    const uint8_t* bt = c.int_vector(0, bt_bits, bt_w);
    const uint8_t* btnr = c.int_vector(1, btnr_bits, w1);
    c.skip_int_vector(0);                                      // btnrp (pointer samples: recomputed by walking)
    c.skip_int_vector(0);                                      // rank samples
    const uint8_t* inv = c.int_vector(1, b3, w3);
    if (!c.ok || bt_w != 6) return false;                      // bits::hi(63) + 1
    const uint64_t n_blocks = bt_bits / 6;
    if (n_blocks != (size + 63) / 63 || b3 != (n_blocks + 31) / 32) return false;
    static uint64_t binom[64][64];
    static uint8_t space[64];
    static bool ready = false;
    if (!ready) {
        for (int nn = 0; nn < 64; ++nn) for (int k = 0; k < 64; ++k) binom[nn][k] = k == 0 ? 1 : (nn == 0 ? 0 : (k > nn ? 0 : binom[nn - 1][k - 1] + binom[nn - 1][k]));
        for (int k = 0; k < 64; ++k) space[k] = binom[63][k] == 1 ? 0 : (uint8_t)(64 - __builtin_clzll(binom[63][k]));
        ready = true;
    }
    bits_out = size;
    plain.assign((size + 63) / 64 + 1, 0);
    uint64_t pos = 0;                                          // bit position in btnr
    for (uint64_t i = 0; i < n_blocks; ++i) {
        const uint32_t stored = (uint32_t)read_packed(bt, i, 6);
        const bool inverted = (inv[(i / 32) >> 3] >> ((i / 32) & 7)) & 1;
        uint32_t k = inverted ? 63 - stored : stored;
        const uint32_t len = space[stored];
        if (pos + len > btnr_bits) return false;
        uint64_t nr = 0;
        if (len) {
            const uint64_t wq = pos >> 6, o = pos & 63;
            uint64_t lo, hi = 0;
            memcpy(&lo, btnr + 8 * wq, 8);
            nr = lo >> o;
            if (o + len > 64) { memcpy(&hi, btnr + 8 * (wq + 1), 8); nr |= hi << (64 - o); }
            if (len < 64) nr &= (1ull << len) - 1;
        }
        pos += len;
        uint64_t bin = 0;
        if (k == 63) bin = (1ull << 63) - 1;
        else for (uint32_t b = 0, nn = 63; b < 63 && k; ++b, --nn) {
            const uint64_t cc = binom[nn - 1][k];
            if (nr >= cc) { nr -= cc; --k; bin |= 1ull << b; }
        }
        const uint64_t at = i * 63;
        if (at >= size) break;
        const uint64_t take = size - at < 63 ? size - at : 63;
        if (take < 63) bin &= (1ull << take) - 1;
        plain[at >> 6] |= bin << (at & 63);
        if ((at & 63) + take > 64) plain[(at >> 6) + 1] |= bin >> (64 - (at & 63));
    }
    return c.ok;
}

static vlg_status sdsl_file_open_impl(const char* path, uint32_t sa_sample_dens, int bv_kind, vlg_sdsl_file** out)
{
    using namespace vlg;
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(VLG_E_INVALID, std::string("cannot open ") + path);
    std::unique_ptr<FILE, int (*)(FILE*)> fp_guard(fp, fclose);
    std::unique_ptr<vlg_sdsl_file> holder(new vlg_sdsl_file());
    vlg_sdsl_file* f = holder.get();
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    f->raw.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = f->raw.empty() ? 0 : fread(f->raw.data(), 1, f->raw.size(), fp);
    fp_guard.reset();
    auto bad = [&](const char* what) { return fail(VLG_E_INVALID, std::string("not a csa_wt<wt_huff<>> file (") + what + ")"); };
    if (got != f->raw.size()) return bad("short read");
    Cursor c{f->raw.data(), f->raw.data() + f->raw.size()};
    // ---- wavelet tree (wt_pc.hpp:638-652) -----------------------------------------------------------
    f->n = c.get<uint64_t>();
    f->wt_sigma = c.get<uint64_t>();
    if (bv_kind == VLG_BV_RRR63) {
        // wt_huff<rrr_vector<63>>: the bit-vector is an rrr_vector, its rank / select supports store nothing (rrr_vector.hpp:511-522)
        if (!c.ok || f->wt_sigma == 0 || f->wt_sigma > 256) return bad("wavelet tree header");
        if (!rrr63_to_plain(c, f->bv_bits, f->bv)) return bad("rrr_vector<63>");
    } else {
        uint8_t w;
        const uint8_t* bvw = c.int_vector(1, f->bv_bits, w);
        if (!c.ok || f->wt_sigma == 0 || f->wt_sigma > 256) return bad("wavelet tree header");
        f->bv.assign((f->bv_bits + 63) / 64 + 1, 0);
        if (f->bv_bits) memcpy(f->bv.data(), bvw, ((f->bv_bits + 63) / 64) * 8);
        c.skip_int_vector(64);                                    // rank_support_v basic blocks (rebuilt on the device as counts)
        c.skip_select_mcl();                                      // select_1
        c.skip_select_mcl();                                      // select_0
    }
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
    *out = holder.release();
    return VLG_OK;
}

// No exception may cross the C boundary: a damaged file that asks for absurd sizes ends as a status, not std::terminate.
extern "C" vlg_status vlg_sdsl_file_open_kind(const char* path, uint32_t sa_sample_dens, int bv_kind, vlg_sdsl_file** out)
{
    using namespace vlg;
    if (!path || !out) return fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    if (bv_kind != VLG_BV_PLAIN && bv_kind != VLG_BV_RRR63) return fail(VLG_E_INVALID, "unknown bit-vector kind");
    try { return sdsl_file_open_impl(path, sa_sample_dens, bv_kind, out); }
    catch (const std::bad_alloc&) { return fail(VLG_E_OOM, "out of host memory while reading the index file"); }
    catch (const std::exception& e) { return fail(VLG_E_INVALID, std::string("not a csa_wt<wt_huff<>> file (") + e.what() + ")"); }
}

extern "C" vlg_status vlg_sdsl_file_open(const char* path, uint32_t sa_sample_dens, vlg_sdsl_file** out)
{
    return vlg_sdsl_file_open_kind(path, sa_sample_dens, VLG_BV_PLAIN, out);
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

extern "C" vlg_status vlg_index_load_sdsl_kind(const char* path, uint32_t sa_sample_dens, int bv_kind, vlg_index** out)
{
    if (!out) return vlg::fail(VLG_E_INVALID, "null argument");
    *out = nullptr;
    vlg_sdsl_file* f = nullptr;
    if (vlg_status st = vlg_sdsl_file_open_kind(path, sa_sample_dens, bv_kind, &f)) return st;
    vlg_index_parts p;
    vlg_sdsl_file_parts(f, &p);
    vlg_index* plain = nullptr;
    vlg_status st = vlg_index_from_parts(&p, &plain);
    vlg_sdsl_file_close(f);
    if (st || bv_kind == VLG_BV_PLAIN) { *out = plain; return st; }
    // a stock rrr image: its blocks were decoded on the host; the device index keeps them rrr-coded in its own block numbering
    st = vlg_index_compress(plain, VLG_BV_RRR63, out);
    vlg_index_destroy(plain);
    return st;
}

extern "C" vlg_status vlg_index_load_sdsl(const char* path, uint32_t sa_sample_dens, vlg_index** out)
{
    return vlg_index_load_sdsl_kind(path, sa_sample_dens, VLG_BV_PLAIN, out);
}

// =============================================================================================
// Writer: the same format, so that stock sdsl can `load_from_file(csa, path)` an index built on the device.
//   rank_support_v<1,1>     include/sdsl/rank_support_v.hpp:67-106      (interleaved absolute / 7 x 9-bit relative counts)
//   select_support_mcl<b,1> include/sdsl/select_support_mcl.hpp:203-262 (the structure init_slow builds; load() and
//                           select() accept it whatever size the vector has -- the reference's init_fast, used from
//                           100 000 bits on, differs only in where it draws the long/mini line for some super-blocks)
//   _isa_sampling           include/sdsl/csa_sampling_strategy.hpp:626-642, density 64 (csa_wt's default t_inv_dens)
// =============================================================================================
extern "C" vlg_status vlg_index_isa_samples(const vlg_index* idx, uint32_t inv_dens, uint64_t* h_out, uint64_t count);

namespace {

struct Sink {
    FILE* fp; bool ok = true;
    void raw(const void* p, size_t bytes) { if (ok && bytes && fwrite(p, 1, bytes, fp) != bytes) ok = false; }
    template <class T> void put(T v) { raw(&v, sizeof v); }
    // int_vector<w>: u64 size in bits [, u8 width when w == 0], ceil(bits/64) words (int_vector.hpp:584-600,1507-1557)
    void int_vector_packed(const std::vector<uint64_t>& values, uint8_t width, bool width_in_stream)
    {
        const uint64_t bits = (uint64_t)values.size() * width;
        std::vector<uint64_t> words((bits + 63) / 64, 0);
        for (uint64_t i = 0; i < values.size(); ++i) {
            const uint64_t bit = i * width, w = bit >> 6, o = bit & 63;
            const uint64_t v = width == 64 ? values[i] : (values[i] & ((1ull << width) - 1));
            words[w] |= v << o;
            if (o + width > 64) words[w + 1] |= v >> (64 - o);
        }
        put<uint64_t>(bits);
        if (width_in_stream) put<uint8_t>(width);
        raw(words.data(), words.size() * 8);
    }
    void bit_vector_words(const uint64_t* words, uint64_t bits) { put<uint64_t>(bits); raw(words, ((bits + 63) / 64) * 8); }
};

inline uint32_t hi_bit(uint64_t x) { return x ? 63 - (uint32_t)__builtin_clzll(x) : 0; }      // bits::hi

inline bool bit_at(const uint64_t* w, uint64_t i) { return (w[i >> 6] >> (i & 63)) & 1; }

// rank_support_v<1,1>(&bv): the int_vector<64> m_basic_block
void write_rank_support_v(Sink& out, const uint64_t* data, uint64_t bits)
{
    std::vector<uint64_t> bb;
    if (bits == 0) { bb.assign(2, 0); out.int_vector_packed(bb, 64, false); return; }
    const uint64_t cap_words = (bits + 63) / 64;                      // capacity() >> 6
    bb.assign((((cap_words * 64) >> 9) + 1) << 1, 0);
    uint64_t i, j = 0, sum = (uint64_t)__builtin_popcountll(data[0]), second = 0;
    for (i = 1; i < cap_words; ++i) {
        if (!(i & 7)) { j += 2; bb[j - 1] = second; bb[j] = bb[j - 2] + sum; second = sum = 0; }
        else second |= sum << (63 - 9 * (i & 7));
        sum += (uint64_t)__builtin_popcountll(data[i]);
    }
    if (i & 7) { second |= sum << (63 - 9 * (i & 7)); bb[j + 1] = second; }
    else { j += 2; bb[j - 1] = second; bb[j] = bb[j - 2] + sum; bb[j + 1] = 0; }
    out.int_vector_packed(bb, 64, false);
}

// select_support_mcl<b,1>: arg count, super-block starts, mini_or_long flags, then per super-block either the 4096 positions
// (long) or the offsets of every 64th argument (mini)
void write_select_mcl(Sink& out, const uint64_t* data, uint64_t bits, bool ones)
{
    const uint64_t kSuper = 4096;
    uint64_t arg_cnt = 0;
    for (uint64_t w = 0; w < (bits + 63) / 64; ++w) {
        uint64_t x = ones ? data[w] : ~data[w];
        if (w == (bits + 63) / 64 - 1 && (bits & 63)) x &= (1ull << (bits & 63)) - 1;
        arg_cnt += (uint64_t)__builtin_popcountll(x);
    }
    out.put<uint64_t>(arg_cnt);
    if (!arg_cnt) return;
    const uint64_t capacity = ((bits + 63) / 64) * 64;
    const uint64_t logn = hi_bit(capacity) + 1, logn4 = logn * logn * logn * logn;
    const uint64_t sb = (arg_cnt + kSuper - 1) / kSuper;
    std::vector<uint64_t> superblock(sb, 0);
    std::vector<uint8_t> is_mini(sb, 0);
    std::vector<std::vector<uint64_t>> payload(sb);
    std::vector<uint8_t> payload_width(sb, 0);
    bool any_long = false;
    std::vector<uint64_t> pos(kSuper);
    uint64_t cnt = 0, sbi = 0;
    for (uint64_t i = 0; i < bits; ++i) {
        if (bit_at(data, i) != ones) continue;
        pos[cnt % kSuper] = i;
        ++cnt;
        if (cnt % kSuper == 0 || cnt == arg_cnt) {
            const uint64_t last = (cnt - 1) % kSuper;
            superblock[sbi] = pos[0];
            const uint64_t diff = pos[last] - pos[0];
            if (diff > logn4) {                                       // long: every position, 4096 entries of hi(last position)+1 bits
                any_long = true;
                payload_width[sbi] = (uint8_t)(hi_bit(pos[last]) + 1);
                payload[sbi].assign(kSuper, 0);
                for (uint64_t j = 0; j <= last; ++j) payload[sbi][j] = pos[j];
            } else {                                                  // mini: offset of every 64th argument, 64 entries
                is_mini[sbi] = 1;
                payload_width[sbi] = (uint8_t)(hi_bit(diff) + 1);
                payload[sbi].assign(64, 0);
                for (uint64_t j = 0; j <= last; j += 64) payload[sbi][j / 64] = pos[j] - pos[0];
            }
            ++sbi;
        }
    }
    out.int_vector_packed(superblock, (uint8_t)logn, true);
    {   // mini_or_long: empty unless some super-block is long
        std::vector<uint64_t> flags(any_long ? (sb + 63) / 64 : 0, 0);
        if (any_long) for (uint64_t i = 0; i < sb; ++i) if (is_mini[i]) flags[i >> 6] |= 1ull << (i & 63);
        out.bit_vector_words(flags.data(), any_long ? sb : 0);
    }
    for (uint64_t i = 0; i < sb; ++i) out.int_vector_packed(payload[i], payload_width[i], true);
}

}  // namespace

extern "C" vlg_status vlg_index_save_sdsl(const vlg_index* idx, const char* path)
{
    using namespace vlg;
    if (!idx || !path) return fail(VLG_E_INVALID, "null argument");
    vlg_index_parts sz;
    if (vlg_status st = vlg_index_export_parts(idx, &sz, nullptr)) return st;
    if (sz.sa_sample_dens != 32) return fail(VLG_E_UNSUPPORTED, "the reference type csa_wt<wt_huff<>,32,64> has SA sample density 32");
    uint8_t c2c[256];
    std::vector<uint64_t> C(257), bv((sz.bv_bits + 63) / 64 + 1, 0), samples(sz.n_samples);
    std::vector<vlg_wt_node> nodes(sz.n_nodes);
    vlg_index_parts_out po{c2c, C.data(), bv.data(), nodes.data(), samples.data()};
    if (vlg_status st = vlg_index_export_parts(idx, &sz, &po)) return st;
    const uint64_t n = sz.n, inv_dens = 64;
    std::vector<uint64_t> isa((n - 1) / inv_dens + 1);
    if (vlg_status st = vlg_index_isa_samples(idx, (uint32_t)inv_dens, isa.data(), isa.size())) return st;
    const HostTree& tree = idx->tree;
    FILE* fp = fopen(path, "wb");
    if (!fp) return fail(VLG_E_INVALID, std::string("cannot create ") + path);
    Sink out{fp};
    // ---- wt_pc::serialize (wt_pc.hpp:638-652) -----------------------------------------------------------
    out.put<uint64_t>(n);
    out.put<uint64_t>(sz.sigma);
    out.bit_vector_words(bv.data(), sz.bv_bits);
    write_rank_support_v(out, bv.data(), sz.bv_bits);
    write_select_mcl(out, bv.data(), sz.bv_bits, true);
    write_select_mcl(out, bv.data(), sz.bv_bits, false);
    out.put<uint64_t>(sz.n_nodes);
    uint16_t c_to_leaf[256];
    for (int c = 0; c < 256; ++c) c_to_leaf[c] = 0xFFFF;
    for (uint32_t v = 0; v < sz.n_nodes; ++v) {
        out.put<uint64_t>(nodes[v].bv_pos); out.put<uint64_t>(nodes[v].bv_pos_rank);
        out.put<uint16_t>(nodes[v].parent); out.put<uint16_t>(nodes[v].child[0]); out.put<uint16_t>(nodes[v].child[1]);
        if (nodes[v].child[0] == 0xFFFF) c_to_leaf[nodes[v].bv_pos_rank & 0xFF] = (uint16_t)v;
    }
    out.raw(c_to_leaf, sizeof c_to_leaf);
    out.raw(tree.paths.data(), 256 * 8);
    // ---- SA samples, ISA samples: int_vector<0> of width hi(n)+1 (csa_sampling_strategy.hpp:85-98, 626-642) ------------
    const uint8_t w = (uint8_t)(hi_bit(n) + 1);
    out.int_vector_packed(samples, w, true);
    out.int_vector_packed(isa, w, true);
    // ---- byte_alphabet (lib/csa_alphabet_strategy.cpp:103-112) -------------------------------------------------
    {
        std::vector<uint64_t> v(256);
        for (int c = 0; c < 256; ++c) v[c] = c2c[c];
        out.int_vector_packed(v, 8, false);
        // comp2char: comp ranks are assigned in byte order (lib/csa_alphabet_strategy.cpp:43-49), so it is the sorted list of
        // the occurring bytes; bytes that do not occur map to 0 in char2comp like the sentinel
        std::vector<uint64_t> occurring;
        occurring.push_back(0);
        for (int c = 1; c < 256; ++c) if (c2c[c]) occurring.push_back(c);
        if (occurring.size() != sz.sigma) { fclose(fp); return fail(VLG_E_INTERNAL, "alphabet does not match sigma"); }
        out.int_vector_packed(occurring, 8, false);
        std::vector<uint64_t> Cv(C.begin(), C.begin() + sz.sigma + 1);
        out.int_vector_packed(Cv, 64, false);
        out.put<uint16_t>((uint16_t)sz.sigma);
    }
    const bool ok = out.ok;
    if (fclose(fp) != 0 || !ok) return fail(VLG_E_INVALID, std::string("write failed: ") + path);
    return VLG_OK;
}
