// oracle/ref_glue.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" glue that instantiates the *reference's own* templates from the
// headers where they lie under /root/reference (nothing is copied into this repo)
// so that tests can pin oracle/vlg_oracle.c against the real implementation.
//
// What is reference code here (compiled from /root/reference/include/sdsl + lib/*.cpp):
//   * wt_huff<> = wt_pc<huff_shape,...>   ctor / rank / inverse_select / operator[]
//                                         (include/sdsl/wt_pc.hpp:197-248,318-402)
//   * rank_support_v<1,1>, rank_support_v5<1,1>, rrr_vector<63>+rank_support_rrr
//                                         (rank_support_v.hpp:67-124, rank_support_v5.hpp:65-134,
//                                          rrr_vector.hpp:145-237,444-480)
//   * byte_alphabet                       (lib/csa_alphabet_strategy.cpp:25-55)
//   * LF trait traverse_csa_wt            (suffix_array_helper.hpp:336-349)
//   * wt_int<bit_vector_il<>, rank_support_il<>>  -- the tree type of vlg_index (vlg_index.hpp:116-119): ctor, operator[],
//                                         the level-concatenated bit-vector `tree`, expand(node) / expand(node, range) / value_range
//                                         (include/sdsl/wt_int.hpp:182-270, 339-361, 824-939; bit_vector_il.hpp)
//   * wt_int<> (bit_vector, rank_support_v)  ctor / rank / inverse_select / tree: the BWT container of csa_wt<wt_int<>>
//                                         (include/sdsl/wt_int.hpp:182-270, 370-395, 405-430)
//   * int_alphabet<>                      (include/sdsl/csa_alphabet_strategy.hpp:394-470; lib/csa_alphabet_strategy.cpp)
//
// What is NOT buildable in this image (see DESIGN.md "Oracle"):
//   csa_wt.hpp / wavelet_trees.hpp / suffix_arrays.hpp / vlg_index.hpp / index_sasearch.hpp all
//   pull construct_sa.hpp, which #includes "divsufsort.h" from an empty git submodule, and
//   suffix_array_algorithm.hpp names csa_wt<> in its signatures, so backward_search / locate /
//   forward_search cannot be instantiated either.  No stand-in is written for any of them.
//   The `ref_csa` adapter below only gives the reference LF trait (traverse_csa_wt) the members
//   it reads (wavelet_tree, C, char2comp); its operator[] is a 5-line restatement of
//   csa_wt.hpp:335-348 + csa_sampling_strategy.hpp:85-111 and is labelled as such.
#include <sdsl/int_vector.hpp>
#include <sdsl/int_vector_buffer.hpp>
#include <sdsl/rank_support.hpp>
#include <sdsl/rank_support_v.hpp>
#include <sdsl/rank_support_v5.hpp>
#include <sdsl/rrr_vector.hpp>
#include <sdsl/wt_huff.hpp>
#include <sdsl/wt_int.hpp>
#include <sdsl/bit_vector_il.hpp>
#include <sdsl/csa_alphabet_strategy.hpp>
#include <sdsl/suffix_array_helper.hpp>
#include <sdsl/io.hpp>
#include <vector>
#include <string>
#include <atomic>
#include <fstream>
#include <cstring>

using namespace sdsl;

namespace {

std::atomic<uint64_t> g_seq(0);

struct ref_base {
    virtual ~ref_base() {}
    virtual uint64_t size() const = 0;
    virtual uint64_t wt_rank(uint64_t i, uint8_t c) const = 0;
    virtual uint64_t inverse_select(uint64_t i, uint8_t* c) const = 0;
    virtual uint64_t bv_size() const = 0;
    virtual uint64_t bv_rank1(uint64_t idx) const = 0;
    virtual int bv_get(uint64_t idx) const = 0;
    virtual uint64_t n_nodes() const = 0;
    virtual void node(uint64_t v, uint64_t* bv_pos, uint64_t* sz, int* leaf, int* sym, int* c0, int* c1) const = 0;
    virtual void alphabet(uint8_t* c2c, uint64_t* C, uint32_t* sigma) const = 0;
    virtual uint64_t sa(uint64_t i) const = 0;
    virtual uint64_t lf_at(uint64_t i) const = 0;
    virtual int write_csa_image(const char* path, const uint64_t* sa, uint64_t n) const = 0;
};

// Minimal csa_tag model for the reference's generic algorithms.
template<class t_wt>
struct ref_csa : ref_base {
    typedef csa_tag                          index_category;
    typedef byte_alphabet_tag                alphabet_category;
    typedef uint64_t                         size_type;
    typedef uint64_t                         value_type;
    typedef uint8_t                          char_type;
    typedef ptrdiff_t                        difference_type;
    typedef byte_alphabet                    alphabet_type;
    typedef t_wt                             wavelet_tree_type;
    enum { sa_sample_dens = 32 };

    t_wt           wavelet_tree;
    byte_alphabet  m_alphabet;
    int_vector<>   sa_sample;   // SA[0], SA[32], ... width bits::hi(n)+1  (csa_sampling_strategy.hpp:85-98)
    const typename byte_alphabet::char2comp_type& char2comp;
    const typename byte_alphabet::comp2char_type& comp2char;
    const typename byte_alphabet::C_type&         C;
    const typename byte_alphabet::sigma_type&     sigma;
    const bwt_of_csa_wt<ref_csa>                  bwt;
    const traverse_csa_wt<ref_csa, false>         lf;

    ref_csa(const uint8_t* bwt_in, const uint64_t* sa_in, uint64_t n)
        : char2comp(m_alphabet.char2comp), comp2char(m_alphabet.comp2char), C(m_alphabet.C),
          sigma(m_alphabet.sigma), bwt(*this), lf(*this)
    {
        std::string f = "@vref_bwt_" + std::to_string(g_seq++);
        {
            int_vector<8> b(n);
            for (uint64_t i = 0; i < n; ++i) b[i] = bwt_in[i];
            store_to_file(b, f);
        }
        {
            int_vector_buffer<8> buf(f);
            byte_alphabet tmp(buf, n);              // reference ctor
            m_alphabet.swap(tmp);
        }
        {
            int_vector_buffer<8> buf(f);
            t_wt tmp(buf, n);                       // reference wt_pc ctor
            wavelet_tree.swap(tmp);
        }
        sdsl::remove(f);
        if (sa_in) {
            sa_sample.width(bits::hi(n) + 1);
            sa_sample.resize((n + sa_sample_dens - 1) / sa_sample_dens);
            for (uint64_t i = 0, j = 0; i < n; i += sa_sample_dens) sa_sample[j++] = sa_in[i];
        }
    }

    uint64_t size() const { return wavelet_tree.size(); }
    size_type rank_bwt(size_type i, const char_type c) const { return wavelet_tree.rank(i, c); }
    size_type select(size_type, const char_type) const { return 0; }

    // csa_wt.hpp:335-348 with _sa_order_sampling::is_sampled (csa_sampling_strategy.hpp:102-111)
    value_type operator[](size_type i) const
    {
        size_type off = 0;
        while (i % sa_sample_dens) { i = lf[i]; ++off; }
        value_type result = sa_sample[i / sa_sample_dens];
        return (result + off < size()) ? result + off : result + off - size();
    }

    uint64_t wt_rank(uint64_t i, uint8_t c) const { return wavelet_tree.rank(i, c); }
    uint64_t inverse_select(uint64_t i, uint8_t* c) const
    {
        auto rc = wavelet_tree.inverse_select(i);
        *c = rc.second;
        return rc.first;
    }
    uint64_t bv_size() const { return wavelet_tree.bv.size(); }
    uint64_t bv_rank1(uint64_t idx) const
    {
        typename t_wt::rank_1_type rs(&wavelet_tree.bv);
        return rs.rank(idx);
    }
    int bv_get(uint64_t idx) const { return wavelet_tree.bv[idx]; }
    uint64_t n_nodes() const { return wavelet_tree.sigma ? 2 * (uint64_t)wavelet_tree.sigma - 1 : 0; }
    void node(uint64_t v, uint64_t* bv_pos, uint64_t* sz, int* leaf, int* sym, int* c0, int* c1) const
    {
        typename t_wt::node_type nv = (typename t_wt::node_type)v;
        *leaf = wavelet_tree.is_leaf(nv);
        *sz = wavelet_tree.size(nv);
        if (*leaf) {
            *sym = wavelet_tree.sym(nv);
            *bv_pos = 0; *c0 = *c1 = -1;
        } else {
            *sym = -1;
            auto ch = wavelet_tree.expand(nv);
            *c0 = ch[0]; *c1 = ch[1];
            *bv_pos = wavelet_tree.bit_vec(nv).begin() - wavelet_tree.bv.begin();
        }
    }
    void alphabet(uint8_t* c2c, uint64_t* Cout, uint32_t* sg) const
    {
        for (int i = 0; i < 256; ++i) c2c[i] = char2comp[i];
        for (uint32_t i = 0; i <= sigma; ++i) Cout[i] = C[i];
        *sg = sigma;
    }
    uint64_t sa(uint64_t i) const { return (*this)[i]; }
    uint64_t lf_at(uint64_t i) const { return lf[i]; }

    // The byte image csa_wt::serialize writes (csa_wt.hpp:374-393): wavelet tree, SA samples, ISA samples, alphabet --
    // every member through the reference's own serialize(); the ISA samples are filled as _isa_sampling's ctor does
    // (csa_sampling_strategy.hpp:626-648, density 64 = csa_wt's default t_inv_dens).
    int write_csa_image(const char* path, const uint64_t* sa_in, uint64_t n) const
    {
        std::ofstream out(path, std::ios::binary);
        if (!out) return -1;
        wavelet_tree.serialize(out);
        sa_sample.serialize(out);
        int_vector<> isa;
        const uint64_t inv_dens = 64;
        if (n >= 1) { isa.width(bits::hi(n) + 1); isa.resize((n - 1) / inv_dens + 1); }
        for (uint64_t i = 0; i < isa.size(); ++i) isa[i] = 0;
        for (uint64_t i = 0; i < n; ++i) if (sa_in[i] % inv_dens == 0) isa[sa_in[i] / inv_dens] = i;
        isa.serialize(out);
        m_alphabet.serialize(out);
        return out ? 0 : -1;
    }
};

typedef wt_huff<bit_vector, rank_support_v<>>  wt_v;
typedef wt_huff<bit_vector, rank_support_v5<>> wt_v5;
typedef wt_huff<rrr_vector<63>>                wt_rrr;

} // namespace

extern "C" {

// variant: 0 = wt_huff<bit_vector,rank_support_v>, 1 = rank_support_v5, 2 = rrr_vector<63>
void* vref_create(const uint8_t* bwt, const uint64_t* sa, uint64_t n, int variant)
{
    try {
        switch (variant) {
            case 0: return new ref_csa<wt_v>(bwt, sa, n);
            case 1: return new ref_csa<wt_v5>(bwt, sa, n);
            case 2: return new ref_csa<wt_rrr>(bwt, sa, n);
        }
    } catch (...) {}
    return nullptr;
}
void vref_destroy(void* h) { delete (ref_base*)h; }
uint64_t vref_size(void* h) { return ((ref_base*)h)->size(); }
uint64_t vref_wt_rank(void* h, uint64_t i, uint8_t c) { return ((ref_base*)h)->wt_rank(i, c); }
uint64_t vref_inverse_select(void* h, uint64_t i, uint8_t* c) { return ((ref_base*)h)->inverse_select(i, c); }
uint64_t vref_bv_size(void* h) { return ((ref_base*)h)->bv_size(); }
uint64_t vref_bv_rank1(void* h, uint64_t idx) { return ((ref_base*)h)->bv_rank1(idx); }
void vref_bv_bits(void* h, uint64_t* words)
{
    ref_base* b = (ref_base*)h;
    uint64_t n = b->bv_size();
    for (uint64_t i = 0; i < (n + 63) / 64; ++i) words[i] = 0;
    for (uint64_t i = 0; i < n; ++i) if (b->bv_get(i)) words[i >> 6] |= 1ULL << (i & 63);
}
uint64_t vref_n_nodes(void* h) { return ((ref_base*)h)->n_nodes(); }
void vref_node(void* h, uint64_t v, uint64_t* bv_pos, uint64_t* sz, int* leaf, int* sym, int* c0, int* c1)
{
    ((ref_base*)h)->node(v, bv_pos, sz, leaf, sym, c0, c1);
}
void vref_alphabet(void* h, uint8_t* c2c, uint64_t* C, uint32_t* sigma) { ((ref_base*)h)->alphabet(c2c, C, sigma); }
uint64_t vref_sa(void* h, uint64_t i) { return ((ref_base*)h)->sa(i); }
uint64_t vref_lf(void* h, uint64_t i) { return ((ref_base*)h)->lf_at(i); }
int vref_write_csa_image(void* h, const char* path, const uint64_t* sa, uint64_t n) { return ((ref_base*)h)->write_csa_image(path, sa, n); }

// Stand-alone bit-vector rank: variant 0 = rank_support_v<1,1>, 1 = rank_support_v5<1,1>, 2 = rrr_vector<63>.
void vref_bitrank(const uint64_t* words, uint64_t nbits, int variant,
                  const uint64_t* idx, uint64_t k, uint64_t* out)
{
    bit_vector bv(nbits, 0);
    for (uint64_t i = 0; i < nbits; ++i) bv[i] = (words[i >> 6] >> (i & 63)) & 1;
    if (variant == 0) {
        rank_support_v<1, 1> rs(&bv);
        for (uint64_t j = 0; j < k; ++j) out[j] = rs.rank(idx[j]);
    } else if (variant == 1) {
        rank_support_v5<1, 1> rs(&bv);
        for (uint64_t j = 0; j < k; ++j) out[j] = rs.rank(idx[j]);
    } else {
        rrr_vector<63> rv(bv);
        rrr_vector<63>::rank_1_type rs(&rv);
        for (uint64_t j = 0; j < k; ++j) out[j] = rs.rank(idx[j]);
    }
}

// Raw rank_support_v block array (rank_support_v.hpp:75-105) for layout parity of the oracle.
uint64_t vref_rank_v_blocks(const uint64_t* words, uint64_t nbits, uint64_t* out, uint64_t cap)
{
    bit_vector bv(nbits, 0);
    for (uint64_t i = 0; i < nbits; ++i) bv[i] = (words[i >> 6] >> (i & 63)) & 1;
    rank_support_v<1, 1> rs(&bv);
    std::stringstream ss;
    rs.serialize(ss);
    std::string s = ss.str();          // int_vector<64>: u64 size-in-bits, then words
    uint64_t bits = 0;
    memcpy(&bits, s.data(), 8);
    uint64_t nw = bits / 64;
    for (uint64_t i = 0; i < nw && i < cap; ++i) memcpy(&out[i], s.data() + 8 + 8 * i, 8);
    return nw;
}

// Loads a csa_wt<wt_huff<>,32,64> file image member by member with the reference's own load() functions (wt_huff incl. its
// rank_support_v and both select_support_mcl, int_vector<> samples, byte_alphabet) and checks it against the truth handed
// in (BWT and SA of the text): access, rank, select through the loaded structures, both sample vectors, the alphabet.
// Returns 0 if everything agrees, otherwise a code telling which check failed first.
int vref_check_csa_image(const char* path, const uint8_t* bwt, const uint64_t* sa, uint64_t n, uint64_t step)
{
    try {
        std::ifstream in(path, std::ios::binary);
        if (!in) return 1;
        wt_v wt;
        wt.load(in);
        int_vector<> sa_sample, isa_sample;
        sa_sample.load(in);
        isa_sample.load(in);
        byte_alphabet alpha;
        alpha.load(in);
        if (!in) return 2;
        in.peek();
        if (!in.eof()) return 3;                                        // trailing bytes
        if (wt.size() != n) return 10;
        if (step == 0) step = 1;
        // access + rank
        std::vector<uint64_t> cnt(256, 0);
        for (uint64_t i = 0; i < n; ++i) {
            if (i % step == 0) {
                if (wt[i] != bwt[i]) return 11;
                if (wt.rank(i, bwt[i]) != cnt[bwt[i]]) return 12;
                auto is = wt.inverse_select(i);
                if (is.second != bwt[i] || is.first != cnt[bwt[i]]) return 13;
            }
            ++cnt[bwt[i]];
            // select through select_support_mcl of the concatenated bit-vector (wt_pc::select)
            if (i % step == 0 && wt.select(cnt[bwt[i]], bwt[i]) != i) return 14;
        }
        // samples
        if (sa_sample.size() != (n + 31) / 32) return 20;
        for (uint64_t j = 0; j < sa_sample.size(); ++j) if (sa_sample[j] != sa[32 * j]) return 21;
        if (isa_sample.size() != (n - 1) / 64 + 1) return 22;
        for (uint64_t j = 0; j < isa_sample.size(); ++j) if (isa_sample[j] >= n || sa[isa_sample[j]] != 64 * j) return 23;
        if (sa_sample.width() != bits::hi(n) + 1 || isa_sample.width() != bits::hi(n) + 1) return 24;
        // alphabet
        uint64_t sigma = 0;
        for (int c = 0; c < 256; ++c) if (cnt[c]) {
            if (alpha.char2comp[c] != sigma || alpha.comp2char[sigma] != c) return 30;
            if (alpha.C[sigma + 1] - alpha.C[sigma] != cnt[c]) return 31;
            ++sigma;
        }
        if (alpha.sigma != sigma || alpha.C[sigma] != n) return 32;
        return 0;
    } catch (...) {
        return 99;
    }
}

// ---- the tree of vlg_index: wt_int<bit_vector_il<>, rank_support_il<>> over a vector of values (the suffix array) ----------------
// Built by the reference's own constructor from an int_vector_buffer, as construct(wts, KEY_SA file) does (vlg_index.hpp:386-387,
// wt_int.hpp:182-270).  count_less / quantile below walk the tree ONLY through the reference's expand(v) / expand(v, range) /
// value_range / size / is_leaf -- the calls vlg_iterator's wt_range_walker makes (wt_helper.hpp:726-785) -- so they say what the
// reference's tree answers for a suffix-array range.
typedef wt_int<bit_vector_il<>, rank_support_il<>> wtsa_ref_type;

void* vrefw_create(const uint64_t* vals, uint64_t n)
{
    try {
        std::string f = "@vref_sa_" + std::to_string(g_seq++);
        {
            int_vector<> v(n, 0, 64);
            for (uint64_t i = 0; i < n; ++i) v[i] = vals[i];
            store_to_file(v, f);
        }
        wtsa_ref_type* wt = nullptr;
        {
            int_vector_buffer<> buf(f);
            wt = new wtsa_ref_type(buf, n);
        }
        sdsl::remove(f);
        return wt;
    } catch (...) {}
    return nullptr;
}
void vrefw_destroy(void* h) { delete (wtsa_ref_type*)h; }
uint64_t vrefw_size(void* h) { return ((wtsa_ref_type*)h)->size(); }
uint32_t vrefw_levels(void* h) { return ((wtsa_ref_type*)h)->max_level; }
uint64_t vrefw_access(void* h, uint64_t i) { return (*(wtsa_ref_type*)h)[i]; }
// the concatenation of all level bit-vectors (wt_int::tree): size * max_level bits, level l = bits [l * size, (l + 1) * size)
void vrefw_tree_bits(void* h, uint64_t* words)
{
    const wtsa_ref_type& wt = *(wtsa_ref_type*)h;
    const uint64_t nb = wt.tree.size();
    for (uint64_t i = 0; i < (nb + 63) / 64; ++i) words[i] = 0;
    for (uint64_t i = 0; i < nb; ++i) if (wt.tree[i]) words[i >> 6] |= 1ULL << (i & 63);
}
// number of values < x among wt[l, l + len)
uint64_t vrefw_count_less(void* h, uint64_t l, uint64_t len, uint64_t x)
{
    const wtsa_ref_type& wt = *(wtsa_ref_type*)h;
    if (!len) return 0;
    auto v = wt.root();
    sdsl::range_type r = {{l, l + len - 1}};
    uint64_t acc = 0;
    while (!wt.is_leaf(v)) {
        auto vr = wt.value_range(v);
        if (x <= vr[0]) return acc;
        if (x > vr[1]) return acc + (r[1] + 1 - r[0]);
        auto ch = wt.expand(v);
        auto rs = wt.expand(v, r);
        const uint64_t left_n = rs[0][1] + 1 - rs[0][0];           // (an empty range has [s, s - 1])
        if (x > wt.value_range(ch[0])[1]) {                          // everything in the left child is smaller
            acc += left_n;
            v = ch[1]; r = rs[1];
        } else { v = ch[0]; r = rs[0]; }
        if (r[1] + 1 == r[0]) return acc;
    }
    return acc + (wt.sym(v) < x ? r[1] + 1 - r[0] : 0);
}
// the q-th smallest (0-based) of wt[l, l + len), q < len
uint64_t vrefw_quantile(void* h, uint64_t l, uint64_t len, uint64_t q)
{
    const wtsa_ref_type& wt = *(wtsa_ref_type*)h;
    auto v = wt.root();
    sdsl::range_type r = {{l, l + len - 1}};
    while (!wt.is_leaf(v)) {
        auto ch = wt.expand(v);
        auto rs = wt.expand(v, r);
        const uint64_t left_n = rs[0][1] + 1 - rs[0][0];
        if (q < left_n) { v = ch[0]; r = rs[0]; }
        else { q -= left_n; v = ch[1]; r = rs[1]; }
    }
    return wt.sym(v);
}

// ---- vlg_iterator over the reference's OWN wt_range_walker -----------------------------------------------------------------------
// vlg_index.hpp itself cannot be included (it pulls suffix_arrays.hpp -> construct_sa.hpp -> divsufsort.h), but everything its iterator
// stands on can: wt_int, wt_node_cache and wt_range_walker (wt_helper.hpp:691-785) are the reference's code, compiled above.  The
// three loops of vlg_iterator -- relax (vlg_index.hpp:227-249), pull_forward (:254-266), next (:269-291) -- and its constructor /
// operator++ (:303-321, 345-352) are RESTATED here line by line over those real walkers; the suffix-array ranges come from the caller
// (forward_search needs the text and the same unbuildable header).  What this pins: the tuples the paper's algorithm yields on the
// reference's tree -- an implementation of the VLG semantics that shares no code with the merge join of oracle/vlg_oracle.c.
// gaps[i] = (lo, hi) as gapped_pattern_query stores them (start-to-start distances, vlg_index.hpp:95).
uint64_t vrefw_vlg_iterate(void* h, uint32_t k, const uint64_t* sp, const uint64_t* ep, const uint64_t* gap_lo, const uint64_t* gap_hi,
                           uint64_t last_subpattern_size, uint64_t* out_tuples, uint64_t cap)
{
    typedef wt_range_walker<wtsa_ref_type> walker;
    const wtsa_ref_type& wt = *(wtsa_ref_type*)h;
    std::vector<walker> lex_ranges;
    auto root_node = wt_node_cache<wtsa_ref_type>(wt.root(), wt);
    for (uint32_t i = 0; i < k; ++i) {
        lex_ranges.emplace_back(wt, range_type({sp[i], ep[i]}), root_node);
        if (sp[i] > ep[i]) return 0;                                 // :315-316 shortcut on empty range
    }
    auto size = [&]() { return lex_ranges.size(); };
    auto relax = [&]() -> bool {                                     // :227-249
        bool redo = true;
        while (redo) {
            redo = false;
            for (size_t i = 1; i < size(); ++i) {
                if (lex_ranges[i - 1].current_node().range_end + gap_hi[i - 1] < lex_ranges[i].current_node().range_begin) {
                    lex_ranges[i - 1].next_right();
                    redo = true;
                    if (!lex_ranges[i - 1].has_more()) return false;
                }
                if (lex_ranges[i - 1].current_node().range_begin + gap_lo[i - 1] > lex_ranges[i].current_node().range_end) {
                    lex_ranges[i].next_right();
                    redo = true;
                    if (!lex_ranges[i].has_more()) return false;
                }
            }
        }
        return true;
    };
    auto pull_forward = [&]() -> bool {                              // :254-266
        auto last_pos = lex_ranges[lex_ranges.size() - 1].current_node().range_begin;
        while (lex_ranges[0].has_more() && lex_ranges[0].current_node().range_end <= last_pos) lex_ranges[0].next_right();
        while (lex_ranges[0].next_leaf() && lex_ranges[0].current_node().range_begin < last_pos + last_subpattern_size) ;
        return lex_ranges[0].has_more();
    };
    bool finished = false;
    auto next = [&]() {                                              // :269-291
        while (relax()) {
            size_t r = 1, j = 0;
            bool found = false;
            for (size_t i = 0; i < size(); ++i) {
                auto lr = lex_ranges[i].current_node().node.size;
                if (lr > r) { r = lr; j = i; found = true; }
            }
            if (found) lex_ranges[j].next_down();
            else return;
        }
        finished = true;
    };
    next();
    uint64_t m = 0;
    while (!finished) {
        if (m < cap && out_tuples) for (uint32_t i = 0; i < k; ++i) out_tuples[m * k + i] = lex_ranges[i].current_node().range_begin;   // operator[] :337-340
        ++m;
        if (pull_forward()) next(); else finished = true;            // operator++ :345-352
    }
    return m;
}

// ---- wt_int<> (bit_vector + rank_support_v), the BWT container of csa_wt<wt_int<>> for integer alphabets ----------------------------
typedef wt_int<> wt_int_plain;
void* vrefi_create(const uint64_t* vals, uint64_t n)
{
    try {
        std::string f = "@vref_ibwt_" + std::to_string(g_seq++);
        {
            int_vector<> v(n, 0, 64);
            for (uint64_t i = 0; i < n; ++i) v[i] = vals[i];
            store_to_file(v, f);
        }
        wt_int_plain* wt = nullptr;
        {
            int_vector_buffer<> buf(f);
            wt = new wt_int_plain(buf, n);
        }
        sdsl::remove(f);
        return wt;
    } catch (...) {}
    return nullptr;
}
void vrefi_destroy(void* h) { delete (wt_int_plain*)h; }
uint32_t vrefi_levels(void* h) { return ((wt_int_plain*)h)->max_level; }
uint64_t vrefi_sigma(void* h) { return ((wt_int_plain*)h)->sigma; }
uint64_t vrefi_rank(void* h, uint64_t i, uint64_t c) { return ((wt_int_plain*)h)->rank(i, c); }
uint64_t vrefi_inverse_select(void* h, uint64_t i, uint64_t* c)
{
    auto rc = ((wt_int_plain*)h)->inverse_select(i);
    *c = rc.second;
    return rc.first;
}
void vrefi_tree_bits(void* h, uint64_t* words)
{
    const wt_int_plain& wt = *(wt_int_plain*)h;
    const uint64_t nb = wt.tree.size();
    for (uint64_t i = 0; i < (nb + 63) / 64; ++i) words[i] = 0;
    for (uint64_t i = 0; i < nb; ++i) if (wt.tree[i]) words[i >> 6] |= 1ULL << (i & 63);
}

// ---- int_alphabet (csa_alphabet_strategy.hpp:394-470): built by the reference's constructor from an integer text -----------------
// out_C must hold sigma + 1 entries, out_comp2char sigma entries (call with nulls first to get sigma).
uint64_t vref_int_alphabet(const uint64_t* text, uint64_t n, uint64_t* out_C, uint64_t* out_comp2char)
{
    try {
        std::string f = "@vref_itext_" + std::to_string(g_seq++);
        {
            int_vector<> v(n, 0, 64);
            for (uint64_t i = 0; i < n; ++i) v[i] = text[i];
            store_to_file(v, f);
        }
        uint64_t sigma = 0;
        {
            int_vector_buffer<> buf(f);
            int_alphabet<> a(buf, n);
            sigma = a.sigma;
            if (out_C) for (uint64_t i = 0; i <= sigma; ++i) out_C[i] = a.C[i];
            if (out_comp2char) for (uint64_t i = 0; i < sigma; ++i) {
                out_comp2char[i] = a.comp2char[i];
                if (a.char2comp[a.comp2char[i]] != i) return ~0ull;
            }
        }
        sdsl::remove(f);
        return sigma;
    } catch (...) {}
    return ~0ull;
}

} // extern "C"
