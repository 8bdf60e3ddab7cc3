/* oracle/vlg_oracle_int.c -- TEST INFRASTRUCTURE ONLY (see vlg_oracle.h).
 *
 * CPU restatement of the reference's FM-index for INTEGER alphabets and of its second SA sampling strategy
 * (SURVEY.md 8f-4), in the reference's own data layout:
 *   csa_wt<wt_int<>, dens, ., sa_order_sa_sampling | text_order_sa_sampling<bit_vector>, ., int_alphabet<>>
 *     int_alphabet ctor / char2comp / C       include/sdsl/csa_alphabet_strategy.hpp:394-470, 496-536
 *     wt_int ctor (level-wise bit emission)   include/sdsl/wt_int.hpp:182-270
 *     wt_int::rank / inverse_select           include/sdsl/wt_int.hpp:370-395, 405-430
 *     LF                                      include/sdsl/suffix_array_helper.hpp:336-349
 *     csa_wt::operator[]                      include/sdsl/csa_wt.hpp:335-348
 *     _sa_order_sampling                      include/sdsl/csa_sampling_strategy.hpp:64-112
 *     _text_order_sampling                    include/sdsl/csa_sampling_strategy.hpp:127-246 (marked bit-vector over SA indices,
 *                                             samples SA/dens in the order of the marked indices)
 *     backward_search / locate                include/sdsl/suffix_array_algorithm.hpp:250-326, 604-619
 *     gapped_pattern_query<int_alphabet_tag>  include/sdsl/vlg_index.hpp:57-69, 72-105 (sub-patterns = whitespace-separated
 *                                             decimals, gaps count symbols)
 * Parity status: int_alphabet and wt_int (ctor bits, rank via expand, operator[]) are PINNED against the reference's own code in
 * oracle/_ref (ref_glue.cpp: vref_int_alphabet, vrefw_*); csa_sampling_strategy.hpp cannot be compiled here (it includes
 * wavelet_trees.hpp -> construct.hpp -> construct_sa.hpp -> divsufsort.h, an empty submodule), so the text-order sampling is
 * pinned by its defining property only: csa[i] == SA[i] for every i, marked[i] <=> SA[i] % dens == 0.  The byte-alphabet variant
 * of text-order sampling reuses vlg_oracle.c's index (vlgo_text_order_* below).
 * Limit of this restatement (not of the reference): the suffix sorter ranks symbols as text[i] + 1 with 0 = "behind the end", so a
 * text must not hold the symbol 2^64 - 1 (it would wrap to 0); every other 64-bit symbol is fine (tests go up to 2^64 - 3).
 */
#include "vlg_oracle.h"
#include <stdlib.h>
#include <string.h>

static inline unsigned hi_bit(uint64_t x) { return x ? 63u - (unsigned)__builtin_clzll(x) : 0u; } /* bits::hi */

struct vlgo_int_index {
    uint64_t n;            /* symbols + 1 (the sentinel 0)                                        */
    uint64_t sigma;        /* distinct symbols of text + sentinel                                 */
    uint64_t* comp2char;   /* [sigma] ascending (int_alphabet: select on m_char, or the identity) */
    uint64_t* C;           /* [sigma + 1]                                                         */
    uint32_t levels;       /* wt_int::max_level = hi(max symbol, at least 1) + 1                  */
    uint64_t* tree;        /* n * levels bits, level l = bits [l n, (l + 1) n)                     */
    uint64_t* rb;          /* rank_support_v block array over tree                                */
    uint32_t dens;
    int text_order;
    uint64_t n_samples;
    uint64_t* samples;     /* sa order: SA[0], SA[dens], ...; text order: SA[i] / dens of the marked i, ascending i */
    uint64_t* marked;      /* text order: bit i = (SA[i] % dens == 0)                             */
    uint64_t* marked_rb;
    uint64_t* bwt;
};

/* ---- suffix array of an integer text (own sorter: prefix doubling; the SA is unique) --------------------------------------- */
typedef struct { uint64_t a, b; uint64_t i; } sa_key;
static int cmp_key(const void* x, const void* y)
{
    const sa_key* p = (const sa_key*)x; const sa_key* q = (const sa_key*)y;
    if (p->a != q->a) return p->a < q->a ? -1 : 1;
    if (p->b != q->b) return p->b < q->b ? -1 : 1;
    return 0;
}
static int cmp_u64(const void* a, const void* b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : (x > y ? 1 : 0); }

/* text[0, n) with text[n-1] = 0 the unique smallest symbol */
int vlgo_int_suffix_array(const uint64_t* text, uint64_t n, uint64_t* sa)
{
    if (!n) return 0;
    sa_key* k = (sa_key*)malloc(n * sizeof(sa_key));
    uint64_t* rank = (uint64_t*)malloc(n * 8);
    if (!k || !rank) { free(k); free(rank); return -1; }
    for (uint64_t i = 0; i < n; ++i) rank[i] = text[i] + 1;                  /* 0 = "behind the end" */
    for (uint64_t h = 1;; h <<= 1) {
        for (uint64_t i = 0; i < n; ++i) { k[i].a = rank[i]; k[i].b = i + h < n ? rank[i + h] : 0; k[i].i = i; }
        qsort(k, n, sizeof(sa_key), cmp_key);
        uint64_t r = 0;
        for (uint64_t j = 0; j < n; ++j) {
            if (j && cmp_key(&k[j - 1], &k[j]) != 0) ++r;
            rank[k[j].i] = r + 1;
        }
        if (r + 1 == n || h >= n) break;
    }
    for (uint64_t j = 0; j < n; ++j) sa[j] = k[j].i;
    free(k); free(rank);
    return 0;
}

static inline int tree_bit(const vlgo_int_index* x, uint64_t p) { return (int)((x->tree[p >> 6] >> (p & 63)) & 1); }
static inline uint64_t tree_rank(const vlgo_int_index* x, uint64_t p) { return vlgo_rank_v(x->tree, x->rb, p); }

void vlgo_int_free(vlgo_int_index* x)
{
    if (!x) return;
    free(x->comp2char); free(x->C); free(x->tree); free(x->rb); free(x->samples); free(x->marked); free(x->marked_rb); free(x->bwt);
    free(x);
}

/* text: n_text symbols, none of them 0 (construct() refuses such a text: include/sdsl/construct.hpp:36-45 contains_no_zero_symbol) */
vlgo_int_index* vlgo_int_build(const uint64_t* text, uint64_t n_text, uint32_t dens, int text_order)
{
    for (uint64_t i = 0; i < n_text; ++i) if (text[i] == 0) return NULL;
    vlgo_int_index* x = (vlgo_int_index*)calloc(1, sizeof *x);
    const uint64_t n = n_text + 1;
    x->n = n; x->dens = dens ? dens : 32; x->text_order = text_order;
    uint64_t* t = (uint64_t*)malloc(n * 8);
    uint64_t* sa = (uint64_t*)malloc(n * 8);
    memcpy(t, text, n_text * 8);
    t[n_text] = 0;
    vlgo_int_suffix_array(t, n, sa);
    x->bwt = (uint64_t*)malloc(n * 8);
    for (uint64_t i = 0; i < n; ++i) x->bwt[i] = sa[i] ? t[sa[i] - 1] : t[n - 1];
    /* ---- int_alphabet over the BWT (csa_alphabet_strategy.hpp:496-536): the std::map D = sorted distinct symbols with counts */
    uint64_t* sorted = (uint64_t*)malloc(n * 8);
    memcpy(sorted, x->bwt, n * 8);
    qsort(sorted, n, 8, cmp_u64);
    uint64_t sigma = 0;
    for (uint64_t i = 0; i < n; ++i) if (!i || sorted[i] != sorted[i - 1]) ++sigma;
    x->sigma = sigma;
    x->comp2char = (uint64_t*)malloc(sigma * 8);
    x->C = (uint64_t*)calloc(sigma + 1, 8);
    for (uint64_t i = 0, c = 0; i < n; ++i) {
        if (!i || sorted[i] != sorted[i - 1]) { x->comp2char[c] = sorted[i]; x->C[c] = i; ++c; }
    }
    x->C[sigma] = n;
    /* ---- wt_int over the BWT (wt_int.hpp:182-270) ------------------------------------------------------------------------------ */
    uint64_t mx = 1;                                                          /* "value_type x = 1" */
    for (uint64_t i = 0; i < n; ++i) if (x->bwt[i] > mx) mx = x->bwt[i];
    x->levels = hi_bit(mx) + 1;
    const uint64_t bits = n * x->levels;
    x->tree = (uint64_t*)calloc(bits / 64 + 2, 8);
    uint64_t* rac = (uint64_t*)malloc(n * 8);
    uint64_t* buf1 = (uint64_t*)malloc(n * 8);
    memcpy(rac, x->bwt, n * 8);
    uint64_t tree_pos = 0;
    uint64_t mask_old = x->levels >= 64 ? 0 : 1ULL << x->levels;
    for (uint32_t k = 0; k < x->levels; ++k) {
        uint64_t start = 0;
        const uint64_t mask_new = 1ULL << (x->levels - k - 1);
        do {
            uint64_t i = start, cnt0 = 0, cnt1 = 0;
            const uint64_t start_value = rac[i] & mask_old;
            uint64_t v;
            while (i < n && ((v = rac[i]) & mask_old) == start_value) {
                if (v & mask_new) { x->tree[tree_pos >> 6] |= 1ULL << (tree_pos & 63); buf1[cnt1++] = v; }
                else rac[start + cnt0++] = v;
                ++tree_pos; ++i;
            }
            if (k + 1 < x->levels) for (uint64_t j = 0; j < cnt1; ++j) rac[start + cnt0 + j] = buf1[j];
            start += cnt0 + cnt1;
        } while (start < n);
        mask_old += mask_new;
    }
    free(rac); free(buf1);
    x->rb = (uint64_t*)calloc(2 * ((bits >> 9) + 2), 8);
    vlgo_rank_v_build(x->tree, bits, x->rb);
    /* ---- SA samples ----------------------------------------------------------------------------------------------------------------- */
    x->n_samples = (n + x->dens - 1) / x->dens;
    x->samples = (uint64_t*)calloc(x->n_samples + 1, 8);
    if (!text_order) {
        for (uint64_t i = 0, j = 0; i < n; i += x->dens) x->samples[j++] = sa[i];                  /* csa_sampling_strategy.hpp:89-98 */
    } else {
        x->marked = (uint64_t*)calloc(n / 64 + 2, 8);
        uint64_t cnt = 0;
        for (uint64_t i = 0; i < n; ++i)                                                            /* :158-164 */
            if (sa[i] % x->dens == 0) { x->marked[i >> 6] |= 1ULL << (i & 63); x->samples[cnt++] = sa[i] / x->dens; }
        x->marked_rb = (uint64_t*)calloc(2 * ((n >> 9) + 2), 8);
        vlgo_rank_v_build(x->marked, n, x->marked_rb);
    }
    free(t); free(sa); free(sorted);
    return x;
}

uint64_t vlgo_int_size(const vlgo_int_index* x) { return x->n; }
uint64_t vlgo_int_sigma(const vlgo_int_index* x) { return x->sigma; }
const uint64_t* vlgo_int_C(const vlgo_int_index* x) { return x->C; }
const uint64_t* vlgo_int_comp2char(const vlgo_int_index* x) { return x->comp2char; }
uint32_t vlgo_int_levels(const vlgo_int_index* x) { return x->levels; }
const uint64_t* vlgo_int_tree(const vlgo_int_index* x) { return x->tree; }
const uint64_t* vlgo_int_bwt(const vlgo_int_index* x) { return x->bwt; }
uint64_t vlgo_int_n_samples(const vlgo_int_index* x) { return x->n_samples; }
const uint64_t* vlgo_int_samples(const vlgo_int_index* x) { return x->samples; }
const uint64_t* vlgo_int_marked(const vlgo_int_index* x) { return x->marked; }

/* int_alphabet::char2comp (csa_alphabet_strategy.hpp:421-437): 0 for a symbol that does not occur (and for the sentinel itself) */
uint64_t vlgo_int_char2comp(const vlgo_int_index* x, uint64_t c)
{
    uint64_t lo = 0, hi = x->sigma;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (x->comp2char[mid] < c) lo = mid + 1; else hi = mid; }
    return lo < x->sigma && x->comp2char[lo] == c ? lo : 0;
}

/* wt_int::rank (wt_int.hpp:370-395) */
uint64_t vlgo_int_rank(const vlgo_int_index* x, uint64_t i, uint64_t c)
{
    if (x->levels < 64 && (1ULL << x->levels) <= c) return 0;
    uint64_t offset = 0, node_size = x->n;
    uint64_t mask = 1ULL << (x->levels - 1);
    for (uint32_t k = 0; k < x->levels && i; ++k) {
        const uint64_t ones_before_o = tree_rank(x, offset);
        const uint64_t ones_before_i = tree_rank(x, offset + i) - ones_before_o;
        const uint64_t ones_before_end = tree_rank(x, offset + node_size) - ones_before_o;
        if (c & mask) { offset += node_size - ones_before_end; node_size = ones_before_end; i = ones_before_i; }
        else { node_size = node_size - ones_before_end; i = i - ones_before_i; }
        offset += x->n;
        mask >>= 1;
    }
    return i;
}

/* wt_int::inverse_select (wt_int.hpp:405-430) */
uint64_t vlgo_int_inverse_select(const vlgo_int_index* x, uint64_t i, uint64_t* c_out)
{
    uint64_t c = 0, node_size = x->n, offset = 0;
    for (uint32_t k = 0; k < x->levels; ++k) {
        const uint64_t ones_before_o = tree_rank(x, offset);
        const uint64_t ones_before_i = tree_rank(x, offset + i) - ones_before_o;
        const uint64_t ones_before_end = tree_rank(x, offset + node_size) - ones_before_o;
        c <<= 1;
        if (tree_bit(x, offset + i)) { offset += node_size - ones_before_end; node_size = ones_before_end; i = ones_before_i; c |= 1; }
        else { node_size = node_size - ones_before_end; i = i - ones_before_i; }
        offset += x->n;
    }
    *c_out = c;
    return i;
}

/* LF (suffix_array_helper.hpp:336-349) */
uint64_t vlgo_int_lf(const vlgo_int_index* x, uint64_t i)
{
    uint64_t c;
    const uint64_t j = vlgo_int_inverse_select(x, i, &c);
    return x->C[vlgo_int_char2comp(x, c)] + j;
}

/* csa_wt::operator[] (csa_wt.hpp:335-348) with either sampling strategy */
uint64_t vlgo_int_sa(const vlgo_int_index* x, uint64_t i, uint64_t* lf_steps)
{
    uint64_t off = 0;
    if (!x->text_order) {
        while (i % x->dens) { i = vlgo_int_lf(x, i); ++off; }                                       /* is_sampled: :102-105 */
    } else {
        while (!((x->marked[i >> 6] >> (i & 63)) & 1)) { i = vlgo_int_lf(x, i); ++off; }            /* :185-188 */
    }
    const uint64_t r = x->text_order ? x->samples[vlgo_rank_v(x->marked, x->marked_rb, i)] * x->dens    /* :191-194 */
                                     : x->samples[i / x->dens];
    if (lf_steps) *lf_steps += off;
    return r + off < x->n ? r + off : r + off - x->n;
}

/* backward_search (suffix_array_algorithm.hpp:250-278, 305-326) */
uint64_t vlgo_int_backward_search(const vlgo_int_index* x, const uint64_t* pat, uint64_t m, uint64_t* lo, uint64_t* ro)
{
    uint64_t l = 0, r = x->n - 1;
    while (m > 0 && r + 1 - l > 0) {
        const uint64_t c = pat[--m];
        const uint64_t cc = vlgo_int_char2comp(x, c);
        if (cc == 0 && c > 0) { l = 1; r = 0; }
        else {
            const uint64_t c_begin = x->C[cc];
            if (l == 0 && r + 1 == x->n) { l = c_begin; r = x->C[cc + 1] - 1; }
            else { const uint64_t nl = c_begin + vlgo_int_rank(x, l, c); r = c_begin + vlgo_int_rank(x, r + 1, c) - 1; l = nl; }
        }
    }
    if (lo) *lo = l;
    if (ro) *ro = r;
    return r + 1 - l;
}

uint64_t vlgo_int_locate(const vlgo_int_index* x, const uint64_t* pat, uint64_t m, uint64_t* out, uint64_t cap)
{
    uint64_t l, r;
    const uint64_t cnt = vlgo_int_backward_search(x, pat, m, &l, &r);
    for (uint64_t j = 0; j < cnt && j < cap; ++j) out[j] = vlgo_int_sa(x, l + j, NULL);
    return cnt;
}

/* gapped_pattern_query<int_alphabet_tag> (vlg_index.hpp:57-69, 72-105): the gap structure is the library dialect's; every sub-pattern
 * is read as whitespace-separated decimals (istringstream >> uint64_t: reading stops at the first token that is not a number);
 * gaps[i] = (min + |s_i|, max + |s_i|) with |s_i| counted in SYMBOLS.  syms receives all symbols, sub_off[k + 1] their offsets. */
int vlgo_parse_int(const uint8_t* re, uint64_t len, vlgo_query* q, uint64_t* syms, uint64_t syms_cap, uint64_t* sub_off)
{
    const int rc = vlgo_parse(re, len, 0, q);
    if (rc != VLGO_OK && rc != VLGO_E_EMPTY_SUBPATTERN) return rc;
    uint64_t ns = 0;
    for (uint32_t i = 0; i < q->k; ++i) {
        sub_off[i] = ns;
        const uint8_t* s = q->sub[i];
        const uint8_t* e = s + q->sub_len[i];
        for (;;) {
            while (s < e && (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r' || *s == '\f' || *s == '\v')) ++s;
            if (s == e || *s < '0' || *s > '9') break;
            uint64_t v = 0;
            int overflow = 0;
            while (s < e && *s >= '0' && *s <= '9') { if (v > (~0ULL - (uint64_t)(*s - '0')) / 10) overflow = 1; v = v * 10 + (uint64_t)(*s - '0'); ++s; }
            if (overflow) break;                                                  /* the stream fails: no more symbols */
            if (ns >= syms_cap) return VLGO_E_PARSE_NUM;
            syms[ns++] = v;
        }
    }
    sub_off[q->k] = ns;
    for (uint32_t i = 0; i < q->k; ++i) if (sub_off[i + 1] == sub_off[i]) return VLGO_E_EMPTY_SUBPATTERN;
    for (uint32_t i = 1; i < q->k; ++i) {
        const uint64_t chars = q->sub_len[i - 1], nsym = sub_off[i] - sub_off[i - 1];
        q->lo[i] = q->lo[i] - chars + nsym;
        q->hi[i] = q->hi[i] - chars + nsym;
    }
    q->end_len = sub_off[q->k] - sub_off[q->k - 1];
    return VLGO_OK;
}

/* the whole path for one integer query: backward_search, locate, sort, join (as vlgo_search does for bytes) */
uint64_t vlgo_int_search(const vlgo_int_index* x, const vlgo_query* q, const uint64_t* syms, const uint64_t* sub_off, uint64_t* out, uint64_t cap,
                         uint64_t* stats)
{
    uint64_t* L[VLGO_MAX_SUB];
    uint64_t len[VLGO_MAX_SUB], l[VLGO_MAX_SUB], r[VLGO_MAX_SUB];
    for (uint32_t i = 0; i < q->k; ++i) {
        len[i] = vlgo_int_backward_search(x, syms + sub_off[i], sub_off[i + 1] - sub_off[i], &l[i], &r[i]);
        if (!len[i]) return 0;                                                    /* vlg_index.hpp:315-316 */
    }
    for (uint32_t i = 0; i < q->k; ++i) {
        L[i] = (uint64_t*)malloc((len[i] + 1) * 8);
        for (uint64_t j = 0; j < len[i]; ++j) L[i][j] = vlgo_int_sa(x, l[i] + j, stats ? &stats[1] : NULL);
        if (stats) stats[0] += len[i];
        qsort(L[i], len[i], 8, cmp_u64);
    }
    const uint64_t m = vlgo_join(q->k, (const uint64_t* const*)L, len, q->lo, q->hi, q->end_len, out, cap);
    for (uint32_t i = 0; i < q->k; ++i) free(L[i]);
    return m;
}

/* ---- text-order sampling for the BYTE index of vlg_oracle.c: the marked bit-vector and the condensed samples, computed from the
 * index alone (every SA value through csa[i]); csa[i] with them restates csa_wt.hpp:335-348 + csa_sampling_strategy.hpp:185-194 */
struct vlgo_text_order {
    uint64_t n, n_samples;
    uint32_t dens;
    uint64_t* marked;
    uint64_t* rb;
    uint64_t* samples;
};
typedef struct vlgo_text_order vlgo_text_order;

vlgo_text_order* vlgo_text_order_build(const vlgo_index* x, uint32_t dens)
{
    vlgo_text_order* t = (vlgo_text_order*)calloc(1, sizeof *t);
    const uint64_t n = vlgo_size(x);
    t->n = n; t->dens = dens; t->n_samples = (n + dens - 1) / dens;
    t->marked = (uint64_t*)calloc(n / 64 + 2, 8);
    t->samples = (uint64_t*)calloc(t->n_samples + 1, 8);
    uint64_t cnt = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t v = vlgo_sa(x, i, NULL, NULL);
        if (v % dens == 0) { t->marked[i >> 6] |= 1ULL << (i & 63); t->samples[cnt++] = v / dens; }
    }
    t->rb = (uint64_t*)calloc(2 * ((n >> 9) + 2), 8);
    vlgo_rank_v_build(t->marked, n, t->rb);
    return t;
}
void vlgo_text_order_free(vlgo_text_order* t) { if (!t) return; free(t->marked); free(t->rb); free(t->samples); free(t); }
const uint64_t* vlgo_text_order_marked(const vlgo_text_order* t) { return t->marked; }
const uint64_t* vlgo_text_order_samples(const vlgo_text_order* t) { return t->samples; }
uint64_t vlgo_text_order_n_samples(const vlgo_text_order* t) { return t->n_samples; }
uint64_t vlgo_text_order_sa(const vlgo_index* x, const vlgo_text_order* t, uint64_t i, uint64_t* lf_steps)
{
    uint64_t off = 0;
    while (!((t->marked[i >> 6] >> (i & 63)) & 1)) { i = vlgo_lf(x, i); ++off; }
    const uint64_t r = t->samples[vlgo_rank_v(t->marked, t->rb, i)] * t->dens;
    if (lf_steps) *lf_steps += off;
    return r + off < t->n ? r + off : r + off - t->n;
}
