/* oracle/vlg_oracle.c -- TEST INFRASTRUCTURE ONLY (see vlg_oracle.h).
 *
 * CPU restatement, in plain C, of the reference's FM-index path for variable-length-gap
 * matching.  Every function cites the reference file:line (relative to /root/reference) whose
 * behaviour it restates.  Data layouts follow the reference so that the
 * timed CPU baseline touches memory the way the reference does:
 *   bit-vector words + rank_support_v block array (25 % overhead), BFS node table, m_path[],
 *   char2comp/C, bit-packed SA samples (width hi(n)+1, density 32).
 */
#include "vlg_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define UNDEF16 0xFFFFu

struct vlgo_index {
    uint64_t n;              /* text length incl. sentinel                          */
    uint32_t sigma;
    uint32_t dens;
    uint8_t  char2comp[256]; /* lib/csa_alphabet_strategy.cpp:25-55                 */
    uint8_t  comp2char[256];
    uint64_t C[257];
    /* wavelet tree (wt_pc.hpp:88-94) */
    uint64_t bv_bits;
    uint64_t* bv;            /* ceil(bv_bits/64)+1 words                            */
    uint64_t* rb;            /* rank_support_v basic blocks                         */
    uint64_t rb_words;
    vlgo_node* nodes;
    uint32_t n_nodes;
    uint16_t c_to_leaf[256];
    uint64_t path[256];
    /* SA samples, bit-packed like int_vector<0> (csa_sampling_strategy.hpp:85-98) */
    uint64_t* smp;
    uint64_t n_samples;
    uint32_t smp_width;
    uint8_t* bwt;            /* kept only when built here */
};

static inline uint64_t popc(uint64_t x) { return (uint64_t)__builtin_popcountll(x); } /* bits.hpp:245-248 */
static inline uint64_t lo_set(unsigned k) { return k >= 64 ? ~0ULL : ((1ULL << k) - 1); } /* lib/bits.cpp lo_set[] */
static inline unsigned hi_bit(uint64_t x) { return x ? 63u - (unsigned)__builtin_clzll(x) : 0u; } /* bits::hi */

/* ------------------------------------------------------------------------------------------
 * rank_support_v<1,1>   include/sdsl/rank_support_v.hpp:67-124
 * ---------------------------------------------------------------------------------------- */
uint64_t vlgo_rank_v_build(const uint64_t* words, uint64_t nbits, uint64_t* B)
{
    uint64_t nwords = (nbits + 63) >> 6;                 /* capacity>>6 */
    uint64_t nsb = (nwords >> 3) + 1;                    /* ((capacity>>9)+1) super-blocks */
    uint64_t cum = 0;
    for (uint64_t s = 0; s < nsb; ++s) {
        B[2 * s] = cum;                                  /* :90  absolute count before the super-block */
        uint64_t packed = 0, in = 0;
        for (unsigned j = 0; j < 8; ++j) {
            uint64_t w = 8 * s + j;
            if (j >= 1 && w <= nwords) packed |= in << (63 - 9 * j);   /* :93,97  9-bit fields */
            if (w < nwords) in += popc(words[w]);
        }
        B[2 * s + 1] = packed;
        cum += in;
    }
    return 2 * nsb;
}

uint64_t vlgo_rank_v(const uint64_t* words, const uint64_t* B, uint64_t idx)
{
    const uint64_t* p = B + ((idx >> 8) & 0xFFFFFFFFFFFFFFFEULL);       /* :117-118 */
    uint64_t r = p[0] + ((p[1] >> (63 - 9 * ((idx & 0x1FF) >> 6))) & 0x1FF);
    if (idx & 0x3F) r += popc(words[idx >> 6] & lo_set((unsigned)(idx & 0x3F)));  /* :119-121 */
    return r;
}

/* ------------------------------------------------------------------------------------------
 * rank_support_v5<1,1>   include/sdsl/rank_support_v5.hpp:65-134
 * ---------------------------------------------------------------------------------------- */
uint64_t vlgo_rank_v5_build(const uint64_t* words, uint64_t nbits, uint64_t* B)
{
    uint64_t nwords = (nbits + 63) >> 6;
    uint64_t nsb = (nwords >> 5) + 1;                    /* ((capacity>>11)+1) */
    uint64_t cum = 0;
    for (uint64_t s = 0; s < nsb; ++s) {
        B[2 * s] = cum;
        uint64_t packed = 0, in = 0;
        for (unsigned j = 0; j < 32; ++j) {
            uint64_t w = 32 * s + j;
            if (j >= 6 && (j % 6) == 0 && w <= nwords) packed |= in << (60 - 12 * (j / 6)); /* :93 */
            if (w < nwords) in += popc(words[w]);
        }
        B[2 * s + 1] = packed;
        cum += in;
    }
    return 2 * nsb;
}

uint64_t vlgo_rank_v5(const uint64_t* words, const uint64_t* B, uint64_t idx)
{
    const uint64_t* p = B + ((idx >> 10) & 0xFFFFFFFFFFFFFFFEULL);      /* :119-120 */
    uint64_t r = p[0] + ((p[1] >> (60 - 12 * ((idx & 0x7FF) / (64 * 6)))) & 0x7FFULL);
    if (idx & 0x3F) r += popc(words[idx >> 6] & lo_set((unsigned)(idx & 0x3F)));
    uint64_t w = idx >> 6;
    unsigned to_do = (unsigned)((w & 0x1F) % 6);                        /* :126 */
    while (to_do) { --w; r += popc(words[w]); --to_do; }                /* :128-132 */
    return r;
}

/* ------------------------------------------------------------------------------------------
 * Suffix array: prefix doubling with counting sorts (own code; SA is unique).
 * Defines what construct_sa (construct_sa.hpp:145-170) must produce.
 * ---------------------------------------------------------------------------------------- */
int vlgo_suffix_array(const uint8_t* text, uint64_t n64, uint64_t* sa_out)
{
    if (n64 == 0) return 0;
    if (n64 >= 0xFFFFFFF0ULL) return -1;
    uint32_t n = (uint32_t)n64;
    uint32_t* sa = (uint32_t*)malloc(4ull * n), *sa2 = (uint32_t*)malloc(4ull * n);
    uint32_t* rk = (uint32_t*)malloc(4ull * n), *rk2 = (uint32_t*)malloc(4ull * n);
    uint32_t nb = n > 256 ? n : 256;
    uint32_t* cnt = (uint32_t*)malloc(4ull * (nb + 1));
    if (!sa || !sa2 || !rk || !rk2 || !cnt) { free(sa); free(sa2); free(rk); free(rk2); free(cnt); return -1; }
    memset(cnt, 0, 4ull * 257);
    for (uint32_t i = 0; i < n; ++i) cnt[text[i] + 1]++;
    for (uint32_t c = 0; c < 256; ++c) cnt[c + 1] += cnt[c];
    for (uint32_t i = 0; i < n; ++i) sa[cnt[text[i]]++] = i;
    rk[sa[0]] = 0;
    for (uint32_t i = 1; i < n; ++i) rk[sa[i]] = rk[sa[i - 1]] + (text[sa[i]] != text[sa[i - 1]]);
    for (uint32_t h = 1; rk[sa[n - 1]] != n - 1; h <<= 1) {
        /* order by second key (rank[i+h], "none" smallest), derived from the current order */
        uint32_t p = 0;
        for (uint32_t i = n - h; i < n; ++i) sa2[p++] = i;
        for (uint32_t i = 0; i < n; ++i) if (sa[i] >= h) sa2[p++] = sa[i] - h;
        /* stable counting sort by first key */
        uint32_t maxr = rk[sa[n - 1]];
        memset(cnt, 0, 4ull * (maxr + 2));
        for (uint32_t i = 0; i < n; ++i) cnt[rk[i] + 1]++;
        for (uint32_t r = 0; r <= maxr; ++r) cnt[r + 1] += cnt[r];
        for (uint32_t i = 0; i < n; ++i) sa[cnt[rk[sa2[i]]]++] = sa2[i];
        rk2[sa[0]] = 0;
        for (uint32_t i = 1; i < n; ++i) {
            uint32_t a = sa[i - 1], b = sa[i];
            uint32_t a2 = a + h < n ? rk[a + h] + 1 : 0, b2 = b + h < n ? rk[b + h] + 1 : 0;
            rk2[b] = rk2[a] + (rk[a] != rk[b] || a2 != b2);
        }
        uint32_t* t = rk; rk = rk2; rk2 = t;
        if (h > n) break;
    }
    for (uint32_t i = 0; i < n; ++i) sa_out[i] = sa[i];
    free(sa); free(sa2); free(rk); free(rk2); free(cnt);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * byte_alphabet   lib/csa_alphabet_strategy.cpp:25-55
 * ---------------------------------------------------------------------------------------- */
static void build_alphabet(vlgo_index* x, const uint8_t* seq, uint64_t n)
{
    uint64_t cnt[256];
    memset(cnt, 0, sizeof cnt);
    for (uint64_t i = 0; i < n; ++i) cnt[seq[i]]++;
    memset(x->char2comp, 0, 256);
    memset(x->comp2char, 0, 256);
    x->sigma = 0;
    for (int c = 0; c < 256; ++c) if (cnt[c]) {
        x->char2comp[c] = (uint8_t)x->sigma;
        x->comp2char[x->sigma] = (uint8_t)c;
        x->C[x->sigma + 1] = cnt[c];
        x->sigma++;
    }
    x->C[0] = 0;
    for (uint32_t i = 1; i <= x->sigma; ++i) x->C[i] += x->C[i - 1];
}

/* ------------------------------------------------------------------------------------------
 * Huffman shape  include/sdsl/wt_huff.hpp:91-117  (min-heap on (freq, node id))
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t freq, sym, parent, child[2]; } pc_node;   /* wt_helper.hpp:56-70 */
#define UNDEF64 0xFFFFFFFFFFFFFFFFULL
typedef struct { uint64_t f, id; } hp_item;
static int hp_less(hp_item a, hp_item b) { return a.f < b.f || (a.f == b.f && a.id < b.id); }
static void hp_push(hp_item* h, uint32_t* n, hp_item v)
{
    uint32_t i = (*n)++;
    h[i] = v;
    while (i && hp_less(h[i], h[(i - 1) / 2])) { hp_item t = h[i]; h[i] = h[(i - 1) / 2]; h[(i - 1) / 2] = t; i = (i - 1) / 2; }
}
static hp_item hp_pop(hp_item* h, uint32_t* n)
{
    hp_item top = h[0];
    h[0] = h[--(*n)];
    uint32_t i = 0;
    for (;;) {
        uint32_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < *n && hp_less(h[l], h[m])) m = l;
        if (r < *n && hp_less(h[r], h[m])) m = r;
        if (m == i) break;
        hp_item t = h[i]; h[i] = h[m]; h[m] = t; i = m;
    }
    return top;
}

/* Tree layout: _byte_tree ctor, include/sdsl/wt_helper.hpp:170-241 (BFS order) */
static uint64_t build_tree(vlgo_index* x, const uint64_t* freq /*[256]*/)
{
    pc_node tmp[512];
    uint32_t nt = 0;
    hp_item heap[512];
    uint32_t hn = 0;
    for (int c = 0; c < 256; ++c) if (freq[c]) {
        hp_item it = { freq[c], nt };
        hp_push(heap, &hn, it);
        pc_node nd = { freq[c], (uint64_t)c, UNDEF64, { UNDEF64, UNDEF64 } };
        tmp[nt++] = nd;
    }
    while (hn > 1) {                                             /* wt_huff.hpp:105-116 */
        hp_item v1 = hp_pop(heap, &hn), v2 = hp_pop(heap, &hn);
        tmp[v1.id].parent = nt; tmp[v2.id].parent = nt;
        hp_item it = { v1.f + v2.f, nt };
        hp_push(heap, &hn, it);
        pc_node nd = { v1.f + v2.f, 0, UNDEF64, { v1.id, v2.id } };
        tmp[nt++] = nd;
    }
    x->n_nodes = nt;
    x->nodes = (vlgo_node*)calloc(nt ? nt : 1, sizeof(vlgo_node));
    if (!nt) return 0;
    /* BFS renumbering; freq is carried in bv_pos until the node is visited (wt_helper.hpp:189-205) */
    uint64_t fr[512], sy[512];
    uint64_t tch[512][2];
    uint32_t q[512], qh = 0, qt = 0;
    fr[0] = tmp[nt - 1].freq; sy[0] = tmp[nt - 1].sym; tch[0][0] = tmp[nt - 1].child[0]; tch[0][1] = tmp[nt - 1].child[1];
    x->nodes[0].parent = UNDEF16;
    q[qt++] = 0;
    uint32_t node_cnt = 1;
    uint64_t bv_size = 0;
    while (qh < qt) {
        uint32_t idx = q[qh++];
        vlgo_node* nd = &x->nodes[idx];
        nd->bv_pos = bv_size;
        int inner = tch[idx][0] != UNDEF64;
        if (inner) bv_size += fr[idx];
        nd->bv_pos_rank = sy[idx];
        if (inner) {
            for (int k = 0; k < 2; ++k) {
                uint64_t t = tch[idx][k];
                fr[node_cnt] = tmp[t].freq; sy[node_cnt] = tmp[t].sym;
                tch[node_cnt][0] = tmp[t].child[0]; tch[node_cnt][1] = tmp[t].child[1];
                x->nodes[node_cnt].parent = (uint16_t)idx;
                q[qt++] = node_cnt;
                nd->child[k] = (uint16_t)node_cnt++;
            }
        } else {
            nd->child[0] = nd->child[1] = UNDEF16;
        }
    }
    /* m_c_to_leaf, m_path (wt_helper.hpp:207-240): first edge in bit 0, length in bits 56..63 */
    for (int c = 0; c < 256; ++c) x->c_to_leaf[c] = UNDEF16;
    for (uint32_t v = 0; v < nt; ++v)
        if (x->nodes[v].child[0] == UNDEF16) x->c_to_leaf[(uint8_t)x->nodes[v].bv_pos_rank] = (uint16_t)v;
    uint64_t prev_c = 0;
    for (int c = 0; c < 256; ++c) {
        if (x->c_to_leaf[c] != UNDEF16) {
            uint32_t v = x->c_to_leaf[c];
            uint64_t pw = 0, pl = 0;
            while (v != 0) {
                pw <<= 1;
                if (x->nodes[x->nodes[v].parent].child[1] == v) pw |= 1ULL;
                ++pl;
                v = x->nodes[v].parent;
            }
            x->path[c] = pw | (pl << 56);
            prev_c = (uint64_t)c;
        } else {
            x->path[c] = prev_c;     /* length 0 */
        }
    }
    return bv_size;
}

static void finish_rank(vlgo_index* x)
{
    uint64_t nwords = (x->bv_bits + 63) >> 6;
    x->rb_words = 2 * ((nwords >> 3) + 1);
    x->rb = (uint64_t*)calloc(x->rb_words, 8);
    vlgo_rank_v_build(x->bv, x->bv_bits, x->rb);
}

/* wt_pc ctor  include/sdsl/wt_pc.hpp:197-248 (bits written symbol by symbol instead of by
 * runs of <= 64 equal symbols -- same bit-vector). */
static void build_wt(vlgo_index* x, const uint8_t* seq, uint64_t n)
{
    uint64_t freq[256];
    memset(freq, 0, sizeof freq);
    for (uint64_t i = 0; i < n; ++i) freq[seq[i]]++;
    x->bv_bits = build_tree(x, freq);
    x->bv = (uint64_t*)calloc(((x->bv_bits + 63) >> 6) + 1, 8);
    uint64_t cur[512];
    for (uint32_t v = 0; v < x->n_nodes; ++v) cur[v] = x->nodes[v].bv_pos;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t p = x->path[seq[i]];
        uint32_t len = (uint32_t)(p >> 56), v = 0;
        for (uint32_t l = 0; l < len; ++l, p >>= 1) {            /* insert_char :110-122 */
            if (p & 1) x->bv[cur[v] >> 6] |= 1ULL << (cur[v] & 63);
            cur[v]++;
            v = x->nodes[v].child[p & 1];
        }
    }
    finish_rank(x);
    /* init_node_ranks  wt_helper.hpp:243-250 */
    for (uint32_t v = 0; v < x->n_nodes; ++v)
        if (x->nodes[v].child[0] != UNDEF16)
            x->nodes[v].bv_pos_rank = vlgo_rank_v(x->bv, x->rb, x->nodes[v].bv_pos);
}

static void set_sample(vlgo_index* x, uint64_t j, uint64_t v)
{
    uint64_t bit = j * x->smp_width;
    uint64_t w = bit >> 6, o = bit & 63;
    x->smp[w] |= v << o;
    if (o + x->smp_width > 64) x->smp[w + 1] |= v >> (64 - o);
}

/* int_vector<0>::operator[] -> bits::read_int   bits.hpp:494-505 */
static inline uint64_t get_sample(const vlgo_index* x, uint64_t j)
{
    uint64_t bit = j * x->smp_width;
    uint64_t w = bit >> 6, o = bit & 63;
    uint64_t v = x->smp[w] >> o;
    if (o + x->smp_width > 64) v |= x->smp[w + 1] << (64 - o);
    return v & lo_set(x->smp_width);
}

static void alloc_samples(vlgo_index* x)
{
    x->smp_width = hi_bit(x->n) + 1;                             /* csa_sampling_strategy.hpp:89 */
    x->n_samples = (x->n + x->dens - 1) / x->dens;               /* :90 */
    x->smp = (uint64_t*)calloc(((x->n_samples * x->smp_width + 63) >> 6) + 2, 8);
}

vlgo_index* vlgo_build_from_bwt(const uint8_t* bwt, const uint64_t* sa, uint64_t n, uint32_t dens)
{
    vlgo_index* x = (vlgo_index*)calloc(1, sizeof *x);
    x->n = n;
    x->dens = dens ? dens : 32;
    build_alphabet(x, bwt, n);
    build_wt(x, bwt, n);
    alloc_samples(x);
    if (sa) for (uint64_t i = 0, j = 0; i < n; i += x->dens) set_sample(x, j++, sa[i]);   /* :92-98 */
    x->bwt = (uint8_t*)malloc(n ? n : 1);
    memcpy(x->bwt, bwt, n);
    return x;
}

vlgo_index* vlgo_build(const uint8_t* text, uint64_t n_text, uint32_t dens)
{
    uint64_t n = n_text + 1;
    uint8_t* t = (uint8_t*)malloc(n);
    memcpy(t, text, n_text);
    t[n_text] = 0;                                               /* construct.hpp:47-52 */
    uint64_t* sa = (uint64_t*)malloc(8 * n);
    if (vlgo_suffix_array(t, n, sa) != 0) { free(t); free(sa); return NULL; }
    uint8_t* bwt = (uint8_t*)malloc(n);
    for (uint64_t i = 0; i < n; ++i) bwt[i] = t[sa[i] ? sa[i] - 1 : n - 1];   /* construct_bwt.hpp:71-75 */
    vlgo_index* x = vlgo_build_from_bwt(bwt, sa, n, dens);
    free(t); free(sa); free(bwt);
    return x;
}

vlgo_index* vlgo_from_parts(uint64_t n, uint32_t sigma, const uint8_t* char2comp, const uint64_t* C,
                            const uint64_t* bv_words, uint64_t bv_bits,
                            const vlgo_node* nodes, uint32_t n_nodes,
                            const uint64_t* samples, uint64_t n_samples, uint32_t dens)
{
    vlgo_index* x = (vlgo_index*)calloc(1, sizeof *x);
    x->n = n; x->sigma = sigma; x->dens = dens ? dens : 32;
    memcpy(x->char2comp, char2comp, 256);
    memcpy(x->C, C, 8 * (sigma + 1));
    for (int c = 0; c < 256; ++c) if (c == 0 || char2comp[c]) x->comp2char[char2comp[c]] = (uint8_t)c;
    x->bv_bits = bv_bits;
    uint64_t nw = (bv_bits + 63) >> 6;
    x->bv = (uint64_t*)calloc(nw + 1, 8);
    memcpy(x->bv, bv_words, 8 * nw);
    x->n_nodes = n_nodes;
    x->nodes = (vlgo_node*)calloc(n_nodes ? n_nodes : 1, sizeof(vlgo_node));
    memcpy(x->nodes, nodes, sizeof(vlgo_node) * n_nodes);
    for (int c = 0; c < 256; ++c) x->c_to_leaf[c] = UNDEF16;
    for (uint32_t v = 0; v < n_nodes; ++v)
        if (nodes[v].child[0] == UNDEF16) x->c_to_leaf[(uint8_t)nodes[v].bv_pos_rank] = (uint16_t)v;
    uint64_t prev_c = 0;
    for (int c = 0; c < 256; ++c) {
        if (x->c_to_leaf[c] != UNDEF16) {
            uint32_t v = x->c_to_leaf[c];
            uint64_t pw = 0, pl = 0;
            while (v != 0) {
                pw <<= 1;
                if (x->nodes[x->nodes[v].parent].child[1] == v) pw |= 1ULL;
                ++pl; v = x->nodes[v].parent;
            }
            x->path[c] = pw | (pl << 56);
            prev_c = (uint64_t)c;
        } else x->path[c] = prev_c;
    }
    finish_rank(x);
    alloc_samples(x);
    if (n_samples != x->n_samples) { vlgo_free(x); return NULL; }
    for (uint64_t j = 0; j < n_samples; ++j) set_sample(x, j, samples[j]);
    return x;
}

void vlgo_free(vlgo_index* x)
{
    if (!x) return;
    free(x->bv); free(x->rb); free(x->nodes); free(x->smp); free(x->bwt); free(x);
}

uint64_t vlgo_size(const vlgo_index* x) { return x->n; }
uint32_t vlgo_sigma(const vlgo_index* x) { return x->sigma; }
const uint8_t* vlgo_char2comp(const vlgo_index* x) { return x->char2comp; }
const uint64_t* vlgo_C(const vlgo_index* x) { return x->C; }
uint64_t vlgo_bv_bits(const vlgo_index* x) { return x->bv_bits; }
const uint64_t* vlgo_bv_words(const vlgo_index* x) { return x->bv; }
uint32_t vlgo_n_nodes(const vlgo_index* x) { return x->n_nodes; }
const vlgo_node* vlgo_nodes(const vlgo_index* x) { return x->nodes; }
const uint64_t* vlgo_paths(const vlgo_index* x) { return x->path; }
uint64_t vlgo_n_samples(const vlgo_index* x) { return x->n_samples; }
uint64_t vlgo_sample(const vlgo_index* x, uint64_t j) { return get_sample(x, j); }
const uint64_t* vlgo_rank_blocks(const vlgo_index* x, uint64_t* nw) { if (nw) *nw = x->rb_words; return x->rb; }
const uint8_t* vlgo_bwt(const vlgo_index* x) { return x->bwt; }

uint64_t vlgo_bv_rank1(const vlgo_index* x, uint64_t idx) { return vlgo_rank_v(x->bv, x->rb, idx); }

/* wt_pc::rank   include/sdsl/wt_pc.hpp:350-373 */
static inline uint64_t wt_rank_cnt(const vlgo_index* x, uint64_t i, uint8_t c, uint64_t* ranks)
{
    if (x->c_to_leaf[c] == UNDEF16) return 0;                    /* :352-354 */
    if (x->sigma == 1) return i;                                 /* :355-357 */
    uint64_t p = x->path[c];
    uint32_t len = (uint32_t)(p >> 56);
    uint64_t res = i;
    uint32_t v = 0;
    for (uint32_t l = 0; l < len && res; ++l, p >>= 1) {
        const vlgo_node* nd = &x->nodes[v];
        uint64_t r1 = vlgo_rank_v(x->bv, x->rb, nd->bv_pos + res) - nd->bv_pos_rank;
        if (ranks) ++*ranks;
        res = (p & 1) ? r1 : res - r1;
        v = nd->child[p & 1];
    }
    return res;
}
uint64_t vlgo_wt_rank(const vlgo_index* x, uint64_t i, uint8_t c) { return wt_rank_cnt(x, i, c, NULL); }

/* wt_pc::inverse_select   include/sdsl/wt_pc.hpp:385-402 */
static inline uint64_t inv_select(const vlgo_index* x, uint64_t i, uint8_t* c, uint64_t* levels)
{
    uint32_t v = 0;
    while (x->nodes[v].child[0] != UNDEF16) {
        const vlgo_node* nd = &x->nodes[v];
        uint64_t pos = nd->bv_pos + i;
        uint64_t r1 = vlgo_rank_v(x->bv, x->rb, pos) - nd->bv_pos_rank;
        if (levels) ++*levels;
        if ((x->bv[pos >> 6] >> (pos & 63)) & 1) { i = r1; v = nd->child[1]; }
        else { i -= r1; v = nd->child[0]; }
    }
    *c = (uint8_t)x->nodes[v].bv_pos_rank;
    return i;
}
uint64_t vlgo_inverse_select(const vlgo_index* x, uint64_t i, uint8_t* c) { return inv_select(x, i, c, NULL); }

/* LF   include/sdsl/suffix_array_helper.hpp:336-349 */
static inline uint64_t lf_step(const vlgo_index* x, uint64_t i, uint64_t* levels)
{
    uint8_t c;
    uint64_t j = inv_select(x, i, &c, levels);
    return x->C[x->char2comp[c]] + j;
}
uint64_t vlgo_lf(const vlgo_index* x, uint64_t i) { return lf_step(x, i, NULL); }

/* csa_wt::operator[]   include/sdsl/csa_wt.hpp:335-348; sampling csa_sampling_strategy.hpp:102-111 */
uint64_t vlgo_sa(const vlgo_index* x, uint64_t i, uint64_t* lf_steps, uint64_t* levels)
{
    uint64_t off = 0;
    while (i % x->dens) { i = lf_step(x, i, levels); ++off; }
    if (lf_steps) *lf_steps += off;
    uint64_t r = get_sample(x, i / x->dens);
    return (r + off < x->n) ? r + off : r + off - x->n;
}

/* backward_search   include/sdsl/suffix_array_algorithm.hpp:250-278 (char), :305-326 (pattern) */
static uint64_t bs_cnt(const vlgo_index* x, const uint8_t* pat, uint64_t m, uint64_t* lo, uint64_t* ro, uint64_t* ranks)
{
    uint64_t l = 0, r = x->n - 1;
    const uint8_t* it = pat + m;
    while (pat < it && r + 1 - l > 0) {
        --it;
        uint8_t c = *it;
        uint64_t cc = x->char2comp[c];
        if (cc == 0 && c > 0) { l = 1; r = 0; }                  /* :263-265 */
        else {
            uint64_t c_begin = x->C[cc];
            if (l == 0 && r + 1 == x->n) { l = c_begin; r = x->C[cc + 1] - 1; }       /* :268-270 */
            else {
                uint64_t nl = c_begin + wt_rank_cnt(x, l, c, ranks);                 /* :272 */
                uint64_t nr = c_begin + wt_rank_cnt(x, r + 1, c, ranks) - 1;         /* :273 */
                l = nl; r = nr;
            }
        }
    }
    *lo = l; *ro = r;
    return r + 1 - l;
}
uint64_t vlgo_backward_search(const vlgo_index* x, const uint8_t* pat, uint64_t m, uint64_t* l, uint64_t* r)
{
    return bs_cnt(x, pat, m, l, r, NULL);
}

/* locate   include/sdsl/suffix_array_algorithm.hpp:604-619 */
uint64_t vlgo_locate(const vlgo_index* x, const uint8_t* pat, uint64_t m, uint64_t* out, uint64_t cap)
{
    uint64_t l, r;
    uint64_t occ = bs_cnt(x, pat, m, &l, &r, NULL);
    for (uint64_t i = 0; i < occ && i < cap; ++i) out[i] = vlgo_sa(x, l + i, NULL, NULL);
    return occ;
}

/* ------------------------------------------------------------------------------------------
 * Query parsing.
 *   dialect 0: gapped_pattern_query(const std::string&)  include/sdsl/vlg_index.hpp:54-105
 *   dialect 1: gapped_pattern(const std::string&, true)  benchmark/gapped-matching/include/utils.hpp:25-70
 *              + the gap/length mapping of index_sasearch::search  index_sasearch.hpp:68-69,113
 * ---------------------------------------------------------------------------------------- */
static int64_t find2(const uint8_t* s, uint64_t len, uint64_t from, char a, char b)
{
    if (b) { for (uint64_t i = from; i + 1 < len; ++i) if (s[i] == (uint8_t)a && s[i + 1] == (uint8_t)b) return (int64_t)i; }
    else   { for (uint64_t i = from; i < len; ++i) if (s[i] == (uint8_t)a) return (int64_t)i; }
    return -1;
}
/* std::stoull on [s,e): optional blanks, optional sign, >=1 digit; trailing junk ignored */
static int parse_u64(const uint8_t* s, const uint8_t* e, uint64_t* out)
{
    while (s < e && (*s == ' ' || (*s >= 9 && *s <= 13))) ++s;
    int neg = 0;
    if (s < e && (*s == '+' || *s == '-')) { neg = *s == '-'; ++s; }
    if (s >= e || *s < '0' || *s > '9') return -1;
    uint64_t v = 0;
    while (s < e && *s >= '0' && *s <= '9') {
        if (v > (0xFFFFFFFFFFFFFFFFULL - (uint64_t)(*s - '0')) / 10) return -1;   /* out_of_range */
        v = v * 10 + (uint64_t)(*s - '0'); ++s;
    }
    *out = neg ? (uint64_t)(-(int64_t)v) : v;
    return 0;
}

int vlgo_parse(const uint8_t* re, uint64_t len, int dialect, vlgo_query* q)
{
    memset(q, 0, sizeof *q);
    uint64_t raw_lo[VLGO_MAX_SUB], raw_hi[VLGO_MAX_SUB];
    uint64_t start = 0;               /* last_gap_end + 1 */
    for (;;) {
        int64_t gp = find2(re, len, start, '.', '{');
        if (gp < 0) break;
        if (q->k + 1 >= VLGO_MAX_SUB) return VLGO_E_PARSE_NUM;
        int64_t ge = find2(re, len, (uint64_t)gp, '}', 0);
        if (ge < 0) return VLGO_E_PARSE_NUM;                      /* reference: stoull("") throws */
        int64_t comma = -1;
        for (int64_t i = gp; i <= ge; ++i) if (re[i] == ',') { comma = i; break; }
        if (comma < 0) return VLGO_E_PARSE_NUM;
        uint64_t a, b;
        if (parse_u64(re + gp + 2, re + comma, &a)) return VLGO_E_PARSE_NUM;
        if (parse_u64(re + comma + 1, re + ge, &b)) return VLGO_E_PARSE_NUM;
        if (a > b) return VLGO_E_PARSE_MINMAX;
        q->sub[q->k] = re + start;
        q->sub_len[q->k] = (uint64_t)gp - start;
        raw_lo[q->k + 1] = a; raw_hi[q->k + 1] = b;
        q->k++;
        if (dialect == 0) {
            if ((uint64_t)ge + 1 == len || re[ge + 1] != '?') return VLGO_E_PARSE_QMARK;
            start = (uint64_t)ge + 2;                             /* last_gap_end = gap_end ('?' index) */
        } else {
            start = (uint64_t)ge + 1;
        }
    }
    q->sub[q->k] = re + start;
    q->sub_len[q->k] = len - start;
    q->k++;
    for (uint32_t i = 0; i < q->k; ++i) if (q->sub_len[i] == 0) return VLGO_E_EMPTY_SUBPATTERN;
    if (dialect == 0) {
        for (uint32_t i = 1; i < q->k; ++i) {                     /* vlg_index.hpp:95 */
            q->lo[i] = raw_lo[i] + q->sub_len[i - 1];
            q->hi[i] = raw_hi[i] + q->sub_len[i - 1];
        }
        q->end_len = q->sub_len[q->k - 1];                        /* vlg_index.hpp:262,306 */
    } else {
        for (uint32_t i = 1; i < q->k; ++i) {                     /* index_sasearch.hpp:68-69 */
            q->lo[i] = raw_lo[1] + q->sub_len[0];
            q->hi[i] = raw_hi[1] + q->sub_len[0];
        }
        q->end_len = q->sub_len[0];                               /* index_sasearch.hpp:113 */
    }
    return VLGO_OK;
}

/* ------------------------------------------------------------------------------------------
 * Merge join   benchmark/gapped-matching/include/index_sasearch.hpp:85-116, with per-gap bounds
 * and the library's non-overlap length (vlg_index.hpp:227-291 semantics).
 * ---------------------------------------------------------------------------------------- */
uint64_t vlgo_join(uint32_t k, const uint64_t* const* L, const uint64_t* len,
                   const uint64_t* lo, const uint64_t* hi, uint64_t end_len,
                   uint64_t* out, uint64_t cap)
{
    uint64_t it[VLGO_MAX_SUB];
    memset(it, 0, sizeof it);
    for (uint32_t i = 0; i < k; ++i) if (len[i] == 0) return 0;   /* vlg_index.hpp:315-316 */
    uint64_t matches = 0;
    while (it[0] != len[0]) {
        uint64_t prev = L[0][it[0]];
        int brk = 0, cont = 0;
        for (uint32_t i = 1; i < k; ++i) {
            while (it[i] != len[i] && prev + lo[i] > L[i][it[i]]) ++it[i];          /* :94 */
            if (it[i] == len[i]) { brk = 1; break; }                                 /* :95 */
            uint64_t np = L[i][it[i]];
            if (prev + hi[i] < np) { ++it[i - 1]; cont = 1; break; }                 /* :99 */
            prev = np;
        }
        if (brk) break;
        if (cont) continue;
        if (out && matches < cap) for (uint32_t i = 0; i < k; ++i) out[matches * k + i] = L[i][it[i]];
        ++matches;
        uint64_t posx = L[k - 1][it[k - 1]] + end_len;                               /* :113 */
        while (it[0] != len[0] && L[0][it[0]] < posx) ++it[0];                       /* :114-115 */
    }
    return matches;
}

static int cmp_u64(const void* a, const void* b)
{
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : x > y;
}

/* index_sasearch::search shape (index_sasearch.hpp:58-118) on the FM path of SURVEY 3.3:
 * per sub-pattern backward_search; if any interval is empty the query has no match and nothing
 * is located (vlg_index.hpp:315-316 returns at the first empty range); otherwise locate every
 * list, std::sort it, then the join. */
uint64_t vlgo_search(const vlgo_index* x, const vlgo_query* q, uint64_t* out, uint64_t cap, uint64_t* stats)
{
    uint64_t* lists[VLGO_MAX_SUB];
    uint64_t lens[VLGO_MAX_SUB], ls[VLGO_MAX_SUB];
    uint64_t st_occ = 0, st_lf = 0, st_lv = 0, st_rk = 0;
    uint32_t k = q->k;
    int empty = (k == 0);
    for (uint32_t i = 0; i < VLGO_MAX_SUB; ++i) { lists[i] = NULL; lens[i] = 0; ls[i] = 0; }
    for (uint32_t i = 0; i < k; ++i) {
        uint64_t r, occ = 0;
        if (q->sub_len[i] < x->n) occ = bs_cnt(x, q->sub[i], q->sub_len[i], &ls[i], &r, &st_rk);
        lens[i] = occ;
        if (occ == 0) empty = 1;
    }
    uint64_t res = 0;
    if (!empty) {
        for (uint32_t i = 0; i < k; ++i) {
            uint64_t occ = lens[i];
            lists[i] = (uint64_t*)malloc(8 * occ);
            for (uint64_t j = 0; j < occ; ++j) lists[i][j] = vlgo_sa(x, ls[i] + j, &st_lf, &st_lv);
            st_occ += occ;
            qsort(lists[i], occ, 8, cmp_u64);                     /* index_sasearch.hpp:80 */
        }
        res = vlgo_join(k, (const uint64_t* const*)lists, lens, q->lo, q->hi, q->end_len, out, cap);
    }
    for (uint32_t i = 0; i < k; ++i) free(lists[i]);
    if (stats) { stats[0] += st_occ; stats[1] += st_lf; stats[2] += st_lv; stats[3] += st_rk; }
    return res;
}

/* ------------------------------------------------------------------------------------------
 * SASEARCH: the benchmark's plain-suffix-array index (benchmark/gapped-matching/include/index_sasearch.hpp:58-118), kept
 * beside the FM path as the fastest CPU algorithm of the reference (CPU baseline (3)).
 *   forward_search over text + SA   include/sdsl/suffix_array_algorithm.hpp:48-112  (two binary searches, the pattern
 *                                    compared with the text suffix, a suffix that ends inside the pattern is smaller)
 *   copy SA[sp..ep], std::sort, merge join   index_sasearch.hpp:76-116
 * text: n bytes incl. the sentinel; sa: n entries (32-bit, n <= 2^32).
 * ---------------------------------------------------------------------------------------- */
static int sa_compare(const uint8_t* text, uint64_t n, const uint32_t* sa, uint64_t i, const uint8_t* pat, uint64_t m)
{
    uint64_t t = sa[i];
    for (uint64_t j = 0; j < m; ++j, ++t) {
        if (t == n) return 1;
        if (text[t] < pat[j]) return 1;
        if (text[t] > pat[j]) return -1;
    }
    return 0;
}

uint64_t vlgo_sa_forward_search(const uint8_t* text, uint64_t n, const uint32_t* sa, const uint8_t* pat, uint64_t m,
                                uint64_t* l_res_out, uint64_t* r_res_out)
{
    uint64_t l_res = 0, r_res = (uint64_t)0 - 1, l_upper = n, r_upper = n;     /* l = 0, r = n - 1 */
    if (m >= n) { *l_res_out = 0; *r_res_out = (uint64_t)0 - 1; return 0; }
    while (l_res < l_upper) {
        uint64_t sample = l_res + (l_upper - l_res) / 2;
        if (sa_compare(text, n, sa, sample, pat, m) == 1) l_res = sample + 1; else l_upper = sample;
    }
    while (r_res + 1 < r_upper) {
        uint64_t sample = r_res + (r_upper - r_res) / 2;
        if (sa_compare(text, n, sa, sample, pat, m) == -1) r_upper = sample; else r_res = sample;
    }
    *l_res_out = l_res; *r_res_out = r_res;
    return r_res - l_res + 1;
}

uint64_t vlgo_sasearch(const uint8_t* text, uint64_t n, const uint32_t* sa, const vlgo_query* q, uint64_t* out, uint64_t cap,
                       uint64_t* stats)
{
    uint64_t* lists[VLGO_MAX_SUB];
    uint64_t lens[VLGO_MAX_SUB];
    uint32_t k = q->k;
    uint64_t res = 0, st_occ = 0;
    for (uint32_t i = 0; i < VLGO_MAX_SUB; ++i) { lists[i] = NULL; lens[i] = 0; }
    for (uint32_t i = 0; i < k; ++i) {                       /* every range is materialised and sorted (index_sasearch.hpp:76-82) */
        uint64_t sp, ep;
        uint64_t occ = vlgo_sa_forward_search(text, n, sa, q->sub[i], q->sub_len[i], &sp, &ep);
        lens[i] = occ;
        lists[i] = (uint64_t*)malloc(8 * (occ ? occ : 1));
        for (uint64_t j = 0; j < occ; ++j) lists[i][j] = sa[sp + j];
        qsort(lists[i], occ, 8, cmp_u64);
        st_occ += occ;
    }
    if (k) res = vlgo_join(k, (const uint64_t* const*)lists, lens, q->lo, q->hi, q->end_len, out, cap);
    for (uint32_t i = 0; i < k; ++i) free(lists[i]);
    if (stats) stats[0] += st_occ;
    return res;
}

/* ------------------------------------------------------------------------------------------
 * Many queries on T threads (bench.py's cpu_baseline, disclosure lines): the reference's driver is one thread
 * (gm_search.cpp:91-121); this is the same per-query code on a pool that draws queries from a shared counter, for the
 * "what would all host cores do" figure.  The index / text are only read.  Stops drawing when budget_s of wall time have passed.
 * stats (4 x u64, summed over the threads): located occurrences, LF steps, WT levels, backward-search ranks; *matches = tuples found.
 * Returns the number of queries finished. */
#include <pthread.h>
#include <time.h>

typedef struct {
    const vlgo_index* x;                                    /* FM path ... */
    const uint8_t* text; uint64_t n; const uint32_t* sa;    /* ... or SASEARCH when x is NULL */
    const uint8_t* blob; const uint64_t* off; uint64_t nq; int dialect;
    double budget_s; struct timespec t0;
    uint64_t next;                                          /* shared cursor (atomic) */
    pthread_mutex_t mu;
    uint64_t stats[4], matches, done;
} many_job;

static double since(const struct timespec* t0)
{
    struct timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0->tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0->tv_nsec);
}

static void* many_worker(void* arg)
{
    many_job* J = (many_job*)arg;
    uint64_t st[4] = {0, 0, 0, 0}, matches = 0, done = 0;
    for (;;) {
        const uint64_t i = __atomic_fetch_add(&J->next, 1, __ATOMIC_RELAXED);
        if (i >= J->nq || since(&J->t0) > J->budget_s) break;
        vlgo_query q;
        if (vlgo_parse(J->blob + J->off[i], J->off[i + 1] - J->off[i], J->dialect, &q) == VLGO_OK)
            matches += J->x ? vlgo_search(J->x, &q, NULL, 0, st) : vlgo_sasearch(J->text, J->n, J->sa, &q, NULL, 0, st);
        ++done;
    }
    pthread_mutex_lock(&J->mu);
    for (int k = 0; k < 4; ++k) J->stats[k] += st[k];
    J->matches += matches;
    J->done += done;
    pthread_mutex_unlock(&J->mu);
    return NULL;
}

static uint64_t run_many(many_job* J, uint32_t n_threads, uint64_t* stats, uint64_t* matches)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    pthread_t th[1024];
    uint32_t started = 0;
    pthread_mutex_init(&J->mu, NULL);
    clock_gettime(CLOCK_MONOTONIC, &J->t0);
    for (; started + 1 < n_threads; ++started)
        if (pthread_create(&th[started], NULL, many_worker, J) != 0) break;
    many_worker(J);                                         /* the caller's thread is one of the pool */
    for (uint32_t t = 0; t < started; ++t) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&J->mu);
    if (stats) for (int k = 0; k < 4; ++k) stats[k] += J->stats[k];
    if (matches) *matches += J->matches;
    return J->done;
}

uint64_t vlgo_search_many(const vlgo_index* x, const uint8_t* blob, const uint64_t* off, uint64_t nq, int dialect, uint32_t n_threads,
                          double budget_s, uint64_t* stats, uint64_t* matches)
{
    many_job J;
    memset(&J, 0, sizeof J);
    J.x = x; J.blob = blob; J.off = off; J.nq = nq; J.dialect = dialect; J.budget_s = budget_s;
    return run_many(&J, n_threads, stats, matches);
}

uint64_t vlgo_sasearch_many(const uint8_t* text, uint64_t n, const uint32_t* sa, const uint8_t* blob, const uint64_t* off, uint64_t nq, int dialect,
                            uint32_t n_threads, double budget_s, uint64_t* stats, uint64_t* matches)
{
    many_job J;
    memset(&J, 0, sizeof J);
    J.text = text; J.n = n; J.sa = sa; J.blob = blob; J.off = off; J.nq = nq; J.dialect = dialect; J.budget_s = budget_s;
    return run_many(&J, n_threads, stats, matches);
}

/* ------------------------------------------------------------------------------------------
 * rrr_vector<63> + rank_support_rrr<1,63>  (SURVEY a-11; BASELINE config 5)
 *   ctor    include/sdsl/rrr_vector.hpp:145-237   (block 63 bits, rank/pointer sample every 32 blocks,
 *                                                  per-superblock inversion of the stored classes)
 *   rank    include/sdsl/rrr_vector.hpp:444-480
 *   coding  include/sdsl/rrr_helper.hpp:173-258 (binomial table), :282-284 (space_for_bt),
 *           :304-320 (bin_to_nr), :411-460 (decode_popcount incl. the binary-search variant for k < 10)
 * Arrays are kept as plain u64/u8 instead of bit-packed int_vectors (storage detail); m_btnr is the
 * reference's concatenated variable-length offset stream.
 * ---------------------------------------------------------------------------------------- */
#define RRR_BS 63
#define RRR_K 32
struct vlgo_rrr {
    uint64_t size;
    uint64_t n_bt;        /* (size+63)/63 blocks incl. the dummy block when size % 63 == 0 */
    uint8_t* bt;          /* stored class (complemented inside inverted superblocks) */
    uint64_t* btnr;       /* offset stream */
    uint64_t btnr_bits;
    uint64_t* btnrp;      /* [ceil(n_bt/32)] */
    uint64_t* rank;       /* [ceil(n_bt/32) + (size % 2016 > 0)] (+1 safety) */
    uint64_t n_rank;
    uint8_t* invert;
};
static uint64_t g_binom[65][65];
static uint16_t g_space[64];
static int g_binom_ready = 0;
static void binom_init(void)
{
    if (g_binom_ready) return;
    for (int k = 0; k <= 64; ++k) g_binom[0][k] = 0;
    for (int nn = 0; nn <= 64; ++nn) g_binom[nn][0] = 1;
    for (int nn = 1; nn <= 64; ++nn)
        for (int k = 1; k <= 64; ++k) g_binom[nn][k] = (k == nn) ? 1 : g_binom[nn - 1][k - 1] + g_binom[nn - 1][k];
    for (int k = 0; k <= 63; ++k) g_space[k] = (g_binom[63][k] == 1) ? 0 : (uint16_t)(hi_bit(g_binom[63][k]) + 1);   /* :241-243 */
    g_binom_ready = 1;
}
static inline uint64_t get_bits(const uint64_t* w, uint64_t pos, unsigned len)      /* bits::read_int */
{
    if (!len) return 0;
    uint64_t i = pos >> 6, o = pos & 63;
    uint64_t v = w[i] >> o;
    if (o + len > 64) v |= w[i + 1] << (64 - o);
    return len == 64 ? v : (v & ((1ULL << len) - 1));
}
static inline void put_bits(uint64_t* w, uint64_t pos, uint64_t v, unsigned len)
{
    if (!len) return;
    uint64_t i = pos >> 6, o = pos & 63;
    w[i] |= v << o;
    if (o + len > 64) w[i + 1] |= v >> (64 - o);
}
static uint64_t bin_to_nr(uint64_t bin)                                              /* rrr_helper.hpp:304-320 */
{
    if (bin == 0 || bin == ((1ULL << 63) - 1)) return 0;
    uint64_t nr = 0;
    unsigned k = (unsigned)popc(bin), nn = RRR_BS;
    while (bin) {
        if (bin & 1) { nr += g_binom[nn - 1][k]; --k; }
        bin >>= 1; --nn;
    }
    return nr;
}
static unsigned decode_popcount(unsigned k, uint64_t nr, unsigned off)                /* rrr_helper.hpp:411-460 */
{
    const unsigned n = RRR_BS;
    if (k == n) return off;
    if (k == 0) return 0;
    if (k == 1) return (n - nr - 1) < off;
    unsigned result = 0, nn = n;
    if (k + 1 < 10 + 1) {                         /* BINARY_SEARCH_THRESHOLD = 63/6 */
        while (k > 1) {
            unsigned lb = k, rb = nn + 1;
            while (lb < rb) {
                unsigned mid = (lb + rb) / 2;
                if (nr >= g_binom[mid - 1][k]) lb = mid + 1; else rb = mid;
            }
            nn = lb - 1;
            if (n - nn >= off) return result;
            ++result;
            nr -= g_binom[nn - 1][k];
            --k; --nn;
        }
    } else {
        unsigned i = 0;
        while (k > 1) {
            if (i >= off) return result;
            if (nr >= g_binom[nn - 1][k]) { nr -= g_binom[nn - 1][k]; --k; ++result; }
            --nn; ++i;
        }
    }
    return result + ((n - nr - 1) < off);
}

vlgo_rrr* vlgo_rrr_build(const uint64_t* words, uint64_t size)
{
    binom_init();
    vlgo_rrr* r = (vlgo_rrr*)calloc(1, sizeof *r);
    r->size = size;
    r->n_bt = (size + RRR_BS) / RRR_BS;
    uint64_t nsb = (r->n_bt + RRR_K - 1) / RRR_K;
    r->bt = (uint8_t*)calloc(r->n_bt + 1, 1);
    r->btnrp = (uint64_t*)calloc(nsb + 1, 8);
    r->n_rank = nsb + ((size % (RRR_K * RRR_BS)) > 0);
    r->rank = (uint64_t*)calloc(r->n_rank + 2, 8);
    r->invert = (uint8_t*)calloc(nsb + 1, 1);
    uint64_t nw = (size + 63) / 64;
    uint64_t* pad = (uint64_t*)calloc(nw + 2, 8);
    memcpy(pad, words, nw * 8);
    /* (1) classes and the length of the offset stream  (:153-166) */
    uint64_t pos = 0, i = 0, btnr_pos = 0;
    while (pos + RRR_BS <= size) { unsigned x = (unsigned)popc(get_bits(pad, pos, RRR_BS)); r->bt[i++] = (uint8_t)x; btnr_pos += g_space[x]; pos += RRR_BS; }
    if (pos < size) { unsigned x = (unsigned)popc(get_bits(pad, pos, (unsigned)(size - pos))); r->bt[i++] = (uint8_t)x; btnr_pos += g_space[x]; }
    r->btnr_bits = btnr_pos > 64 ? btnr_pos : 64;
    r->btnr = (uint64_t*)calloc((r->btnr_bits + 63) / 64 + 2, 8);
    /* (2) offsets, pointers, rank samples, inversion  (:175-234) */
    pos = 0; i = 0; btnr_pos = 0;
    uint64_t sum_rank = 0;
    int inv = 0;
    while (pos + RRR_BS <= size) {
        if (i % RRR_K == 0) {
            r->btnrp[i / RRR_K] = btnr_pos; r->rank[i / RRR_K] = sum_rank;
            if (i + RRR_K <= r->n_bt) {
                unsigned gt = 0;
                for (uint64_t j = i; j < i + RRR_K; ++j) if (r->bt[j] > RRR_BS / 2) ++gt;
                if (gt > RRR_K / 2) { r->invert[i / RRR_K] = 1; for (uint64_t j = i; j < i + RRR_K; ++j) r->bt[j] = (uint8_t)(RRR_BS - r->bt[j]); inv = 1; }
                else inv = 0;
            } else inv = 0;
        }
        unsigned x = r->bt[i++];
        unsigned sp = g_space[x];
        sum_rank += inv ? (RRR_BS - x) : x;
        if (sp) put_bits(r->btnr, btnr_pos, bin_to_nr(get_bits(pad, pos, RRR_BS)), sp);
        btnr_pos += sp;
        pos += RRR_BS;
    }
    if (pos < size) {
        if (i % RRR_K == 0) { r->btnrp[i / RRR_K] = btnr_pos; r->rank[i / RRR_K] = sum_rank; r->invert[i / RRR_K] = 0; inv = 0; }
        unsigned x = r->bt[i++];
        unsigned sp = g_space[x];
        sum_rank += inv ? (RRR_BS - x) : x;
        if (sp) put_bits(r->btnr, btnr_pos, bin_to_nr(get_bits(pad, pos, (unsigned)(size - pos))), sp);
        btnr_pos += sp;
    }
    r->rank[r->n_rank - 1] = sum_rank;                      /* :233 */
    free(pad);
    return r;
}

uint64_t vlgo_rrr_rank(const vlgo_rrr* r, uint64_t i)        /* rrr_vector.hpp:444-480 */
{
    uint64_t bt_idx = i / RRR_BS, sample = bt_idx / RRR_K;
    uint64_t btnrp = r->btnrp[sample], rank = r->rank[sample];
    if (sample + 1 < r->n_rank) {
        uint64_t diff = r->rank[sample + 1] - rank;
        if (diff == 0) return rank;
        if (diff == (uint64_t)RRR_BS * RRR_K) return rank + i - sample * RRR_K * RRR_BS;
    }
    const int inv = r->invert[sample];
    for (uint64_t j = sample * RRR_K; j < bt_idx; ++j) {
        unsigned x = r->bt[j];
        rank += inv ? RRR_BS - x : x;
        btnrp += g_space[x];
    }
    unsigned off = (unsigned)(i % RRR_BS);
    if (!off) return rank;
    unsigned bt = inv ? RRR_BS - r->bt[bt_idx] : r->bt[bt_idx];
    unsigned len = g_space[bt];
    uint64_t nr = get_bits(r->btnr, btnrp, len);
    return rank + decode_popcount(bt, nr, off);
}

uint64_t vlgo_rrr_bits(const vlgo_rrr* r) { return 6 * r->n_bt + r->btnr_bits + 64 * ((r->n_bt + 31) / 32) * 2 + (r->n_bt + 31) / 32; }
void vlgo_rrr_free(vlgo_rrr* r) { if (!r) return; free(r->bt); free(r->btnr); free(r->btnrp); free(r->rank); free(r->invert); free(r); }
