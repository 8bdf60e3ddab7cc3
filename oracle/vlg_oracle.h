/* oracle/vlg_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's FM-index VLG path (olydis/vlg_matching, an
 * sdsl-lite fork).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use it; the product (vlg_matching_amd/, include/) never links, loads or calls it.
 *
 * Parity status: PINNED at component level against the reference's own code compiled from
 * /root/reference (oracle/_ref, see ref_glue.cpp: wt_huff ctor/rank/inverse_select,
 * rank_support_v/v5, byte_alphabet, LF) and at VLG level against the known answers the survey
 * captured from sdsl::locate/count(vlg_index, query) (tests/golden/vlg_known_answers.json).
 * backward_search / locate / the merge join cannot be compiled from the reference in this image
 * (divsufsort.h absent) and are pinned by those goldens plus brute-force scans.
 */
#ifndef VLG_ORACLE_H
#define VLG_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vlgo_index vlgo_index;

/* One prefix-code tree node, reference layout wt_helper.hpp:73-131 (BFS order, node 0 = root). */
typedef struct {
    uint64_t bv_pos;       /* inner: start of the node's bits in the WT bit-vector          */
    uint64_t bv_pos_rank;  /* inner: rank1(bv_pos); leaf: the symbol (wt_pc.hpp:334-335)     */
    uint16_t parent;       /* 0xFFFF = none                                                */
    uint16_t child[2];     /* 0xFFFF = leaf                                                */
} vlgo_node;

#define VLGO_OK 0
#define VLGO_E_PARSE_QMARK 1   /* "expected '?'"      vlg_index.hpp:97-99 */
#define VLGO_E_PARSE_MINMAX 2  /* "min-gap > max-gap" vlg_index.hpp:92-94 / utils.hpp:57-59 */
#define VLGO_E_PARSE_NUM 3     /* std::stoull would throw                                  */
#define VLGO_E_EMPTY_SUBPATTERN 4
#define VLGO_MAX_SUB 64

typedef struct {
    uint32_t k;                          /* number of sub-patterns                           */
    const uint8_t* sub[VLGO_MAX_SUB];    /* pointers into the caller's pattern string        */
    uint64_t sub_len[VLGO_MAX_SUB];
    uint64_t lo[VLGO_MAX_SUB];           /* lo[i], hi[i] (i>=1): start-to-start distance     */
    uint64_t hi[VLGO_MAX_SUB];           /*   bounds between sub-pattern i-1 and i           */
    uint64_t end_len;                    /* non-overlap length added to the last position    */
} vlgo_query;

/* ---- construction --------------------------------------------------------------------- */
/* Suffix array of text[0,n) where text[n-1]==0 is the unique smallest byte (own sorter;
 * the SA of a text is unique, so any correct sorter yields the reference's SA). */
int vlgo_suffix_array(const uint8_t* text, uint64_t n, uint64_t* sa);

/* Build from raw text (no zero byte; the 0 sentinel is appended: construct.hpp:47-52). */
vlgo_index* vlgo_build(const uint8_t* text, uint64_t n_text, uint32_t dens);
/* Build from a BWT (n bytes incl. the single 0) and, optionally, the full SA (n entries). */
vlgo_index* vlgo_build_from_bwt(const uint8_t* bwt, const uint64_t* sa, uint64_t n, uint32_t dens);
/* Adopt already-built parts in the reference's layout (used by bench.py's cpu_baseline leg:
 * the parts come from the device builder, whose output tests prove bit-identical to vlgo_build). */
vlgo_index* vlgo_from_parts(uint64_t n, uint32_t sigma, const uint8_t* char2comp, const uint64_t* C,
                            const uint64_t* bv_words, uint64_t bv_bits,
                            const vlgo_node* nodes, uint32_t n_nodes,
                            const uint64_t* samples, uint64_t n_samples, uint32_t dens);
void vlgo_free(vlgo_index*);

/* ---- parts accessors -------------------------------------------------------------------- */
uint64_t vlgo_size(const vlgo_index*);            /* n = |text|+1 */
uint32_t vlgo_sigma(const vlgo_index*);
const uint8_t* vlgo_char2comp(const vlgo_index*); /* [256] */
const uint64_t* vlgo_C(const vlgo_index*);        /* [sigma+1] */
uint64_t vlgo_bv_bits(const vlgo_index*);
const uint64_t* vlgo_bv_words(const vlgo_index*);
uint32_t vlgo_n_nodes(const vlgo_index*);
const vlgo_node* vlgo_nodes(const vlgo_index*);
const uint64_t* vlgo_paths(const vlgo_index*);    /* m_path[256] */
uint64_t vlgo_n_samples(const vlgo_index*);
uint64_t vlgo_sample(const vlgo_index*, uint64_t j);
const uint64_t* vlgo_rank_blocks(const vlgo_index*, uint64_t* n_words);  /* rank_support_v array */
const uint8_t* vlgo_bwt(const vlgo_index*);       /* only if built from text/BWT, else NULL */

/* ---- primitives ------------------------------------------------------------------------- */
uint64_t vlgo_bv_rank1(const vlgo_index*, uint64_t idx);                 /* rank_support_v::rank */
uint64_t vlgo_wt_rank(const vlgo_index*, uint64_t i, uint8_t c);         /* wt_pc::rank */
uint64_t vlgo_inverse_select(const vlgo_index*, uint64_t i, uint8_t* c); /* wt_pc::inverse_select */
uint64_t vlgo_lf(const vlgo_index*, uint64_t i);
uint64_t vlgo_sa(const vlgo_index*, uint64_t i, uint64_t* lf_steps, uint64_t* levels); /* csa[i] */
/* backward_search(csa,0,n-1,pat): returns count, (l,r) as the reference leaves them. */
uint64_t vlgo_backward_search(const vlgo_index*, const uint8_t* pat, uint64_t m, uint64_t* l, uint64_t* r);
/* locate: occ[j] = csa[l+j] (SA order, unsorted). Returns count; writes min(count,cap). */
uint64_t vlgo_locate(const vlgo_index*, const uint8_t* pat, uint64_t m, uint64_t* out, uint64_t cap);

/* Stand-alone restatements of the bit-vector rank structures. */
uint64_t vlgo_rank_v_build(const uint64_t* words, uint64_t nbits, uint64_t* blocks /* 2*((cap>>9)+1) */);
uint64_t vlgo_rank_v(const uint64_t* words, const uint64_t* blocks, uint64_t idx);
uint64_t vlgo_rank_v5_build(const uint64_t* words, uint64_t nbits, uint64_t* blocks /* 2*((cap>>11)+1) */);
uint64_t vlgo_rank_v5(const uint64_t* words, const uint64_t* blocks, uint64_t idx);

/* rrr_vector<63> + rank_support_rrr<1,63> (include/sdsl/rrr_vector.hpp:145-237,444-480; rrr_helper.hpp:173-460) */
typedef struct vlgo_rrr vlgo_rrr;
vlgo_rrr* vlgo_rrr_build(const uint64_t* words, uint64_t nbits);
uint64_t vlgo_rrr_rank(const vlgo_rrr*, uint64_t i);
uint64_t vlgo_rrr_bits(const vlgo_rrr*);      /* compressed size in bits (classes + offsets + samples) */
void vlgo_rrr_free(vlgo_rrr*);

/* ---- queries ---------------------------------------------------------------------------- */
/* dialect 0 = library  (gapped_pattern_query, vlg_index.hpp:54-105: '?' mandatory, per-gap bounds
 *                       + |s_{i-1}|, non-overlap by |s_last|)
 * dialect 1 = benchmark (gapped_pattern utils.hpp:25-70 + index_sasearch.hpp:68-69,113: no '?',
 *                       gaps[0] and |s_0| for every gap and for non-overlap)                 */
int vlgo_parse(const uint8_t* regexp, uint64_t len, int dialect, vlgo_query* q);

/* Gap-bounded merge join over k ascending lists (index_sasearch.hpp:85-116 generalised to per-gap
 * bounds).  tuples_out receives k positions per match (may be NULL).
 * Returns the number of matches; writes at most cap matches. */
uint64_t vlgo_join(uint32_t k, const uint64_t* const* lists, const uint64_t* lens,
                   const uint64_t* lo, const uint64_t* hi, uint64_t end_len,
                   uint64_t* tuples_out, uint64_t cap);

/* Full FM path for one query: backward_search + locate + sort + join.
 * stats (optional, 4 x u64): located occurrences, LF steps, WT levels walked, backward-search ranks */
uint64_t vlgo_search(const vlgo_index*, const vlgo_query* q, uint64_t* tuples_out, uint64_t cap, uint64_t* stats);

/* SASEARCH (benchmark/gapped-matching/include/index_sasearch.hpp:58-118): plain suffix array + text; forward_search of
 * suffix_array_algorithm.hpp:48-112, every range copied and sorted, then the same join.  text has n bytes incl. the sentinel. */
uint64_t vlgo_sa_forward_search(const uint8_t* text, uint64_t n, const uint32_t* sa, const uint8_t* pat, uint64_t m,
                                uint64_t* l_res, uint64_t* r_res);
uint64_t vlgo_sasearch(const uint8_t* text, uint64_t n, const uint32_t* sa, const vlgo_query* q, uint64_t* tuples_out, uint64_t cap,
                       uint64_t* stats);

/* Many queries (blob + offsets, nq + 1 of them) on n_threads threads drawing from a shared counter, until budget_s seconds have
 * passed: what all host cores would make of the reference's single-threaded loop (gm_search.cpp:91-121).  stats / matches are
 * summed over the threads (may be NULL); returns the number of queries finished. */
uint64_t vlgo_search_many(const vlgo_index*, const uint8_t* blob, const uint64_t* off, uint64_t nq, int dialect, uint32_t n_threads,
                          double budget_s, uint64_t* stats, uint64_t* matches);
uint64_t vlgo_sasearch_many(const uint8_t* text, uint64_t n, const uint32_t* sa, const uint8_t* blob, const uint64_t* off, uint64_t nq,
                            int dialect, uint32_t n_threads, double budget_s, uint64_t* stats, uint64_t* matches);

/* ---- integer alphabets and text-order SA sampling (vlg_oracle_int.c) --------------------------------------------
 * csa_wt<wt_int<>, dens, ., sa_order | text_order sampling, ., int_alphabet<>> restated in the reference's layout. */
typedef struct vlgo_int_index vlgo_int_index;
int vlgo_int_suffix_array(const uint64_t* text, uint64_t n, uint64_t* sa);     /* text[n-1] = 0, the unique smallest symbol */
vlgo_int_index* vlgo_int_build(const uint64_t* text, uint64_t n_text, uint32_t dens, int text_order);   /* NULL: a 0 symbol in the text */
void vlgo_int_free(vlgo_int_index*);
uint64_t vlgo_int_size(const vlgo_int_index*);
uint64_t vlgo_int_sigma(const vlgo_int_index*);
const uint64_t* vlgo_int_C(const vlgo_int_index*);
const uint64_t* vlgo_int_comp2char(const vlgo_int_index*);
uint32_t vlgo_int_levels(const vlgo_int_index*);
const uint64_t* vlgo_int_tree(const vlgo_int_index*);          /* n * levels bits */
const uint64_t* vlgo_int_bwt(const vlgo_int_index*);
uint64_t vlgo_int_n_samples(const vlgo_int_index*);
const uint64_t* vlgo_int_samples(const vlgo_int_index*);
const uint64_t* vlgo_int_marked(const vlgo_int_index*);        /* text-order sampling only */
uint64_t vlgo_int_char2comp(const vlgo_int_index*, uint64_t c);
uint64_t vlgo_int_rank(const vlgo_int_index*, uint64_t i, uint64_t c);                    /* wt_int::rank */
uint64_t vlgo_int_inverse_select(const vlgo_int_index*, uint64_t i, uint64_t* c);         /* wt_int::inverse_select */
uint64_t vlgo_int_lf(const vlgo_int_index*, uint64_t i);
uint64_t vlgo_int_sa(const vlgo_int_index*, uint64_t i, uint64_t* lf_steps);              /* csa[i] */
uint64_t vlgo_int_backward_search(const vlgo_int_index*, const uint64_t* pat, uint64_t m, uint64_t* l, uint64_t* r);
uint64_t vlgo_int_locate(const vlgo_int_index*, const uint64_t* pat, uint64_t m, uint64_t* out, uint64_t cap);
/* gapped_pattern_query<int_alphabet_tag>: q->sub / sub_len point into the regexp (characters), syms / sub_off hold the symbols */
int vlgo_parse_int(const uint8_t* regexp, uint64_t len, vlgo_query* q, uint64_t* syms, uint64_t syms_cap, uint64_t* sub_off /* [k+1] */);
uint64_t vlgo_int_search(const vlgo_int_index*, const vlgo_query* q, const uint64_t* syms, const uint64_t* sub_off, uint64_t* tuples_out,
                         uint64_t cap, uint64_t* stats);
/* text-order sampling on top of the byte index */
typedef struct vlgo_text_order vlgo_text_order;
vlgo_text_order* vlgo_text_order_build(const vlgo_index*, uint32_t dens);
void vlgo_text_order_free(vlgo_text_order*);
const uint64_t* vlgo_text_order_marked(const vlgo_text_order*);
const uint64_t* vlgo_text_order_samples(const vlgo_text_order*);
uint64_t vlgo_text_order_n_samples(const vlgo_text_order*);
uint64_t vlgo_text_order_sa(const vlgo_index*, const vlgo_text_order*, uint64_t i, uint64_t* lf_steps);
#ifdef __cplusplus
}
#endif
#endif
