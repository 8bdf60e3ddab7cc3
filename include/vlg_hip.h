/* include/vlg_hip.h -- C-ABI of the MI355X-native variable-length-gap (VLG) matcher.
 *
 * This is the drop-in boundary for the FM-index VLG hot path of olydis/vlg_matching
 * (an sdsl-lite fork).  The reference has no FFI: the path sits behind C++ template concepts
 * (SURVEY.md 8b).  Each entry point below names the reference interface it replaces
 * (file:line relative to the reference root); INTEGRATION.md shows the C++ adapter a reference
 * maintainer would add to route `index_*::search` / `sdsl::locate` through it.
 *
 * Conventions
 *   - plain C, no exceptions: every call returns a vlg_status; vlg_last_error() gives the text.
 *   - "h_" arguments are host pointers, "d_" arguments are device (HBM) pointers of the current
 *     HIP device; the caller owns every buffer it passes in.
 *   - positions/counts are uint64_t at the boundary, like sdsl's size_type; all arithmetic is
 *     unsigned integer / bit manipulation (no floating point anywhere on the path).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - a vlg_index is immutable after creation; batched calls on different streams may share it.
 *     A vlg_workspace / vlg_result belongs to one caller thread at a time.
 *   - there is no CPU fallback: without a usable HIP device every compute call fails with
 *     VLG_E_NO_DEVICE.
 */
#ifndef VLG_HIP_H
#define VLG_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    VLG_OK = 0,
    VLG_E_INVALID = 1,       /* bad argument                                                   */
    VLG_E_NO_DEVICE = 2,     /* no HIP device / HIP runtime error                              */
    VLG_E_OOM = 3,           /* device or host allocation failed                               */
    VLG_E_PARSE = 4,         /* gap syntax: the reference throws std::runtime_error
                                (include/sdsl/vlg_index.hpp:92-99)                             */
    VLG_E_ZERO_BYTE = 5,     /* text contains a 0 byte: std::logic_error in the reference
                                (include/sdsl/construct.hpp:36-45)                             */
    VLG_E_UNSUPPORTED = 6,   /* e.g. text longer than this build handles                       */
    VLG_E_WORKSPACE = 7,     /* one query needs more workspace than the configured cap          */
    VLG_E_INTERNAL = 8
} vlg_status;

typedef struct vlg_index vlg_index;          /* FM-index resident in HBM                       */
typedef struct vlg_bitvector vlg_bitvector;  /* stand-alone rank-enabled bit-vector in HBM     */
typedef struct vlg_queries vlg_queries;      /* a parsed query batch resident in HBM           */
typedef struct vlg_result vlg_result;        /* results of one vlg_search_batch, in HBM        */
typedef struct vlg_workspace vlg_workspace;  /* scratch HBM + stream + per-kernel statistics   */

const char* vlg_last_error(void);            /* thread-local text of the last failure          */
const char* vlg_version(void);
vlg_status vlg_device_count(int* n);
vlg_status vlg_set_device(int ordinal);

/* ------------------------------------------------------------------------------------------
 * Index parts in the REFERENCE's own layout (host memory) -- what a loaded
 * sdsl::csa_wt<wt_huff<>,32,64> exposes (include/sdsl/csa_wt.hpp:120-152):
 *   wavelet_tree.bv (wt_pc.hpp:185), the _byte_tree nodes (wt_helper.hpp:73-131, BFS order),
 *   char2comp / C (csa_wt.hpp:139-142), sa_sample (csa_wt.hpp:150; csa_sampling_strategy.hpp:85-111).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t bv_pos;        /* inner node: first bit of the node in wavelet_tree.bv             */
    uint64_t bv_pos_rank;   /* inner node: rank1(bv_pos); leaf: the symbol                      */
    uint16_t parent;        /* 0xFFFF for the root                                             */
    uint16_t child[2];      /* 0xFFFF,0xFFFF for a leaf                                        */
} vlg_wt_node;

typedef struct {
    uint64_t n;                   /* csa.size() = |text| + 1 (sentinel)                         */
    uint32_t sigma;               /* csa.sigma                                                 */
    uint32_t sa_sample_dens;      /* t_dens, 32 in the reference default (csa_wt.hpp:61)        */
    const uint8_t* char2comp;     /* [256]                                                     */
    const uint64_t* C;            /* [sigma+1]                                                 */
    const uint64_t* bv_words;     /* ceil(bv_bits/64) words, bit i = word[i>>6] >> (i&63)      */
    uint64_t bv_bits;
    const vlg_wt_node* nodes;     /* [n_nodes], node 0 = root                                  */
    uint32_t n_nodes;             /* 2*sigma-1                                                 */
    const uint64_t* sa_samples;   /* [ceil(n/dens)] unpacked: SA[0], SA[dens], ...              */
    uint64_t n_samples;
} vlg_index_parts;

typedef struct {
    uint64_t n;                   /* text length + 1                                           */
    uint32_t sigma;
    uint32_t sa_sample_dens;
    uint32_t n_nodes;
    uint32_t max_code_len;        /* deepest leaf of the Huffman-shaped tree                    */
    uint64_t wt_bits;             /* sum of node sizes = n * (mean code length)                 */
    uint64_t n_blocks;            /* 32-byte super-blocks in HBM                               */
    uint64_t n_samples;
    uint64_t hbm_bytes;           /* total device bytes of the index                           */
    uint32_t pos_bytes;           /* 4 (n <= 2^32) or 8: width of SA samples / SA indices; text positions inside the
                                     kernels are 32-bit up to n = 2^32 + 1 whatever this says     */
    uint32_t bv_kind;             /* VLG_BV_PLAIN or VLG_BV_RRR63                               */
    uint32_t sampling;            /* VLG_SAMPLING_SA_ORDER or VLG_SAMPLING_TEXT_ORDER           */
    uint32_t reserved;
} vlg_index_info;

/* Build on the device from raw text (no 0 byte; the sentinel is appended like
 * sdsl::construct, include/sdsl/construct.hpp:47-52,112-159): suffix sort, BWT, Huffman-shaped
 * wavelet tree (wt_pc.hpp:197-248), SA sampling -- all in HBM.  Replaces
 * `index_*(collection&)` (benchmark/gapped-matching/include/index_sasearch.hpp:23-31) and
 * `construct(idx, file, num_bytes)` (include/sdsl/vlg_index.hpp:375-392). */
vlg_status vlg_index_build(const uint8_t* h_text, uint64_t n_text, uint32_t sa_sample_dens, vlg_index** out);
/* Same, text already in HBM. */
vlg_status vlg_index_build_device(const uint8_t* d_text, uint64_t n_text, uint32_t sa_sample_dens,
                                  void* stream, vlg_index** out);
/* Adopt an index built by the reference (host arrays in its layout) -- replaces
 * `idx.load(istream)` (index_sasearch.hpp:41-45; csa_wt.hpp:395-407).
 * Limits: n <= 2^36, and no wavelet-tree node may hold 2^32 or more 1-bits (the super-block counts are 32-bit and
 * node-relative): always true for n <= 2^32, otherwise checked against the symbol counts -> VLG_E_UNSUPPORTED. */
vlg_status vlg_index_from_parts(const vlg_index_parts* h_parts, vlg_index** out);
/* Export to the reference layout (two-phase: pass NULL buffers in `out` to get sizes only). -- replaces
 * `idx.serialize(ostream)` (index_sasearch.hpp:33-39).  Buffers in `out` are caller-allocated host arrays. */
typedef struct {
    uint8_t* char2comp;           /* [256]                                                     */
    uint64_t* C;                  /* [257]                                                     */
    uint64_t* bv_words;           /* [ceil(bv_bits/64)]                                        */
    vlg_wt_node* nodes;           /* [n_nodes]                                                 */
    uint64_t* sa_samples;         /* [n_samples]                                               */
} vlg_index_parts_out;
vlg_status vlg_index_export_parts(const vlg_index* idx, vlg_index_parts* sizes, vlg_index_parts_out* out);
/* A second index over the same text whose wavelet-tree bit-vectors are H0-compressed -- csa_wt<wt_huff<rrr_vector<63>>>
 * (BASELINE config 5; include/sdsl/rrr_vector.hpp).  Every search entry point accepts it and returns identical results.
 * Blocks of 63 bits are stored as a 6-bit class and an offset of ceil(log2 C(63,k)) bits -- the sizes of rrr_vector<63>; the
 * offset numbers the blocks of a class by halves (csrc/rrr_code.hpp) instead of bit by bit, so that a rank decodes in a fixed
 * short sequence of table lookups.  The device image is this library's own (blob magic "VGLB2"); a stock
 * csa_wt<wt_huff<rrr_vector<63>>> file is read by vlg_index_load_sdsl_kind, which decodes its blocks and ends here.  `src` must be
 * a plain index.  An integer-alphabet index (vlg_index_build_int) is compressed the same way, level by level of its wavelet matrix:
 * csa_wt<wt_int<rrr_vector<63>>, ., ., ., ., int_alphabet<>> (test/csa_int_test.cpp:32); vlg_index_info then says
 * VLG_BV_INT_MATRIX_RRR63. */
#define VLG_BV_PLAIN 0
#define VLG_BV_RRR63 1
#define VLG_BV_INT_MATRIX 2          /* integer-alphabet index (vlg_index_build_int): bv_kind of vlg_index_info */
#define VLG_BV_INT_MATRIX_RRR63 3    /* ... with rrr-63 levels (vlg_index_compress of one) */
vlg_status vlg_index_compress(const vlg_index* src, int bv_kind, vlg_index** out);
/* The other SA sampling strategy of the reference, and other densities: a second index over the same BWT whose SA samples are
 *   VLG_SAMPLING_SA_ORDER    SA[0], SA[d], SA[2d], ...                   sa_order_sa_sampling, csa_wt's default
 *                            (include/sdsl/csa_sampling_strategy.hpp:64-112)
 *   VLG_SAMPLING_TEXT_ORDER  the SA values that are multiples of d, found through a marked bit-vector over the SA indices:
 *                            is_sampled(i) = marked[i], csa[i] = samples[rank_marked(i)] * d   text_order_sa_sampling (:127-246) --
 *                            the locate indexes of benchmark/indexing_locate/index.config:9-12; a walk takes SA[i] % d LF steps
 * Every search entry point accepts it and returns identical results.  `src` must be SA-order sampled (plain or rrr); the new index keeps
 * its samples as wide as the source does (8 bytes when SA indices need 33 bits: n > 2^32, or VLG_FORCE_POS64);
 * VLG_SAMPLING_SA_ORDER with d = 1 (also vlg_index_build* with sa_sample_dens = 1, any n) keeps the whole suffix array in HBM
 * (csa_wt<wt_huff<>, 1, .>: 4 B per text character up to 4 GiB of text, 8 B beyond): csa[i] is one read (csa_wt.hpp:335-348 with
 * zero LF steps), vlg_search_batch's locate stage becomes a coalesced copy of the SA intervals and keeps no trail table;
 * for a text-order index vlg_index_export_parts' sa_samples receives the condensed values SA / d, vlg_index_export_marked the
 * marks (bit i = h_words[i >> 6] >> (i & 63), ceil(n / 64) words). */
#define VLG_SAMPLING_SA_ORDER 0
#define VLG_SAMPLING_TEXT_ORDER 1
vlg_status vlg_index_resample(const vlg_index* src, int sampling, uint32_t sa_sample_dens, vlg_index** out);
vlg_status vlg_index_export_marked(const vlg_index* idx, uint64_t* h_words);
vlg_status vlg_index_get_info(const vlg_index* idx, vlg_index_info* info);

/* FM-index of an INTEGER text -- csa_wt<wt_int<>, dens, ., sa_order_sa_sampling, ., int_alphabet<>>, the csa the reference builds for
 * vlg_index<int_alphabet_tag> (include/sdsl/vlg_index.hpp:383-384) and its word-level experiments: int_alphabet
 * (include/sdsl/csa_alphabet_strategy.hpp:394-470), BWT in a level-wise tree (wt_int.hpp; here a wavelet matrix over the compact
 * symbols: one super-block read per level), LF / csa[i] / backward_search as for bytes.  h_text: n_symbols uint32_t, none of them 0
 * (VLG_E_ZERO_BYTE: construct.hpp:36-45); n_symbols < 2^32 / 5.  The handle is a vlg_index: vlg_search_batch (with a batch parsed
 * by vlg_queries_parse_int), vlg_queries_intervals / _occurrences, vlg_backward_search_batch (patterns = little-endian uint32_t
 * symbols), vlg_sa_batch, vlg_locate_batch, blob export / attach / broadcast take it; the byte-only entry points refuse it. */
vlg_status vlg_index_build_int(const uint32_t* h_text, uint64_t n_symbols, uint32_t sa_sample_dens, vlg_index** out);
/* 64-bit symbols.  gapped_pattern_query<int_alphabet_tag> reads uint64_t tokens (vlg_index.hpp:57-69) and int_vector<64> texts hold
 * 64-bit symbols; the device indexes (vlg_index_build_int, vlg_wtsa_build_int) hold uint32_t.  A symbol map carries the sorted
 * distinct symbols of a 64-bit text and sends a symbol to its rank + 1: dense, order-preserving, never 0 -- so the suffix order, every
 * SA interval and every result of the mapped text are those of the original.  vlg_symbol_map_apply maps a text (or any symbols:
 * one that does not occur in the map's text becomes sigma + 1, which occurs nowhere in the mapped text), the mapped text goes to
 * vlg_index_build_int / vlg_wtsa_build_int, and vlg_queries_parse_int_mapped parses a batch as vlg_queries_parse_int does -- tokens
 * of any 64-bit value -- through the same map.  Host only (no GPU needed for the map itself). */
typedef struct vlg_symbol_map vlg_symbol_map;
vlg_status vlg_symbol_map_create(const uint64_t* h_text, uint64_t n_symbols, vlg_symbol_map** out);
uint64_t vlg_symbol_map_sigma(const vlg_symbol_map* map);
vlg_status vlg_symbol_map_symbols(const vlg_symbol_map* map, uint64_t* h_out /* [sigma] ascending */);
vlg_status vlg_symbol_map_apply(const vlg_symbol_map* map, const uint64_t* h_in, uint64_t n, uint32_t* h_out);
void vlg_symbol_map_destroy(vlg_symbol_map* map);
/* int_alphabet of such an index: *sigma symbols (comp 0 = the sentinel), h_C[sigma + 1], h_comp2char[sigma]; null buffers: sigma only */
vlg_status vlg_index_export_int_alphabet(const vlg_index* idx, uint64_t* sigma, uint64_t* h_C, uint64_t* h_comp2char);
/* wt_int::rank(i, c) (include/sdsl/wt_int.hpp:370-395) on its BWT: out[j] = #d_sym[j] in BWT[0, d_i[j]) */
vlg_status vlg_int_rank_batch(const vlg_index* idx, const uint64_t* d_i, const uint32_t* d_sym, uint64_t* d_out, uint64_t count, void* stream);
void vlg_index_destroy(vlg_index* idx);

/* Suffix array of text + sentinel on the device (what sdsl::construct_sa computes, include/sdsl/construct_sa.hpp:145-170): d_sa
 * receives n_text + 1 entries, d_sa[0] = n_text.  The same prefix-doubling sorter the index builder uses; n_text < 2^32 - 1. */
vlg_status vlg_suffix_array_device(const uint8_t* d_text, uint64_t n_text, uint32_t* d_sa, void* stream);
/* ISA samples as csa_wt keeps them (include/sdsl/csa_sampling_strategy.hpp:626-642): h_out[j] = the SA index i with SA[i] = j * inv_dens,
 * count = (n-1)/inv_dens + 1.  Computed from the index alone by walking LF from every SA sample. */
vlg_status vlg_index_isa_samples(const vlg_index* idx, uint32_t inv_dens, uint64_t* h_out, uint64_t count);
/* Store the index in the reference's on-disk format of csa_wt<wt_huff<>,32,64> (csa_wt.hpp:374-393) so that stock sdsl
 * can load_from_file() it: wavelet tree with rank_support_v and both select_support_mcl, SA samples, ISA samples (density
 * 64), byte_alphabet.  The index must be a plain one with SA sample density 32. */
vlg_status vlg_index_save_sdsl(const vlg_index* idx, const char* path);
/* Load an index stored by stock sdsl: the file written by `store_to_file(csa, file)` / `csa.serialize(out)` for
 * csa_wt<wt_huff<>, t_dens, t_inv_dens> with the default sampling strategies and byte_alphabet
 * (include/sdsl/csa_wt.hpp:374-393; member formats: wt_pc.hpp:638-652, int_vector.hpp:584-600, rank_support_v.hpp:134-148,
 * select_support_mcl.hpp:424-494, wt_helper.hpp:112-131,275-301, lib/csa_alphabet_strategy.cpp:103-121).
 * t_dens is a template parameter of the reference type and is not stored: pass it (0 = 32, the reference default).
 * vlg_sdsl_file_* parse on the host only (no GPU needed); vlg_index_load_sdsl = open + parts + vlg_index_from_parts. */
typedef struct vlg_sdsl_file vlg_sdsl_file;
vlg_status vlg_sdsl_file_open(const char* path, uint32_t sa_sample_dens, vlg_sdsl_file** out);
vlg_status vlg_sdsl_file_parts(const vlg_sdsl_file* f, vlg_index_parts* parts);   /* pointers stay valid until close */
void vlg_sdsl_file_close(vlg_sdsl_file* f);
vlg_status vlg_index_load_sdsl(const char* path, uint32_t sa_sample_dens, vlg_index** out);
/* The same for the file of csa_wt<wt_huff<rrr_vector<63>>, t_dens, t_inv_dens> (bv_kind = VLG_BV_RRR63; the index type of
 * benchmark/indexing_count/index.config:10 and BASELINE config 5): the wavelet tree's bit-vector is an rrr_vector<63>
 * (include/sdsl/rrr_vector.hpp:349-372: size, block classes, offsets, pointer and rank samples, inversion bits; its rank / select
 * supports store nothing).  Every block is decoded on the host (rrr_helper.hpp:304-375) and the index is kept rrr-coded on the
 * device in this library's block numbering (vlg_index_compress): same sizes, same answers.  VLG_BV_PLAIN = vlg_index_load_sdsl. */
vlg_status vlg_sdsl_file_open_kind(const char* path, uint32_t sa_sample_dens, int bv_kind, vlg_sdsl_file** out);
vlg_status vlg_index_load_sdsl_kind(const char* path, uint32_t sa_sample_dens, int bv_kind, vlg_index** out);

/* One contiguous device image of the read-only index, for replication across the GPUs of a node
 * (SURVEY.md 8e): the owner exports it into caller-provided HBM, the caller moves it with RCCL
 * (ncclBroadcast, or torch.distributed.broadcast on a uint8 tensor), every other rank attaches.
 * The attached index borrows the blob: keep it alive until vlg_index_destroy. */
vlg_status vlg_index_blob_bytes(const vlg_index* idx, uint64_t* bytes);
vlg_status vlg_index_blob_export(const vlg_index* idx, void* d_blob, uint64_t bytes, void* stream);
vlg_status vlg_index_attach_blob(const void* d_blob, uint64_t bytes, vlg_index** out);
/* The same for ONE process that drives several GPUs of a node (the C++ host's `gm_search_gpu -g N`): a copy of the index in the
 * HBM of `device`, moved by a peer copy over xGMI; the new handle owns its memory.  Calls on it must run with `device`
 * current (vlg_set_device is per host thread). */
vlg_status vlg_index_replicate(const vlg_index* src, int device, vlg_index** out);

/* ------------------------------------------------------------------------------------------
 * X1: RCCL (one process per GPU over xGMI; SURVEY.md 8b "broadcast(handle, ncclComm)", 8e).  An ncclComm_t crosses this ABI as
 * void*.  The caller may bring its own communicator (created with the RCCL its process holds: the library binds RCCL at run time
 * and calls the one already loaded, else librccl.so.1), or make one here: rank 0 calls vlg_comm_unique_id, the host hands the 128
 * bytes to every rank (a pipe, a file, MPI ...), every rank calls vlg_comm_create with its device current.
 * ---------------------------------------------------------------------------------------- */
const char* vlg_comm_library(void);                               /* path of the RCCL that was bound ("" = none)  */
typedef struct { char bytes[128]; } vlg_comm_id;                  /* ncclUniqueId                                */
vlg_status vlg_comm_unique_id(vlg_comm_id* out);                  /* ncclGetUniqueId                             */
vlg_status vlg_comm_create(const vlg_comm_id* id, int n_ranks, int rank, void** nccl_comm);   /* ncclCommInitRank */
vlg_status vlg_comm_info(void* nccl_comm, int* n_ranks, int* rank);
void vlg_comm_destroy(void* nccl_comm);
/* The read-only index of rank `root` in every rank's HBM: one ncclBroadcast of its contiguous image (size first), received
 * straight into the allocation the new index owns -- `idx.load(istream)` on every rank (gm_search.cpp:68-80) replaced by one load
 * + one collective.  Root: pass the index, *out (optional) receives the same handle; other ranks: pass NULL, *out receives a new
 * index to destroy.  Collective: every rank of the communicator calls it. */
vlg_status vlg_index_broadcast(const vlg_index* idx_or_null, void* nccl_comm, int root, void* stream, vlg_index** out);
/* Sum over the ranks, in place, modulo 2^64 -- num_results / checksum (gm_search.cpp:110-114) / located occurrences of a sharded
 * batch.  Host values in, host values out. */
vlg_status vlg_comm_allreduce_sum_u64(void* nccl_comm, uint64_t* h_vals, uint32_t count, void* stream);
/* All-gather of device buffers of different sizes: rank r contributes h_counts[r] elements of elem_bytes (d_send); every rank
 * ends with all of them in rank order in d_recv.  The exchange step of the list-sharded locate (vlg_workspace option "comm"). */
vlg_status vlg_comm_allgatherv(void* nccl_comm, const void* d_send, const uint64_t* h_counts, uint32_t elem_bytes, void* d_recv,
                               void* stream);

/* ------------------------------------------------------------------------------------------
 * K1: batched bit-rank.  rank_support_v<1,1>::rank / rank_support_v5<1,1>::rank
 * (include/sdsl/rank_support_v.hpp:114-124, rank_support_v5.hpp:116-134) on a plain bit_vector,
 * re-laid out as 256-bit super-blocks {32-bit count, 224 data bits}.
 * out[j] = number of 1 bits in bv[0, idx[j]),  0 <= idx[j] <= nbits.
 * ---------------------------------------------------------------------------------------- */
vlg_status vlg_bitvector_create(const uint64_t* h_words, uint64_t nbits, vlg_bitvector** out);
vlg_status vlg_bitvector_rank_batch(const vlg_bitvector* bv, const uint64_t* d_idx, uint64_t* d_out,
                                    uint64_t count, void* stream);
uint64_t vlg_bitvector_hbm_bytes(const vlg_bitvector* bv);
void vlg_bitvector_destroy(vlg_bitvector* bv);

/* ------------------------------------------------------------------------------------------
 * K6: the same batched rank on an H0-compressed bit-vector: rrr_vector<63> + rank_support_rrr<1,63>
 * (include/sdsl/rrr_vector.hpp:145-237, 444-480; block coding include/sdsl/rrr_helper.hpp:304-320, 411-460).
 * One 32-byte header per 32 blocks of 63 bits {ones before, offset position, 32 x 6-bit classes} + offset stream;
 * blocks are decoded on the fly against the binomial table staged in LDS.  Results equal vlg_bitvector_rank_batch.
 * ---------------------------------------------------------------------------------------- */
typedef struct vlg_rrr_bitvector vlg_rrr_bitvector;
vlg_status vlg_rrr_bitvector_create(const uint64_t* h_words, uint64_t nbits, vlg_rrr_bitvector** out);
vlg_status vlg_rrr_bitvector_rank_batch(const vlg_rrr_bitvector* bv, const uint64_t* d_idx, uint64_t* d_out,
                                        uint64_t count, void* stream);
uint64_t vlg_rrr_bitvector_hbm_bytes(const vlg_rrr_bitvector* bv);
void vlg_rrr_bitvector_destroy(vlg_rrr_bitvector* bv);

/* ------------------------------------------------------------------------------------------
 * K2 / K3 primitives on the index (device-resident arguments).
 * ---------------------------------------------------------------------------------------- */
/* wt_pc::rank(i, c) (include/sdsl/wt_pc.hpp:350-373): out[j] = #c[j] in BWT[0, i[j]). */
vlg_status vlg_wt_rank_batch(const vlg_index* idx, const uint64_t* d_i, const uint8_t* d_c, uint64_t* d_out,
                             uint64_t count, void* stream);
/* backward_search(csa, 0, n-1, pat.begin(), pat.end(), l, r)
 * (include/sdsl/suffix_array_algorithm.hpp:305-326): pattern p = d_blob[d_off[p], d_off[p+1]).
 * d_l/d_r receive the interval exactly as the reference leaves it; occurrences = r+1-l. */
vlg_status vlg_backward_search_batch(const vlg_index* idx, const uint8_t* d_blob, const uint64_t* d_off,
                                     uint64_t n_patterns, uint64_t* d_l, uint64_t* d_r, void* stream);
/* csa[i] (include/sdsl/csa_wt.hpp:335-348) for arbitrary SA indices: out[j] = SA[d_i[j]]. */
vlg_status vlg_sa_batch(const vlg_index* idx, const uint64_t* d_i, uint64_t* d_out, uint64_t count, void* stream);
/* locate (include/sdsl/suffix_array_algorithm.hpp:604-619): for pattern p the occurrences
 * csa[l[p]..r[p]] are written, in SA order (unsorted, like the reference), to
 * d_out[d_out_off[p] ...]; d_out_off = exclusive prefix sum of (r+1-l), n_patterns+1 entries. */
vlg_status vlg_locate_batch(const vlg_index* idx, const uint64_t* d_l, const uint64_t* d_r,
                            const uint64_t* d_out_off, uint64_t n_patterns, uint64_t total,
                            uint64_t* d_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Query batches.  Parsing mirrors the reference's two dialects:
 *   VLG_DIALECT_LIBRARY   gapped_pattern_query(const std::string&)   include/sdsl/vlg_index.hpp:54-105
 *   VLG_DIALECT_BENCHMARK gapped_pattern(const std::string&, true) + index_sasearch's gap mapping
 *                         benchmark/gapped-matching/include/utils.hpp:25-70, index_sasearch.hpp:68-69,113
 * ---------------------------------------------------------------------------------------- */
#define VLG_DIALECT_LIBRARY 0
#define VLG_DIALECT_BENCHMARK 1
#define VLG_MAX_SUBPATTERNS 64

/* Host-only view of one parsed query (no device involved): sub-pattern i = re[sub_off[i], sub_off[i]+sub_len[i]);
 * lo/hi[i] (i >= 1) = start-to-start distance bounds between sub-patterns i-1 and i; end_len = length added to the
 * last position for the non-overlap rule.  Returns VLG_E_PARSE / VLG_E_INVALID like vlg_queries_parse. */
typedef struct {
    uint32_t k;
    uint32_t reserved;
    uint64_t sub_off[VLG_MAX_SUBPATTERNS], sub_len[VLG_MAX_SUBPATTERNS];
    uint64_t lo[VLG_MAX_SUBPATTERNS], hi[VLG_MAX_SUBPATTERNS];
    uint64_t end_len;
} vlg_parsed_query;
vlg_status vlg_parse_query(const char* h_regexp, uint64_t len, int dialect, vlg_parsed_query* out);

/* Parse `n_queries` regexps (concatenated in h_text; query q = h_text[h_off[q], h_off[q+1])) and upload.
 * h_status (optional, n_queries ints) receives VLG_OK / VLG_E_PARSE / VLG_E_INVALID per query; an
 * unparsable query stays in the batch as an always-empty query (the reference driver skips such
 * lines, utils.hpp:94-99).  Returns VLG_E_PARSE if any query failed and h_status is NULL. */
vlg_status vlg_queries_parse(const char* h_text, const uint64_t* h_off, uint64_t n_queries, int dialect,
                             int* h_status, vlg_queries** out);
/* Already-parsed form: sub-pattern bytes + per-gap start-to-start bounds (lo,hi) + non-overlap length.
 * Query q has sub-patterns [h_qsub[q], h_qsub[q+1]); sub-pattern s = h_blob[h_suboff[s], h_suboff[s+1]);
 * h_lo/h_hi are indexed by sub-pattern (entry of a query's first sub-pattern is ignored). */
vlg_status vlg_queries_create(const uint8_t* h_blob, const uint64_t* h_suboff, const uint64_t* h_qsub,
                              const uint64_t* h_lo, const uint64_t* h_hi, const uint64_t* h_end_len,
                              uint64_t n_queries, vlg_queries** out);
/* sdsl::count(csa, sub-pattern) (include/sdsl/suffix_array_algorithm.hpp:516-529) for every sub-pattern of the batch: one
 * backward-search pass, h_occ[s] = r + 1 - l (vlg_queries_subpatterns(q) entries, in batch order).  What SURVEY.md 8(e) shards
 * a batch by: the sum over a query's sub-patterns estimates its locate + join work. */
vlg_status vlg_queries_occurrences(const vlg_index* idx, const vlg_queries* q, uint64_t* h_occ, void* stream);
/* The same pass, returning the SA interval [l, r] of every sub-pattern as backward_search leaves it (r + 1 - l occurrences;
 * suffix_array_algorithm.hpp:305-326).  Equal intervals are the same occurrence list: a host that shards a batch can keep the
 * queries that share their longest list on one GPU, so that the list is located once (vlg_matching_amd.dist.shard_by_affinity). */
vlg_status vlg_queries_intervals(const vlg_index* idx, const vlg_queries* q, uint64_t* h_l, uint64_t* h_r, void* stream);
uint64_t vlg_queries_count(const vlg_queries* q);
uint64_t vlg_queries_subpatterns(const vlg_queries* q);
/* h_k[q] = number of sub-patterns of query q (0 for a query that failed to parse). */
vlg_status vlg_queries_k(const vlg_queries* q, uint32_t* h_k);
void vlg_queries_destroy(vlg_queries* q);

/* ------------------------------------------------------------------------------------------
 * Workspace, fused search, results.
 * ---------------------------------------------------------------------------------------- */
/* max_hbm_bytes bounds the scratch used per chunk of queries (0 = default 8 GiB). */
vlg_status vlg_workspace_create(uint64_t max_hbm_bytes, void* stream, vlg_workspace** out);
void vlg_workspace_destroy(vlg_workspace* ws);

/* The whole hot path for a batch, everything resident in HBM:
 *   per sub-pattern backward_search -> locate -> sort ascending -> gap-bounded merge join
 * = `idx.search(pat)` for every pattern (benchmark/gapped-matching/src/gm_search.cpp:91-121,
 *   index_sasearch.hpp:58-118) and `sdsl::locate(idx, query)` (include/sdsl/vlg_index.hpp:395-401).
 * Matches are the left-most, lazy, non-overlapping tuples of SURVEY.md Appendix C, bit-exact. */
vlg_status vlg_search_batch(const vlg_index* idx, const vlg_queries* q, vlg_workspace* ws, vlg_result** out);

/* K4 stand-alone (std::sort of one occurrence list, benchmark/gapped-matching/include/index_sasearch.hpp:80): every list
 * d_pos[h_off[l], h_off[l + 1]) of 32-bit positions < 2^position_bits is sorted ascending in place by the per-list sort of the batch
 * path.  mode 1: as "list_sort" = 1 above, mode 2: as 2.  n_clustered (may be null) receives the number of long lists whose
 * positions were too clustered for the window pass (they took the four full passes).  Synchronises `stream` before it returns. */
vlg_status vlg_sort_lists_u32(uint32_t* d_pos, const uint64_t* h_off, uint64_t n_lists, uint32_t position_bits, int mode,
                              uint64_t* n_clustered, void* stream);
/* K5 on its own: the gap-bounded merge join of benchmark/gapped-matching/include/index_sasearch.hpp:85-116 (semantics of
 * vlg_iterator, include/sdsl/vlg_index.hpp:227-291) over caller-provided occurrence lists in HBM.
 *   d_lists      all lists concatenated, u64 positions, every list ascending (what std::sort leaves, index_sasearch.hpp:80);
 *                a position may not exceed 2^63
 *   h_list_off   [n_lists+1] element offsets of the lists inside d_lists
 *   h_join_list  [n_joins+1] join j uses the lists [h_join_list[j], h_join_list[j+1]) as its sub-patterns 0..k-1 (a list may
 *                not be shared between joins here: pass it twice); k <= VLG_MAX_SUBPATTERNS
 *   h_lo, h_hi   [n_lists] start-to-start distance bounds between a list and the previous one of its join (the entry of a
 *                join's first list is ignored); lo <= hi < 2^63
 *   h_end_len    [n_joins] length added to the last position of a match for the non-overlap rule (|s_{k-1}| for the library,
 *                |s_0| for index_sasearch.hpp:113)
 * The result is read like a search result: counts / offsets / first positions / tuples per join.  A join with an empty list
 * has no match (vlg_index.hpp:315-316).  Uses the workspace's window filter and chunking like vlg_search_batch. */
vlg_status vlg_join_batch(const uint64_t* d_lists, const uint64_t* h_list_off, uint64_t n_lists, const uint64_t* h_join_list,
                          const uint64_t* h_lo, const uint64_t* h_hi, const uint64_t* h_end_len, uint64_t n_joins,
                          vlg_workspace* ws, vlg_result** out);

typedef struct {
    uint64_t n_queries;
    uint64_t n_matches;           /* = gm_search's num_results (gm_search.cpp:110-114)          */
    uint64_t checksum;            /* = gm_search's checksum: sum of first positions mod 2^64    */
    uint64_t n_tuple_values;      /* sum over matches of k(query)                              */
    uint64_t located_occurrences; /* occurrences materialised by locate (each distinct SA interval of the
                                     batch is located once and shared by the queries that use it)     */
    uint64_t lf_steps;            /* LF steps taken by locate (csa_wt.hpp:338-341)              */
    uint64_t wt_levels_locate;    /* 32-byte super-block reads in locate                       */
    uint64_t wt_levels_bsearch;   /* 32-byte super-block reads in backward search              */
    uint64_t n_chunks;
    uint64_t logical_occurrences; /* sum of the list lengths the joins consumed (what the reference
                                     would locate query by query)                                 */
    uint64_t join_slots;          /* list elements the join passes evaluated: the non-final lists of every live
                                     query, after the window filter                               */
    uint64_t locate_mode;         /* how the last super-chunk's occurrences were located: VLG_LOCATE_*  */
} vlg_result_summary;
/* csa[i] = iterated LF to a sampled index (csa_wt.hpp:335-348), organised per batch as
 *   WALKS     one walk per occurrence (few occurrences; integer alphabets; workspace option "sweep" = 0)
 *   SWEEP     all occurrences advance together in SA order, walks that meet share their LF steps ("trail" = 1)
 *   UNSAMPLE  a batch that locates a large part of all text positions ("unsample_pct" per cent, default 40): one walker per SA
 *             sample rebuilds the whole suffix array -- n LF steps whatever the batch, each SA index visited once -- and the
 *             lists are copied out of it; recomputed for every batch, nothing is kept
 *   COPY      an index that keeps every SA value (sa_sample_dens = 1): no LF step at all                               */
enum { VLG_LOCATE_NONE = 0, VLG_LOCATE_WALKS = 1, VLG_LOCATE_SWEEP = 2, VLG_LOCATE_UNSAMPLE = 3, VLG_LOCATE_COPY = 4 };

vlg_status vlg_result_summary_get(const vlg_result* r, vlg_result_summary* s);
/* Copy to host.  counts[q] = matches of query q; offsets = exclusive prefix sum (n_queries+1);
 * first_positions = gapped_search_result::positions of every query, concatenated (utils.hpp:73-80);
 * tuples = every sub-pattern position of every match (vlg_iterator::operator[], vlg_index.hpp:337-340),
 * query-major, k(q) values per match.  Any pointer may be NULL. */
vlg_status vlg_result_fetch(const vlg_result* r, uint64_t* h_counts, uint64_t* h_offsets,
                            uint64_t* h_first_positions, uint64_t* h_tuples);
/* The same positions as 32-bit values, for callers that keep them that way: results of a text whose positions fit 32 bits
 * (vlg_index_info.n <= 2^32 + 1) are held 4 bytes wide in HBM and cross PCIe that way -- vlg_result_fetch widens them on the
 * host (threads, pinned staging blocks), this entry point copies them as they are.  VLG_E_INVALID when the result's positions
 * are 64 bits wide (larger texts, VLG_FORCE_POS64=1, vlg_wtsa_* results).  Either pointer may be NULL. */
vlg_status vlg_result_fetch32(const vlg_result* r, uint32_t* h_first_positions, uint32_t* h_tuples);
void vlg_result_destroy(vlg_result* r);

/* ------------------------------------------------------------------------------------------
 * The paper's index: vlg_index<alphabet_tag, wt_int<>> -- the text plus a wavelet tree over its suffix
 * array (include/sdsl/vlg_index.hpp:109-198, construct :375-392), searched lazily by vlg_iterator (:209-373): no occurrence list
 * is ever located or sorted; a match costs a few root-to-leaf walks of the tree, so the first matches of a query come cheaply
 * however long its lists are.  Byte texts (byte_alphabet_tag) and integer texts (int_alphabet_tag: symbols are uint32_t, a
 * query's sub-patterns whitespace-separated decimals, vlg_index.hpp:63-69).
 * On the device the tree is level-contiguous: level l is one rank-enabled bit-vector of 256-bit super-blocks (as K1), nodes are
 * intervals of it (wt_int, include/sdsl/wt_int.hpp:215-255).  Results equal sdsl::locate(vlg_index, query) -- the same tuples as
 * vlg_search_batch on the FM-index -- cut after max_matches_per_query matches of each query when that is not 0 (what a caller
 * that stops iterating early sees, include/sdsl/vlg_index.hpp:357-363).  One corner is defined away: for a query of one
 * sub-pattern of one symbol the reference's iterator drops an occurrence at 2j + 1 that directly follows a match at 2j
 * (vlg_index.hpp:254-266 with wt_helper.hpp:776-779); like the benchmark's merge join this library reports both.
 * ---------------------------------------------------------------------------------------- */
typedef struct vlg_wtsa vlg_wtsa;
typedef struct {
    uint64_t n;                   /* symbols + 1 (sentinel) = size of the suffix array                */
    uint32_t symbol_bytes;        /* 1 or 4                                                           */
    uint32_t levels;              /* bits per suffix-array value = levels of the tree                 */
    uint64_t blocks_per_level;    /* 32-byte super-blocks per level                                   */
    uint64_t hbm_bytes;           /* text + tree                                                      */
} vlg_wtsa_info;
/* symbol_bytes 1: h_text is uint8_t[n_symbols] without a 0 byte (VLG_E_ZERO_BYTE); 4: uint32_t[n_symbols], any values. */
vlg_status vlg_wtsa_build(const void* h_text, uint64_t n_symbols, uint32_t symbol_bytes, vlg_wtsa** out);
vlg_status vlg_wtsa_get_info(const vlg_wtsa* idx, vlg_wtsa_info* info);
void vlg_wtsa_destroy(vlg_wtsa* idx);
/* wt[i] (wt_int::operator[], include/sdsl/wt_int.hpp:339-361) = SA[i] for arbitrary indices. */
vlg_status vlg_wtsa_sa_batch(const vlg_wtsa* idx, const uint64_t* d_i, uint64_t* d_out, uint64_t count, void* stream);
/* The two walks every search of this index is made of, on arbitrary suffix-array ranges [l, l + len): what the reference gets from
 * wt_int::expand(v) / expand(v, range) as vlg_iterator's wt_range_walker descends (include/sdsl/wt_int.hpp:824-939,
 * wt_helper.hpp:726-785).  quantile == 0: out[j] = number of SA[l, l + len) smaller than x[j]; quantile != 0: out[j] = the x[j]-th
 * smallest of them (0-based, x[j] < len[j]).  A range outside the suffix array (or x[j] >= len[j]) gives ~0. */
vlg_status vlg_wtsa_range_walk_batch(const vlg_wtsa* idx, const uint64_t* d_l, const uint64_t* d_len, const uint64_t* d_x, int quantile,
                                     uint64_t* d_out, uint64_t count, void* stream);
/* Level `level` of the tree as plain words (bit i = h_words[i >> 6] >> (i & 63), ceil(n / 64) words): bits [level * n, (level + 1) * n)
 * of wt_int::tree (include/sdsl/wt_int.hpp:162, 215-255) -- what m_wt.serialize stores (vlg_index.hpp:181-190). */
vlg_status vlg_wtsa_export_level(const vlg_wtsa* idx, uint32_t level, uint64_t* h_words);
/* forward_search(text.begin(), text.end(), wt, 0, wt.size()-1, pat.begin(), pat.end(), sp, ep)
 * (include/sdsl/suffix_array_algorithm.hpp:48-112) for every sub-pattern of the batch: h_sp/h_ep receive the suffix-array
 * range [sp, ep] (sp = ep + 1: no occurrence). */
vlg_status vlg_wtsa_ranges(const vlg_wtsa* idx, const vlg_queries* q, uint64_t* h_sp, uint64_t* h_ep, void* stream);
/* Integer-alphabet query batch: like vlg_queries_parse with VLG_DIALECT_LIBRARY, sub-patterns parsed as the reference parses them
 * for int_alphabet_tag (whitespace-separated decimals; gaps count symbols).  Only vlg_wtsa_* entry points accept such a batch. */
vlg_status vlg_queries_parse_int(const char* h_text, const uint64_t* h_off, uint64_t n_queries, int* h_status, vlg_queries** out);
/* the same through a symbol map (above): tokens are the ORIGINAL 64-bit symbols, the batch holds their mapped values */
vlg_status vlg_queries_parse_int_mapped(const vlg_symbol_map* map, const char* h_text, const uint64_t* h_off, uint64_t n_queries,
                                        int* h_status, vlg_queries** out);
/* sdsl::locate / count on the batch, at most max_matches_per_query matches each (0 = all). */
vlg_status vlg_wtsa_search_batch(const vlg_wtsa* idx, const vlg_queries* q, uint64_t max_matches_per_query, vlg_workspace* ws,
                                 vlg_result** out);

/* Per-kernel accounting of the last calls on this workspace (HIP events on the workspace stream).
 * Enabled with vlg_workspace_profile(ws, 1); reset by vlg_workspace_profile(ws, 1) again. */
typedef struct {
    char name[32];                /* "backward_search", "locate", "sort", "join", ...           */
    uint64_t launches;
    double total_ms;              /* sum of event-timed launch durations                       */
    uint64_t algorithmic_bytes;   /* SURVEY.md 8(d) accounting, 0 if not defined for the kernel */
} vlg_kernel_stat;
vlg_status vlg_workspace_profile(vlg_workspace* ws, int enable);
/* Options: "dedup" (default 1): share one located+sorted list between all sub-patterns of a batch that
 * have the same SA interval; 0 = locate every sub-pattern of every query separately like the reference.
 * "sweep" (default 1): locate by the synchronous sorted LF sweep (coalesced super-block reads) when the batch has
 * at least "sweep_min" occurrences (default 2^22); the last "sweep_tail" (default 2^22) stragglers and smaller
 * batches use the one-lane-per-occurrence random-access kernel.
 * "trail" (default 1, needs "dedup"): inside a sorted sweep an occurrence that steps onto an SA index another occurrence
 * has visited stops there and takes that occurrence's position plus the distance (csa[i] = csa[LF(i)] + 1 shared between
 * lanes), so a batch walks every LF trail once; 0 = every occurrence walks to its own sample like csa_wt::operator[].
 * "list_sort" (default 1): with 32-bit positions every occurrence list is sorted inside itself (short lists in LDS by value ranges;
 * long ones by two LSD radix passes over the top 16 position bits + one LDS pass over windows of whole groups; 2: four LSD passes
 * over all position bits, what lists with clustered positions take anyway); 0, or 64-bit positions: "global_sort_min" (default 2^20): from this many
 * occurrences on all lists are sorted by one radix sort of (list, position) keys instead of one segmented sort.
 * "filter" (default 1): before the join drop the list elements whose gap windows hold no element of the neighbouring
 * lists (they are in no match); "filter_min" (default 2^12) = join slots below which a query is joined as it is
 * ("filter_stream_min", default 2^16, for the queries filtered by streaming sweeps);
 * "filter_pivot" (default 1): filter outwards from the shortest list of a query when it is >= 6x shorter than all of
 * them together ("filter_pivot_ratio", default 6), otherwise (or with 0) by streaming sweeps over block bitmaps;
 * "filter_group_bytes" (default 0 = a third of the join scratch) caps the filter state of the queries filtered together.
 * "reserve" = bytes of scratch to allocate right away (at most the workspace's cap) instead of on first use.
 * "tuples" (default 1): materialise every sub-pattern position of every match (what sdsl::locate returns); 0 = first
 * positions only, which is all the benchmark's gapped_search_result holds (index_sasearch.hpp:58-118): n_tuple_values
 * is 0 and vlg_result_fetch refuses a tuples buffer.
 * Counts, first positions, tuples and the checksum are identical whatever the other options are. */
vlg_status vlg_workspace_set_option(vlg_workspace* ws, const char* name, int64_t value);
vlg_status vlg_workspace_kernel_stats(vlg_workspace* ws, vlg_kernel_stat* out, uint32_t cap, uint32_t* n);

/* ------------------------------------------------------------------------------------------
 * Collective search (one process per GPU): ONE batch answered by all ranks of a communicator.
 * Every rank passes the same index image and the same query batch to vlg_search_batch, with a workspace of the same cap.  A batch
 * is cheap on one GPU because equal SA intervals are located once (vlg_result_summary.located_occurrences << logical_occurrences);
 * sharding only the query loop of gm_search.cpp:91-121 makes every rank locate the frequent lists again.  Here the DISTINCT LISTS are
 * sharded for locate + sort (a contiguous share of the lists in SA order per rank, equal occurrences), the sorted lists are exchanged
 * (one in-place all-gather of positions: the exchange step of this path), and the QUERIES are sharded for filter + join (contiguous
 * pieces of equal join work).  The result holds this rank's queries (vlg_result_owned_queries; counts of the others are 0); counts,
 * checksums and located occurrences are summed over the ranks by the caller (vlg_comm_allreduce_sum_u64).
 * vlg_workspace_set_comm: exchange by RCCL (vlg_comm_allgatherv); NULL = back to single-GPU searches.
 * vlg_workspace_set_exchange: the caller moves the bytes -- `fn` must all-gather IN PLACE: d_buf holds this rank's piece at the offset
 * of the pieces before it (h_counts[r] elements of elem_bytes per rank r), afterwards every piece; work it enqueues must be ordered
 * with `stream`.  (Hosts without RCCL between their ranks; rehearsals with several ranks on one device.) */
typedef int (*vlg_exchange_fn)(void* ctx, void* d_buf, const uint64_t* h_counts, uint32_t elem_bytes, int n_ranks, int rank, void* stream);
vlg_status vlg_workspace_set_comm(vlg_workspace* ws, void* nccl_comm);
vlg_status vlg_workspace_set_exchange(vlg_workspace* ws, int n_ranks, int rank, vlg_exchange_fn fn, void* ctx);
/* The exchange step as it runs by default over a communicator: PAIRWISE and NEEDED-ONLY.  Every rank knows the whole plan of the
 * batch (which rank sorts which list, which rank joins which queries), so a sorted list travels only to the ranks whose queries
 * use it: the lists a rank owes a peer are packed into one buffer per peer and sent point to point -- vlg_comm_alltoallv, one
 * grouped ncclSend / ncclRecv pair per peer, one xGMI link each -- instead of every list visiting every rank in a ring.  Workspace
 * option "exchange_all" = 1 goes back to the in-place all-gather of everything (vlg_comm_allgatherv).
 * vlg_workspace_set_exchange_alltoall: the same with the caller moving the bytes (hosts without RCCL between their ranks,
 * rehearsals on one device): fn sends h_send_counts[r] elements to rank r (packed in rank order at d_send) and receives
 * h_recv_counts[r] from it (packed at d_recv), its own piece included, ordered with `stream`.
 * Before the payload moves, the ranks all-gather ONE status word each: a rank whose part of the batch failed up to there (workspace
 * too small for its share, out of memory, a failed kernel) tells the others, and vlg_search_batch returns an error on EVERY rank --
 * its own on the failed one, VLG_E_INTERNAL "rank r failed ..." on its peers -- instead of leaving them inside a collective.
 * An error after the exchange (in a rank's joins) is that rank's alone: the product issues no further collective; a caller that
 * reduces counters afterwards (vlg_comm_allreduce_sum_u64) has to handle it like any failure of one of its own ranks. */
vlg_status vlg_comm_alltoallv(void* nccl_comm, const void* d_send, const uint64_t* h_send_counts, void* d_recv,
                              const uint64_t* h_recv_counts, uint32_t elem_bytes, void* stream);
typedef int (*vlg_alltoall_fn)(void* ctx, const void* d_send, const uint64_t* h_send_counts, void* d_recv, const uint64_t* h_recv_counts,
                               uint32_t elem_bytes, int n_ranks, int rank, void* stream);
vlg_status vlg_workspace_set_exchange_alltoall(vlg_workspace* ws, int n_ranks, int rank, vlg_alltoall_fn fn, void* ctx);
/* [begin, end) pairs of the queries this rank joined in a collective search (n_ranges = 0: a single-GPU search, all of them) */
vlg_status vlg_result_owned_queries(const vlg_result* r, uint64_t* h_ranges, uint32_t cap_ranges, uint32_t* n_ranges);

#ifdef __cplusplus
}
#endif
#endif /* VLG_HIP_H */
