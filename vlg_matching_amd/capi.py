"""ctypes binding of include/vlg_hip.h (the C-ABI drop-in boundary)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

LOCATE_NONE, LOCATE_WALKS, LOCATE_SWEEP, LOCATE_UNSAMPLE, LOCATE_COPY = range(5)
OK, E_INVALID, E_NO_DEVICE, E_OOM, E_PARSE, E_ZERO_BYTE, E_UNSUPPORTED, E_WORKSPACE, E_INTERNAL = range(9)
DIALECT_LIBRARY, DIALECT_BENCHMARK = 0, 1


class VlgError(RuntimeError):
    def __init__(self, status, text):
        self.status = status
        super().__init__("vlg status %d: %s" % (status, text))


class WtNode(C.Structure):
    _fields_ = [("bv_pos", C.c_uint64), ("bv_pos_rank", C.c_uint64), ("parent", C.c_uint16), ("child", C.c_uint16 * 2)]


class IndexParts(C.Structure):
    _fields_ = [("n", C.c_uint64), ("sigma", C.c_uint32), ("sa_sample_dens", C.c_uint32), ("char2comp", C.c_void_p),
                ("C", C.c_void_p), ("bv_words", C.c_void_p), ("bv_bits", C.c_uint64), ("nodes", C.c_void_p),
                ("n_nodes", C.c_uint32), ("sa_samples", C.c_void_p), ("n_samples", C.c_uint64)]


class IndexPartsOut(C.Structure):
    _fields_ = [("char2comp", C.c_void_p), ("C", C.c_void_p), ("bv_words", C.c_void_p), ("nodes", C.c_void_p),
                ("sa_samples", C.c_void_p)]


class IndexInfo(C.Structure):
    _fields_ = [("n", C.c_uint64), ("sigma", C.c_uint32), ("sa_sample_dens", C.c_uint32), ("n_nodes", C.c_uint32),
                ("max_code_len", C.c_uint32), ("wt_bits", C.c_uint64), ("n_blocks", C.c_uint64), ("n_samples", C.c_uint64),
                ("hbm_bytes", C.c_uint64), ("pos_bytes", C.c_uint32), ("bv_kind", C.c_uint32), ("sampling", C.c_uint32), ("reserved", C.c_uint32)]


class ResultSummary(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_queries", "n_matches", "checksum", "n_tuple_values", "located_occurrences",
                                           "lf_steps", "wt_levels_locate", "wt_levels_bsearch", "n_chunks",
                                           "logical_occurrences", "join_slots", "locate_mode")]


class ParsedQuery(C.Structure):
    _fields_ = [("k", C.c_uint32), ("reserved", C.c_uint32), ("sub_off", C.c_uint64 * 64), ("sub_len", C.c_uint64 * 64),
                ("lo", C.c_uint64 * 64), ("hi", C.c_uint64 * 64), ("end_len", C.c_uint64)]


class WtsaInfo(C.Structure):
    _fields_ = [("n", C.c_uint64), ("symbol_bytes", C.c_uint32), ("levels", C.c_uint32), ("blocks_per_level", C.c_uint64),
                ("hbm_bytes", C.c_uint64)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint64), ("total_ms", C.c_double), ("algorithmic_bytes", C.c_uint64)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_int, C.c_int, C.c_void_p)
ALLTOALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_int, C.c_int,
                          C.c_void_p)

# every symbol include/vlg_hip.h declares: (name, restype, argtypes)
_P, _U64, _I = C.c_void_p, C.c_uint64, C.c_int
SYMBOLS = [
    ("vlg_last_error", C.c_char_p, []),
    ("vlg_version", C.c_char_p, []),
    ("vlg_device_count", _I, [C.POINTER(C.c_int)]),
    ("vlg_set_device", _I, [_I]),
    ("vlg_index_build", _I, [_P, _U64, C.c_uint32, C.POINTER(_P)]),
    ("vlg_index_build_device", _I, [_P, _U64, C.c_uint32, _P, C.POINTER(_P)]),
    ("vlg_index_from_parts", _I, [C.POINTER(IndexParts), C.POINTER(_P)]),
    ("vlg_index_export_parts", _I, [_P, C.POINTER(IndexParts), C.POINTER(IndexPartsOut)]),
    ("vlg_index_compress", _I, [_P, _I, C.POINTER(_P)]),
    ("vlg_index_resample", _I, [_P, _I, C.c_uint32, C.POINTER(_P)]),
    ("vlg_index_export_marked", _I, [_P, _P]),
    ("vlg_index_get_info", _I, [_P, C.POINTER(IndexInfo)]),
    ("vlg_index_build_int", _I, [_P, _U64, C.c_uint32, C.POINTER(_P)]),
    ("vlg_index_export_int_alphabet", _I, [_P, C.POINTER(_U64), _P, _P]),
    ("vlg_int_rank_batch", _I, [_P, _P, _P, _P, _U64, _P]),
    ("vlg_index_destroy", None, [_P]),
    ("vlg_sdsl_file_open", _I, [C.c_char_p, C.c_uint32, C.POINTER(_P)]),
    ("vlg_sdsl_file_parts", _I, [_P, C.POINTER(IndexParts)]),
    ("vlg_sdsl_file_close", None, [_P]),
    ("vlg_index_load_sdsl", _I, [C.c_char_p, C.c_uint32, C.POINTER(_P)]),
    ("vlg_sdsl_file_open_kind", _I, [C.c_char_p, C.c_uint32, _I, C.POINTER(_P)]),
    ("vlg_index_load_sdsl_kind", _I, [C.c_char_p, C.c_uint32, _I, C.POINTER(_P)]),
    ("vlg_index_save_sdsl", _I, [_P, C.c_char_p]),
    ("vlg_suffix_array_device", _I, [_P, _U64, _P, _P]),
    ("vlg_index_isa_samples", _I, [_P, C.c_uint32, _P, _U64]),
    ("vlg_index_blob_bytes", _I, [_P, C.POINTER(_U64)]),
    ("vlg_index_blob_export", _I, [_P, _P, _U64, _P]),
    ("vlg_index_attach_blob", _I, [_P, _U64, C.POINTER(_P)]),
    ("vlg_index_replicate", _I, [_P, _I, C.POINTER(_P)]),
    ("vlg_comm_library", C.c_char_p, []),
    ("vlg_comm_unique_id", _I, [_P]),
    ("vlg_comm_create", _I, [_P, _I, _I, C.POINTER(_P)]),
    ("vlg_comm_info", _I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("vlg_comm_destroy", None, [_P]),
    ("vlg_index_broadcast", _I, [_P, _P, _I, _P, C.POINTER(_P)]),
    ("vlg_comm_allreduce_sum_u64", _I, [_P, _P, C.c_uint32, _P]),
    ("vlg_comm_allgatherv", _I, [_P, _P, _P, C.c_uint32, _P, _P]),
    ("vlg_comm_alltoallv", _I, [_P, _P, _P, _P, _P, C.c_uint32, _P]),
    ("vlg_bitvector_create", _I, [_P, _U64, C.POINTER(_P)]),
    ("vlg_bitvector_rank_batch", _I, [_P, _P, _P, _U64, _P]),
    ("vlg_bitvector_hbm_bytes", _U64, [_P]),
    ("vlg_bitvector_destroy", None, [_P]),
    ("vlg_rrr_bitvector_create", _I, [_P, _U64, C.POINTER(_P)]),
    ("vlg_rrr_bitvector_rank_batch", _I, [_P, _P, _P, _U64, _P]),
    ("vlg_rrr_bitvector_hbm_bytes", _U64, [_P]),
    ("vlg_rrr_bitvector_destroy", None, [_P]),
    ("vlg_wt_rank_batch", _I, [_P, _P, _P, _P, _U64, _P]),
    ("vlg_backward_search_batch", _I, [_P, _P, _P, _U64, _P, _P, _P]),
    ("vlg_sa_batch", _I, [_P, _P, _P, _U64, _P]),
    ("vlg_locate_batch", _I, [_P, _P, _P, _P, _U64, _U64, _P, _P]),
    ("vlg_parse_query", _I, [C.c_char_p, _U64, _I, C.POINTER(ParsedQuery)]),
    ("vlg_queries_parse", _I, [C.c_char_p, _P, _U64, _I, _P, C.POINTER(_P)]),
    ("vlg_queries_create", _I, [_P, _P, _P, _P, _P, _P, _U64, C.POINTER(_P)]),
    ("vlg_queries_occurrences", _I, [_P, _P, _P, _P]),
    ("vlg_queries_intervals", _I, [_P, _P, _P, _P, _P]),
    ("vlg_queries_count", _U64, [_P]),
    ("vlg_queries_subpatterns", _U64, [_P]),
    ("vlg_queries_k", _I, [_P, _P]),
    ("vlg_queries_destroy", None, [_P]),
    ("vlg_workspace_create", _I, [_U64, _P, C.POINTER(_P)]),
    ("vlg_workspace_destroy", None, [_P]),
    ("vlg_search_batch", _I, [_P, _P, _P, C.POINTER(_P)]),
    ("vlg_join_batch", _I, [_P, _P, _U64, _P, _P, _P, _P, _U64, _P, C.POINTER(_P)]),
    ("vlg_sort_lists_u32", _I, [_P, _P, _U64, C.c_uint32, _I, C.POINTER(_U64), _P]),
    ("vlg_result_summary_get", _I, [_P, C.POINTER(ResultSummary)]),
    ("vlg_result_fetch", _I, [_P, _P, _P, _P, _P]),
    ("vlg_result_fetch32", _I, [_P, _P, _P]),
    ("vlg_result_destroy", None, [_P]),
    ("vlg_wtsa_build", _I, [_P, _U64, C.c_uint32, C.POINTER(_P)]),
    ("vlg_wtsa_get_info", _I, [_P, C.POINTER(WtsaInfo)]),
    ("vlg_wtsa_destroy", None, [_P]),
    ("vlg_wtsa_sa_batch", _I, [_P, _P, _P, _U64, _P]),
    ("vlg_wtsa_ranges", _I, [_P, _P, _P, _P, _P]),
    ("vlg_wtsa_range_walk_batch", _I, [_P, _P, _P, _P, _I, _P, _U64, _P]),
    ("vlg_wtsa_export_level", _I, [_P, C.c_uint32, _P]),
    ("vlg_queries_parse_int", _I, [C.c_char_p, _P, _U64, _P, C.POINTER(_P)]),
    ("vlg_queries_parse_int_mapped", _I, [_P, _P, _P, _U64, _P, C.POINTER(_P)]),
    ("vlg_symbol_map_create", _I, [_P, _U64, C.POINTER(_P)]),
    ("vlg_symbol_map_sigma", _U64, [_P]),
    ("vlg_symbol_map_symbols", _I, [_P, _P]),
    ("vlg_symbol_map_apply", _I, [_P, _P, _U64, _P]),
    ("vlg_symbol_map_destroy", None, [_P]),
    ("vlg_wtsa_search_batch", _I, [_P, _P, _U64, _P, C.POINTER(_P)]),
    ("vlg_workspace_profile", _I, [_P, _I]),
    ("vlg_workspace_set_option", _I, [_P, C.c_char_p, C.c_int64]),
    ("vlg_workspace_kernel_stats", _I, [_P, C.POINTER(KernelStat), C.c_uint32, C.POINTER(C.c_uint32)]),
    ("vlg_workspace_set_comm", _I, [_P, _P]),
    ("vlg_workspace_set_exchange", _I, [_P, _I, _I, _P, _P]),
    ("vlg_workspace_set_exchange_alltoall", _I, [_P, _I, _I, _P, _P]),
    ("vlg_result_owned_queries", _I, [_P, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
]


def library_path():
    """The in-tree library; VLG_HIP_LIBRARY names another build of it (kernel experiments)."""
    return os.environ.get("VLG_HIP_LIBRARY") or os.path.join(_HERE, "libvlg_hip.so")


def build_library(jobs=4):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j%d" % jobs], stdout=subprocess.DEVNULL)


def lib():
    """The loaded C-ABI library.  Fails loudly when it is missing: there is no CPU fallback."""
    global _LIB
    if _LIB is None:
        p = library_path()
        if not os.path.exists(p):
            raise VlgError(E_NO_DEVICE, "HIP extension %s is not built (run __graft_entry__.build())" % p)
        L = C.CDLL(p)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def check(status):
    if status != OK:
        raise VlgError(status, (lib().vlg_last_error() or b"").decode("utf-8", "replace"))
