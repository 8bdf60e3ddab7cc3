"""ctypes access to the TEST-ONLY checkers (oracle/libvlgoracle.so and oracle/_ref/libvlgref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (vlg_matching_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)
MAX_SUB = 64


class Node(C.Structure):
    _fields_ = [("bv_pos", C.c_uint64), ("bv_pos_rank", C.c_uint64),
                ("parent", C.c_uint16), ("child", C.c_uint16 * 2)]


class Query(C.Structure):
    _fields_ = [("k", C.c_uint32), ("sub", C.c_void_p * MAX_SUB), ("sub_len", C.c_uint64 * MAX_SUB),
                ("lo", C.c_uint64 * MAX_SUB), ("hi", C.c_uint64 * MAX_SUB), ("end_len", C.c_uint64)]


def build(force=False):
    """Compile the C restatement (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "libvlgoracle.so")
    src = [os.path.join(_HERE, f) for f in ("vlg_oracle.c", "vlg_oracle_int.c", "vlg_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "libvlgoracle.so", "-B"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libvlgref.so")
    if os.path.isdir("/root/reference/include/sdsl"):
        glue = os.path.join(_HERE, "ref_glue.cpp")
        if force or not os.path.exists(ref_so) or os.path.getmtime(glue) > os.path.getmtime(ref_so):
            subprocess.check_call(["make", "-C", _HERE, "_ref/libvlgref.so", "-B"], stdout=subprocess.DEVNULL)


def _np_u8(a):
    return np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a,
                                dtype=np.uint8)


def lib():
    global _LIB
    if _LIB is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "libvlgoracle.so"))
        L.vlgo_suffix_array.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.vlgo_suffix_array.restype = C.c_int
        L.vlgo_build.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.vlgo_build.restype = C.c_void_p
        L.vlgo_build_from_bwt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
        L.vlgo_build_from_bwt.restype = C.c_void_p
        L.vlgo_from_parts.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                      C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32]
        L.vlgo_from_parts.restype = C.c_void_p
        L.vlgo_free.argtypes = [C.c_void_p]
        for name, res in [("vlgo_size", C.c_uint64), ("vlgo_sigma", C.c_uint32), ("vlgo_char2comp", u8p),
                          ("vlgo_C", u64p), ("vlgo_bv_bits", C.c_uint64), ("vlgo_bv_words", u64p),
                          ("vlgo_n_nodes", C.c_uint32), ("vlgo_nodes", C.POINTER(Node)), ("vlgo_paths", u64p),
                          ("vlgo_n_samples", C.c_uint64), ("vlgo_bwt", u8p)]:
            f = getattr(L, name)
            f.argtypes = [C.c_void_p]
            f.restype = res
        L.vlgo_sample.argtypes = [C.c_void_p, C.c_uint64]
        L.vlgo_sample.restype = C.c_uint64
        L.vlgo_rank_blocks.argtypes = [C.c_void_p, u64p]
        L.vlgo_rank_blocks.restype = u64p
        L.vlgo_bv_rank1.argtypes = [C.c_void_p, C.c_uint64]
        L.vlgo_bv_rank1.restype = C.c_uint64
        L.vlgo_wt_rank.argtypes = [C.c_void_p, C.c_uint64, C.c_uint8]
        L.vlgo_wt_rank.restype = C.c_uint64
        L.vlgo_inverse_select.argtypes = [C.c_void_p, C.c_uint64, u8p]
        L.vlgo_inverse_select.restype = C.c_uint64
        L.vlgo_lf.argtypes = [C.c_void_p, C.c_uint64]
        L.vlgo_lf.restype = C.c_uint64
        L.vlgo_sa.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p]
        L.vlgo_sa.restype = C.c_uint64
        L.vlgo_backward_search.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, u64p, u64p]
        L.vlgo_backward_search.restype = C.c_uint64
        L.vlgo_locate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.vlgo_locate.restype = C.c_uint64
        for nm in ("vlgo_rank_v_build", "vlgo_rank_v5_build"):
            getattr(L, nm).argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
            getattr(L, nm).restype = C.c_uint64
        for nm in ("vlgo_rank_v", "vlgo_rank_v5"):
            getattr(L, nm).argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
            getattr(L, nm).restype = C.c_uint64
        L.vlgo_rrr_build.argtypes = [C.c_void_p, C.c_uint64]
        L.vlgo_rrr_build.restype = C.c_void_p
        L.vlgo_rrr_rank.argtypes = [C.c_void_p, C.c_uint64]
        L.vlgo_rrr_rank.restype = C.c_uint64
        L.vlgo_rrr_bits.argtypes = [C.c_void_p]
        L.vlgo_rrr_bits.restype = C.c_uint64
        L.vlgo_rrr_free.argtypes = [C.c_void_p]
        L.vlgo_parse.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(Query)]
        L.vlgo_parse.restype = C.c_int
        L.vlgo_join.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                C.c_void_p, C.c_uint64]
        L.vlgo_join.restype = C.c_uint64
        L.vlgo_search.argtypes = [C.c_void_p, C.POINTER(Query), C.c_void_p, C.c_uint64, C.c_void_p]
        L.vlgo_search.restype = C.c_uint64
        L.vlgo_sasearch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(Query), C.c_void_p, C.c_uint64, C.c_void_p]
        L.vlgo_sasearch.restype = C.c_uint64
        L.vlgo_search_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_double, C.c_void_p, C.c_void_p]
        L.vlgo_search_many.restype = C.c_uint64
        L.vlgo_sasearch_many.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_double,
                                         C.c_void_p, C.c_void_p]
        L.vlgo_sasearch_many.restype = C.c_uint64
        L.vlgo_sa_forward_search.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.vlgo_sa_forward_search.restype = C.c_uint64
        # integer alphabets / text-order sampling (vlg_oracle_int.c)
        P, U = C.c_void_p, C.c_uint64
        L.vlgo_int_suffix_array.argtypes = [P, U, P]
        L.vlgo_int_build.argtypes = [P, U, C.c_uint32, C.c_int]
        L.vlgo_int_build.restype = P
        L.vlgo_int_free.argtypes = [P]
        for nm in ("vlgo_int_size", "vlgo_int_sigma", "vlgo_int_n_samples"):
            getattr(L, nm).argtypes = [P]
            getattr(L, nm).restype = U
        L.vlgo_int_levels.argtypes = [P]
        L.vlgo_int_levels.restype = C.c_uint32
        for nm in ("vlgo_int_C", "vlgo_int_comp2char", "vlgo_int_tree", "vlgo_int_bwt", "vlgo_int_samples", "vlgo_int_marked"):
            getattr(L, nm).argtypes = [P]
            getattr(L, nm).restype = C.POINTER(C.c_uint64)
        L.vlgo_int_char2comp.argtypes = [P, U]
        L.vlgo_int_char2comp.restype = U
        L.vlgo_int_rank.argtypes = [P, U, U]
        L.vlgo_int_rank.restype = U
        L.vlgo_int_inverse_select.argtypes = [P, U, C.POINTER(U)]
        L.vlgo_int_inverse_select.restype = U
        L.vlgo_int_lf.argtypes = [P, U]
        L.vlgo_int_lf.restype = U
        L.vlgo_int_sa.argtypes = [P, U, C.POINTER(U)]
        L.vlgo_int_sa.restype = U
        L.vlgo_int_backward_search.argtypes = [P, P, U, C.POINTER(U), C.POINTER(U)]
        L.vlgo_int_backward_search.restype = U
        L.vlgo_int_locate.argtypes = [P, P, U, P, U]
        L.vlgo_int_locate.restype = U
        L.vlgo_parse_int.argtypes = [C.c_char_p, U, C.POINTER(Query), P, U, P]
        L.vlgo_parse_int.restype = C.c_int
        L.vlgo_int_search.argtypes = [P, C.POINTER(Query), P, P, P, U, P]
        L.vlgo_int_search.restype = U
        L.vlgo_text_order_build.argtypes = [P, C.c_uint32]
        L.vlgo_text_order_build.restype = P
        L.vlgo_text_order_free.argtypes = [P]
        for nm in ("vlgo_text_order_marked", "vlgo_text_order_samples"):
            getattr(L, nm).argtypes = [P]
            getattr(L, nm).restype = C.POINTER(C.c_uint64)
        L.vlgo_text_order_n_samples.argtypes = [P]
        L.vlgo_text_order_n_samples.restype = U
        L.vlgo_text_order_sa.argtypes = [P, P, U, C.POINTER(U)]
        L.vlgo_text_order_sa.restype = U
        _LIB = L
    return _LIB


def ref():
    """The reference's own code (oracle/_ref); None when it was never built."""
    global _REF
    if _REF is None:
        build()
        p = os.path.join(_HERE, "_ref", "libvlgref.so")
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.vref_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        R.vref_create.restype = C.c_void_p
        R.vref_destroy.argtypes = [C.c_void_p]
        for nm in ("vref_size", "vref_bv_size", "vref_n_nodes"):
            getattr(R, nm).argtypes = [C.c_void_p]
            getattr(R, nm).restype = C.c_uint64
        R.vref_wt_rank.argtypes = [C.c_void_p, C.c_uint64, C.c_uint8]
        R.vref_wt_rank.restype = C.c_uint64
        R.vref_inverse_select.argtypes = [C.c_void_p, C.c_uint64, u8p]
        R.vref_inverse_select.restype = C.c_uint64
        R.vref_bv_rank1.argtypes = [C.c_void_p, C.c_uint64]
        R.vref_bv_rank1.restype = C.c_uint64
        R.vref_bv_bits.argtypes = [C.c_void_p, C.c_void_p]
        R.vref_node.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                C.POINTER(C.c_int), C.POINTER(C.c_int)]
        R.vref_alphabet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        R.vref_sa.argtypes = [C.c_void_p, C.c_uint64]
        R.vref_sa.restype = C.c_uint64
        R.vref_lf.argtypes = [C.c_void_p, C.c_uint64]
        R.vref_lf.restype = C.c_uint64
        R.vref_write_csa_image.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        R.vref_write_csa_image.restype = C.c_int
        R.vref_check_csa_image.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        R.vref_check_csa_image.restype = C.c_int
        R.vref_bitrank.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
        R.vref_rank_v_blocks.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        R.vref_rank_v_blocks.restype = C.c_uint64
        # wt_int<bit_vector_il<>, rank_support_il<>> (the tree of vlg_index) and int_alphabet
        R.vrefw_create.argtypes = [C.c_void_p, C.c_uint64]
        R.vrefw_create.restype = C.c_void_p
        R.vrefw_destroy.argtypes = [C.c_void_p]
        R.vrefw_size.argtypes = [C.c_void_p]
        R.vrefw_size.restype = C.c_uint64
        R.vrefw_levels.argtypes = [C.c_void_p]
        R.vrefw_levels.restype = C.c_uint32
        R.vrefw_access.argtypes = [C.c_void_p, C.c_uint64]
        R.vrefw_access.restype = C.c_uint64
        R.vrefw_tree_bits.argtypes = [C.c_void_p, C.c_void_p]
        for nm in ("vrefw_count_less", "vrefw_quantile"):
            getattr(R, nm).argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
            getattr(R, nm).restype = C.c_uint64
        R.vrefi_create.argtypes = [C.c_void_p, C.c_uint64]
        R.vrefi_create.restype = C.c_void_p
        R.vrefi_destroy.argtypes = [C.c_void_p]
        R.vrefi_levels.argtypes = [C.c_void_p]
        R.vrefi_levels.restype = C.c_uint32
        R.vrefi_sigma.argtypes = [C.c_void_p]
        R.vrefi_sigma.restype = C.c_uint64
        R.vrefi_rank.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        R.vrefi_rank.restype = C.c_uint64
        R.vrefi_inverse_select.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        R.vrefi_inverse_select.restype = C.c_uint64
        R.vrefi_tree_bits.argtypes = [C.c_void_p, C.c_void_p]
        R.vrefw_vlg_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        R.vrefw_vlg_iterate.restype = C.c_uint64
        R.vref_int_alphabet.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        R.vref_int_alphabet.restype = C.c_uint64
        _REF = R
    return _REF


def suffix_array(text_with_sentinel):
    t = _np_u8(text_with_sentinel)
    sa = np.empty(len(t), dtype=np.uint64)
    rc = lib().vlgo_suffix_array(t.ctypes.data, len(t), sa.ctypes.data)
    assert rc == 0
    return sa


class ParseError(ValueError):
    def __init__(self, code):
        self.code = code
        super().__init__({1: "invalid gap description: expected '?' (lazy semantics)",
                          2: "invalid gap description: min-gap > max-gap",
                          3: "invalid gap description", 4: "empty sub-pattern"}.get(code, str(code)))


def parse(regexp, dialect=0):
    """-> (Query struct, backing bytes).  Raises ParseError like the reference throws."""
    raw = regexp.encode("latin-1") if isinstance(regexp, str) else bytes(regexp)
    buf = C.create_string_buffer(raw, len(raw) + 1)
    q = Query()
    rc = lib().vlgo_parse(C.addressof(buf), len(raw), dialect, C.byref(q))
    if rc:
        raise ParseError(rc)
    q._buf = buf
    return q


def query_fields(q):
    base = C.addressof(q._buf)
    subs = [bytes(q._buf.raw[q.sub[i] - base: q.sub[i] - base + q.sub_len[i]]) for i in range(q.k)]
    return subs, [int(q.lo[i]) for i in range(1, q.k)], [int(q.hi[i]) for i in range(1, q.k)], int(q.end_len)


def join(lists, lo, hi, end_len, cap=None):
    """lists: k ascending uint64 arrays; lo/hi: k-1 start-to-start bounds. -> (matches, tuples[m,k])"""
    k = len(lists)
    arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in lists]
    ptrs = (C.c_void_p * k)(*[a.ctypes.data for a in arrs])
    lens = np.array([len(a) for a in arrs], dtype=np.uint64)
    lo_a = np.array([0] + list(lo), dtype=np.uint64)
    hi_a = np.array([0] + list(hi), dtype=np.uint64)
    if cap is None:
        cap = len(arrs[0])
    out = np.empty((max(cap, 1), k), dtype=np.uint64)
    m = lib().vlgo_join(k, ptrs, lens.ctypes.data, lo_a.ctypes.data, hi_a.ctypes.data, end_len,
                        out.ctypes.data, cap)
    return int(m), out[:min(m, cap)]


class Index:
    """CPU restatement of csa_wt<wt_huff<>,32,64> + the VLG search on top of it."""

    def __init__(self, handle, dens=32):
        assert handle
        self.h = handle
        self.dens = int(dens) if dens else 32              # t_dens of csa_wt<> (csa_wt.hpp:60-72)

    @classmethod
    def from_text(cls, text, dens=32):
        t = _np_u8(text)
        assert not (t == 0).any(), "text must not contain a zero byte (construct.hpp:36-45)"
        return cls(lib().vlgo_build(t.ctypes.data, len(t), dens), dens)

    @classmethod
    def from_bwt(cls, bwt, sa, dens=32):
        b = _np_u8(bwt)
        s = np.ascontiguousarray(sa, dtype=np.uint64)
        return cls(lib().vlgo_build_from_bwt(b.ctypes.data, s.ctypes.data, len(b), dens), dens)

    @classmethod
    def from_parts(cls, p):
        nodes = np.ascontiguousarray(p["nodes"])
        assert nodes.dtype.itemsize == C.sizeof(Node)
        c2c = np.ascontiguousarray(p["char2comp"], dtype=np.uint8)
        Cc = np.ascontiguousarray(p["C"], dtype=np.uint64)
        bv = np.ascontiguousarray(p["bv_words"], dtype=np.uint64)
        smp = np.ascontiguousarray(p["samples"], dtype=np.uint64)
        h = lib().vlgo_from_parts(int(p["n"]), int(p["sigma"]), c2c.ctypes.data, Cc.ctypes.data, bv.ctypes.data,
                                  int(p["bv_bits"]), nodes.ctypes.data, len(nodes), smp.ctypes.data, len(smp),
                                  int(p.get("dens", 32)))
        return cls(h, int(p.get("dens", 32)))

    def __del__(self):
        try:
            if self.h:
                lib().vlgo_free(self.h)
                self.h = None
        except Exception:
            pass

    # parts ------------------------------------------------------------------
    @property
    def n(self):
        return int(lib().vlgo_size(self.h))

    @property
    def sigma(self):
        return int(lib().vlgo_sigma(self.h))

    def parts(self):
        L = lib()
        n_nodes = L.vlgo_n_nodes(self.h)
        nd = np.ctypeslib.as_array(C.cast(L.vlgo_nodes(self.h), u8p), shape=(max(n_nodes, 1) * C.sizeof(Node),))
        nodes = np.frombuffer(bytes(nd[: n_nodes * C.sizeof(Node)]), dtype=NODE_DTYPE).copy()
        bits = int(L.vlgo_bv_bits(self.h))
        nw = (bits + 63) // 64
        ns = int(L.vlgo_n_samples(self.h))
        return {
            "n": self.n, "sigma": self.sigma, "dens": self.dens,
            "char2comp": np.ctypeslib.as_array(L.vlgo_char2comp(self.h), shape=(256,)).copy(),
            "C": np.ctypeslib.as_array(L.vlgo_C(self.h), shape=(self.sigma + 1,)).copy(),
            "bv_bits": bits,
            "bv_words": np.ctypeslib.as_array(L.vlgo_bv_words(self.h), shape=(max(nw, 1),))[:nw].copy(),
            "nodes": nodes,
            "paths": np.ctypeslib.as_array(L.vlgo_paths(self.h), shape=(256,)).copy(),
            "samples": np.array([L.vlgo_sample(self.h, j) for j in range(ns)], dtype=np.uint64),
        }

    def bwt(self):
        p = lib().vlgo_bwt(self.h)
        return np.ctypeslib.as_array(p, shape=(self.n,)).copy()

    def rank_blocks(self):
        nw = C.c_uint64()
        p = lib().vlgo_rank_blocks(self.h, C.byref(nw))
        return np.ctypeslib.as_array(p, shape=(nw.value,)).copy()

    # primitives ---------------------------------------------------------------
    def bv_rank1(self, idx):
        return int(lib().vlgo_bv_rank1(self.h, int(idx)))

    def wt_rank(self, i, c):
        return int(lib().vlgo_wt_rank(self.h, int(i), int(c)))

    def inverse_select(self, i):
        c = C.c_uint8()
        r = lib().vlgo_inverse_select(self.h, int(i), C.byref(c))
        return int(r), int(c.value)

    def lf(self, i):
        return int(lib().vlgo_lf(self.h, int(i)))

    def sa(self, i):
        return int(lib().vlgo_sa(self.h, int(i), None, None))

    def backward_search(self, pat):
        p = _np_u8(pat)
        l, r = C.c_uint64(), C.c_uint64()
        cnt = lib().vlgo_backward_search(self.h, p.ctypes.data, len(p), C.byref(l), C.byref(r))
        return int(cnt), int(l.value), int(r.value)

    def locate(self, pat):
        p = _np_u8(pat)
        cnt = lib().vlgo_locate(self.h, p.ctypes.data, len(p), None, 0)
        out = np.empty(max(cnt, 1), dtype=np.uint64)
        lib().vlgo_locate(self.h, p.ctypes.data, len(p), out.ctypes.data, cnt)
        return out[:cnt]

    def search(self, regexp, dialect=0, stats=None):
        """-> tuples array [matches, k] (all sub-pattern start positions)."""
        q = parse(regexp, dialect)
        st = np.zeros(4, dtype=np.uint64)
        m = lib().vlgo_search(self.h, C.byref(q), None, 0, st.ctypes.data)
        out = np.empty((max(m, 1), q.k), dtype=np.uint64)
        lib().vlgo_search(self.h, C.byref(q), out.ctypes.data, m, None)
        if stats is not None:
            stats += st
        return out[:m]


def _blob(queries):
    raws = [q.encode("latin-1") if isinstance(q, str) else bytes(q) for q in queries]
    off = np.zeros(len(raws) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in raws])
    return np.frombuffer(b"".join(raws) + b"\0", dtype=np.uint8), off


def search_many(index, queries, threads=1, budget_s=1e9, dialect=0):
    """The queries on `threads` native threads (vlgo_search_many; an Index or a SaSearch) -> (queries finished, matches, stats[4])."""
    blob, off = _blob(queries)
    st = np.zeros(4, dtype=np.uint64)
    m = C.c_uint64(0)
    if isinstance(index, SaSearch):
        done = lib().vlgo_sasearch_many(index.text.ctypes.data, len(index.text), index.sa.ctypes.data, blob.ctypes.data, off.ctypes.data, len(queries),
                                        dialect, threads, float(budget_s), st.ctypes.data, C.byref(m))
    else:
        done = lib().vlgo_search_many(index.h, blob.ctypes.data, off.ctypes.data, len(queries), dialect, threads, float(budget_s), st.ctypes.data,
                                      C.byref(m))
    return int(done), int(m.value), st


class SaSearch:
    """The benchmark's SASEARCH index (index_sasearch.hpp): text + plain suffix array, forward_search, sort, join."""

    def __init__(self, text_with_sentinel, sa):
        self.text = _np_u8(text_with_sentinel)
        self.sa = np.ascontiguousarray(sa, dtype=np.uint32)
        assert len(self.text) == len(self.sa)

    def count(self, pat):
        p = _np_u8(pat)
        l, r = C.c_uint64(), C.c_uint64()
        return int(lib().vlgo_sa_forward_search(self.text.ctypes.data, len(self.text), self.sa.ctypes.data, p.ctypes.data, len(p),
                                                C.byref(l), C.byref(r)))

    def search(self, regexp, dialect=0, stats=None):
        q = parse(regexp, dialect)
        st = np.zeros(4, dtype=np.uint64)
        m = lib().vlgo_sasearch(self.text.ctypes.data, len(self.text), self.sa.ctypes.data, C.byref(q), None, 0, st.ctypes.data)
        out = np.empty((max(m, 1), q.k), dtype=np.uint64)
        lib().vlgo_sasearch(self.text.ctypes.data, len(self.text), self.sa.ctypes.data, C.byref(q), out.ctypes.data, m, None)
        if stats is not None:
            stats += st
        return out[:m]


NODE_DTYPE = np.dtype([("bv_pos", "<u8"), ("bv_pos_rank", "<u8"), ("parent", "<u2"), ("child", "<u2", (2,))],
                      align=True)
assert NODE_DTYPE.itemsize == C.sizeof(Node), (NODE_DTYPE.itemsize, C.sizeof(Node))


class RefIndex:
    """The reference's own wt_huff / byte_alphabet / LF (oracle/_ref), for pinning the oracle."""

    def __init__(self, bwt, sa, variant=0):
        R = ref()
        assert R is not None
        b = _np_u8(bwt)
        s = np.ascontiguousarray(sa, dtype=np.uint64)
        self.h = R.vref_create(b.ctypes.data, s.ctypes.data, len(b), variant)
        assert self.h

    def __del__(self):
        try:
            if self.h:
                ref().vref_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def wt_rank(self, i, c):
        return int(ref().vref_wt_rank(self.h, int(i), int(c)))

    def inverse_select(self, i):
        c = C.c_uint8()
        r = ref().vref_inverse_select(self.h, int(i), C.byref(c))
        return int(r), int(c.value)

    def bv_size(self):
        return int(ref().vref_bv_size(self.h))

    def bv_rank1(self, idx):
        return int(ref().vref_bv_rank1(self.h, int(idx)))

    def bv_words(self):
        nw = (self.bv_size() + 63) // 64
        w = np.zeros(max(nw, 1), dtype=np.uint64)
        ref().vref_bv_bits(self.h, w.ctypes.data)
        return w[:nw]

    def nodes(self):
        R = ref()
        out = []
        for v in range(R.vref_n_nodes(self.h)):
            bp, sz = C.c_uint64(), C.c_uint64()
            leaf, sym, c0, c1 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            R.vref_node(self.h, v, C.byref(bp), C.byref(sz), C.byref(leaf), C.byref(sym), C.byref(c0), C.byref(c1))
            out.append(dict(bv_pos=bp.value, size=sz.value, leaf=leaf.value, sym=sym.value, c0=c0.value, c1=c1.value))
        return out

    def alphabet(self):
        c2c = np.zeros(256, dtype=np.uint8)
        Cc = np.zeros(257, dtype=np.uint64)
        sg = C.c_uint32()
        ref().vref_alphabet(self.h, c2c.ctypes.data, Cc.ctypes.data, C.byref(sg))
        return c2c, Cc[: sg.value + 1], sg.value

    def sa(self, i):
        return int(ref().vref_sa(self.h, int(i)))

    def lf(self, i):
        return int(ref().vref_lf(self.h, int(i)))

    def write_csa_image(self, path, sa):
        """The file stock sdsl would store for this csa_wt<wt_huff<>,32,64> (members serialised by the reference itself)."""
        s = np.ascontiguousarray(sa, dtype=np.uint64)
        rc = ref().vref_write_csa_image(self.h, str(path).encode(), s.ctypes.data, len(s))
        assert rc == 0


def ref_check_csa_image(path, bwt, sa, step=1):
    """Load a csa_wt<wt_huff<>,32,64> file member by member with the reference's own load() functions and check access, rank,
    inverse_select, select, both sample vectors and the alphabet against the BWT / SA given.  0 = all good."""
    b = _np_u8(bwt)
    s = np.ascontiguousarray(sa, dtype=np.uint64)
    return int(ref().vref_check_csa_image(str(path).encode(), b.ctypes.data, s.ctypes.data, len(b), int(step)))


def ref_bitrank(words, nbits, variant, idx):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    ix = np.ascontiguousarray(idx, dtype=np.uint64)
    out = np.empty(len(ix), dtype=np.uint64)
    ref().vref_bitrank(w.ctypes.data, nbits, variant, ix.ctypes.data, len(ix), out.ctypes.data)
    return out


def ref_rank_v_blocks(words, nbits):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    cap = 2 * (((nbits + 63) // 64 >> 3) + 1) + 4
    out = np.zeros(cap, dtype=np.uint64)
    nw = ref().vref_rank_v_blocks(w.ctypes.data, nbits, out.ctypes.data, cap)
    return out[:nw]


def rrr_rank(words, nbits, idx):
    """rrr_vector<63> restatement: build from plain words, rank at every idx."""
    w = np.concatenate([np.ascontiguousarray(words, dtype=np.uint64), np.zeros(2, np.uint64)])
    h = lib().vlgo_rrr_build(w.ctypes.data, nbits)
    out = np.array([lib().vlgo_rrr_rank(h, int(i)) for i in idx], dtype=np.uint64)
    bits = int(lib().vlgo_rrr_bits(h))
    lib().vlgo_rrr_free(h)
    return out, bits


class RefWtInt:
    """The reference's own wt_int<bit_vector_il<>, rank_support_il<>> (the tree type of vlg_index, vlg_index.hpp:116-119) over a
    vector of values, built by its constructor; count_less / quantile walk it through the reference's expand() only."""

    def __init__(self, values):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        self.h = ref().vrefw_create(v.ctypes.data, len(v))
        assert self.h, "reference wt_int construction failed"
        self.n = len(v)
        self.levels = int(ref().vrefw_levels(self.h))
        assert int(ref().vrefw_size(self.h)) == self.n

    def __del__(self):
        try:
            if self.h:
                ref().vrefw_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def __getitem__(self, i):
        return int(ref().vrefw_access(self.h, int(i)))

    def level_bits(self):
        """[levels][n] array of bits: level l of wt_int::tree is its bits [l * n, (l + 1) * n)"""
        nb = self.n * self.levels
        w = np.zeros((nb + 63) // 64 + 1, dtype=np.uint64)
        ref().vrefw_tree_bits(self.h, w.ctypes.data)
        bits = np.unpackbits(w.view(np.uint8), bitorder="little")[:nb]
        return bits.reshape(self.levels, self.n)

    def vlg_iterate(self, ranges, lo, hi, last_len):
        """vlg_iterator's loops (vlg_index.hpp:227-291) restated over the reference's own wt_range_walker on this tree: ranges = SA ranges
        [(sp, ep)] of the sub-patterns, lo / hi = the start-to-start gap bounds as gapped_pattern_query stores them -> tuples [m, k]"""
        k = len(ranges)
        sp = np.array([r[0] for r in ranges], dtype=np.uint64)
        ep = np.array([r[1] for r in ranges], dtype=np.uint64)
        glo = np.array(list(lo) + [0], dtype=np.uint64)
        ghi = np.array(list(hi) + [0], dtype=np.uint64)
        cap = 1 << 12
        while True:
            out = np.zeros(cap * k, dtype=np.uint64)
            m = int(ref().vrefw_vlg_iterate(self.h, k, sp.ctypes.data, ep.ctypes.data, glo.ctypes.data, ghi.ctypes.data, int(last_len), out.ctypes.data, cap))
            if m <= cap:
                return out[: m * k].reshape(m, k)
            cap = m

    def count_less(self, l, length, x):
        return int(ref().vrefw_count_less(self.h, int(l), int(length), int(x)))

    def quantile(self, l, length, q):
        return int(ref().vrefw_quantile(self.h, int(l), int(length), int(q)))


def ref_int_alphabet(text):
    """int_alphabet<> built by the reference's constructor (csa_alphabet_strategy.hpp:394-470) -> (C[sigma+1], comp2char[sigma])"""
    t = np.ascontiguousarray(text, dtype=np.uint64)
    sigma = int(ref().vref_int_alphabet(t.ctypes.data, len(t), None, None))
    assert sigma != (1 << 64) - 1
    Cc, c2c = np.zeros(sigma + 1, np.uint64), np.zeros(max(sigma, 1), np.uint64)
    assert int(ref().vref_int_alphabet(t.ctypes.data, len(t), Cc.ctypes.data, c2c.ctypes.data)) == sigma
    return Cc, c2c[:sigma]


class IntIndex:
    """csa_wt<wt_int<>, dens, ., sa_order | text_order sampling, ., int_alphabet<>> restated (vlg_oracle_int.c): the FM-index of an
    integer text.  Symbols are positive integers (construct() refuses a 0, construct.hpp:36-45)."""

    def __init__(self, text, dens=32, text_order=False):
        t = np.ascontiguousarray(text, dtype=np.uint64)
        self.h = lib().vlgo_int_build(t.ctypes.data, len(t), dens, 1 if text_order else 0)
        if not self.h:
            raise ValueError("integer text contains the symbol 0")
        self.n, self.sigma, self.levels = int(lib().vlgo_int_size(self.h)), int(lib().vlgo_int_sigma(self.h)), int(lib().vlgo_int_levels(self.h))
        self.dens, self.text_order = dens, bool(text_order)

    def __del__(self):
        try:
            if self.h:
                lib().vlgo_int_free(self.h)
                self.h = None
        except Exception:
            pass

    def _arr(self, fn, count):
        return np.ctypeslib.as_array(fn(self.h), shape=(max(count, 1),))[:count].copy()

    def C(self):
        return self._arr(lib().vlgo_int_C, self.sigma + 1)

    def comp2char(self):
        return self._arr(lib().vlgo_int_comp2char, self.sigma)

    def bwt(self):
        return self._arr(lib().vlgo_int_bwt, self.n)

    def level_bits(self):
        nb = self.n * self.levels
        w = self._arr(lib().vlgo_int_tree, nb // 64 + 1)
        return np.unpackbits(w.view(np.uint8), bitorder="little")[:nb].reshape(self.levels, self.n)

    def samples(self):
        return self._arr(lib().vlgo_int_samples, int(lib().vlgo_int_n_samples(self.h)))

    def marked(self):
        w = self._arr(lib().vlgo_int_marked, self.n // 64 + 1)
        return np.unpackbits(w.view(np.uint8), bitorder="little")[: self.n]

    def rank(self, i, c):
        return int(lib().vlgo_int_rank(self.h, int(i), int(c)))

    def inverse_select(self, i):
        c = C.c_uint64()
        r = lib().vlgo_int_inverse_select(self.h, int(i), C.byref(c))
        return int(r), int(c.value)

    def lf(self, i):
        return int(lib().vlgo_int_lf(self.h, int(i)))

    def sa(self, i):
        return int(lib().vlgo_int_sa(self.h, int(i), None))

    def backward_search(self, pat):
        p = np.ascontiguousarray(pat, dtype=np.uint64)
        l, r = C.c_uint64(), C.c_uint64()
        cnt = lib().vlgo_int_backward_search(self.h, p.ctypes.data, len(p), C.byref(l), C.byref(r))
        return int(cnt), int(l.value), int(r.value)

    def search(self, regexp, stats=None):
        """tuples [matches, k] of the integer-alphabet query (library dialect)"""
        raw = regexp.encode("latin-1") if isinstance(regexp, str) else bytes(regexp)
        q = Query()
        syms = np.zeros(len(raw) + 2, dtype=np.uint64)
        sub_off = np.zeros(66, dtype=np.uint64)
        rc = lib().vlgo_parse_int(raw, len(raw), C.byref(q), syms.ctypes.data, len(syms), sub_off.ctypes.data)
        if rc:
            raise ParseError(rc)
        cap = 1 << 12
        while True:
            out = np.zeros(cap * q.k, dtype=np.uint64)
            st = np.zeros(4, dtype=np.uint64)
            m = int(lib().vlgo_int_search(self.h, C.byref(q), syms.ctypes.data, sub_off.ctypes.data, out.ctypes.data, cap, st.ctypes.data))
            if m <= cap:
                if stats is not None:
                    stats += st
                return out[: m * q.k].reshape(m, q.k)
            cap = m


class TextOrder:
    """text_order_sa_sampling on top of the byte index (csa_sampling_strategy.hpp:127-246): marked bit-vector + condensed samples"""

    def __init__(self, index, dens=32):
        self.index, self.dens = index, dens
        self.h = lib().vlgo_text_order_build(index.h, dens)

    def __del__(self):
        try:
            if self.h:
                lib().vlgo_text_order_free(self.h)
                self.h = None
        except Exception:
            pass

    def marked(self):
        n = self.index.n
        w = np.ctypeslib.as_array(lib().vlgo_text_order_marked(self.h), shape=(n // 64 + 1,)).copy()
        return np.unpackbits(w.view(np.uint8), bitorder="little")[:n]

    def samples(self):
        k = int(lib().vlgo_text_order_n_samples(self.h))
        return np.ctypeslib.as_array(lib().vlgo_text_order_samples(self.h), shape=(max(k, 1),))[:k].copy()

    def sa(self, i, steps=None):
        st = C.c_uint64(0)
        v = int(lib().vlgo_text_order_sa(self.index.h, self.h, int(i), C.byref(st)))
        if steps is not None:
            steps[0] += int(st.value)
        return v


class RefWtIntPlain:
    """The reference's own wt_int<> (bit_vector + rank_support_v) over a vector of integers: what csa_wt<wt_int<>> keeps its BWT in."""

    def __init__(self, values):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        self.h = ref().vrefi_create(v.ctypes.data, len(v))
        assert self.h
        self.n, self.levels, self.sigma = len(v), int(ref().vrefi_levels(self.h)), int(ref().vrefi_sigma(self.h))

    def __del__(self):
        try:
            if self.h:
                ref().vrefi_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def rank(self, i, c):
        return int(ref().vrefi_rank(self.h, int(i), int(c)))

    def inverse_select(self, i):
        c = C.c_uint64()
        r = ref().vrefi_inverse_select(self.h, int(i), C.byref(c))
        return int(r), int(c.value)

    def level_bits(self):
        nb = self.n * self.levels
        w = np.zeros((nb + 63) // 64 + 1, dtype=np.uint64)
        ref().vrefi_tree_bits(self.h, w.ctypes.data)
        return np.unpackbits(w.view(np.uint8), bitorder="little")[:nb].reshape(self.levels, self.n)
