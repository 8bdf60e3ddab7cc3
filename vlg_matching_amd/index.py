"""Host-side mirror of the reference interface for the VLG hot path.

  VlgIndex          ~ sdsl::vlg_index<> / the benchmark index concept (construct, serialize/load via parts, search)
  count / locate    ~ sdsl::count / sdsl::locate (include/sdsl/vlg_index.hpp:395-411)
  VlgIndex.search   ~ index_*::search for a batch of gapped_pattern (benchmark/gapped-matching/src/gm_search.cpp:91-121)
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import VlgError, check, lib

NODE_DTYPE = np.dtype([("bv_pos", "<u8"), ("bv_pos_rank", "<u8"), ("parent", "<u2"), ("child", "<u2", (2,))], align=True)
assert NODE_DTYPE.itemsize == C.sizeof(capi.WtNode)


def _u8(a):
    if isinstance(a, (bytes, bytearray)):
        return np.frombuffer(bytes(a), dtype=np.uint8)
    if isinstance(a, str):
        return np.frombuffer(a.encode("latin-1"), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


class SearchResult:
    """Per-query match counts, first positions (gapped_search_result::positions) and full tuples."""

    def __init__(self, handle, ks):
        self._h = handle
        self._ks = ks
        s = capi.ResultSummary()
        check(lib().vlg_result_summary_get(handle, C.byref(s)))
        self.summary = {k: int(getattr(s, k)) for k, _ in capi.ResultSummary._fields_}
        self._fetched = None

    def __del__(self):
        try:
            if self._h:
                lib().vlg_result_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def owned_queries(self):
        """[(begin, end), ...] of the queries this rank joined in a collective search; [] after a single-GPU search (all of them)"""
        n = C.c_uint32()
        check(lib().vlg_result_owned_queries(self._h, None, 0, C.byref(n)))
        r = np.zeros(2 * max(n.value, 1), dtype=np.uint64)
        check(lib().vlg_result_owned_queries(self._h, r.ctypes.data, n.value, C.byref(n)))
        return [(int(r[2 * i]), int(r[2 * i + 1])) for i in range(n.value)]

    def fetch(self):
        if self._fetched is None:
            nq = self.summary["n_queries"]
            counts = np.zeros(max(nq, 1), dtype=np.uint64)
            offsets = np.zeros(nq + 1, dtype=np.uint64)
            first = np.zeros(max(self.summary["n_matches"], 1), dtype=np.uint64)
            tuples = np.zeros(max(self.summary["n_tuple_values"], 1), dtype=np.uint64)
            has_tuples = self.summary["n_tuple_values"] or not self.summary["n_matches"]       # workspace option "tuples" = 0: none
            check(lib().vlg_result_fetch(self._h, counts.ctypes.data, offsets.ctypes.data, first.ctypes.data,
                                         tuples.ctypes.data if has_tuples else None))
            self._fetched = (counts[:nq], offsets, first[: self.summary["n_matches"]], tuples[: self.summary["n_tuple_values"]])
        return self._fetched

    def fetch_into(self, counts_ptr, offsets_ptr, first_ptr, tuples_ptr):
        """vlg_result_fetch into caller-provided host buffers (e.g. pinned memory); any pointer may be None."""
        check(lib().vlg_result_fetch(self._h, counts_ptr, offsets_ptr, first_ptr, tuples_ptr))

    def fetch32_into(self, first_ptr, tuples_ptr):
        """vlg_result_fetch32: the positions as they are held in HBM when they fit 32 bits (VlgError otherwise)."""
        check(lib().vlg_result_fetch32(self._h, first_ptr, tuples_ptr))

    def fetch32(self):
        first = np.zeros(max(self.summary["n_matches"], 1), dtype=np.uint32)
        tuples = np.zeros(max(self.summary["n_tuple_values"], 1), dtype=np.uint32)
        self.fetch32_into(first.ctypes.data, tuples.ctypes.data if self.summary["n_tuple_values"] else None)
        return first[: self.summary["n_matches"]], tuples[: self.summary["n_tuple_values"]]

    @property
    def counts(self):
        return self.fetch()[0]

    def positions(self, q):
        _, off, first, _ = self.fetch()
        return first[int(off[q]): int(off[q + 1])]

    def tuples(self, q):
        counts, _, _, tup = self.fetch()
        if self.summary["n_matches"] and not self.summary["n_tuple_values"]:
            raise VlgError(capi.E_INVALID, "tuples were not materialised (workspace option \"tuples\" is 0)")
        ks = np.asarray(self._ks, dtype=np.uint64)
        toff = np.concatenate([[0], np.cumsum(counts * ks)]).astype(np.int64)
        k = int(ks[q])
        return tup[toff[q]: toff[q + 1]].reshape(-1, max(k, 1)) if k else np.zeros((0, 0), np.uint64)


def read_sdsl_file(path, dens=32, rrr=False):
    """Host-only parse of a stock sdsl csa_wt<wt_huff<>> file (rrr: csa_wt<wt_huff<rrr_vector<63>>>, its blocks decoded back to
    plain bits) -> parts dict (same keys as VlgIndex.export_parts())."""
    f = C.c_void_p()
    check(lib().vlg_sdsl_file_open_kind(str(path).encode(), dens, 1 if rrr else 0, C.byref(f)))
    try:
        p = capi.IndexParts()
        check(lib().vlg_sdsl_file_parts(f, C.byref(p)))
        nw = (p.bv_bits + 63) // 64
        return {"n": int(p.n), "sigma": int(p.sigma), "dens": int(p.sa_sample_dens),
                "char2comp": np.ctypeslib.as_array(C.cast(p.char2comp, C.POINTER(C.c_uint8)), shape=(256,)).copy(),
                "C": np.ctypeslib.as_array(C.cast(p.C, C.POINTER(C.c_uint64)), shape=(p.sigma + 1,)).copy(),
                "bv_bits": int(p.bv_bits),
                "bv_words": np.ctypeslib.as_array(C.cast(p.bv_words, C.POINTER(C.c_uint64)), shape=(max(nw, 1),))[:nw].copy(),
                "nodes": np.frombuffer(C.string_at(p.nodes, p.n_nodes * C.sizeof(capi.WtNode)), dtype=NODE_DTYPE).copy(),
                "samples": np.ctypeslib.as_array(C.cast(p.sa_samples, C.POINTER(C.c_uint64)), shape=(max(p.n_samples, 1),))[: p.n_samples].copy()}
    finally:
        lib().vlg_sdsl_file_close(f)


class Workspace:
    def __init__(self, max_hbm_bytes=0, stream=None):
        h = C.c_void_p()
        check(lib().vlg_workspace_create(int(max_hbm_bytes), stream, C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self._h:
                lib().vlg_workspace_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_option(self, name, value):
        check(lib().vlg_workspace_set_option(self._h, name.encode(), int(value)))

    def set_comm(self, comm):
        """Collective searches over the ranks of an RCCL communicator (vlg_matching_amd.dist.Comm): the distinct lists of a batch are
        sharded for locate + sort and all-gathered, the queries are sharded for the joins.  None: single-GPU searches again."""
        check(lib().vlg_workspace_set_comm(self._h, comm._h if comm is not None else None))
        self._comm = comm

    def set_exchange(self, n_ranks, rank, callback):
        """The same with the caller moving the bytes: callback(d_buf, counts, elem_bytes, n_ranks, rank, stream) -> 0, an in-place
        all-gather of device pieces (vlg_workspace_set_exchange)."""
        def trampoline(ctx, d_buf, counts_ptr, elem_bytes, n, r, stream):
            try:
                return int(callback(d_buf, [int(counts_ptr[i]) for i in range(n)], int(elem_bytes), int(n), int(r), stream) or 0)
            except Exception as e:                                   # an exception must not cross the C frame
                import sys
                print("exchange callback failed: %r" % (e,), file=sys.stderr)
                return 1
        self._exchange_cb = capi.EXCHANGE_FN(trampoline)
        check(lib().vlg_workspace_set_exchange(self._h, int(n_ranks), int(rank), self._exchange_cb, None))

    def set_exchange_alltoall(self, n_ranks, rank, callback):
        """The pairwise exchange with the caller moving the bytes (vlg_workspace_set_exchange_alltoall):
        callback(d_send, send_counts, d_recv, recv_counts, elem_bytes, n_ranks, rank, stream) -> 0 sends send_counts[r] elements to rank r
        (packed in rank order at d_send) and receives recv_counts[r] from it (packed at d_recv)."""
        def trampoline(ctx, d_send, sc, d_recv, rc, elem_bytes, n, r, stream):
            try:
                return int(callback(d_send or 0, [int(sc[i]) for i in range(n)], d_recv or 0, [int(rc[i]) for i in range(n)], int(elem_bytes),
                                    int(n), int(r), stream) or 0)
            except Exception as e:                                   # an exception must not cross the C frame
                import sys
                print("exchange callback failed: %r" % (e,), file=sys.stderr)
                return 1
        self._exchange_cb = capi.ALLTOALL_FN(trampoline)
        check(lib().vlg_workspace_set_exchange_alltoall(self._h, int(n_ranks), int(rank), self._exchange_cb, None))

    def profile(self, enable=True):
        check(lib().vlg_workspace_profile(self._h, 1 if enable else 0))

    def kernel_stats(self):
        arr = (capi.KernelStat * 32)()
        n = C.c_uint32()
        check(lib().vlg_workspace_kernel_stats(self._h, arr, 32, C.byref(n)))
        return {arr[i].name.decode(): dict(launches=int(arr[i].launches), total_ms=float(arr[i].total_ms),
                                           algorithmic_bytes=int(arr[i].algorithmic_bytes)) for i in range(min(n.value, 32))}


class SymbolMap:
    """vlg_symbol_map: the sorted distinct symbols of a 64-bit integer text, symbol -> rank + 1 (dense, order-preserving, never 0).
    The device indexes hold uint32 symbols; a text with larger ones is mapped, indexed, and queried through the same map."""

    def __init__(self, text):
        t = np.ascontiguousarray(text, dtype=np.uint64)
        h = C.c_void_p()
        check(lib().vlg_symbol_map_create(t.ctypes.data if len(t) else None, len(t), C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self._h:
                lib().vlg_symbol_map_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def sigma(self):
        return int(lib().vlg_symbol_map_sigma(self._h))

    def symbols(self):
        out = np.zeros(self.sigma, dtype=np.uint64)
        check(lib().vlg_symbol_map_symbols(self._h, out.ctypes.data if len(out) else None))
        return out

    def apply(self, symbols):
        """-> uint32 array: rank + 1 of every symbol (sigma + 1 for a symbol the map's text does not hold)"""
        t = np.ascontiguousarray(symbols, dtype=np.uint64)
        out = np.zeros(len(t), dtype=np.uint32)
        check(lib().vlg_symbol_map_apply(self._h, t.ctypes.data if len(t) else None, len(t), out.ctypes.data if len(t) else None))
        return out

    def queries(self, regexps, strict=True):
        return Queries.from_int(regexps, strict=strict, symbol_map=self)


def parse_query(regexp, dialect=capi.DIALECT_LIBRARY):
    """gapped_pattern_query / gapped_pattern on the host: -> (sub-patterns, lo[], hi[], end_len).  Raises VlgError(E_PARSE)."""
    raw = regexp.encode("latin-1") if isinstance(regexp, str) else bytes(regexp)
    p = capi.ParsedQuery()
    check(lib().vlg_parse_query(raw, len(raw), dialect, C.byref(p)))
    subs = [raw[p.sub_off[i]: p.sub_off[i] + p.sub_len[i]] for i in range(p.k)]
    return subs, [int(p.lo[i]) for i in range(1, p.k)], [int(p.hi[i]) for i in range(1, p.k)], int(p.end_len)


class Queries:
    """A parsed query batch resident in HBM."""

    def __init__(self, regexps, dialect=capi.DIALECT_LIBRARY, strict=True):
        raws = [r.encode("latin-1") if isinstance(r, str) else bytes(r) for r in regexps]
        off = np.zeros(len(raws) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in raws])
        text = b"".join(raws)
        h = C.c_void_p()
        status = np.zeros(max(len(raws), 1), dtype=np.int32)
        check(lib().vlg_queries_parse(text, off.ctypes.data, len(raws), dialect, None if strict else status.ctypes.data, C.byref(h)))
        self._h = h
        self.status = status[: len(raws)]
        self.n = len(raws)

    @classmethod
    def from_blob(cls, text, off, dialect=capi.DIALECT_LIBRARY):
        """The regexps already concatenated (query i = text[off[i], off[i+1])): one vlg_queries_parse call, nothing else."""
        self = cls.__new__(cls)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        h = C.c_void_p()
        check(lib().vlg_queries_parse(text, off.ctypes.data, len(off) - 1, dialect, None, C.byref(h)))
        self._h = h
        self.n = len(off) - 1
        self.status = np.zeros(self.n, dtype=np.int32)
        return self

    @classmethod
    def from_int(cls, regexps, strict=True, symbol_map=None):
        """Integer-alphabet batch (gapped_pattern_query<int_alphabet_tag>): sub-patterns are whitespace-separated decimals, gaps count
        symbols -- for an integer-alphabet index (VlgIndex.build_int) or WtsaIndex over an integer text.  symbol_map: the SymbolMap
        the index's text went through (64-bit symbols): tokens are the original symbols."""
        raws = [r.encode("latin-1") if isinstance(r, str) else bytes(r) for r in regexps]
        off = np.zeros(len(raws) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in raws])
        self = cls.__new__(cls)
        h = C.c_void_p()
        status = np.zeros(max(len(raws), 1), dtype=np.int32)
        st = None if strict else status.ctypes.data
        if symbol_map is not None:
            check(lib().vlg_queries_parse_int_mapped(symbol_map._h, b"".join(raws), off.ctypes.data, len(raws), st, C.byref(h)))
        else:
            check(lib().vlg_queries_parse_int(b"".join(raws), off.ctypes.data, len(raws), st, C.byref(h)))
        self._h, self.n, self.status = h, len(raws), status[: len(raws)]
        return self

    def subpattern_range(self):
        """qsub[nq+1]: query i owns the sub-patterns [qsub[i], qsub[i+1]) of the batch"""
        return np.concatenate([[0], np.cumsum(self.ks.astype(np.int64))])

    @classmethod
    def from_arrays(cls, subpatterns, lo, hi, end_len):
        """subpatterns: list (per query) of lists of bytes; lo/hi: per query lists of k-1 start-to-start bounds."""
        self = cls.__new__(cls)
        blob, suboff, qsub, flo, fhi = [], [0], [0], [], []
        for subs, l, h in zip(subpatterns, lo, hi):
            for i, s in enumerate(subs):
                blob.append(bytes(s))
                suboff.append(suboff[-1] + len(s))
                flo.append(0 if i == 0 else int(l[i - 1]))
                fhi.append(0 if i == 0 else int(h[i - 1]))
            qsub.append(len(suboff) - 1)
        b = np.frombuffer(b"".join(blob) + b"\0", dtype=np.uint8)
        a = [np.asarray(x, dtype=np.uint64) for x in (suboff, qsub, flo + [0], fhi + [0], list(end_len) + [0])]
        hq = C.c_void_p()
        check(lib().vlg_queries_create(b.ctypes.data, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                                       a[4].ctypes.data, len(subpatterns), C.byref(hq)))
        self._h = hq
        self.n = len(subpatterns)
        self.status = np.zeros(self.n, dtype=np.int32)
        return self

    @property
    def ks(self):
        """sub-patterns per query (0 for a query that failed to parse)"""
        if getattr(self, "_ks", None) is None:
            k = np.zeros(max(self.n, 1), dtype=np.uint32)
            check(lib().vlg_queries_k(self._h, k.ctypes.data))
            self._ks = k[: self.n]
        return self._ks

    def __del__(self):
        try:
            if self._h:
                lib().vlg_queries_destroy(self._h)
                self._h = None
        except Exception:
            pass


class VlgIndex:
    """FM-index (csa_wt<wt_huff<>>-equivalent) resident in HBM."""

    def __init__(self, handle, keep=None):
        self._h = handle
        self._keep = keep          # e.g. the torch tensor backing an attached blob
        self._ws = None

    # -- construction ---------------------------------------------------------------------------
    @classmethod
    def build(cls, text, dens=32):
        """sdsl::construct equivalent, on the device (suffix sort, BWT, wavelet tree, sampling)."""
        t = _u8(text)
        h = C.c_void_p()
        check(lib().vlg_index_build(t.ctypes.data if len(t) else None, len(t), dens, C.byref(h)))
        return cls(h)

    @classmethod
    def build_int(cls, text, dens=32):
        """FM-index of an integer text (csa_wt<wt_int<>, dens, ..., int_alphabet<>>): symbols uint32, none of them 0."""
        t = np.ascontiguousarray(text, dtype=np.uint32)
        h = C.c_void_p()
        check(lib().vlg_index_build_int(t.ctypes.data if len(t) else None, len(t), dens, C.byref(h)))
        return cls(h)

    def int_alphabet(self):
        """(C[sigma + 1], comp2char[sigma]) of an integer-alphabet index"""
        sg = C.c_uint64()
        check(lib().vlg_index_export_int_alphabet(self._h, C.byref(sg), None, None))
        Cc, c2c = np.zeros(sg.value + 1, np.uint64), np.zeros(max(sg.value, 1), np.uint64)
        check(lib().vlg_index_export_int_alphabet(self._h, C.byref(sg), Cc.ctypes.data, c2c.ctypes.data))
        return Cc, c2c[: sg.value]

    @classmethod
    def build_device(cls, d_text_ptr, n_text, dens=32, stream=None):
        h = C.c_void_p()
        check(lib().vlg_index_build_device(d_text_ptr, n_text, dens, stream, C.byref(h)))
        return cls(h)

    @classmethod
    def from_parts(cls, p):
        """Adopt an index in the reference's own layout (dict as produced by export_parts())."""
        nodes = np.ascontiguousarray(p["nodes"])
        c2c = np.ascontiguousarray(p["char2comp"], dtype=np.uint8)
        Cc = np.ascontiguousarray(p["C"], dtype=np.uint64)
        bv = np.ascontiguousarray(p["bv_words"], dtype=np.uint64)
        smp = np.ascontiguousarray(p["samples"], dtype=np.uint64)
        parts = capi.IndexParts(int(p["n"]), int(p["sigma"]), int(p.get("dens", 32)), c2c.ctypes.data, Cc.ctypes.data,
                                bv.ctypes.data if len(bv) else None, int(p["bv_bits"]), nodes.ctypes.data, len(nodes),
                                smp.ctypes.data, len(smp))
        h = C.c_void_p()
        check(lib().vlg_index_from_parts(C.byref(parts), C.byref(h)))
        return cls(h)

    @classmethod
    def load_sdsl(cls, path, dens=32, rrr=False):
        """An index stored by stock sdsl: a csa_wt<wt_huff<>> file, or (rrr) a csa_wt<wt_huff<rrr_vector<63>>> file."""
        h = C.c_void_p()
        check(lib().vlg_index_load_sdsl_kind(str(path).encode(), dens, 1 if rrr else 0, C.byref(h)))
        return cls(h)

    def save_sdsl(self, path):
        """Store in the reference's on-disk format of csa_wt<wt_huff<>,32,64> (stock sdsl can load_from_file it)."""
        check(lib().vlg_index_save_sdsl(self._h, str(path).encode()))

    def isa_samples(self, inv_dens=64):
        """isa_sample of csa_wt: out[j] = SA index of text position j * inv_dens."""
        n = self.info()["n"]
        out = np.zeros((n - 1) // inv_dens + 1, dtype=np.uint64)
        check(lib().vlg_index_isa_samples(self._h, inv_dens, out.ctypes.data, len(out)))
        return out

    @classmethod
    def attach_blob(cls, d_ptr, nbytes, keep=None):
        h = C.c_void_p()
        check(lib().vlg_index_attach_blob(d_ptr, nbytes, C.byref(h)))
        return cls(h, keep)

    def __del__(self):
        try:
            if self._h:
                lib().vlg_index_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- introspection ----------------------------------------------------------------------------
    def info(self):
        i = capi.IndexInfo()
        check(lib().vlg_index_get_info(self._h, C.byref(i)))
        return {k: int(getattr(i, k)) for k, _ in capi.IndexInfo._fields_}

    def compress(self, bv_kind=1):
        """A csa_wt<wt_huff<rrr_vector<63>>>-equivalent of this (plain) index (wt_int<rrr_vector<63>> for an integer index); same answers, compressed bit-vectors."""
        h = C.c_void_p()
        check(lib().vlg_index_compress(self._h, bv_kind, C.byref(h)))
        return VlgIndex(h)

    def resample(self, text_order=True, dens=32):
        """A second index over the same BWT with text_order_sa_sampling (or SA-order sampling of another density); same answers."""
        h = C.c_void_p()
        check(lib().vlg_index_resample(self._h, 1 if text_order else 0, int(dens), C.byref(h)))
        return VlgIndex(h)

    def marked(self):
        """text-order sampling: the marks over the SA indices -> uint8 array of n zeros / ones"""
        n = self.info()["n"]
        w = np.zeros((n + 63) // 64, dtype=np.uint64)
        check(lib().vlg_index_export_marked(self._h, w.ctypes.data))
        return np.unpackbits(w.view(np.uint8), bitorder="little")[:n]

    def export_parts(self):
        sz = capi.IndexParts()
        check(lib().vlg_index_export_parts(self._h, C.byref(sz), None))
        c2c = np.zeros(256, np.uint8)
        Cc = np.zeros(257, np.uint64)
        bv = np.zeros(max((sz.bv_bits + 63) // 64, 1), np.uint64)
        nodes = np.zeros(max(sz.n_nodes, 1), dtype=NODE_DTYPE)
        smp = np.zeros(max(sz.n_samples, 1), np.uint64)
        out = capi.IndexPartsOut(c2c.ctypes.data, Cc.ctypes.data, bv.ctypes.data, nodes.ctypes.data, smp.ctypes.data)
        check(lib().vlg_index_export_parts(self._h, C.byref(sz), C.byref(out)))
        return {"n": int(sz.n), "sigma": int(sz.sigma), "dens": int(sz.sa_sample_dens), "char2comp": c2c,
                "C": Cc[: sz.sigma + 1], "bv_bits": int(sz.bv_bits), "bv_words": bv[: (sz.bv_bits + 63) // 64],
                "nodes": nodes[: sz.n_nodes], "samples": smp[: sz.n_samples]}

    def blob_bytes(self):
        b = C.c_uint64()
        check(lib().vlg_index_blob_bytes(self._h, C.byref(b)))
        return int(b.value)

    def blob_export(self, d_ptr, nbytes, stream=None):
        check(lib().vlg_index_blob_export(self._h, d_ptr, nbytes, stream))

    # -- search -----------------------------------------------------------------------------------
    def _queries(self, queries, dialect=capi.DIALECT_LIBRARY, strict=True):
        if isinstance(queries, Queries):
            return queries
        if self.info()["bv_kind"] in (2, 3):                       # integer-alphabet index: the integer query dialect
            return Queries.from_int(queries, strict)
        return Queries(queries, dialect, strict)

    def occurrences(self, queries, dialect=capi.DIALECT_LIBRARY):
        """sdsl::count of every sub-pattern of the batch (one backward-search pass) -> uint64[n sub-patterns]"""
        q = self._queries(queries, dialect)
        nsub = int(lib().vlg_queries_subpatterns(q._h))
        occ = np.zeros(max(nsub, 1), dtype=np.uint64)
        check(lib().vlg_queries_occurrences(self._h, q._h, occ.ctypes.data, None))
        return occ[:nsub], q

    def intervals(self, queries, dialect=capi.DIALECT_LIBRARY):
        """SA interval [l, r] of every sub-pattern of the batch (one backward-search pass) -> (l[], r[], Queries)"""
        q = self._queries(queries, dialect)
        nsub = int(lib().vlg_queries_subpatterns(q._h))
        l, r = np.zeros(max(nsub, 1), dtype=np.uint64), np.zeros(max(nsub, 1), dtype=np.uint64)
        check(lib().vlg_queries_intervals(self._h, q._h, l.ctypes.data, r.ctypes.data, None))
        return l[:nsub], r[:nsub], q

    def query_weights(self, queries, dialect=capi.DIALECT_LIBRARY):
        """Estimated work per query = sum of the SA-interval sizes of its sub-patterns (0 when one of them does not occur: such a
        query locates nothing) -- what vlg_matching_amd.dist.shard_by_work balances."""
        occ, q = self.occurrences(queries, dialect)
        qsub = q.subpattern_range()
        w = np.zeros(q.n, dtype=np.float64)
        if len(occ):
            cs = np.concatenate([[0], np.cumsum(occ.astype(np.float64))])
            w = cs[qsub[1:]] - cs[qsub[:-1]]
            dead = np.zeros(q.n, dtype=bool)
            zero = np.concatenate([[0], np.cumsum(occ == 0)])
            dead = (zero[qsub[1:]] - zero[qsub[:-1]]) > 0
            w[dead] = 0.0
        return w + 1.0                                         # every query costs something (parse, plan)

    def workspace(self, max_hbm_bytes=0):
        if self._ws is None:
            self._ws = Workspace(max_hbm_bytes)
        return self._ws

    def search(self, queries, dialect=capi.DIALECT_LIBRARY, workspace=None, strict=True):
        """Batched `idx.search(pat)`: queries is a list of regexps or a Queries object."""
        q = self._queries(queries, dialect, strict)
        ws = workspace or self.workspace()
        h = C.c_void_p()
        check(lib().vlg_search_batch(self._h, q._h, ws._h, C.byref(h)))
        return SearchResult(h, q.ks)


class WtsaIndex:
    """sdsl::vlg_index<alphabet_tag, wt_int<>> in HBM: the text + a wavelet tree over its suffix array, searched lazily
    (include/sdsl/vlg_index.hpp:109-373).  `text`: bytes / uint8 array (byte alphabet) or a uint32 array (integer alphabet)."""

    def __init__(self, text):
        if isinstance(text, np.ndarray) and text.dtype != np.uint8:
            t = np.ascontiguousarray(text, dtype=np.uint32)
            self.symbol_bytes = 4
        else:
            t = _u8(text)
            self.symbol_bytes = 1
        h = C.c_void_p()
        check(lib().vlg_wtsa_build(t.ctypes.data if len(t) else None, len(t), self.symbol_bytes, C.byref(h)))
        self._h = h
        self._ws = None

    def __del__(self):
        try:
            if self._h:
                lib().vlg_wtsa_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def info(self):
        i = capi.WtsaInfo()
        check(lib().vlg_wtsa_get_info(self._h, C.byref(i)))
        return {k: int(getattr(i, k)) for k, _ in capi.WtsaInfo._fields_}

    def queries(self, regexps):
        """a parsed batch for this index's alphabet (integer alphabet: sub-patterns are whitespace-separated decimals)"""
        if isinstance(regexps, Queries):
            return regexps
        if self.symbol_bytes == 1:
            return Queries(regexps)
        raws = [r.encode("latin-1") if isinstance(r, str) else bytes(r) for r in regexps]
        off = np.zeros(len(raws) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in raws])
        q = Queries.__new__(Queries)
        h = C.c_void_p()
        check(lib().vlg_queries_parse_int(b"".join(raws), off.ctypes.data, len(raws), None, C.byref(h)))
        q._h, q.n, q.status = h, len(raws), np.zeros(len(raws), dtype=np.int32)
        return q

    def sa_device(self, d_idx_ptr, d_out_ptr, count, stream=None):
        check(lib().vlg_wtsa_sa_batch(self._h, d_idx_ptr, d_out_ptr, count, stream))

    def range_walk_device(self, d_l_ptr, d_len_ptr, d_x_ptr, quantile, d_out_ptr, count, stream=None):
        """count_less (quantile False) / quantile (True) on suffix-array ranges, device pointers"""
        check(lib().vlg_wtsa_range_walk_batch(self._h, d_l_ptr, d_len_ptr, d_x_ptr, 1 if quantile else 0, d_out_ptr, count, stream))

    def level_bits(self, level):
        """bits of one level of the tree -> uint8 array of n zeros / ones (wt_int::tree[level * n : (level + 1) * n])"""
        n = self.info()["n"]
        w = np.zeros((n + 63) // 64, dtype=np.uint64)
        check(lib().vlg_wtsa_export_level(self._h, int(level), w.ctypes.data))
        return np.unpackbits(w.view(np.uint8), bitorder="little")[:n]

    def ranges(self, queries):
        """forward_search of every sub-pattern -> (sp[], ep[]) suffix-array ranges (sp = ep + 1: no occurrence)"""
        q = self.queries(queries)
        nsub = int(lib().vlg_queries_subpatterns(q._h))
        sp, ep = np.zeros(max(nsub, 1), np.uint64), np.zeros(max(nsub, 1), np.uint64)
        check(lib().vlg_wtsa_ranges(self._h, q._h, sp.ctypes.data, ep.ctypes.data, None))
        return sp[:nsub], ep[:nsub]

    def search(self, queries, max_matches=0, workspace=None):
        """sdsl::locate(idx, query) for a batch, lazily: at most max_matches matches per query (0 = all)."""
        q = self.queries(queries)
        if workspace is None:
            if self._ws is None:
                self._ws = Workspace()
            workspace = self._ws
        h = C.c_void_p()
        check(lib().vlg_wtsa_search_batch(self._h, q._h, int(max_matches), workspace._h, C.byref(h)))
        return SearchResult(h, q.ks)


def join_batch(d_lists_ptr, list_off, join_list, lo, hi, end_len, workspace, ks=None):
    """vlg_join_batch: the gap-bounded merge join (index_sasearch.hpp:85-116) over caller-provided sorted u64 lists in HBM.
    list_off[n_lists+1], join_list[n_joins+1], lo/hi[n_lists], end_len[n_joins] are host arrays.  -> SearchResult"""
    a = [np.ascontiguousarray(x, dtype=np.uint64) for x in (list_off, join_list, lo, hi, end_len)]
    n_lists, n_joins = len(a[0]) - 1, len(a[1]) - 1
    if len(a[2]) < max(n_lists, 1):
        a[2] = np.concatenate([a[2], np.zeros(max(n_lists, 1) - len(a[2]), np.uint64)])
    if len(a[3]) < max(n_lists, 1):
        a[3] = np.concatenate([a[3], np.zeros(max(n_lists, 1) - len(a[3]), np.uint64)])
    if len(a[4]) < max(n_joins, 1):
        a[4] = np.concatenate([a[4], np.ones(max(n_joins, 1) - len(a[4]), np.uint64)])
    h = C.c_void_p()
    check(lib().vlg_join_batch(d_lists_ptr, a[0].ctypes.data, n_lists, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                               a[4].ctypes.data, n_joins, workspace._h, C.byref(h)))
    return SearchResult(h, np.diff(a[1].astype(np.int64)).astype(np.uint32) if ks is None else ks)


def locate(idx, query):
    """sdsl::locate(idx, query): tuples [matches, k] of sub-pattern start positions (library dialect)."""
    r = idx.search([query])
    return r.tuples(0)


def count(idx, query):
    """sdsl::count(idx, query)."""
    return int(idx.search([query]).counts[0])


class BitVector:
    """rank_support_v-equivalent on a plain bit-vector, resident in HBM (K1)."""

    def __init__(self, words, nbits):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        h = C.c_void_p()
        check(lib().vlg_bitvector_create(w.ctypes.data if len(w) else None, int(nbits), C.byref(h)))
        self._h = h
        self.nbits = int(nbits)

    def __del__(self):
        try:
            if self._h:
                lib().vlg_bitvector_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def rank_device(self, d_idx_ptr, d_out_ptr, count, stream=None):
        check(lib().vlg_bitvector_rank_batch(self._h, d_idx_ptr, d_out_ptr, count, stream))

    def hbm_bytes(self):
        return int(lib().vlg_bitvector_hbm_bytes(self._h))


class RrrBitVector:
    """rrr_vector<63> + rank_support_rrr equivalent in HBM (K6): on-the-fly block decode, binomial table in LDS."""

    def __init__(self, words, nbits):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        h = C.c_void_p()
        check(lib().vlg_rrr_bitvector_create(w.ctypes.data if len(w) else None, int(nbits), C.byref(h)))
        self._h = h
        self.nbits = int(nbits)

    def __del__(self):
        try:
            if self._h:
                lib().vlg_rrr_bitvector_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def rank_device(self, d_idx_ptr, d_out_ptr, count, stream=None):
        check(lib().vlg_rrr_bitvector_rank_batch(self._h, d_idx_ptr, d_out_ptr, count, stream))

    def hbm_bytes(self):
        return int(lib().vlg_rrr_bitvector_hbm_bytes(self._h))
