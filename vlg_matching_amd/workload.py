"""Seeded synthetic texts and query batches.  Host-side bench/test infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# SURVEY.md 8(d) workload table
CONFIGS = {
    "C1": dict(kind="dna", n=1 << 20, seed=1, nq=4, k=2, m=5, gap=(0, 100), qseed=11),
    "C2": dict(kind="dna_pc", n=100 << 20, seed=2, nq=10000, k=2, m=10, gap=(0, 100), qseed=12),
    "C3": dict(kind="english", n=1 << 30, seed=3, nq=100000, k=3, m=6, gap=(0, 1000), qseed=13),
    "C4": dict(kind="protein", n=1 << 32, seed=4, nq=1000000, k=2, m=5, gap=(0, 100), qseed=14),
    "C5": dict(kind="dna_pc", n=1 << 30, seed=5, nq=100000, k=2, m=12, gap=(0, 100), qseed=15),
}

# What the configs' batches return (gen_queries(text, nq, k, m, gap, qseed) of the config, library dialect) -- exact integers, the
# same on every run: bench.py's `other_configs` and tests/test_gpu_fullsize.py both assert them.  None = not recorded yet.
EXPECTED = {
    "C2": {"n_matches": 144, "checksum": 6970539074, "located_occurrences": 2551964},
    "C3": {"n_matches": 305002331, "checksum": 163759923662180697, "located_occurrences": 637097700},
    "C4": {"n_matches": 255088, "checksum": 547974346109916, "located_occurrences": 2798227671},
    "C5": {"n_matches": 80, "checksum": 42340285501, "located_occurrences": 17372402},
}


def check_expected(name, summary):
    """The batch of config `name` returned what it always returns (exact integers)."""
    want = EXPECTED[name]
    got = {k: summary[k] for k in want}
    assert got == want, (name, got, want)

_PROTEIN = [("A", 825), ("R", 553), ("N", 406), ("D", 545), ("C", 137), ("Q", 393), ("E", 675), ("G", 707), ("H", 227),
            ("I", 596), ("L", 966), ("K", 584), ("M", 242), ("F", 386), ("P", 470), ("S", 656), ("T", 534), ("W", 108),
            ("Y", 292), ("V", 687)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libvlg_workload.so")
        src = os.path.join(_HERE, "csrc", "workload.c")
        if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, src])
        L = C.CDLL(so)
        L.vlgw_gen_iid.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32]
        L.vlgw_gen_zipf_words.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.vlgw_gen_zipf_words.restype = C.c_int
        L.vlgw_gen_positions.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        _LIB = L
    return _LIB


def _iid(n, seed, symbols, weights):
    tot = sum(weights)
    cum, acc = [], 0
    for w in weights:
        acc += w
        cum.append((acc << 32) // tot)
    cum[-1] = 1 << 32
    alpha = np.frombuffer("".join(symbols).encode(), dtype=np.uint8).copy()
    cum = np.array(cum, dtype=np.uint64)
    out = np.empty(n, dtype=np.uint8)
    lib().vlgw_gen_iid(out.ctypes.data, n, seed, alpha.ctypes.data, cum.ctypes.data, len(alpha))
    return out


def gen_text(kind, n, seed):
    """-> uint8 array of n bytes, none of them 0."""
    if kind == "dna":
        return _iid(n, seed, "ACGT", [1, 1, 1, 1])
    if kind == "dna_pc":
        return _iid(n, seed, "ACGT", [29, 21, 21, 29])
    if kind == "protein":
        return _iid(n, seed, [p[0] for p in _PROTEIN], [p[1] for p in _PROTEIN])
    if kind == "english":
        out = np.empty(n, dtype=np.uint8)
        rc = lib().vlgw_gen_zipf_words(out.ctypes.data, n, seed, 50000, 2, 10)
        assert rc == 0
        return out
    raise ValueError(kind)


def gen_query_parts(text, nq, k, m, seed):
    """sub-patterns = text[p, p+m) with p uniform (k per query).  -> list of lists of bytes"""
    pos = np.empty(nq * k, dtype=np.uint64)
    lib().vlgw_gen_positions(pos.ctypes.data, nq * k, seed, len(text), m)
    t = np.ascontiguousarray(text)
    subs = [t[int(p): int(p) + m].tobytes() for p in pos]
    return [subs[i * k:(i + 1) * k] for i in range(nq)]


def gen_queries(text, nq, k, m, gap, seed, dialect=0):
    """Regexps `s0.{a,b}?s1...` (library dialect) or without '?' (benchmark dialect)."""
    g = ".{%d,%d}%s" % (gap[0], gap[1], "?" if dialect == 0 else "")
    return [g.join(s.decode("latin-1") for s in subs) for subs in gen_query_parts(text, nq, k, m, seed)]


def config(name, scale=1.0):
    """Workload of SURVEY.md 8(d); scale < 1 shrinks text and query count proportionally (tests)."""
    c = dict(CONFIGS[name])
    if scale != 1.0:
        c["n"] = max(1024, int(c["n"] * scale))
        c["nq"] = max(16, int(c["nq"] * scale))
    return c
