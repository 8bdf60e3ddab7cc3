"""GPU parity tests: the HIP path, called through the C-ABI (include/vlg_hip.h), against the CPU oracle
on the same seeded inputs and against the committed known answers.  Bit-exact everywhere (integer work)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

from util import bwt_from_sa, dna_text, skewed_text

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "vlg_known_answers.json")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def V():
    import vlg_matching_amd as v
    v.lib()                      # fails loudly if the HIP extension is missing
    return v


def dev_u64(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).cuda()


def host_u64(t):
    return t.cpu().numpy().view(np.uint64)


def bits_to_words(bits):
    nbits = len(bits)
    nw = (nbits + 63) // 64
    b = np.zeros(nw * 64, dtype=bool)
    b[:nbits] = bits
    return np.packbits(b.reshape(-1, 64)[:, ::-1], axis=1).view(">u8").ravel().astype(np.uint64) if nw else np.zeros(0, np.uint64)


@pytest.mark.parametrize("nbits", [0, 1, 63, 64, 223, 224, 225, 447, 448, 4096, 100000, 1 << 20])
def test_k1_bitvector_rank(torch_cuda, V, oracle, nbits):
    """rank_support_test.cpp:70-87 restated: rank(j) for every j (sampled for big vectors)."""
    torch = torch_cuda
    rng = np.random.default_rng(nbits + 1)
    for dens in (0.0, 0.1, 0.5, 1.0):
        bits = rng.random(nbits) < dens
        words = bits_to_words(bits)
        bv = V.BitVector(words, nbits)
        idx = np.arange(nbits + 1, dtype=np.uint64) if nbits <= 5000 else \
            np.unique(np.concatenate([rng.integers(0, nbits + 1, 20000), [0, nbits, 224, 223, nbits - 1]])).astype(np.uint64)
        d_idx = dev_u64(torch, idx)
        d_out = torch.zeros_like(d_idx)
        bv.rank_device(d_idx.data_ptr(), d_out.data_ptr(), len(idx))
        torch.cuda.synchronize()
        got = host_u64(d_out)
        L = oracle.lib()
        wpad = np.concatenate([words, np.zeros(1, np.uint64)])
        blocks = np.zeros(2 * ((len(words) >> 3) + 1), dtype=np.uint64)
        L.vlgo_rank_v_build(wpad.ctypes.data, nbits, blocks.ctypes.data)
        want = np.array([L.vlgo_rank_v(wpad.ctypes.data, blocks.ctypes.data, int(i)) for i in idx[:3000]], dtype=np.uint64)
        assert (got[:3000] == want).all()
        # a-4: the same super-block layout replaces rank_support_v5 too (rank_support_v5.hpp:116-134, restated and pinned in the oracle)
        w5 = np.concatenate([words, np.zeros(40, np.uint64)])
        blocks5 = np.zeros(2 * ((len(words) >> 5) + 1) + 8, dtype=np.uint64)
        L.vlgo_rank_v5_build(w5.ctypes.data, nbits, blocks5.ctypes.data)
        want5 = np.array([L.vlgo_rank_v5(w5.ctypes.data, blocks5.ctypes.data, int(i)) for i in idx[:3000]], dtype=np.uint64)
        assert (got[:3000] == want5).all()
        truth = np.concatenate([[0], np.cumsum(bits)])[idx.astype(np.int64)]
        assert (got == truth).all()


TEXTS = {
    "abracadabra": lambda: b"abracadabrasimsalabim",
    "one_byte": lambda: b"a",
    "100a": lambda: b"a" * 100,
    "all_symbols": lambda: bytes(range(1, 256)),
    "dna_50k": lambda: dna_text(50000, 1).tobytes(),
    "dna_skew": lambda: dna_text(30000, 2, (0.7, 0.1, 0.1, 0.1)).tobytes(),
    "zipf40": lambda: skewed_text(40000, 3).tobytes(),
}


def assert_parts_equal(a, b, check_rank=True):
    assert a["n"] == b["n"] and a["sigma"] == b["sigma"] and a["bv_bits"] == b["bv_bits"]
    assert (np.asarray(a["char2comp"]) == np.asarray(b["char2comp"])).all()
    assert (np.asarray(a["C"]) == np.asarray(b["C"])).all()
    assert (np.asarray(a["bv_words"]) == np.asarray(b["bv_words"])).all(), "wavelet-tree bits differ"
    assert (np.asarray(a["samples"]) == np.asarray(b["samples"])).all(), "SA samples differ"
    na, nb = a["nodes"], b["nodes"]
    assert len(na) == len(nb)
    for f in ("bv_pos", "parent", "child") + (("bv_pos_rank",) if check_rank else ()):
        assert (na[f] == nb[f]).all(), f


@pytest.mark.parametrize("name", list(TEXTS))
def test_from_parts_export_and_blob(torch_cuda, V, oracle, name):
    torch = torch_cuda
    text = TEXTS[name]()
    o = oracle.Index.from_text(text)
    parts = o.parts()
    idx = V.VlgIndex.from_parts(parts)
    assert_parts_equal(idx.export_parts(), parts)
    nb = idx.blob_bytes()
    blob = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    idx.blob_export(blob.data_ptr(), nb)
    blob2 = blob.clone()                                  # as if it had been broadcast to another GPU
    idx2 = V.VlgIndex.attach_blob(blob2.data_ptr(), nb, keep=blob2)
    assert_parts_equal(idx2.export_parts(), parts)
    assert idx2.info() == idx.info()


@pytest.mark.parametrize("name", list(TEXTS) + ["empty"])
def test_device_build_equals_oracle_build(V, oracle, name):
    """SA (csa_byte_test.cpp:136-147 via samples), BWT -> identical Huffman WT bits, tree, C, samples."""
    text = b"" if name == "empty" else TEXTS[name]()
    idx = V.VlgIndex.build(text)
    want = oracle.Index.from_text(text).parts()
    assert_parts_equal(idx.export_parts(), want)


def test_device_build_rejects_zero_byte(V):
    with pytest.raises(V.VlgError) as e:
        V.VlgIndex.build(b"ab\0cd")
    assert e.value.status == 5


def test_device_build_larger_dna(V, oracle):
    text = dna_text(1 << 20, 77).tobytes()
    idx = V.VlgIndex.build(text)
    assert_parts_equal(idx.export_parts(), oracle.Index.from_text(text).parts())


@pytest.mark.parametrize("name", ["abracadabra", "100a", "dna_50k", "zipf40", "all_symbols"])
def test_k2_k3_primitives(torch_cuda, V, oracle, name):
    torch = torch_cuda
    text = TEXTS[name]()
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.from_parts(o.parts())
    n = o.n
    rng = np.random.default_rng(5)
    L = V.lib()
    # wt_pc::rank
    m = 4000
    pos = rng.integers(0, n + 1, m).astype(np.uint64)
    syms = rng.choice(np.unique(np.concatenate([np.frombuffer(text, np.uint8), [0, 254]])), m).astype(np.uint8)
    d_pos, d_sym = dev_u64(torch, pos), torch.from_numpy(syms).cuda()
    d_out = torch.zeros_like(d_pos)
    V.capi.check(L.vlg_wt_rank_batch(idx._h, d_pos.data_ptr(), d_sym.data_ptr(), d_out.data_ptr(), m, None))
    torch.cuda.synchronize()
    want = np.array([o.wt_rank(int(p), int(c)) for p, c in zip(pos, syms)], dtype=np.uint64)
    assert (host_u64(d_out) == want).all()
    # csa[i] for every i (small) or a sample
    ii = np.arange(n, dtype=np.uint64) if n <= 60000 else rng.integers(0, n, 60000).astype(np.uint64)
    d_i = dev_u64(torch, ii)
    d_o = torch.zeros_like(d_i)
    V.capi.check(L.vlg_sa_batch(idx._h, d_i.data_ptr(), d_o.data_ptr(), len(ii), None))
    torch.cuda.synchronize()
    sa = oracle.suffix_array(np.frombuffer(text + b"\0", np.uint8))
    assert (host_u64(d_o) == sa[ii.astype(np.int64)]).all()
    # backward_search + locate of a pattern batch
    pats = [text[s:s + k] for s, k in zip(rng.integers(0, max(1, len(text) - 9), 300), rng.integers(1, 9, 300))]
    pats += [b"\xfe\xfe", b"zq", text, text + b"x", text[:1]]
    blob = np.frombuffer(b"".join(pats) + b"\0", np.uint8)
    off = np.concatenate([[0], np.cumsum([len(p) for p in pats])]).astype(np.uint64)
    d_blob, d_off = torch.from_numpy(blob.copy()).cuda(), dev_u64(torch, off)
    d_l, d_r = torch.zeros(len(pats), dtype=torch.int64, device="cuda"), torch.zeros(len(pats), dtype=torch.int64, device="cuda")
    V.capi.check(L.vlg_backward_search_batch(idx._h, d_blob.data_ptr(), d_off.data_ptr(), len(pats), d_l.data_ptr(), d_r.data_ptr(), None))
    torch.cuda.synchronize()
    l, r = host_u64(d_l), host_u64(d_r)
    ref = [o.backward_search(p) for p in pats]
    assert [(int(a), int(b)) for a, b in zip(l, r)] == [(x[1], x[2]) for x in ref]
    occ = (r + np.uint64(1) - l).astype(np.uint64)
    ooff = np.concatenate([[0], np.cumsum(occ)]).astype(np.uint64)
    total = int(ooff[-1])
    d_ooff = dev_u64(torch, ooff)
    d_loc = torch.zeros(max(total, 1), dtype=torch.int64, device="cuda")
    V.capi.check(L.vlg_locate_batch(idx._h, d_l.data_ptr(), d_r.data_ptr(), d_ooff.data_ptr(), len(pats), total, d_loc.data_ptr(), None))
    torch.cuda.synchronize()
    loc = host_u64(d_loc)
    for j, p in enumerate(pats):
        assert (loc[int(ooff[j]): int(ooff[j + 1])] == o.locate(p)).all(), p      # SA order, like the reference


def test_known_answers_on_gpu(V):
    cases = json.load(open(GOLD))["cases"]
    for c in cases:
        idx = V.VlgIndex.build(c["text"].encode())
        if "error" in c:
            with pytest.raises(V.VlgError) as e:
                V.locate(idx, c["query"])
            assert e.value.status == 4 and c["error"] in str(e.value)
        else:
            assert V.locate(idx, c["query"]).tolist() == c["tuples"], c
            assert V.count(idx, c["query"]) == len(c["tuples"])


def random_queries(text, rng, nq, kmax=5, mmax=5, gapmax=60, gaplo=30):
    qs = []
    for _ in range(nq):
        k = int(rng.integers(1, kmax + 1))
        subs = []
        for _ in range(k):
            s = int(rng.integers(0, len(text) - mmax - 1))
            subs.append(text[s:s + int(rng.integers(1, mmax + 1))])
        q = subs[0].decode("latin-1")
        for sp in subs[1:]:
            a = int(rng.integers(0, gaplo))
            q += ".{%d,%d}?%s" % (a, a + int(rng.integers(0, gapmax)), sp.decode("latin-1"))
        qs.append(q)
    return qs


@pytest.mark.parametrize("name,seed", [("dna_50k", 1), ("dna_skew", 2), ("zipf40", 3), ("100a", 4)])
def test_search_batch_vs_oracle(V, oracle, name, seed):
    text = TEXTS[name]()
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.build(text)
    rng = np.random.default_rng(seed)
    qs = random_queries(text, rng, 400)
    qs += ["\xfe.{0,5}?" + qs[0][:1], qs[1][:1] + ".{0,5}?\xfe"]            # empty first / last list
    res = idx.search(qs)
    total, chk, occ = 0, 0, np.zeros(4, dtype=np.uint64)
    for i, q in enumerate(qs):
        want = o.search(q, stats=occ)
        got = res.tuples(i)
        assert got.tolist() == want.tolist(), q
        assert (res.positions(i) == want[:, 0]).all() if len(want) else len(res.positions(i)) == 0
        total += len(want)
        chk = (chk + int(want[:, 0].sum())) & (2 ** 64 - 1) if len(want) else chk
    s = res.summary
    assert s["n_matches"] == total and s["checksum"] == chk
    assert s["logical_occurrences"] == int(occ[0]) and s["located_occurrences"] <= int(occ[0])
    assert s["wt_levels_bsearch"] == int(occ[3])
    # without interval sharing the GPU walks exactly the LF steps / tree levels the reference path walks
    from vlg_matching_amd.index import Workspace
    ws = Workspace()
    ws.set_option("dedup", 0)
    res2 = idx.search(qs, workspace=ws)
    s2 = res2.summary
    assert s2["located_occurrences"] == int(occ[0]) and s2["lf_steps"] == int(occ[1]) and s2["wt_levels_locate"] == int(occ[2])
    assert s2["n_matches"] == total and s2["checksum"] == chk
    for x, y in zip(res.fetch(), res2.fetch()):
        assert (x == y).all()


def test_search_benchmark_dialect_and_bad_queries(V, oracle):
    text = TEXTS["dna_50k"]()
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.build(text)
    qs = ["ACG.{0,50}TT.{0,50}GA", "A.{3,9}C", "TTT", "AC.{5,1}GT", "ACGT.{0,10}", "G.{0,4}?C"]
    res = idx.search(qs, dialect=V.capi.DIALECT_BENCHMARK, strict=False)
    for i, q in enumerate(qs):
        try:
            want = o.search(q, dialect=1)
        except oracle.ParseError:
            want = np.zeros((0, 1), np.uint64)
        assert res.tuples(i).tolist() == want.tolist() or (len(want) == 0 and int(res.counts[i]) == 0), q


@pytest.mark.parametrize("cap_mb", [48])
def test_search_chunked_equals_unchunked(V, cap_mb):
    """A tiny workspace forces many chunks; results must not change."""
    text = TEXTS["dna_50k"]()
    idx = V.VlgIndex.build(text)
    qs = random_queries(text, np.random.default_rng(9), 300, kmax=3, mmax=3)
    from vlg_matching_amd.index import Workspace
    wa, wb = Workspace(), Workspace(max_hbm_bytes=(cap_mb << 20))
    wb.set_option("reserve", cap_mb << 20)                # scratch allocated up front instead of on first use
    a = idx.search(qs, workspace=wa)
    b = idx.search(qs, workspace=wb)
    assert b.summary["n_chunks"] > a.summary["n_chunks"]
    for k in ("n_matches", "checksum", "n_tuple_values", "logical_occurrences"):
        assert a.summary[k] == b.summary[k], k
    fa, fb = a.fetch(), b.fetch()
    for x, y in zip(fa, fb):
        assert (x == y).all()


@pytest.mark.parametrize("name,seed", [("dna_50k", 21), ("zipf40", 22), ("100a", 23)])
def test_locate_sorted_sweep_equals_random_access_kernel(V, oracle, name, seed):
    """The sorted LF sweep (K3s) and the persistent random-access kernel (K3) must agree with each other and the oracle,
    including LF-step and tree-level counts (the algorithmic-bytes accounting)."""
    from vlg_matching_amd.index import Workspace
    text = TEXTS[name]()
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.build(text)
    qs = random_queries(text, np.random.default_rng(seed), 300, kmax=4, mmax=4)
    ws_a, ws_b = Workspace(), Workspace()
    for ws in (ws_a, ws_b):
        ws.set_option("dedup", 0)
    ws_a.set_option("sweep", 0)
    ws_b.set_option("sweep_min", 1)
    ws_b.set_option("sweep_tail", 100)
    a, b = idx.search(qs, workspace=ws_a), idx.search(qs, workspace=ws_b)
    for k in ("n_matches", "checksum", "located_occurrences", "lf_steps", "wt_levels_locate"):
        assert a.summary[k] == b.summary[k], k
    for x, y in zip(a.fetch(), b.fetch()):
        assert (x == y).all()
    occ = np.zeros(4, dtype=np.uint64)
    for i, q in enumerate(qs):
        assert b.tuples(i).tolist() == o.search(q, stats=occ).tolist()
    assert b.summary["lf_steps"] == int(occ[1]) and b.summary["wt_levels_locate"] == int(occ[2])
    assert "locate_partition" in ws_b.kernel_stats() and ws_b.kernel_stats()["locate_partition"]["launches"] > 0


def _bin(name):
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vlg_matching_amd", "bin", name)


def test_cpp_example_prints_reference_answers(V):
    """vlg_matching_example mirrors examples/vlg_matching.cpp:10-22; expected values are the reference's known answers."""
    import subprocess
    out = subprocess.run([_bin("vlg_matching_example")], capture_output=True, text=True, check=True).stdout
    want = """
count(ac.{2,5}?a.{4,8}?b)=1
locate(ac.{2,5}?a.{4,8}?b)=
  1. occ starting at position 3
     Subpattern positions: 3 10 18

count(a.{0,10}?a.{0,10}?a)=2
locate(a.{0,10}?a.{0,10}?a)=
  1. occ starting at position 0
     Subpattern positions: 0 3 5
  2. occ starting at position 7
     Subpattern positions: 7 10 15

count(foo.{0,10}?bar)=0
locate(foo.{0,10}?bar)=

count(97 99 .{2,5}? 97 .{4,8}? 98)=1
locate(97 99 .{2,5}? 97 .{4,8}? 98)=
  1. occ starting at position 3
     Subpattern positions: 3 10 18

count(97 .{0,10}? 97 .{0,10}? 97)=2
locate(97 .{0,10}? 97 .{0,10}? 97)=
  1. occ starting at position 0
     Subpattern positions: 0 3 5
  2. occ starting at position 7
     Subpattern positions: 7 10 15

count(1337 .{0,10}? 42)=0
locate(1337 .{0,10}? 42)=
"""
    assert out == want


def test_cpp_example_lazy_iterator_on_a_file(V, oracle, tmp_path):
    """vlg_matching_example <file> <query>: more matches than the iterator's first request (16), so it asks again (x4) on the way;
    the printed tuples equal the oracle's."""
    import subprocess
    text = dna_text(4000, 77).tobytes()
    (tmp_path / "t.txt").write_bytes(text)
    q = "AC.{0,9}?G"
    out = subprocess.run([_bin("vlg_matching_example"), str(tmp_path / "t.txt"), q], capture_output=True, text=True, check=True).stdout
    want = oracle.Index.from_text(text).search(q).tolist()
    assert len(want) > 100
    assert ("count(%s)=%d" % (q, len(want))) in out
    got = [[int(x) for x in l.split(":")[1].split()] for l in out.splitlines() if "Subpattern positions" in l]
    assert got == want


def test_c1_plumbing(V, oracle, tmp_path):
    """BASELINE config 1: 2^20 characters of uniform DNA (seed 1), the three queries of
    examples/vlg_matching.cpp:40-42 (lower case: no match on this text) and ACGTA.{0,100}?TTGCA -- through the example binary
    given the text as a file, through VlgIndex.search (FM-index path) and through the paper's index (vlg_wtsa_*), all equal to
    the oracle; the one query that matches is also checked against a plain scan of the text."""
    import subprocess
    from vlg_matching_amd import workload
    from vlg_matching_amd.index import WtsaIndex
    cfg = workload.config("C1")
    assert cfg["kind"] == "dna" and cfg["n"] == 1 << 20 and cfg["seed"] == 1
    text = workload.gen_text(cfg["kind"], cfg["n"], cfg["seed"]).tobytes()
    assert len(text) == 1 << 20 and set(text) == set(b"ACGT")
    queries = ["ac.{2,5}?a.{4,8}?b", "a.{0,10}?a.{0,10}?a", "foo.{0,10}?bar", "ACGTA.{0,100}?TTGCA"]
    o = oracle.Index.from_text(text)
    want = [o.search(q).tolist() for q in queries]
    assert [len(w) for w in want[:3]] == [0, 0, 0] and len(want[3]) > 50
    # independent truth for the matching query: left-most, lazy, non-overlapping pairs from a scan of the text
    a = np.array([i for i in range(len(text) - 4) if text[i:i + 5] == b"ACGTA"], dtype=np.int64)
    b = np.array([i for i in range(len(text) - 4) if text[i:i + 5] == b"TTGCA"], dtype=np.int64)
    scan, nxt = [], 0
    for x in a:
        if x < nxt:
            continue
        j = np.searchsorted(b, x + 5)
        if j < len(b) and b[j] <= x + 5 + 100:
            scan.append([int(x), int(b[j])])
            nxt = int(b[j]) + 5
    assert scan == want[3]
    (tmp_path / "c1.txt").write_bytes(text)
    out = subprocess.run([_bin("vlg_matching_example"), str(tmp_path / "c1.txt")] + queries, capture_output=True, text=True, check=True).stdout
    blocks = out.split("\ncount(")[1:]
    assert len(blocks) == len(queries)
    for q, w, blk in zip(queries, want, blocks):
        assert blk.startswith("%s)=%d\n" % (q, len(w)))
        got = [[int(x) for x in l.split(":")[1].split()] for l in blk.splitlines() if "Subpattern positions" in l]
        first = [int(l.rsplit(" ", 1)[1]) for l in blk.splitlines() if "occ starting at position" in l]
        assert got == w and first == [t[0] for t in w]
    idx = V.VlgIndex.build(text)
    res = idx.search(queries)
    for i, w in enumerate(want):
        assert res.tuples(i).tolist() == w
        assert V.count(idx, queries[i]) == len(w)
    wres = WtsaIndex(text).search(queries)
    for i, w in enumerate(want):
        assert wres.tuples(i).tolist() == w


def test_cpp_driver_pipeline_matches_oracle(V, oracle, tmp_path):
    """gm_index_gpu + gm_search_gpu on a generated collection: the machine-readable lines of gm_search.cpp:142-160."""
    import subprocess
    raw = dna_text(200000, 31).tobytes().replace(b"GATTACA", b"GAT\nACA")     # newline -> space like create_collection
    (tmp_path / "raw.txt").write_bytes(raw)
    col = str(tmp_path / "col")
    built = subprocess.run([_bin("gm_index_gpu"), "-c", col, "-i", str(tmp_path / "raw.txt"), "-s"], check=True, capture_output=True, text=True).stdout
    text = raw.replace(b"\n", b" ")
    # -s: the same index in the reference's own csa_wt<wt_huff<>,32,64> format
    sdsl_file = [l.split(" = ")[1] for l in built.splitlines() if l.startswith("# sdsl_file = ")][0]
    assert_parts_equal(V.VlgIndex.load_sdsl(sdsl_file).export_parts(), V.VlgIndex.build(text).export_parts())
    rng = np.random.default_rng(8)
    pats = []
    for _ in range(200):
        k = int(rng.integers(2, 4))
        subs = [text[s:s + 5].decode() for s in rng.integers(0, len(text) - 6, k)]
        pats.append((".{%d,%d}" % (0, int(rng.integers(10, 300)))).join(subs))
    pats.insert(7, "AC.{9,3}GT")                                                  # unparsable line is skipped
    (tmp_path / "pats.txt").write_text("\n".join(pats) + "\n", encoding="latin-1")
    out = subprocess.run([_bin("gm_search_gpu"), "-c", col, "-p", str(tmp_path / "pats.txt")], check=True, capture_output=True, text=True).stdout
    kv = dict(l[2:].split(" = ") for l in out.splitlines() if l.startswith("# ") and " = " in l)
    o = oracle.Index.from_text(text)
    n, chk = 0, 0
    for p in pats:
        try:
            t = o.search(p, dialect=1)
        except oracle.ParseError:
            continue
        n += len(t)
        chk = (chk + int(t[:, 0].sum())) % (1 << 64) if len(t) else chk
    assert int(kv["num_results"]) == n and int(kv["checksum"]) == chk
    assert int(kv["num_patterns"]) == len(pats) - 1
    for key in ("total_time_mus", "min_time_mus", "qrt_1st_time_mus", "mean_time_mus", "median_time_mus", "qrt_3rd_time_mus", "max_time_mus"):
        assert key in kv
    # batch mode: one clock around the batch, shared out over the queries by their work (1 + sum of occurrence counts, 1 for a query
    # with an empty list) -- the TIMING lines follow the pattern file's order, add up to the batch time and say how they were made
    timing = [int(l.split(" = ")[1]) for l in out.splitlines() if l.startswith("TIMING = ")]
    assert kv["timing_mode"] == "batch_apportioned_by_occurrences" and len(timing) == len(pats) - 1
    w = []
    for p in pats:
        try:
            subs = oracle.query_fields(oracle.parse(p, 1))[0]
        except oracle.ParseError:
            continue
        occ = [o.backward_search(sp)[0] for sp in subs]
        w.append(1 + (0 if min(occ) == 0 else sum(occ)))
    total = int(kv["total_time_mus"])
    assert total - len(timing) <= sum(timing) <= total
    assert all(abs(t - total * wi / sum(w)) <= 1 for t, wi in zip(timing, w))
    assert int(kv["max_time_mus"]) == max(timing) and int(kv["min_time_mus"]) == min(timing)
    out1 = subprocess.run([_bin("gm_search_gpu"), "-c", col, "-p", str(tmp_path / "pats.txt"), "-1"], check=True, capture_output=True, text=True).stdout
    kv1 = dict(l[2:].split(" = ") for l in out1.splitlines() if l.startswith("# ") and " = " in l)
    assert kv1["num_results"] == kv["num_results"] and kv1["checksum"] == kv["checksum"] and kv1["timing_mode"] == "per_query_clock"
    # -g N: the pattern file sharded over N device slices (index replicated by a peer copy, slices cut by work, one host thread
    # each; on a one-GPU box the slices share the device, which still runs replicate + shard + merge)
    for g in ("2", "3"):
        outg = subprocess.run([_bin("gm_search_gpu"), "-c", col, "-p", str(tmp_path / "pats.txt"), "-g", g], check=True, capture_output=True, text=True).stdout
        kvg = dict(l[2:].split(" = ") for l in outg.splitlines() if l.startswith("# ") and " = " in l)
        assert kvg["num_results"] == kv["num_results"] and kvg["checksum"] == kv["checksum"] and kvg["num_gpus"] == g
    # -g 1 -P: one PROCESS per GPU over RCCL (forked before any GPU call; vlg_comm_create from a unique id passed through a pipe made before the fork,
    # vlg_index_broadcast, counters through vlg_comm_allreduce_sum_u64) -- with the one rank a one-GPU box can host
    outp = subprocess.run([_bin("gm_search_gpu"), "-c", col, "-p", str(tmp_path / "pats.txt"), "-g", "1", "-P"], check=True, capture_output=True,
                          text=True, timeout=300).stdout
    kvp = dict(l[2:].split(" = ") for l in outp.splitlines() if l.startswith("# ") and " = " in l)
    assert kvp["num_results"] == kv["num_results"] and kvp["checksum"] == kv["checksum"] and kvp["gpu_mode"] == "processes"
    # -S: the loaded t_dens = 32 index expanded to the whole suffix array in HBM (index_fm_gpu::keep_suffix_array), alone, sharded
    # over device slices and in the process-per-GPU mode: same lines
    for extra in ([], ["-g", "2"], ["-g", "1", "-P"]):
        outs = subprocess.run([_bin("gm_search_gpu"), "-c", col, "-p", str(tmp_path / "pats.txt"), "-S"] + extra, check=True, capture_output=True,
                              text=True, timeout=300).stdout
        kvs = dict(l[2:].split(" = ") for l in outs.splitlines() if l.startswith("# ") and " = " in l)
        assert kvs["num_results"] == kv["num_results"] and kvs["checksum"] == kv["checksum"], extra


@pytest.mark.parametrize("name,seed,tail", [("dna_50k", 71, 16), ("zipf40", 72, 1), ("100a", 73, 4), ("dna_skew", 74, 1000), ("abracadabra", 75, 1)])
def test_locate_trail_sharing_equals_plain_locate_and_oracle(V, oracle, name, seed, tail):
    """Sorted sweep with shared LF trails (an element stops where another one has stood and takes its position plus the
    distance) against the random-access kernel and the oracle; nested SA intervals put the same index into two lists."""
    from vlg_matching_amd.index import Workspace
    text = TEXTS[name]()
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.build(text)
    rng = np.random.default_rng(seed)
    qs = random_queries(text, rng, 300, kmax=4, mmax=4)
    t = text.decode("latin-1")
    qs += [t[:1], t[:2], t[:3], t[1:2], t[:1] + ".{0,9}?" + t[:2], t[-1:], t[-2:]]           # prefixes of each other: nested intervals
    ws_a, ws_b, ws_c = Workspace(), Workspace(), Workspace()
    ws_a.set_option("sweep", 0)
    for ws in (ws_b, ws_c):
        ws.set_option("sweep_min", 1)
        ws.set_option("sweep_tail", tail)
    ws_c.set_option("trail", 0)
    ws_b.set_option("global_sort_min", 1)                 # one radix sort of (list, position) keys instead of a sort per list
    a, b, c = idx.search(qs, workspace=ws_a), idx.search(qs, workspace=ws_b), idx.search(qs, workspace=ws_c)
    assert b.summary["lf_steps"] <= c.summary["lf_steps"] == a.summary["lf_steps"]
    for k in ("n_matches", "checksum", "n_tuple_values", "located_occurrences", "logical_occurrences"):
        assert a.summary[k] == b.summary[k] == c.summary[k], k
    for x, y, z in zip(a.fetch(), b.fetch(), c.fetch()):
        assert (x == y).all() and (x == z).all()
    for i in list(range(0, len(qs), 6)) + list(range(len(qs) - 7, len(qs))):
        assert b.tuples(i).tolist() == o.search(qs[i]).tolist(), qs[i]


@pytest.mark.parametrize("name,seed,tail,force", [("dna_50k", 81, 16, ""), ("zipf40", 82, 1, ""), ("100a", 83, 4, ""), ("dna_skew", 84, 1 << 30, ""),
                                                  ("zipf40", 86, 64, "2"), ("dna_50k", 87, 1, "2")])
def test_locate_by_unsampling_equals_sweep_walks_and_oracle(V, oracle, monkeypatch, name, seed, tail, force):
    """K3u: a dense batch rebuilds the whole suffix array from the SA samples -- one walker per sample, every SA index visited once,
    n - n_samples LF steps whatever the batch -- and copies its intervals out of it.  Same positions, tuples and checksums as the
    sorted sweep with shared trails, the per-occurrence walks and the oracle; on plain and rrr bit-vectors, with narrow and (force =
    "2": BASELINE config 4's mix) 33-bit SA indices, with the walkers finished by the sorted rounds or by the stragglers' kernel."""
    from vlg_matching_amd.index import Workspace
    if force:
        monkeypatch.setenv("VLG_FORCE_POS64", force)
    text = TEXTS[name]()
    o = oracle.Index.from_text(text)
    rng = np.random.default_rng(seed)
    qs = random_queries(text, rng, 300, kmax=4, mmax=4)
    t = text.decode("latin-1")
    qs += [t[:1], t[:2], t[:3], t[1:2], t[:1] + ".{0,9}?" + t[:2], t[-1:], t[-2:]]
    want = [o.search(q).tolist() for q in qs]
    plain = V.VlgIndex.build(text)
    for idx in (plain, plain.compress(), V.VlgIndex.build(text, dens=5)):
        ws_w, ws_s, ws_u = Workspace(), Workspace(), Workspace()
        ws_w.set_option("sweep", 0)
        for ws in (ws_s, ws_u):
            ws.set_option("sweep_min", 1)
            ws.set_option("sweep_tail", 16)
        ws_s.set_option("unsample_pct", 0)
        for k_, v_ in (("unsample_pct", 1), ("unsample_min", 1), ("unsample_tail", tail)):
            ws_u.set_option(k_, v_)
        w, s_, u = idx.search(qs, workspace=ws_w), idx.search(qs, workspace=ws_s), idx.search(qs, workspace=ws_u)
        assert (w.summary["locate_mode"], s_.summary["locate_mode"], u.summary["locate_mode"]) == (V.capi.LOCATE_WALKS, V.capi.LOCATE_SWEEP, V.capi.LOCATE_UNSAMPLE)
        info = idx.info()
        # LF is a permutation of the SA indices and every walker ends ON a sampled index: each index is reached by exactly one step
        assert u.summary["lf_steps"] == info["n"] or info["sigma"] == 1
        st = ws_u.kernel_stats()
        assert st["locate_resolve"]["launches"] == 0 and st["locate"]["launches"] >= 2
        for k in ("n_matches", "checksum", "n_tuple_values", "located_occurrences", "logical_occurrences"):
            assert w.summary[k] == s_.summary[k] == u.summary[k], k
        for x, y, z in zip(w.fetch(), s_.fetch(), u.fetch()):
            assert (x == y).all() and (x == z).all()
        for i in range(len(qs)):
            assert u.tuples(i).tolist() == want[i], qs[i]


def test_first_positions_only(V, oracle):
    """Workspace option "tuples" = 0: what the benchmark's gapped_search_result holds (index_sasearch.hpp:58-118) -- counts,
    first positions and checksum as with tuples, no tuple values, and a tuples buffer is refused."""
    from vlg_matching_amd.index import Workspace
    from vlg_matching_amd.capi import VlgError
    text = TEXTS["dna_50k"]()
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.build(text)
    qs = random_queries(text, np.random.default_rng(77), 200, kmax=4, mmax=4)
    ws_a, ws_b = Workspace(), Workspace()
    ws_b.set_option("tuples", 0)
    a, b = idx.search(qs, workspace=ws_a), idx.search(qs, workspace=ws_b)
    assert b.summary["n_tuple_values"] == 0 and a.summary["n_tuple_values"] > 0
    for k in ("n_matches", "checksum", "located_occurrences", "logical_occurrences"):
        assert a.summary[k] == b.summary[k], k
    assert (a.counts == b.counts).all()
    for i in range(len(qs)):
        assert b.positions(i).tolist() == [t[0] for t in o.search(qs[i]).tolist()], qs[i]
    assert b.summary["n_matches"] > 0
    with pytest.raises(VlgError):
        b.tuples(0)
    from vlg_matching_amd.capi import lib, E_INVALID
    buf = np.zeros(8, dtype=np.uint64)
    assert lib().vlg_result_fetch(b._h, None, None, None, buf.ctypes.data) == E_INVALID


@pytest.mark.parametrize("pivot", [1, 0])
@pytest.mark.parametrize("name,seed,kmax,gapmax,cap_mb", [("dna_50k", 61, 3, 300, 0), ("dna_skew", 62, 6, 40, 0), ("zipf40", 63, 4, 2000, 0),
                                                         ("100a", 64, 3, 5, 0), ("dna_50k", 65, 8, 600, 0), ("dna_50k", 66, 3, 300, 40)])
def test_window_filter_equals_unfiltered_join_and_oracle(V, oracle, name, seed, kmax, gapmax, cap_mb, pivot):
    """Dropping the list elements whose gap windows are empty (streaming sweeps over block bitmaps, or outwards from the
    shortest list of the query) changes no match."""
    from vlg_matching_amd.index import Workspace
    text = TEXTS[name]()
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.build(text)
    rng = np.random.default_rng(seed)
    qs = random_queries(text, rng, 250, kmax=kmax, mmax=3, gapmax=gapmax, gaplo=8)
    qs += random_queries(text, rng, 60, kmax=2, mmax=1, gapmax=3, gaplo=2)
    qs += ["\xfe.{0,5}?" + qs[0][:1], qs[1][:1] + ".{0,5}?\xfe", qs[2][:1]]
    ws_n = Workspace()
    ws_f = Workspace(max_hbm_bytes=(cap_mb << 20)) if cap_mb else Workspace()
    ws_n.set_option("filter", 0)
    ws_f.set_option("filter_min", 0)
    ws_f.set_option("filter_stream_min", 0)
    ws_f.set_option("filter_pivot", pivot)
    ws_f.set_option("pivot_rungs", 2 if seed % 2 else 0)                 # index ranges through the ladder / by bisection
    ws_f.set_option("global_sort_min", 1 if pivot else 1 << 40)      # all lists sorted at once / one sort per list ...
    ws_f.set_option("list_sort", seed % 2)                           # ... when the sort inside every list is off
    if seed in (62, 65):
        ws_f.set_option("filter_group_bytes", 1 << 16)                # many small filter groups inside one batch
    a, b = idx.search(qs, workspace=ws_n), idx.search(qs, workspace=ws_f)
    assert not [k for k, v in ws_n.kernel_stats().items() if k.startswith("filter_") and v["launches"]]
    assert ws_f.kernel_stats()["filter_compact"]["launches"] > 0
    assert 0 < b.summary["join_slots"] <= a.summary["join_slots"] + 64 * kmax * b.summary["n_chunks"]      # (+ class alignment per chunk)
    for k in ("n_matches", "checksum", "n_tuple_values", "logical_occurrences", "located_occurrences"):
        assert a.summary[k] == b.summary[k], k
    for x, y in zip(a.fetch(), b.fetch()):
        assert (x == y).all()
    for i in list(range(0, len(qs), 5)) + [len(qs) - 3, len(qs) - 2, len(qs) - 1]:
        assert b.tuples(i).tolist() == o.search(qs[i]).tolist(), qs[i]
    for dialect in (V.capi.DIALECT_BENCHMARK,):
        bq = [x.replace("?", "") for x in qs[:120]]
        ra = idx.search(bq, dialect=dialect, strict=False, workspace=ws_n)
        rb = idx.search(bq, dialect=dialect, strict=False, workspace=ws_f)
        for x, y in zip(ra.fetch(), rb.fetch()):
            assert (x == y).all()


def test_window_filter_with_nothing_to_mark(V, oracle):
    """Two lists, the first one short: it would be the pivot (kept whole) and the last list is never filtered, so the filter has
    nothing to mark for such a query -- alone in a batch (once: zero-sized launches) and next to a query that is filtered."""
    from vlg_matching_amd.index import Workspace
    rng = np.random.default_rng(77)
    text = bytes(rng.choice(np.frombuffer(b"aaaaaaab", np.uint8), 60000)) + b"qaab" + bytes(rng.choice(np.frombuffer(b"ab", np.uint8), 500)) + b"q"
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.build(text)
    for qs in (["q.{0,300}?a"], ["q.{0,300}?a", "q.{0,9}?b", "b.{0,40}?a.{0,40}?q", "q"]):
        ws = Workspace()
        for k_, v_ in (("filter_min", 0), ("filter_stream_min", 0), ("filter_pivot", 1), ("filter_pivot_ratio", 1)):
            ws.set_option(k_, v_)
        r = idx.search(qs, workspace=ws)
        for i, qy in enumerate(qs):
            assert r.tuples(i).tolist() == o.search(qy).tolist(), qy


@pytest.mark.parametrize("pivot", [1, 0])
def test_window_filter_extreme_gaps(V, oracle, pivot):
    """Windows wider than the text, windows that start beyond it, exact-distance windows, many sub-patterns."""
    from vlg_matching_amd.index import Workspace
    text = TEXTS["dna_50k"]()
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.build(text)
    t = text.decode()
    a, b, c = t[100:104], t[30000:30003], t[49000:49005]
    qs = [a + ".{0,100000000}?" + b, a + ".{0,4000000000000000000}?" + b + ".{0,7}?" + c, a + ".{60000,70000}?" + b,
          a + ".{29890,29900}?" + b, b + ".{18990,19005}?" + c, "A.{0,0}?C.{0,0}?G", "AC.{5,5}?GT.{5,5}?AC.{5,5}?GT",
          ".{1,2}?".join(["A", "C", "G", "T"] * 8), "T.{49990,49999}?A", c + ".{0,3}?" + a, a + ".{0,49999}?" + a + ".{0,49999}?" + a]
    ws_n, ws_f = Workspace(), Workspace()
    ws_n.set_option("filter", 0)
    ws_f.set_option("filter_min", 0)
    ws_f.set_option("filter_stream_min", 0)
    ws_f.set_option("filter_pivot", pivot)
    ws_f.set_option("pivot_rungs", 2)
    ra, rb = idx.search(qs, workspace=ws_n), idx.search(qs, workspace=ws_f)
    assert ws_f.kernel_stats()["filter_compact"]["launches"] > 0
    for i, q in enumerate(qs):
        want = o.search(q).tolist()
        assert ra.tuples(i).tolist() == want, q
        assert rb.tuples(i).tolist() == want, q


def test_join_many_tiles_single_pattern(V, oracle):
    """k = 1 and k = 2 on a list spanning many tiles (non-overlap chains cross tile borders all the time)."""
    text = (b"ab" * 30000) + dna_text(5000, 3).tobytes() + (b"aab" * 9000)
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.build(text)
    from vlg_matching_amd.index import Workspace
    qs = ["a", "ab", "aba", "a.{0,0}?b", "a.{1,3}?a", "b.{0,5}?a.{0,2}?b", "ab.{2,2}?ab", "a.{0,1}?a.{0,1}?a.{0,1}?a"]
    for filt in (0, 1):
        ws = Workspace()
        ws.set_option("filter", filt)
        ws.set_option("filter_min", 0)
        ws.set_option("filter_stream_min", 0)
        res = idx.search(qs, workspace=ws)
        for i, qq in enumerate(qs):
            assert res.tuples(i).tolist() == o.search(qq).tolist(), (qq, filt)


@pytest.mark.parametrize("nbits", [0, 1, 62, 63, 64, 2015, 2016, 2017, 4096, 100000, 1 << 20])
def test_k6_rrr_bitvector_rank(torch_cuda, V, oracle, nbits):
    """rank_support_test.cpp:70-87 for rank_support_rrr<1,63>: device decode vs the oracle's rrr restatement and the truth."""
    torch = torch_cuda
    rng = np.random.default_rng(nbits + 7)
    for dens in (0.0, 0.03, 0.5, 0.97, 1.0):
        bits = rng.random(nbits) < dens
        words = bits_to_words(bits)
        bv = V.RrrBitVector(words, nbits)
        idx = np.arange(nbits + 1, dtype=np.uint64) if nbits <= 5000 else \
            np.unique(np.concatenate([rng.integers(0, nbits + 1, 20000), [0, nbits, 63, 62, 2016, nbits - 1]])).astype(np.uint64)
        d_idx = dev_u64(torch, idx)
        d_out = torch.zeros_like(d_idx)
        bv.rank_device(d_idx.data_ptr(), d_out.data_ptr(), len(idx))
        torch.cuda.synchronize()
        got = host_u64(d_out)
        truth = np.concatenate([[0], np.cumsum(bits)])[idx.astype(np.int64)]
        assert (got == truth).all(), (nbits, dens)
        want, _ = oracle.rrr_rank(words, nbits, idx[:1500])
        assert (got[:1500] == want).all()
        if dens in (0.03, 0.97) and nbits >= 100000:
            assert bv.hbm_bytes() < 0.6 * (nbits / 8)          # H0 compression actually happens on skewed bits


@pytest.mark.parametrize("name,seed", [("dna_50k", 61), ("zipf40", 62), ("100a", 63), ("all_symbols", 64), ("abracadabra", 65)])
def test_rrr_index_variant_equals_plain_and_oracle(torch_cuda, V, oracle, name, seed):
    """BASELINE config 5 shape: csa_wt<wt_huff<rrr_vector<63>>>.  Same answers from compressed bit-vectors (K6 inside K2/K3)."""
    torch = torch_cuda
    from vlg_matching_amd.index import Workspace
    text = TEXTS[name]()
    o = oracle.Index.from_text(text)
    plain = V.VlgIndex.build(text)
    rrr = plain.compress()
    assert rrr.info()["bv_kind"] == 1 and plain.info()["bv_kind"] == 0
    n = o.n
    rng = np.random.default_rng(seed)
    L = V.lib()
    # wt_pc::rank and csa[i] through the rrr bit-vectors
    m = 3000
    pos = rng.integers(0, n + 1, m).astype(np.uint64)
    syms = rng.choice(np.unique(np.concatenate([np.frombuffer(text, np.uint8), [0, 254]])), m).astype(np.uint8)
    d_pos, d_sym = dev_u64(torch, pos), torch.from_numpy(syms).cuda()
    d_out = torch.zeros_like(d_pos)
    V.capi.check(L.vlg_wt_rank_batch(rrr._h, d_pos.data_ptr(), d_sym.data_ptr(), d_out.data_ptr(), m, None))
    torch.cuda.synchronize()
    assert (host_u64(d_out) == np.array([o.wt_rank(int(p), int(c)) for p, c in zip(pos, syms)], dtype=np.uint64)).all()
    ii = np.arange(n, dtype=np.uint64) if n <= 60000 else rng.integers(0, n, 60000).astype(np.uint64)
    d_i = dev_u64(torch, ii)
    d_o = torch.zeros_like(d_i)
    V.capi.check(L.vlg_sa_batch(rrr._h, d_i.data_ptr(), d_o.data_ptr(), len(ii), None))
    torch.cuda.synchronize()
    sa = oracle.suffix_array(np.frombuffer(text + b"\0", np.uint8))
    assert (host_u64(d_o) == sa[ii.astype(np.int64)]).all()
    # whole searches, both locate kernels
    qs = random_queries(text, rng, 200, kmax=4, mmax=4) if len(text) > 30 else ["a.{0,10}?a.{0,10}?a", "ac.{2,5}?a.{4,8}?b", "abra"]
    a = plain.search(qs)
    for opts in ({}, {"sweep_min": 1, "sweep_tail": 50, "trail": 0}, {"sweep_min": 1, "sweep_tail": 50}):
        ws = Workspace()
        for k_, v_ in opts.items():
            ws.set_option(k_, v_)
        b = rrr.search(qs, workspace=ws)
        shared = "sweep_min" in opts and "trail" not in opts        # shared LF trails: fewer steps, same positions
        for k in ("n_matches", "checksum", "located_occurrences", "wt_levels_bsearch") + (() if shared else ("lf_steps", "wt_levels_locate")):
            assert a.summary[k] == b.summary[k], (k, opts)
        assert not shared or b.summary["lf_steps"] <= a.summary["lf_steps"]
        for x, y in zip(a.fetch(), b.fetch()):
            assert (x == y).all()
    for i in range(0, len(qs), 9):
        assert a.tuples(i).tolist() == o.search(qs[i]).tolist()


def test_rrr_index_is_smaller_on_skewed_text(V):
    text = (b"a" * 2000000) + dna_text(3000, 1).tobytes() + (b"b" * 1000000)
    plain = V.VlgIndex.build(text)
    rrr = plain.compress()
    pi, ri = plain.info(), rrr.info()
    assert pi["hbm_bytes"] - ri["hbm_bytes"] > 0.5 * pi["n_blocks"] * 32      # the bit-vector part shrinks by more than half
    assert V.count(rrr, "aaaa.{0,5}?ab") == V.count(plain, "aaaa.{0,5}?ab")


def test_edge_cases_empty_inputs_and_long_queries(V, oracle):
    """Empty text, empty batch, all-dead batch, patterns longer than the text, 40 and 64 sub-patterns, 65 rejected."""
    # empty text: nothing ever matches
    e = V.VlgIndex.build(b"")
    r = e.search(["a", "a.{0,3}?b"])
    assert r.summary["n_matches"] == 0 and r.counts.tolist() == [0, 0]
    # empty batch and a batch in which every query dies in backward search
    idx = V.VlgIndex.build(b"a" * 100)
    assert idx.search([]).summary["n_queries"] == 0
    r = idx.search(["b", "a.{0,5}?b", "b.{0,5}?a", "a" * 101, "a" * 101 + ".{0,1}?a"])
    assert r.summary["n_matches"] == 0 and r.summary["located_occurrences"] == 0
    r = idx.search(["a" * 100 + ".{0,1}?a"])                     # both lists exist (1 and 100 occurrences) but no match fits
    assert r.summary["n_matches"] == 0 and r.summary["located_occurrences"] == 101
    # many sub-patterns (dense join path: k > 8)
    o = oracle.Index.from_text(b"a" * 100)
    for k in (40, 64):
        q = ".{0,1}?".join(["a"] * k)
        assert idx.search([q]).tuples(0).tolist() == o.search(q).tolist()
        assert V.count(idx, q) == len(o.search(q)) >= 1
    with pytest.raises(V.VlgError) as ex:
        idx.search([".{0,1}?".join(["a"] * 65)])
    assert ex.value.status == V.capi.E_INVALID
    # whole text as a pattern, and patterns running over the end
    t = b"abracadabrasimsalabim"
    idx2, o2 = V.VlgIndex.build(t), oracle.Index.from_text(t)
    for q in [t.decode(), t.decode() + "x", "bim", "bimx", "m.{0,0}?x", "abra.{3,3}?abra", "a.{0,100000}?m", "a.{100000,200000}?m"]:
        assert idx2.search([q]).tuples(0).tolist() == o2.search(q).tolist(), q


def test_load_index_stored_by_stock_sdsl(V, refmod, tmp_path):
    """SURVEY 8f-2 end to end: reference-serialised csa_wt<wt_huff<>> file -> vlg_index_load_sdsl -> same answers."""
    O = refmod
    from util import bwt_from_sa
    text = skewed_text(60000, 12).tobytes()
    tz = np.frombuffer(text + b"\0", dtype=np.uint8)
    sa = O.suffix_array(tz)
    path = tmp_path / "idx.sdsl"
    O.RefIndex(bwt_from_sa(tz, sa), sa, 0).write_csa_image(path, sa)
    idx = V.VlgIndex.load_sdsl(path)
    built = V.VlgIndex.build(text)
    assert_parts_equal(idx.export_parts(), built.export_parts())
    o = O.Index.from_text(text)
    qs = random_queries(text, np.random.default_rng(2), 120, kmax=3, mmax=3)
    res = idx.search(qs)
    for i, q in enumerate(qs):
        assert res.tuples(i).tolist() == o.search(q).tolist(), q


def test_load_rrr_index_stored_by_stock_sdsl(V, refmod, tmp_path):
    """BASELINE config 5's index type from a stock file: csa_wt<wt_huff<rrr_vector<63>>> serialised by the reference's own code
    (oracle/_ref) -> vlg_index_load_sdsl_kind(VLG_BV_RRR63) -> an rrr index on the device with the plain index's answers."""
    O = refmod
    from util import bwt_from_sa
    text = skewed_text(60000, 12).tobytes()
    tz = np.frombuffer(text + bytes(1), dtype=np.uint8)
    sa = O.suffix_array(tz)
    path = tmp_path / "idx.rrr.sdsl"
    O.RefIndex(bwt_from_sa(tz, sa), sa, 2).write_csa_image(path, sa)
    idx = V.VlgIndex.load_sdsl(path, rrr=True)
    assert idx.info()["bv_kind"] == 1
    built = V.VlgIndex.build(text)
    qs = random_queries(text, np.random.default_rng(2), 150, kmax=3, mmax=3)
    a, b = idx.search(qs), built.search(qs)
    for x, y in zip(a.fetch(), b.fetch()):
        assert (x == y).all()
    o = O.Index.from_text(text)
    for i in (0, 33, 149):
        assert a.tuples(i).tolist() == o.search(qs[i]).tolist()


@pytest.mark.parametrize("name", ["abracadabra", "one_byte", "100a", "all_symbols", "dna_50k", "zipf40"])
def test_suffix_array_device(torch_cuda, V, oracle, name):
    """The device suffix sorter on its own (what construct_sa computes): SA of text + sentinel."""
    torch = torch_cuda
    text = TEXTS[name]()
    tz = np.frombuffer(text + b"\0", np.uint8)
    d_text = torch.from_numpy(np.frombuffer(text, np.uint8).copy()).cuda()
    d_sa = torch.zeros(len(tz), dtype=torch.int32, device="cuda")
    V.capi.check(V.lib().vlg_suffix_array_device(d_text.data_ptr(), len(text), d_sa.data_ptr(), None))
    torch.cuda.synchronize()
    assert (d_sa.cpu().numpy().view(np.uint32).astype(np.uint64) == oracle.suffix_array(tz)).all()


@pytest.mark.parametrize("name", ["abracadabra", "one_byte", "100a", "all_symbols", "dna_20k", "dna_50k", "zipf40"])
def test_save_sdsl_is_loadable_by_the_reference(V, oracle, refmod, tmp_path, name):
    """SURVEY 8f-2, the writer: a device-built index stored with vlg_index_save_sdsl is (a) loaded member by member by the
    reference's own load() functions and answers access / rank / inverse_select / select / samples / alphabet correctly,
    (b) byte-identical to the image the reference's serialize() functions write where the reference builds its select
    structures the same way (bit-vectors below 100 000 bits), (c) read back by vlg_index_load_sdsl into the same index."""
    text = dna_text(20000, 5).tobytes() if name == "dna_20k" else TEXTS[name]()
    tz = np.frombuffer(text + b"\0", np.uint8)
    sa = oracle.suffix_array(tz)
    bwt = bwt_from_sa(tz, sa)
    idx = V.VlgIndex.build(text)
    path = tmp_path / "gpu.sdsl"
    idx.save_sdsl(path)
    assert oracle.ref_check_csa_image(path, bwt, sa, step=1 if len(tz) <= 30000 else 7) == 0
    isa = idx.isa_samples(64)
    assert (sa[isa.astype(np.int64)] == 64 * np.arange(len(isa), dtype=np.uint64)).all()
    ref_path = tmp_path / "ref.sdsl"
    R = oracle.RefIndex(bwt, sa, 0)
    R.write_csa_image(ref_path, sa)
    if R.bv_size() < 100000:
        assert open(path, "rb").read() == open(ref_path, "rb").read()
    back = V.VlgIndex.load_sdsl(path)
    assert_parts_equal(back.export_parts(), idx.export_parts())
    qs = random_queries(text, np.random.default_rng(3), 40, kmax=3, mmax=3) if len(text) > 30 else [text[:1].decode("latin-1")]
    a, b = idx.search(qs), back.search(qs)
    for x, y in zip(a.fetch(), b.fetch()):
        assert (x == y).all()


@pytest.mark.parametrize("force", ["1", "2"])
def test_64bit_position_kernels(torch_cuda, V, oracle, monkeypatch, force):
    """Texts beyond 4 GiB run on the pos_t = uint64_t instantiations (locate, expand, sort, join, gather, samples).
    VLG_FORCE_POS64=1 selects them on a small text so that they are exercised bit for bit against the oracle; =2 selects the mix
    BASELINE config 4 (n = 2^32 + 1) runs on: 64-bit samples and 33-bit SA indices inside locate, 32-bit text positions -- and
    with them the per-list sort, fences, filter and join of the 32-bit path -- behind it."""
    torch = torch_cuda
    from vlg_matching_amd.index import Workspace
    monkeypatch.setenv("VLG_FORCE_POS64", force)
    text = TEXTS["zipf40"]()
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.build(text)
    assert idx.info()["pos_bytes"] == 8
    assert_parts_equal(idx.export_parts(), o.parts())
    idx2 = V.VlgIndex.from_parts(o.parts())
    assert idx2.info()["pos_bytes"] == 8
    rng = np.random.default_rng(77)
    qs = random_queries(text, rng, 250, kmax=5, mmax=4)
    for sweep_min in (1 << 22, 1):                                           # random-access / sorted-sweep locate
        ws = Workspace()
        ws.set_option("sweep_min", sweep_min)
        ws.set_option("sweep_tail", 64)
        ws.set_option("dedup", 0)
        res = idx.search(qs, workspace=ws)
        occ = np.zeros(4, dtype=np.uint64)
        for i, q in enumerate(qs):
            assert res.tuples(i).tolist() == o.search(q, stats=occ).tolist(), (q, sweep_min)
        assert res.summary["logical_occurrences"] == int(occ[0]) == res.summary["located_occurrences"]
        assert res.summary["lf_steps"] == int(occ[1]) and res.summary["wt_levels_locate"] == int(occ[2])
    # the batch-level shortcuts on 64-bit positions: shared intervals, shared LF trails, one-pass sort, window filter (both modes)
    want = [o.search(q).tolist() for q in qs]
    for pivot in (1, 0):
        ws = Workspace()
        for k_, v_ in (("sweep_min", 1), ("sweep_tail", 16), ("global_sort_min", 1), ("filter_min", 0), ("filter_stream_min", 0), ("filter_pivot", pivot), ("pivot_rungs", 2)):
            ws.set_option(k_, v_)
        res = idx.search(qs, workspace=ws)
        assert ws.kernel_stats()["filter_compact"]["launches"] > 0 and ws.kernel_stats()["locate_resolve"]["launches"] > 0
        for i in range(len(qs)):
            assert res.tuples(i).tolist() == want[i], (qs[i], pivot)
    # csa[i] through the 64-bit locate kernel, and the rrr variant on top of it
    L = V.lib()
    ii = rng.integers(0, o.n, 20000).astype(np.uint64)
    sa = oracle.suffix_array(np.frombuffer(text + b"\0", np.uint8))
    for h in (idx, idx.compress()):
        d_i = dev_u64(torch, ii)
        d_o = torch.zeros_like(d_i)
        V.capi.check(L.vlg_sa_batch(h._h, d_i.data_ptr(), d_o.data_ptr(), len(ii), None))
        torch.cuda.synchronize()
        assert (host_u64(d_o) == sa[ii.astype(np.int64)]).all()


def test_strategy_fuzz_short(V, monkeypatch, capsys):
    """A few seconds of tools/fuzz_strategies.py: random texts and batches, all shortcuts off vs a random combination of them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_strategies", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                 "tools", "fuzz_strategies.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["fuzz_strategies.py", "8", "20261003"])
    mod.main()
    assert "fuzz ok" in capsys.readouterr().out


# ---- K5 on its own: vlg_join_batch on caller-made lists vs the oracle's merge join (index_sasearch.hpp:85-116) ---------------
def _random_join_case(rng, k, sizes, span, gapmax, dense=False):
    lists = []
    for i in range(k):
        n = int(sizes[i])
        if n == 0:
            lists.append(np.zeros(0, np.uint64))
        elif dense:
            lists.append(np.sort(rng.integers(0, span, n)).astype(np.uint64))            # duplicates allowed: ascending, not strictly
        else:
            lists.append(np.sort(rng.choice(span, size=min(n, span), replace=False)).astype(np.uint64))
    lo = [int(rng.integers(0, gapmax // 2 + 1)) for _ in range(k - 1)]
    hi = [l + int(rng.integers(0, gapmax + 1)) for l in lo]
    return lists, lo, hi, int(rng.integers(1, 9))


def _run_join_batch(torch, V, cases, ws):
    from vlg_matching_amd.index import join_batch
    flat, list_off, join_list, lo, hi, end_len = [], [0], [0], [], [], []
    for lists, l, h, e in cases:
        for i, a in enumerate(lists):
            flat.append(a)
            list_off.append(list_off[-1] + len(a))
            lo.append(0 if i == 0 else l[i - 1])
            hi.append(0 if i == 0 else h[i - 1])
        join_list.append(len(list_off) - 1)
        end_len.append(e)
    allv = np.concatenate(flat) if flat else np.zeros(0, np.uint64)
    d = dev_u64(torch, allv if len(allv) else np.zeros(1, np.uint64))
    return join_batch(d.data_ptr(), list_off, join_list, lo, hi, end_len, ws)


@pytest.mark.parametrize("filt", [1, 0])
def test_join_batch_vs_oracle_join(torch_cuda, V, oracle, filt):
    from vlg_matching_amd.index import Workspace
    rng = np.random.default_rng(2024 + filt)
    cases = []
    # lists longer than one tile (1024) and one run (2048); k = 1; empty lists in every place; k up to 8; wide and zero gaps
    cases.append(_random_join_case(rng, 1, [5000], 40000, 10))
    cases.append(_random_join_case(rng, 2, [7000, 9000], 100000, 30))
    cases.append(_random_join_case(rng, 3, [300, 20000, 4000], 200000, 500))               # a short pivot list between long ones
    cases.append(_random_join_case(rng, 3, [20000, 15000, 200], 200000, 300))
    cases.append(_random_join_case(rng, 2, [0, 50], 1000, 10))
    cases.append(_random_join_case(rng, 2, [50, 0], 1000, 10))
    cases.append(_random_join_case(rng, 3, [40, 0, 40], 1000, 10))
    cases.append(_random_join_case(rng, 1, [0], 1000, 10))
    cases.append(_random_join_case(rng, 8, [900] * 8, 30000, 60))
    cases.append(_random_join_case(rng, 5, [3000, 50, 3000, 50, 3000], 60000, 200))
    cases.append(_random_join_case(rng, 2, [6000, 6000], 8000, 3, dense=True))             # duplicates inside a list
    cases.append(([np.arange(0, 50000, 5, dtype=np.uint64), np.arange(2, 50000, 5, dtype=np.uint64)], [2], [2], 1))    # every element matches
    cases.append(([np.array([5, 1 << 40, (1 << 62) + 3], np.uint64), np.array([9, (1 << 40) + 7, (1 << 62) + 4], np.uint64)], [1], [10], 3))
    for _ in range(40):
        k = int(rng.integers(1, 6))
        cases.append(_random_join_case(rng, k, rng.integers(0, 400, k), 5000, 80))
    ws = Workspace()
    ws.set_option("filter", filt)
    ws.set_option("filter_min", 0)
    ws.set_option("filter_stream_min", 0)
    res = _run_join_batch(torch_cuda, V, cases, ws)
    total, chk = 0, 0
    for j, (lists, lo, hi, e) in enumerate(cases):
        m, want = oracle.join(lists, lo, hi, e)
        assert int(res.counts[j]) == m, j
        assert res.tuples(j).tolist() == want.tolist(), j
        assert res.positions(j).tolist() == want[:, 0].tolist() if m else len(res.positions(j)) == 0
        total += m
        chk = (chk + int(want[:, 0].sum())) % (1 << 64) if m else chk
    assert res.summary["n_matches"] == total and res.summary["checksum"] == chk
    # first positions only
    ws.set_option("tuples", 0)
    res2 = _run_join_batch(torch_cuda, V, cases, ws)
    assert (res2.counts == res.counts).all() and res2.summary["checksum"] == chk and res2.summary["n_tuple_values"] == 0


def test_join_batch_rejects_bad_input_and_chunks(torch_cuda, V, oracle):
    from vlg_matching_amd.index import Workspace, join_batch
    from vlg_matching_amd.capi import VlgError
    torch = torch_cuda
    ws = Workspace()
    d = dev_u64(torch, np.array([5, 3, 9, 1, 2, 3], np.uint64))
    with pytest.raises(VlgError):                                          # first list not ascending
        join_batch(d.data_ptr(), [0, 3, 6], [0, 2], [0, 0], [0, 5], [1], ws)
    d = dev_u64(torch, np.array([1, 3, 9, 1, 2, 3], np.uint64))
    with pytest.raises(VlgError):                                          # lo > hi
        join_batch(d.data_ptr(), [0, 3, 6], [0, 2], [0, 7], [0, 5], [1], ws)
    with pytest.raises(VlgError):                                          # end_len 0: the reference's loop would not advance
        join_batch(d.data_ptr(), [0, 3, 6], [0, 2], [0, 0], [0, 5], [0], ws)
    r = join_batch(d.data_ptr(), [0, 3, 6], [0, 2], [0, 0], [0, 5], [1], ws)
    m, want = oracle.join([np.array([1, 3, 9]), np.array([1, 2, 3])], [0], [5], 1)
    assert r.tuples(0).tolist() == want.tolist()
    # a small workspace cap: many chunks, same answers
    rng = np.random.default_rng(77)
    cases = [_random_join_case(rng, int(rng.integers(1, 4)), rng.integers(1000, 30000, 3), 400000, 300) for _ in range(60)]
    big = _run_join_batch(torch, V, cases, Workspace())
    small_ws = Workspace(40 << 20)
    small = _run_join_batch(torch, V, cases, small_ws)
    assert small.summary["n_chunks"] > big.summary["n_chunks"]
    for x, y in zip(big.fetch(), small.fetch()):
        assert (x == y).all()
    for j in (0, 7, 33, 59):
        lists, lo, hi, e = cases[j]
        assert big.tuples(j).tolist() == oracle.join(lists, lo, hi, e)[1].tolist()


def test_queries_occurrences_equal_oracle_counts(V, oracle):
    text = TEXTS["dna_50k"]()
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.build(text)
    rng = np.random.default_rng(8)
    qs = random_queries(text, rng, 200) + ["\xfe.{0,3}?A"]
    occ, q = idx.occurrences(qs)
    want = []
    for qq in qs:
        subs, _, _, _ = oracle.query_fields(oracle.parse(qq))
        want += [o.backward_search(sp)[0] for sp in subs]
    assert occ.tolist() == want
    w = idx.query_weights(qs)
    assert len(w) == len(qs) and w[-1] == 1.0 and (w >= 1.0).all()


# ---- multi-GPU path rehearsed on one device: 2 gloo ranks on cuda:0 --------------------------------------------------------
def _replica_worker(rank, world, port, text, queries, out_q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vlg_matching_amd as V
    from vlg_matching_amd import dist as vdist
    dev = torch.device("cuda", 0)
    idx = V.VlgIndex.build(text) if rank == 0 else None
    # gloo moves CUDA tensors through the host: the calls are the ones the RCCL path makes (broadcast of the blob, attach)
    idx = vdist.replicate_index(idx, dist, dev, src=0)
    w = idx.query_weights(queries)                                       # every rank can weigh the batch: the index is replicated
    b, e = vdist.shard_by_work(w, world)[rank]
    r = idx.search(queries[b:e])
    tot = torch.tensor([r.summary["n_matches"], r.summary["located_occurrences"]], dtype=torch.int64)
    dist.all_reduce(tot)
    chk = vdist.reduce_checksum(r.summary["checksum"], dist)
    out_q.put((rank, b, e, [int(c) for c in r.counts], int(tot[0]), chk, idx.info()["hbm_bytes"]))
    dist.barrier()
    dist.destroy_process_group()


def test_replicate_index_two_ranks_on_one_device(V, oracle):
    """SURVEY.md 8(e) on one GPU: rank 0 builds, the blob is broadcast, rank 1 attaches; the batch is sharded by work;
    totals equal the 1-rank run and the oracle."""
    import socket
    import torch.multiprocessing as mp
    text = dna_text(60000, 9).tobytes()
    rng = np.random.default_rng(19)
    queries = random_queries(text, rng, 300, kmax=3, mmax=4)
    one = V.VlgIndex.build(text).search(queries)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, text, queries, out_q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(out_q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == len(queries)
    assert got[0][3] + got[1][3] == [int(c) for c in one.counts]
    for g in got:
        assert g[4] == one.summary["n_matches"] and g[5] == one.summary["checksum"]
    assert got[0][6] == got[1][6]
    o = oracle.Index.from_text(text)
    for i in (0, 10, 150, 299):
        assert int(one.counts[i]) == len(o.search(queries[i]))


def _nccl_one_rank_worker(port, text, queries, out_q):
    """Everything bench.py does with the `nccl` backend, in a 1-rank process group on cuda:0, plus the product's own RCCL entry
    points (vlg_comm_*, vlg_index_broadcast, vlg_comm_allgatherv, vlg_comm_alltoallv)."""
    import torch
    import torch.distributed as dist
    import vlg_matching_amd as V
    from vlg_matching_amd import dist as vdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)              # bench.py: init_process_group("nccl", device_id=...)
    assert dist.get_backend() == "nccl"
    idx0 = V.VlgIndex.build(text)
    # torch-side collectives of bench.py: replicate (broadcast of a uint8 tensor), scatter_object_list, all_reduce MAX / SUM, all_gather
    idx = vdist.replicate_index(idx0, dist, dev, src=0)
    mine = [None]
    dist.scatter_object_list(mine, [queries], src=0)
    assert mine[0] == queries
    r = idx.search(mine[0])
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tot = torch.tensor([r.summary["n_matches"], r.summary["located_occurrences"]], dtype=torch.int64, device=dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    outl = [torch.zeros_like(t)]
    dist.all_gather(outl, t)
    chk = vdist.reduce_checksum(r.summary["checksum"], dist, dev)
    dist.barrier()
    # the product's RCCL: communicator through the C-ABI, index broadcast, counter reduction, variable all-gather
    comm = vdist.Comm.from_torch_dist(dist)
    n_ranks, rank = comm.info()
    same = comm.broadcast_index(idx0, root=0)
    sums = comm.allreduce_sum_u64([r.summary["n_matches"], (1 << 64) - 1, 7])
    send = torch.arange(1000, dtype=torch.int32, device=dev)
    recv = torch.zeros(1000, dtype=torch.int32, device=dev)
    comm.allgatherv(send.data_ptr(), [1000], 4, recv.data_ptr())
    recv2 = torch.zeros(1000, dtype=torch.int32, device=dev)
    comm.alltoallv(send.data_ptr(), [1000], recv2.data_ptr(), [1000], 4)      # (one rank: its own piece, a device copy inside the group)
    torch.cuda.synchronize()
    r2 = vdist.replicate_index(idx0, dist, dev, src=0, comm=comm).search(queries)
    # collective search over the (one-rank) communicator: list-sharded locate, exchange through vlg_comm_allgatherv, query-sharded joins
    from vlg_matching_amd.index import Workspace
    wsx = Workspace()
    wsx.set_comm(comm)
    r3 = idx0.search(queries, workspace=wsx)
    owned = r3.owned_queries()
    wsx.set_comm(None)
    lib_path = comm.library()
    comm.close()
    out_q.put({"counts": [int(c) for c in r.counts], "tot": [int(x) for x in tot.tolist()], "max": float(t.item()), "chk": chk,
               "gathered": float(outl[0].item()), "comm": (n_ranks, rank), "same": same is idx0, "sums": sums,
               "recv_ok": bool((recv == send).all().item()) and bool((recv2 == send).all().item()), "counts2": [int(c) for c in r2.counts], "rccl": lib_path,
               "counts3": [int(c) for c in r3.counts], "owned": owned, "chk3": int(r3.summary["checksum"]),
               "summary": {k: int(v) for k, v in r.summary.items()}})
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_branch_runs_with_one_rank(V, oracle):
    """The `nccl` (= RCCL) branch of bench.py and of the product must have executed before an 8-GPU node ever sees it: a
    world-size-1 process group on cuda:0 runs replicate_index, scatter_object_list, all_reduce, all_gather, reduce_checksum exactly
    as bench.py calls them, and the C-ABI's own RCCL path (vlg_comm_create, vlg_index_broadcast, vlg_comm_allreduce_sum_u64,
    vlg_comm_allgatherv); results equal the plain single-process run and the oracle."""
    import socket
    import torch.multiprocessing as mp
    text = dna_text(40000, 21).tobytes()
    rng = np.random.default_rng(23)
    queries = random_queries(text, rng, 120, kmax=3, mmax=4)
    one = V.VlgIndex.build(text).search(queries)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    p = ctx.Process(target=_nccl_one_rank_worker, args=(port, text, queries, out_q))
    p.start()
    got = out_q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert got["counts"] == [int(c) for c in one.counts] == got["counts2"]
    assert got["tot"] == [one.summary["n_matches"], one.summary["located_occurrences"]]
    assert got["chk"] == one.summary["checksum"] and got["max"] == 1.5 and got["gathered"] == 1.5
    assert got["comm"] == (1, 0) and got["same"] and got["recv_ok"]
    assert got["sums"] == [one.summary["n_matches"], (1 << 64) - 1, 7]
    assert got["counts3"] == got["counts"] and got["owned"] == [] and got["chk3"] == one.summary["checksum"]
    assert "rccl" in got["rccl"].lower()
    o = oracle.Index.from_text(text)
    for i in (0, 17, 60, 119):
        assert got["counts"][i] == len(o.search(queries[i]))


@pytest.mark.parametrize("name,dens,force", [("dna_50k", 32, ""), ("zipf40", 8, ""), ("100a", 4, ""), ("abracadabra", 1, ""), ("dna_skew", 64, ""),
                                             ("zipf40", 16, "2"), ("dna_50k", 5, "1"), ("abracadabra", 2, "2")])
def test_text_order_sa_sampling(torch_cuda, V, oracle, monkeypatch, name, dens, force):
    """text_order_sa_sampling (include/sdsl/csa_sampling_strategy.hpp:127-246) as an index variant made by
    vlg_index_resample: the marked bit-vector and the condensed samples equal the oracle's restatement, csa[i] == SA[i] for every i
    through vlg_sa_batch, every search mode returns the tuples of the SA-order index and the oracle -- with exactly
    sum(SA[i] % dens) LF steps when nothing is shared (the strategy's defining property) -- and it stacks on the rrr variant.
    force: VLG_FORCE_POS64 -- the strategy is width-agnostic in the reference; "2" runs it on the kernels of BASELINE config 4 (33-bit
    SA indices and 64-bit samples inside locate, 32-bit positions behind it), "1" on 64-bit positions throughout."""
    torch = torch_cuda
    from vlg_matching_amd.index import Workspace
    if force:
        monkeypatch.setenv("VLG_FORCE_POS64", force)
    text = TEXTS[name]()
    o = oracle.Index.from_text(text)
    sa = oracle.suffix_array(np.frombuffer(text + bytes(1), np.uint8)).astype(np.int64)
    base = V.VlgIndex.build(text)
    idx = base.resample(text_order=True, dens=dens)
    info = idx.info()
    assert info["sampling"] == 1 and info["sa_sample_dens"] == dens and info["n_samples"] == (len(sa) + dens - 1) // dens
    assert info["pos_bytes"] == (8 if force else 4)
    to = oracle.TextOrder(o, dens)
    assert (idx.marked() == to.marked()).all() and (idx.marked() == (sa % dens == 0)).all()
    want_samples = to.samples()
    got_samples = idx.export_parts()["samples"]
    assert got_samples[: len(want_samples)].tolist() == want_samples.tolist()
    L = V.lib()
    ii = np.arange(len(sa), dtype=np.uint64)
    for h in (idx, base.compress().resample(text_order=True, dens=dens)):
        d_i = dev_u64(torch, ii)
        d_o = torch.zeros_like(d_i)
        V.capi.check(L.vlg_sa_batch(h._h, d_i.data_ptr(), d_o.data_ptr(), len(ii), None))
        torch.cuda.synchronize()
        assert (host_u64(d_o).astype(np.int64) == sa).all()
    rng = np.random.default_rng(5)
    qs = random_queries(text, rng, 150, kmax=4, mmax=4)
    want = [o.search(q).tolist() for q in qs]
    ref = base.search(qs)
    for opts in ({"sweep_min": 1 << 30, "dedup": 0}, {"sweep_min": 1, "sweep_tail": 16, "dedup": 0}, {"sweep_min": 1, "sweep_tail": 16},
                 {"sweep_min": 1, "sweep_tail": 1 << 30}):
        ws = Workspace()
        for k_, v_ in opts.items():
            ws.set_option(k_, v_)
        res = idx.search(qs, workspace=ws)
        for i in range(len(qs)):
            assert res.tuples(i).tolist() == want[i], (qs[i], opts)
        assert res.summary["checksum"] == ref.summary["checksum"] and res.summary["n_matches"] == ref.summary["n_matches"]
        if opts.get("dedup", 1) == 0:
            # every occurrence walks SA[i] % dens steps to its sample: count them from the located positions themselves
            steps = 0
            for q in qs:
                subs, _, _, _ = oracle.query_fields(oracle.parse(q))
                occs = [o.backward_search(sp) for sp in subs]
                if min(c for c, _, _ in occs) == 0:
                    continue
                for c, l, r in occs:
                    steps += int((sa[l:r + 1] % dens).sum())
            assert res.summary["lf_steps"] == steps, opts
    # SA-order resampling with another density gives the samples the builder would
    re8 = base.resample(text_order=False, dens=8)
    assert_parts_equal(re8.export_parts(), V.VlgIndex.build(text, dens=8).export_parts())
    # a text-order index travels as a blob like any other
    blob = torch.empty(idx.blob_bytes(), dtype=torch.uint8, device="cuda")
    idx.blob_export(blob.data_ptr(), blob.numel())
    att = V.VlgIndex.attach_blob(blob.data_ptr(), blob.numel(), keep=blob)
    assert att.info()["sampling"] == 1
    r2 = att.search(qs)
    assert (r2.counts == ref.counts).all() and r2.summary["checksum"] == ref.summary["checksum"]


@pytest.mark.parametrize("name", ["dna_50k", "zipf40", "100a", "abracadabra", "dna_skew"])
def test_dense_suffix_array_index(torch_cuda, V, oracle, monkeypatch, name):
    """csa_wt<wt_huff<>, 1, .>: the whole suffix array resident in HBM (vlg_index_resample(SA order, 1)).  Its samples
    ARE the suffix array, locate copies SA intervals (sa_dense_copy_kernel: zero LF steps, no trails, no records), and every search
    mode -- the sweep's fast path, the lane-per-occurrence kernel, narrow and forced-wide positions -- returns the tuples of the
    oracle and the checksum of the benchmark's t_dens = 32 index (csa_wt.hpp:335-348 with a sample at every index)."""
    torch = torch_cuda
    from vlg_matching_amd.index import Workspace
    text = TEXTS[name]()
    o = oracle.Index.from_text(text)
    sa = oracle.suffix_array(np.frombuffer(text + bytes(1), np.uint8)).astype(np.int64)
    base = V.VlgIndex.build(text)
    rng = np.random.default_rng(11)
    qs = random_queries(text, rng, 150, kmax=4, mmax=4)
    want = [o.search(q).tolist() for q in qs]
    ref = base.search(qs)
    for src in (base, base.compress()):
        idx = src.resample(text_order=False, dens=1)
        info = idx.info()
        assert info["sampling"] == 0 and info["sa_sample_dens"] == 1 and info["n_samples"] == len(sa)
        if src is base:                                          # (an rrr index does not export its parts)
            assert idx.export_parts()["samples"][: len(sa)].astype(np.int64).tolist() == sa.tolist()
            want_parts = oracle.Index.from_text(text, dens=1).parts()   # the restated csa_wt<wt_huff<>, 1, .>
            assert idx.export_parts()["dens"] == want_parts["dens"] == 1
            assert_parts_equal(idx.export_parts(), want_parts)
        d_i = dev_u64(torch, np.arange(len(sa), dtype=np.uint64))
        d_o = torch.zeros_like(d_i)
        V.capi.check(V.lib().vlg_sa_batch(idx._h, d_i.data_ptr(), d_o.data_ptr(), len(sa), None))
        torch.cuda.synchronize()
        assert (host_u64(d_o).astype(np.int64) == sa).all()
        for opts in ({"sweep_min": 1}, {"sweep_min": 1, "dedup": 0}, {"sweep_min": 1 << 30}, {"sweep_min": 1, "trail": 0}):
            ws = Workspace()
            for k_, v_ in opts.items():
                ws.set_option(k_, v_)
            res = idx.search(qs, workspace=ws)
            for i in range(len(qs)):
                assert res.tuples(i).tolist() == want[i], (qs[i], opts)
            assert res.summary["checksum"] == ref.summary["checksum"] and res.summary["n_matches"] == ref.summary["n_matches"]
            assert res.summary["lf_steps"] == 0, opts
    # built with density 1 right away, with 64-bit samples and wide SA indices (=1: 64-bit positions, =2: 32-bit positions)
    for force in ("1", "2"):
        monkeypatch.setenv("VLG_FORCE_POS64", force)
        wide = V.VlgIndex.build(text, dens=1)
        assert wide.info()["pos_bytes"] == 8 and wide.info()["sa_sample_dens"] == 1
        ws = Workspace()
        ws.set_option("sweep_min", 1)
        res = wide.search(qs, workspace=ws)
        for i in range(len(qs)):
            assert res.tuples(i).tolist() == want[i], (qs[i], force)
        assert res.summary["checksum"] == ref.summary["checksum"] and res.summary["lf_steps"] == 0


def _int_texts():
    rng = np.random.default_rng(31)
    return {
        "survey": np.array([5, 6, 7, 5, 6, 7, 1000, 5], dtype=np.uint32),
        "abra": np.frombuffer(b"abracadabrasimsalabim", dtype=np.uint8).astype(np.uint32),
        "sparse": rng.choice(np.array([3, 7, 7, 19, 1000, 70000, 2 ** 31 + 5], dtype=np.uint32), 5000),
        "words": (1 + rng.zipf(1.3, 20000) % 3000).astype(np.uint32),          # word-level text: large alphabet, Zipf frequencies
        "one": np.array([42], dtype=np.uint32),
        "run": np.full(300, 9, dtype=np.uint32),
        # the reference's own integer fixture (test/test_cases/keeper.int, csa_int_test.config:7: 63 symbols of 8 bytes), kept as data
        "keeper": np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keeper.int"), dtype="<u8").astype(np.uint32),
    }


def _int_queries(text, rng, nq, kmax=3, mmax=3, gapmax=40):
    t = text.tolist()
    qs = []
    for _ in range(nq):
        k = int(rng.integers(1, kmax + 1))
        subs = [t[s:s + int(rng.integers(1, mmax + 1))] for s in rng.integers(0, max(len(t) - mmax, 1), k)]
        q = " ".join(map(str, subs[0]))
        for sp in subs[1:]:
            a = int(rng.integers(0, 10))
            q += " .{%d,%d}? %s" % (a, a + int(rng.integers(0, gapmax)), " ".join(map(str, sp)))
        qs.append(q)
    return qs


def _sort_cases():
    """(name, lists, position_bits, clustered long lists expected with mode 1)"""
    rng = np.random.default_rng(404)
    cases = []
    # positions spread over the text, every size class: empty, single, short (LDS), long (windows)
    lens = [0, 1, 2, 63, 64, 65, 255, 256, 257, 511, 1000, 2047, 2048, 2049, 4095, 4096, 4097, 6143, 8192, 12289, 50000, 300001, 1 << 20]
    cases.append(("spread", [rng.choice(1 << 30, n, replace=False).astype(np.uint32) for n in lens], 30, 0))
    # the same in a text whose positions need all 32 bits, the largest position present
    hi = [np.unique(np.concatenate([rng.integers(0, 2 ** 32 - 1, n, dtype=np.uint64), [2 ** 32 - 2, 0]])).astype(np.uint32) for n in (3000, 70000, 900000)]
    cases.append(("32bit", [rng.permutation(h) for h in hi], 32, 0))
    # clustered: runs of consecutive positions (a^n), a dense island in a sparse list, everything inside one group of the top 16 bits
    run = np.arange(5_000_000, 5_000_000 + 400000, dtype=np.uint32)
    island = np.concatenate([rng.choice(1 << 30, 60000, replace=False), np.arange(777_000_000, 777_000_000 + 9000)]).astype(np.uint32)
    island = np.unique(island)
    tiny_range = (123_456_000 + rng.choice(12000, 9000, replace=False)).astype(np.uint32)
    short_cluster = np.concatenate([np.arange(1000, 4000), [1 << 29]]).astype(np.uint32)          # <= 4096 keys, all but one in one bin
    cases.append(("clustered", [rng.permutation(run), rng.permutation(island), rng.permutation(tiny_range), rng.permutation(short_cluster),
                                rng.choice(1 << 30, 20000, replace=False).astype(np.uint32)], 30, 3))
    # duplicates (not what locate produces, but the sort must not depend on distinct keys)
    dup = rng.integers(0, 5000, 30000).astype(np.uint32)
    dup2 = np.repeat(rng.choice(1 << 28, 3000, replace=False), 3).astype(np.uint32)
    cases.append(("duplicates", [dup, rng.permutation(dup2), np.full(5000, 7, np.uint32), np.full(300, 2 ** 28 - 1, np.uint32)], 28, None))
    # few position bits: the plain passes (no room for 16 top bits), and the smallest text the windows run on
    cases.append(("16bit", [rng.permutation(np.arange(60000, dtype=np.uint32)), rng.choice(1 << 16, 5000, replace=False).astype(np.uint32)], 16, 0))
    cases.append(("17bit", [rng.permutation(np.arange(131000, dtype=np.uint32)), rng.choice(1 << 17, 9000, replace=False).astype(np.uint32)], 17, None))
    return cases


@pytest.mark.parametrize("case", _sort_cases(), ids=lambda c: c[0])
def test_k4_list_sort_stand_alone(torch_cuda, V, case):
    """K4 (std::sort of every occurrence list, index_sasearch.hpp:80) on its own: vlg_sort_lists_u32 sorts every list of a batch in
    place -- numpy's sort is the truth.  Mode 1 (value-range buckets in LDS for short lists; two radix passes over the top 16 bits +
    windows for long ones; clustered lists flagged and sorted by the four full passes) and mode 2 (the full passes) must agree, and the
    number of flagged lists is what the construction of the case says."""
    torch = torch_cuda
    name, lists, bits, want_clustered = case
    off = np.zeros(len(lists) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in lists])
    flat = np.concatenate(lists).astype(np.uint32)
    want = np.concatenate([np.sort(x) for x in lists])
    L = V.lib()
    import ctypes as C
    for mode in (1, 2):
        d = torch.from_numpy(flat.view(np.int32).copy()).cuda()
        nc = C.c_uint64(99)
        V.capi.check(L.vlg_sort_lists_u32(d.data_ptr(), off.ctypes.data, len(lists), bits, mode, C.byref(nc), None))
        got = d.cpu().numpy().view(np.uint32)
        assert (got == want).all(), (name, mode, int(np.argmax(got != want)))
        if mode == 2:
            assert nc.value == 0
        elif want_clustered is not None:
            assert nc.value == want_clustered, (name, nc.value)
    with pytest.raises(V.VlgError):
        V.capi.check(L.vlg_sort_lists_u32(None, off.ctypes.data, len(lists), 33, 1, None, None))


@pytest.mark.parametrize("name", ["survey", "abra", "sparse", "words", "one", "run", "keeper"])
def test_integer_alphabet_fm_index(torch_cuda, V, oracle, name):
    """csa_wt<wt_int<>, 32, ., ., ., int_alphabet<>> on the device (vlg_index_build_int) against its restatement in the
    reference's layout (oracle/vlg_oracle_int.c, pinned by the reference's own wt_int<> / int_alphabet<>): alphabet, wt_int::rank
    for present and absent symbols, csa[i] for every i, backward_search intervals, and the tuples of the whole path -- which must
    also equal the paper's index (vlg_wtsa_*) on the same text."""
    torch = torch_cuda
    from vlg_matching_amd.index import Queries, WtsaIndex
    text = _int_texts()[name]
    o = oracle.IntIndex(text.astype(np.uint64), dens=32)
    idx = V.VlgIndex.build_int(text)
    info = idx.info()
    assert info["n"] == len(text) + 1 and info["bv_kind"] == 2 and info["pos_bytes"] == 4
    Cc, c2c = idx.int_alphabet()
    assert Cc.tolist() == o.C().tolist() and c2c.tolist() == o.comp2char().tolist()
    rng = np.random.default_rng(12)
    L = V.lib()
    m = 600
    syms = np.concatenate([rng.choice(text, m - 6), np.array([1, 2, 4, 123456, 2 ** 32 - 1, int(text[0])], dtype=np.uint32)]).astype(np.uint32)
    pos = rng.integers(0, o.n + 1, m).astype(np.uint64)
    d_p, d_s = dev_u64(torch, pos), torch.from_numpy(syms.view(np.int32)).cuda()
    d_o = torch.zeros(m, dtype=torch.int64, device="cuda")
    V.capi.check(L.vlg_int_rank_batch(idx._h, d_p.data_ptr(), d_s.data_ptr(), d_o.data_ptr(), m, None))
    torch.cuda.synchronize()
    assert host_u64(d_o).tolist() == [o.rank(int(i), int(c)) for i, c in zip(pos, syms)]
    tz = np.concatenate([text.astype(np.int64), [0]])
    sa = np.array(sorted(range(len(tz)), key=lambda i: tz[i:].tolist()), dtype=np.uint64) if len(tz) <= 6000 else \
        np.array([o.sa(i) for i in range(o.n)], dtype=np.uint64)
    d_i = dev_u64(torch, np.arange(o.n, dtype=np.uint64))
    d_v = torch.zeros_like(d_i)
    V.capi.check(L.vlg_sa_batch(idx._h, d_i.data_ptr(), d_v.data_ptr(), o.n, None))
    torch.cuda.synchronize()
    assert (host_u64(d_v) == sa).all()
    qs = _int_queries(text, rng, 120) + ["%d .{0,5}? 999999" % int(text[0]), "999999", "%d" % int(text[-1])]
    l, r, q = idx.intervals(qs)
    # sub-pattern intervals in batch order
    import re
    pats = []
    for qq in qs:
        for part in re.split(r"\.\{\d+,\d+\}\?", qq):
            pats.append([int(x) for x in part.split()])
    assert len(pats) == len(l)
    for p, a, b in zip(pats, l, r):
        cnt, ol, orr = o.backward_search(p)
        assert int(b) + 1 - int(a) == cnt, p
        if cnt:
            assert (int(a), int(b)) == (ol, orr), p
    res = idx.search(qs)
    want = [o.search(qq).tolist() for qq in qs]
    for i in range(len(qs)):
        assert res.tuples(i).tolist() == want[i], qs[i]
    assert res.summary["n_matches"] == sum(len(w) for w in want)
    wres = WtsaIndex(text).search(qs)
    for i in range(len(qs)):
        assert wres.tuples(i).tolist() == want[i], qs[i]
    # the sorted sweep on the wavelet matrix (sigma <= 65534: a 16-bit partition key), with walks that stop at elements of the batch,
    # finished by the sorted rounds or by the stragglers' kernel: same tuples; with nothing shared, the reference's LF steps
    from vlg_matching_amd.index import Workspace
    st0 = np.zeros(4, dtype=np.uint64)
    for qq in qs:
        o.search(qq, stats=st0)
    for opts in ({"sweep_min": 1, "sweep_tail": 16}, {"sweep_min": 1, "sweep_tail": 1 << 30}, {"sweep_min": 1, "sweep_tail": 4, "trail": 0},
                 {"sweep_min": 1, "sweep_tail": 16, "dedup": 0}):
        wsx = Workspace()
        for k_, v_ in opts.items():
            wsx.set_option(k_, v_)
        rx = idx.search(qs, workspace=wsx)
        assert rx.summary["locate_mode"] == V.capi.LOCATE_SWEEP or rx.summary["located_occurrences"] == 0, opts
        for i in range(len(qs)):
            assert rx.tuples(i).tolist() == want[i], (qs[i], opts)
        if opts.get("dedup", 1) == 0:
            assert rx.summary["lf_steps"] == int(st0[1]), opts             # (the matrix reads one block per level and step: levels are its own count)
            assert rx.summary["wt_levels_locate"] == int(st0[1]) * idx.info()["max_code_len"]
        elif opts.get("trail", 1):
            assert rx.summary["lf_steps"] <= int(st0[1])
    # LF-step parity with the restated reference (no interval sharing: every occurrence walks to its own sample)
    ws = Workspace()
    ws.set_option("dedup", 0)
    st = np.zeros(4, dtype=np.uint64)
    for qq in qs:
        o.search(qq, stats=st)
    r2 = idx.search(qs, workspace=ws)
    assert r2.summary["located_occurrences"] == int(st[0]) and r2.summary["lf_steps"] == int(st[1])
    # the image travels like a byte index's
    blob = torch.empty(idx.blob_bytes(), dtype=torch.uint8, device="cuda")
    idx.blob_export(blob.data_ptr(), blob.numel())
    att = V.VlgIndex.attach_blob(blob.data_ptr(), blob.numel(), keep=blob)
    r3 = att.search(qs)
    assert (r3.counts == res.counts).all() and r3.summary["checksum"] == res.summary["checksum"]


@pytest.mark.parametrize("name", ["survey", "sparse", "words", "one", "run", "keeper"])
def test_integer_alphabet_rrr_index(torch_cuda, V, oracle, name):
    """csa_wt<wt_int<rrr_vector<63>>, ., ., ., ., int_alphabet<>> (test/csa_int_test.cpp:32): vlg_index_compress of an integer index
    re-encodes every level of the wavelet matrix as rrr-63 (the byte index's encoder and rank, K6).  Same alphabet, wt_int::rank,
    csa[i], intervals, tuples and LF steps as the plain index and the restated reference structure; the image travels; on a skewed
    word-level text it is smaller."""
    torch = torch_cuda
    from vlg_matching_amd.index import Workspace
    text = _int_texts()[name]
    o = oracle.IntIndex(text.astype(np.uint64), dens=32)
    plain = V.VlgIndex.build_int(text)
    idx = plain.compress()
    info = idx.info()
    assert info["n"] == len(text) + 1 and info["bv_kind"] == 3 and info["pos_bytes"] == 4
    Cc, c2c = idx.int_alphabet()
    assert Cc.tolist() == o.C().tolist() and c2c.tolist() == o.comp2char().tolist()
    rng = np.random.default_rng(13)
    L = V.lib()
    m = 600
    syms = np.concatenate([rng.choice(text, m - 6), np.array([1, 2, 4, 123456, 2 ** 32 - 1, int(text[0])], dtype=np.uint32)]).astype(np.uint32)
    pos = np.concatenate([rng.integers(0, o.n + 1, m - 2), [0, o.n]]).astype(np.uint64)
    d_p, d_s = dev_u64(torch, pos), torch.from_numpy(syms.view(np.int32)).cuda()
    d_o = torch.zeros(m, dtype=torch.int64, device="cuda")
    V.capi.check(L.vlg_int_rank_batch(idx._h, d_p.data_ptr(), d_s.data_ptr(), d_o.data_ptr(), m, None))
    torch.cuda.synchronize()
    assert host_u64(d_o).tolist() == [o.rank(int(i), int(c)) for i, c in zip(pos, syms)]
    d_i = dev_u64(torch, np.arange(o.n, dtype=np.uint64))
    d_v, d_w = torch.zeros_like(d_i), torch.zeros_like(d_i)
    V.capi.check(L.vlg_sa_batch(idx._h, d_i.data_ptr(), d_v.data_ptr(), o.n, None))
    V.capi.check(L.vlg_sa_batch(plain._h, d_i.data_ptr(), d_w.data_ptr(), o.n, None))
    torch.cuda.synchronize()
    assert (host_u64(d_v) == host_u64(d_w)).all() and sorted(host_u64(d_v).tolist()) == list(range(o.n))
    qs = _int_queries(text, rng, 100) + ["%d .{0,5}? 999999" % int(text[0]), "999999", "%d" % int(text[-1])]
    l, r, _ = idx.intervals(qs)
    lp, rp, _ = plain.intervals(qs)
    assert l.tolist() == lp.tolist() and r.tolist() == rp.tolist()
    want = [o.search(qq).tolist() for qq in qs]
    res = idx.search(qs)
    for i in range(len(qs)):
        assert res.tuples(i).tolist() == want[i], qs[i]
    st0 = np.zeros(4, dtype=np.uint64)
    for qq in qs:
        o.search(qq, stats=st0)
    for opts in ({"sweep_min": 1, "sweep_tail": 16}, {"sweep_min": 1, "sweep_tail": 1 << 30}, {"sweep_min": 1, "sweep_tail": 4, "trail": 0},
                 {"sweep_min": 1, "sweep_tail": 16, "dedup": 0}, {"sweep": 0, "dedup": 0}):
        wsx = Workspace()
        for k_, v_ in opts.items():
            wsx.set_option(k_, v_)
        rx = idx.search(qs, workspace=wsx)
        for i in range(len(qs)):
            assert rx.tuples(i).tolist() == want[i], (qs[i], opts)
        if opts.get("dedup", 1) == 0:
            assert rx.summary["located_occurrences"] == int(st0[0]) and rx.summary["lf_steps"] == int(st0[1]), opts
    blob = torch.empty(idx.blob_bytes(), dtype=torch.uint8, device="cuda")
    idx.blob_export(blob.data_ptr(), blob.numel())
    att = V.VlgIndex.attach_blob(blob.data_ptr(), blob.numel(), keep=blob)
    assert att.info()["bv_kind"] == 3
    r3 = att.search(qs)
    assert (r3.counts == res.counts).all() and r3.summary["checksum"] == res.summary["checksum"]


def test_integer_alphabet_rrr_index_is_smaller_on_skewed_text(V):
    """What the compression is for: a word-level text with Zipf frequencies (most of the upper levels' bits are runs)."""
    rng = np.random.default_rng(5)
    text = (1 + rng.zipf(1.6, 400000) % 50000).astype(np.uint32)
    plain = V.VlgIndex.build_int(text)
    idx = plain.compress()
    samples = 4 * ((len(text) + 1 + 31) // 32)                     # the SA samples are the same in both: compare what was compressed
    assert idx.blob_bytes() - samples < 0.8 * (plain.blob_bytes() - samples), (idx.blob_bytes(), plain.blob_bytes(), samples)
    qs = _int_queries(text, rng, 50)
    a, b = plain.search(qs), idx.search(qs)
    assert (a.counts == b.counts).all() and a.summary["checksum"] == b.summary["checksum"] and a.summary["n_matches"] > 0


def test_integer_alphabet_64bit_symbols_through_a_symbol_map(V, oracle):
    """gapped_pattern_query<int_alphabet_tag> reads uint64_t tokens (vlg_index.hpp:57-69).  A text whose symbols need more than 32 bits
    goes through vlg_symbol_map (rank + 1 of every symbol: dense, order-preserving), the mapped text into vlg_index_build_int and
    vlg_wtsa_build_int, and queries -- written with the ORIGINAL 64-bit symbols -- through vlg_queries_parse_int_mapped: same tuples
    as the restated reference structure built over the 64-bit text itself, for both indexes; symbols the text does not hold match
    nothing; without a map a token above 2^32 - 1 is refused."""
    from vlg_matching_amd.index import Queries, WtsaIndex
    rng = np.random.default_rng(77)
    # (2^64 - 1 itself is left out: the oracle's suffix sorter ranks symbols as x + 1 -- oracle/vlg_oracle_int.c)
    vocab = np.array([3, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 45 + 9, 2 ** 63 + 5, 2 ** 64 - 3, 12], dtype=np.uint64)
    text = vocab[rng.choice(len(vocab), 4000, p=np.array([6, 5, 4, 4, 3, 2, 1, 5]) / 30.0)]
    o = oracle.IntIndex(text, dens=32)                                           # the oracle's symbols are 64-bit
    m = V.SymbolMap(text)
    mapped = m.apply(text)
    idx = V.VlgIndex.build_int(mapped)
    t = [int(x) for x in text]
    qs = []
    for _ in range(150):
        k = int(rng.integers(1, 4))
        subs = [t[s:s + int(rng.integers(1, 4))] for s in rng.integers(0, len(t) - 4, k)]
        q = " ".join(map(str, subs[0]))
        for sp in subs[1:]:
            a = int(rng.integers(0, 10))
            q += " .{%d,%d}? %s" % (a, a + int(rng.integers(0, 40)), " ".join(map(str, sp)))
        qs.append(q)
    qs += ["%d .{0,9}? 4242" % t[0], "18446744073709551614", "%d %d" % (2 ** 63 + 5, 2 ** 64 - 3)]      # absent symbols; a rare pair
    want = [o.search(q).tolist() for q in qs]
    assert sum(len(w) for w in want) > 500
    batch = m.queries(qs)
    res = idx.search(batch)
    wres = WtsaIndex(mapped).search(batch)
    for i in range(len(qs)):
        assert res.tuples(i).tolist() == want[i], qs[i]
        assert wres.tuples(i).tolist() == want[i], qs[i]
    # intervals are those of the 64-bit text: the map preserves the suffix order
    l, r, _ = idx.intervals(batch)
    import re
    pats = [[int(x) for x in part.split()] for qq in qs for part in re.split(r"\.\{\d+,\d+\}\?", qq)]
    for p_, a, b in zip(pats, l, r):
        cnt, ol, orr = o.backward_search(p_)
        assert int(b) + 1 - int(a) == cnt and (not cnt or (int(a), int(b)) == (ol, orr)), p_
    with pytest.raises(V.VlgError) as e:
        Queries.from_int(["5 4294967296"])
    assert e.value.status == V.capi.E_INVALID and "vlg_symbol_map" in str(e.value)


def test_integer_alphabet_fm_index_known_answers_and_refusals(V):
    gold = json.load(open(GOLD))
    for case in gold["int_cases"]:
        idx = V.VlgIndex.build_int(np.array(case["int_text"], dtype=np.uint32))
        r = idx.search([case["query"]])
        assert r.tuples(0).tolist() == case["tuples"], case
    with pytest.raises(V.VlgError) as e:
        V.VlgIndex.build_int(np.array([4, 0, 4], dtype=np.uint32))                 # construct.hpp:36-45: a 0 symbol is refused
    assert e.value.status == V.capi.E_ZERO_BYTE
    idx = V.VlgIndex.build_int(np.array([5, 6, 7, 5], dtype=np.uint32))
    from vlg_matching_amd.index import Queries
    with pytest.raises(V.VlgError):
        idx.search(Queries(["ab.{0,3}?c"]))                                       # a byte batch on an integer index
    with pytest.raises(V.VlgError):
        V.VlgIndex.build(b"abcabc").search(Queries.from_int(["5 6"]))             # and the other way round
    with pytest.raises(V.VlgError):
        idx.compress().compress()                                                 # the source must be a plain index
    with pytest.raises(V.VlgError):
        idx.export_parts()


def _collective_worker(rank, world, port, text, queries, opts, out_q):
    import torch
    import torch.distributed as dist
    import vlg_matching_amd as V
    from vlg_matching_amd import dist as vdist
    from vlg_matching_amd.index import Workspace
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = vdist.replicate_index(V.VlgIndex.build(text) if rank == 0 else None, dist, torch.device("cuda", 0), src=0)
    opts = dict(opts)
    mode = opts.pop("_exchange", "alltoall")
    cap = opts.pop("_cap_of_rank", {}).get(rank, 0)
    ws = Workspace(cap)
    for k_, v_ in opts.items():
        ws.set_option(k_, v_)
    if mode == "alltoall":
        ws.set_exchange_alltoall(world, rank, vdist.host_alltoall(dist))       # needed lists only, pairwise (the default over RCCL)
    else:
        ws.set_exchange(world, rank, vdist.host_exchange(dist))                # every list to every rank: in-place all-gather
    try:
        r = idx.search(queries, workspace=ws)
    except V.VlgError as e:                                                    # (the agreement test: every rank reports how it failed)
        out_q.put((rank, "error", e.status, str(e)))
        dist.barrier()
        dist.destroy_process_group()
        return
    owned = r.owned_queries()
    counts = [int(c) for c in r.counts]
    tuples = {}
    for a, b in owned:
        for i in range(a, b):
            tuples[i] = r.tuples(i).tolist()
    st = ws.kernel_stats()
    out_q.put((rank, owned, counts, tuples, {k: int(v) for k, v in r.summary.items()}, st["exchange"]["launches"], st["exchange"]["algorithmic_bytes"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,opts", [(2, {}), (3, {"sweep_min": 1, "sweep_tail": 64, "filter_min": 0, "filter_stream_min": 0}),
                                        (4, {"sweep_min": 1, "sweep_tail": 16}),
                                        (2, {"list_sort": 0, "global_sort_min": 1 << 40}), (2, {"_exchange": "allgather"}),
                                        (3, {"_exchange": "allgather", "sweep_min": 1, "sweep_tail": 64, "filter_min": 0, "filter_stream_min": 0})])
def test_collective_search_shards_lists_and_queries(V, oracle, world, opts):
    """The exchange step (SURVEY.md 8e strong scaling): `world` ranks on ONE device, the pieces moved through gloo on the host --
    every rank locates + sorts only its share of the distinct lists (the shares add up to the 1-GPU figure: nothing is located
    twice), receives sorted lists of the others, joins its piece of the queries; pieces are disjoint and cover the batch, tuples
    equal the single-process run and the oracle.  Default: the pairwise, needed-only exchange (vlg_workspace_set_exchange_alltoall:
    a list travels only to the ranks whose queries use it -- fewer bytes received than the all-gather's); "_exchange": "allgather" =
    every list to every rank (vlg_workspace_set_exchange)."""
    import socket
    import torch.multiprocessing as mp
    text = skewed_text(60000, 41).tobytes()
    rng = np.random.default_rng(43)
    queries = random_queries(text, rng, 400, kmax=4, mmax=3)
    one = V.VlgIndex.build(text).search(queries)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    procs = [ctx.Process(target=_collective_worker, args=(r, world, port, text, queries, opts, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(out_q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    covered = []
    total = one.summary["located_occurrences"]
    for rank, owned, counts, tuples, summ, xl, xbytes in got:
        assert len(owned) == 1 and xl >= 1
        # bytes received: everything but the own share with the all-gather, no more than that -- and less, unless the rank's queries
        # happen to use every list -- with the needed-only exchange
        if opts.get("_exchange") == "allgather":
            assert xbytes == 4 * (total - summ["located_occurrences"])
        else:
            assert xbytes <= 4 * (total - summ["located_occurrences"])
        a, b = owned[0]
        covered.append((a, b))
        assert all(c == 0 for i, c in enumerate(counts) if not (a <= i < b))
        for i in range(a, b):
            assert counts[i] == int(one.counts[i]) and tuples[i] == one.tuples(i).tolist(), (rank, i, queries[i])
    covered.sort()
    assert covered[0][0] == 0 and covered[-1][1] == len(queries)
    for (a0, b0), (a1, b1) in zip(covered, covered[1:]):
        assert b0 == a1
    assert sum(g[4]["n_matches"] for g in got) == one.summary["n_matches"]
    assert sum(g[4]["checksum"] for g in got) % (1 << 64) == one.summary["checksum"]
    located = [g[4]["located_occurrences"] for g in got]
    assert sum(located) == one.summary["located_occurrences"]                 # every distinct list located exactly once
    if opts.get("_exchange") != "allgather":
        assert sum(g[6] for g in got) < 4 * (world - 1) * total                # (all-gather: (world - 1) x every occurrence)
    assert max(located) <= one.summary["located_occurrences"] / world * 2 + 20000     # shares are cut between lists: balanced up to one list
    o = oracle.Index.from_text(text)
    for i in (0, 99, 250, 399):
        assert one.tuples(i).tolist() == o.search(queries[i]).tolist()


@pytest.mark.parametrize("mode", ["alltoall", "allgather"])
def test_collective_search_agrees_on_a_failed_rank(V, mode):
    """A rank whose share of a collective search cannot run (here: a workspace cap too small for its plan) says so in the status
    all-gather that precedes every exchange: vlg_search_batch returns an error on EVERY rank -- VLG_E_WORKSPACE on the one that
    failed, VLG_E_INTERNAL naming it on its peers -- and nobody is left waiting inside a collective."""
    import socket
    import torch.multiprocessing as mp
    text = skewed_text(60000, 41).tobytes()
    queries = random_queries(text, np.random.default_rng(43), 400, kmax=4, mmax=3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    opts = {"_exchange": mode, "_cap_of_rank": {1: 17 << 20}}                 # 17 MiB: 16 MiB are fixed overhead, the lists do not fit the rest
    procs = [ctx.Process(target=_collective_worker, args=(r, 2, port, text, queries, opts, out_q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(out_q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [g[1] for g in got] == ["error", "error"], got
    assert got[1][2] == V.capi.E_WORKSPACE
    assert got[0][2] == V.capi.E_INTERNAL and "rank 1 failed" in got[0][3]


def test_narrow_results_widen_on_fetch(torch_cuda, V, oracle, monkeypatch):
    """Positions that fit 32 bits are held 4 bytes wide in HBM: vlg_result_fetch hands out the 64-bit values of
    gapped_search_result::positions (utils.hpp:73-80) -- widened by host threads from pinned staging blocks, many blocks per thread
    on a result of a few million matches -- and vlg_result_fetch32 the same values as they are stored; results with 64-bit
    positions (VLG_FORCE_POS64=1, the wtsa index) refuse the narrow fetch."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    text = rng.choice(np.frombuffer(b"ab", np.uint8), 50_000_000, p=[0.7, 0.3]).tobytes()
    idx = V.VlgIndex.build(text)
    qs = ["ab.{0,3}?a", "a", "ba.{1,2}?b.{0,2}?a", "bbbbbbbbbbbbbbbbbbbbbbbbbbbbbbbbbbbbbb", "b.{0,0}?b"]
    res = idx.search(qs)
    counts, off, first, tup = res.fetch()
    assert res.summary["n_matches"] > 16 * (8 << 20) // 4                 # more than one staging block for each of the fetch threads
    t = np.frombuffer(text, np.uint8)
    a_at = np.flatnonzero(t == ord("a")).astype(np.uint64)
    assert (res.positions(1) == a_at).all()                                # every 'a' of the text, in order
    bb = np.flatnonzero((t[:-1] == ord("b")) & (t[1:] == ord("b")))
    keep, last = [], -2
    for p in bb.tolist():                                                  # non-overlapping, left to right
        if p > last + 1:
            keep.append(p); last = p
    assert res.positions(4).tolist() == keep
    assert int(first.sum(dtype=np.uint64)) == res.summary["checksum"] and counts[3] == 0
    f32, t32 = res.fetch32()
    assert f32.dtype == np.uint32 and (f32 == first).all() and (t32 == tup).all()
    small = idx.search(qs[:1] + qs[2:4])                                   # and a result far smaller than a staging block
    o = oracle.Index.from_text(text[:200_000])
    sidx = V.VlgIndex.build(text[:200_000])
    r2 = sidx.search(qs)
    for i, q in enumerate(qs):
        assert r2.tuples(i).tolist() == o.search(q).tolist(), q
    assert (r2.fetch32()[0] == r2.fetch()[2]).all() and small.summary["n_matches"] == int(counts[0] + counts[2])
    monkeypatch.setenv("VLG_FORCE_POS64", "1")
    wide = V.VlgIndex.build(text[:200_000]).search(qs)
    for x, y in zip(wide.fetch(), r2.fetch()):
        assert (x == y).all()
    with pytest.raises(V.VlgError):
        wide.fetch32()


@pytest.mark.parametrize("force", ["0", "1"])
def test_pivot_filter_through_the_ladder(torch_cuda, V, monkeypatch, force):
    """The pivot filter finds its index ranges by descending the 4-ary ladder over the sorted lists (one 16-byte load per level)
    or, with workspace option pivot_rungs = 0, by bracket searches and bisection: lists of a few elements up to millions, at every
    alignment inside the shared array, 32- and 64-bit positions -- same survivors' joins, same matches as the unfiltered join."""
    from vlg_matching_amd.index import Workspace
    monkeypatch.setenv("VLG_FORCE_POS64", force)
    rng = np.random.default_rng(31)
    n = 3_000_000
    text = bytearray(rng.choice(np.frombuffer(b"abc", np.uint8), n, p=[0.86, 0.12, 0.02]).tobytes())
    for p in rng.integers(0, n - 8, 40):
        text[p:p + 3] = b"xyz"                                             # a rare sub-pattern: pivot lists of a few elements
    text = bytes(text)
    idx = V.VlgIndex.build(text)
    qs = []
    subs = ["a", "b", "c", "ab", "ba", "aa", "cb", "bc", "xyz", "abc", "aab", "cc", "bbb", "ca", "aaaa", "acb"]
    for _ in range(300):
        k = int(rng.integers(2, 5))
        parts = [subs[int(rng.integers(0, len(subs)))] for _ in range(k)]
        q = parts[0]
        for s in parts[1:]:
            lo = int(rng.integers(0, 30))
            q += ".{%d,%d}?%s" % (lo, lo + int(rng.choice([0, 4, 60, 900, 20000])), s)
        qs.append(q)
    res = {}
    for name, opts in (("plain", {"filter": 0}), ("bisect", {"pivot_rungs": 0}), ("ladder", {"pivot_rungs": 2})):
        ws = Workspace()
        for k_, v_ in {"filter_min": 0, "filter_stream_min": 0, "filter_pivot": 1, "filter_pivot_ratio": 2, **opts}.items():
            ws.set_option(k_, v_)
        res[name] = idx.search(qs, workspace=ws)
        if name != "plain":
            assert ws.kernel_stats()["filter_pivot"]["launches"] > 0 and ws.kernel_stats()["filter_compact"]["launches"] > 0
    assert res["ladder"].summary["join_slots"] == res["bisect"].summary["join_slots"] < res["plain"].summary["join_slots"]
    for name in ("bisect", "ladder"):
        for k_ in ("n_matches", "checksum", "n_tuple_values", "logical_occurrences"):
            assert res[name].summary[k_] == res["plain"].summary[k_], (name, k_)
        for x, y in zip(res[name].fetch(), res["plain"].fetch()):
            assert (x == y).all(), name


def test_window_filter_is_skipped_when_it_cannot_pay(torch_cuda, V, oracle):
    """A batch of very many small queries (BASELINE config 4 in miniature): the candidates of the window filter hold too few join slots
    to pay its per-query host cost, so the batch is joined as it is -- same matches; filter_min = 0 forces the filter on it."""
    from vlg_matching_amd.index import Workspace
    rng = np.random.default_rng(12)
    text = rng.choice(np.frombuffer(b"abcdefgh", np.uint8), 400_000).tobytes()
    o = oracle.Index.from_text(text)
    idx = V.VlgIndex.build(text)
    qs = []
    for _ in range(6000):                                                  # sub-patterns of 3..5 characters: lists of a few hundred elements
        parts = []
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(0, len(text) - 6))
            parts.append(text[at:at + int(rng.integers(3, 6))].decode())
        q = parts[0]
        for sp in parts[1:]:
            q += ".{%d,%d}?%s" % (1, 1 + int(rng.integers(0, 40)), sp)
        qs.append(q)
    qs[7] = "a.{0,60}?b.{0,60}?c"                                           # a few heavy queries among them: they alone would be filtered
    qs[1900] = "e.{1,9}?a"
    ws_auto, ws_forced = Workspace(), Workspace()
    ws_forced.set_option("filter_min", 0)
    ws_forced.set_option("filter_stream_min", 0)
    a, b = idx.search(qs, workspace=ws_auto), idx.search(qs, workspace=ws_forced)
    assert ws_auto.kernel_stats()["filter_compact"]["launches"] == 0 and ws_forced.kernel_stats()["filter_compact"]["launches"] > 0
    assert b.summary["join_slots"] < a.summary["join_slots"]
    for x, y in zip(a.fetch(), b.fetch()):
        assert (x == y).all()
    for i in (7, 1900, 0, 5999):
        assert a.tuples(i).tolist() == o.search(qs[i]).tolist(), qs[i]
