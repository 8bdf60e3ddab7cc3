"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/vlg_hip.h declares, and refuses to compute without a GPU (no silent fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def V():
    import vlg_matching_amd as v
    if not os.path.exists(v.library_path()):
        v.build_library()
    return v


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "vlg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vlg_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound(V):
    L = C.CDLL(V.library_path())
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "library does not export " + n
    bound = {s[0] for s in V.capi.SYMBOLS}
    assert set(names) == bound, set(names) ^ bound


def test_no_gpu_means_loud_failure(V):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(V.VlgError) as e:
        V.VlgIndex.build(b"abracadabra")
    assert e.value.status == V.capi.E_NO_DEVICE
    with pytest.raises(V.VlgError):
        V.BitVector(np.zeros(1, np.uint64), 10)
    with pytest.raises(V.VlgError):
        V.index.Workspace()


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under vlg_matching_amd/ or include/ may reference it."""
    bad = []
    for base in ("vlg_matching_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".c", "Makefile")):
                    s = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"oracle|vlgo_|libvlgref|/root/reference", s):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_host_parser_matches_oracle_parser(V, oracle):
    """vlg_parse_query (host logic of the product) against the oracle's restatement of both reference dialects."""
    cases = ["ab.{2,5}?cde.{0,7}?f", "a", "abc.{0,0}?abc", "a.{1,2}b", "a.{3,2}?b", "a.{x,2}?b", "a.{1,2", ".{1,2}?b", "a.{1,2}?",
             "a.{ 1, 2}?b", "a.{+1,2}?b", "a.{1,2}?b.{3,4}c", "x.{0,18446744073709551616}?y", "x.{0,4611686018427387903}?y",
             "ab.{2,5}cde.{0,7}f", "A.{0,100}C", "A.{0,100}?C", "q.{1}?r", "a.{1,2}?b.{5,6}?c.{7,8}?d"]
    for dialect in (0, 1):
        for c in cases:
            try:
                want = oracle.query_fields(oracle.parse(c, dialect))
            except oracle.ParseError as e:
                with pytest.raises(V.VlgError) as ee:
                    V.parse_query(c, dialect)
                assert ee.value.status == (V.capi.E_INVALID if e.code == 4 else V.capi.E_PARSE), c
                continue
            subs, lo, hi, end = V.parse_query(c, dialect)
            assert (subs, lo, hi, end) == (want[0], want[1], want[2], want[3]), (c, dialect)


def test_huge_gap_bounds_are_rejected_not_wrapped(V):
    """The reference wraps `max + |s|` modulo 2^64 (vlg_index.hpp:95); the product rejects bounds >= 2^62."""
    for c in ("x.{0,18446744073709551615}?y", "x.{0,4611686018427387904}?y"):
        with pytest.raises(V.VlgError) as e:
            V.parse_query(c, 0)
        assert e.value.status == V.capi.E_INVALID


def test_cpp_host_binaries_fail_loudly_without_gpu(V):
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = os.path.join(ROOT, "vlg_matching_amd", "bin", "vlg_matching_example")
    if not os.path.exists(exe):
        V.build_library()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device" in r.stderr


def test_sdsl_file_reader_on_reference_written_image(V, refmod, tmp_path):
    """a csa_wt<wt_huff<>> file whose members were serialised by the reference's own code parses into exactly
    the parts the oracle builds (host-only logic, no GPU)."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from util import bwt_from_sa, dna_text, skewed_text
    O = refmod
    for name, text in [("abra", b"abracadabrasimsalabim"), ("dna", dna_text(20000, 3).tobytes()), ("zipf", skewed_text(30000, 4).tobytes()),
                       ("allsym", bytes(range(1, 256)) * 3), ("one", b"a")]:
        tz = np.frombuffer(text + b"\0", dtype=np.uint8)
        sa = O.suffix_array(tz)
        R = O.RefIndex(bwt_from_sa(tz, sa), sa, 0)
        path = tmp_path / (name + ".sdsl")
        R.write_csa_image(path, sa)
        got = V.index.read_sdsl_file(path)
        want = O.Index.from_text(text).parts()
        assert got["n"] == want["n"] and got["sigma"] == want["sigma"] and got["bv_bits"] == want["bv_bits"]
        assert (got["char2comp"] == want["char2comp"]).all() and (got["C"] == want["C"]).all()
        assert (got["bv_words"] == want["bv_words"]).all() and (got["samples"] == want["samples"]).all()
        for f in ("bv_pos", "bv_pos_rank", "parent", "child"):
            assert (got["nodes"][f] == want["nodes"][f]).all(), f
        # the CPU algorithm runs on the parsed parts
        o2 = O.Index.from_parts(got)
        assert o2.search("a.{0,10}?a").tolist() == O.Index.from_text(text).search("a.{0,10}?a").tolist()
    (tmp_path / "junk.sdsl").write_bytes(b"not an index at all" * 10)
    with pytest.raises(V.VlgError):
        V.index.read_sdsl_file(tmp_path / "junk.sdsl")
    with pytest.raises(V.VlgError):
        V.index.read_sdsl_file(tmp_path / "dna.sdsl", dens=16)     # wrong density for this file


def test_sdsl_file_reader_survives_damaged_files(V, refmod, tmp_path):
    """Truncated and bit-flipped csa_wt files: vlg_sdsl_file_open either parses them or reports VLG_E_INVALID -- no crash,
    no runaway allocation.  Run in a child process so that a crash is a test failure."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from util import bwt_from_sa, dna_text
    O = refmod
    text = dna_text(3000, 8).tobytes()
    tz = np.frombuffer(text + b"\0", dtype=np.uint8)
    sa = O.suffix_array(tz)
    good = tmp_path / "good.sdsl"
    O.RefIndex(bwt_from_sa(tz, sa), sa, 0).write_csa_image(good, sa)
    script = r'''
import sys, os, numpy as np
sys.path.insert(0, sys.argv[1])
import vlg_matching_amd as V
raw = bytearray(open(sys.argv[2], "rb").read())
rng = np.random.default_rng(5)
tmp = sys.argv[2] + ".bad"
ok = bad = 0
cases = [bytes(raw[:n]) for n in list(range(0, 64)) + [int(x) for x in rng.integers(64, len(raw), 150)]]
for _ in range(400):
    b = bytearray(raw)
    for pos in rng.integers(0, len(b), int(rng.integers(1, 4))):
        b[pos] = int(rng.integers(0, 256))
    cases.append(bytes(b))
for i, v in enumerate([0, 1, 2**63, 2**64 - 1, 2**40]):             # absurd sizes in the first header words
    for off in (0, 8, 16):
        b = bytearray(raw); b[off:off + 8] = int(v).to_bytes(8, "little"); cases.append(bytes(b))
for c in cases:
    open(tmp, "wb").write(c)
    try:
        V.index.read_sdsl_file(tmp)
        ok += 1
    except V.VlgError:
        bad += 1
print("done", ok, bad)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", script, root, str(good)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("done"), (r.returncode, r.stdout[-300:], r.stderr[-600:])
    ok, bad = (int(x) for x in r.stdout.split()[1:3])
    assert bad > 200                                          # most damage is detected; a flipped payload bit can still parse


def test_sdsl_file_reader_on_reference_written_rrr_image(V, refmod, tmp_path):
    """A csa_wt<wt_huff<rrr_vector<63>>> image whose wavelet tree was built and serialised by the reference's OWN rrr_vector<63>
    (oracle/_ref, variant 2: include/sdsl/rrr_vector.hpp:145-237, 349-372) parses -- every block decoded from its class and offset,
    inverted super-blocks included -- into exactly the plain bit-vector, tree, samples and alphabet of the plain index."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from util import bwt_from_sa, dna_text, skewed_text
    O = refmod
    rng = np.random.default_rng(7)
    dense = bytes(rng.choice(np.frombuffer(b"aaaaaaaaaaaaaaab", dtype=np.uint8), 40000).tolist())   # long runs: inverted super-blocks
    for name, text in [("abra", b"abracadabrasimsalabim"), ("dna", dna_text(20000, 3).tobytes()), ("zipf", skewed_text(30000, 4).tobytes()),
                       ("dense", dense), ("one", b"a"), ("63", b"ab" * 63), ("allsym", bytes(range(1, 256)) * 3)]:
        tz = np.frombuffer(text + b"\0", dtype=np.uint8)
        sa = O.suffix_array(tz)
        R = O.RefIndex(bwt_from_sa(tz, sa), sa, 2)
        path = tmp_path / (name + ".rrr.sdsl")
        R.write_csa_image(path, sa)
        got = V.index.read_sdsl_file(path, rrr=True)
        want = O.Index.from_text(text).parts()
        assert got["n"] == want["n"] and got["sigma"] == want["sigma"] and got["bv_bits"] == want["bv_bits"], name
        assert (got["bv_words"] == want["bv_words"]).all(), name
        assert (got["char2comp"] == want["char2comp"]).all() and (got["C"] == want["C"]).all() and (got["samples"] == want["samples"]).all()
        for f in ("bv_pos", "parent", "child"):
            assert (got["nodes"][f] == want["nodes"][f]).all(), f
    with pytest.raises(V.VlgError):
        V.index.read_sdsl_file(tmp_path / "dna.rrr.sdsl", rrr=False)        # an rrr image is not a plain one


def test_rccl_loader_reports_a_missing_library(V):
    """vlg_comm_* bind RCCL at run time.  When the chosen library cannot be opened every entry point returns VLG_E_UNSUPPORTED
    with the loader's message (it used to read dlerror() twice and crash on the second, cleared, answer).  VLG_RCCL_LIBRARY is an
    explicit choice: nothing else is probed behind it.  Child process: the binding is made once per process."""
    import subprocess
    import sys
    script = r'''
import sys, ctypes as C
sys.path.insert(0, sys.argv[1])
import vlg_matching_amd as V
L = V.lib()
buf = (C.c_uint8 * 128)()
s = L.vlg_comm_unique_id(C.cast(buf, C.c_void_p))
print("status", s, (L.vlg_last_error() or b"").decode())
s2 = L.vlg_comm_info(C.c_void_p(1), None, None)
print("status2", s2)
'''
    env = dict(os.environ, VLG_RCCL_LIBRARY="/nonexistent/librccl_bogus.so")
    r = subprocess.run([sys.executable, "-c", script, ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-600:])
    lines = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines())
    assert lines["status"].startswith("%d " % V.capi.E_UNSUPPORTED), r.stdout
    assert "RCCL not found" in lines["status"] and "librccl_bogus" in lines["status"]
    assert lines["status2"].strip() == str(V.capi.E_UNSUPPORTED)


def test_symbol_map_is_dense_and_order_preserving(V):
    """vlg_symbol_map (64-bit symbols for the uint32 integer indexes): symbol -> rank + 1 among the text's distinct symbols -- dense,
    ascending with the symbol (so the mapped text has the original's suffix order), never 0; a symbol the text does not hold maps to
    sigma + 1.  Host-only: works without a GPU."""
    rng = np.random.default_rng(2)
    vocab = np.array([0, 1, 7, 2 ** 32 - 1, 2 ** 32, 2 ** 40 + 3, 2 ** 63, 2 ** 64 - 1], dtype=np.uint64)
    text = vocab[rng.integers(0, len(vocab), 5000)]
    m = V.SymbolMap(text)
    assert m.sigma == len(vocab) and m.symbols().tolist() == sorted(vocab.tolist())
    mapped = m.apply(text)
    assert mapped.dtype == np.uint32 and mapped.min() == 1 and mapped.max() == m.sigma
    order = np.argsort(vocab)
    rank = {int(vocab[i]): r + 1 for r, i in enumerate(order)}
    assert mapped.tolist() == [rank[int(x)] for x in text]
    assert m.apply(np.array([5, 2 ** 50, 2 ** 64 - 2], dtype=np.uint64)).tolist() == [m.sigma + 1] * 3
    big = rng.integers(0, 2 ** 63, 3_000_000, dtype=np.uint64)                 # the threaded path
    mb = V.SymbolMap(big)
    got = mb.apply(big)
    assert (mb.symbols()[got.astype(np.int64) - 1] == big).all()
    assert V.SymbolMap(np.zeros(0, dtype=np.uint64)).sigma == 0
