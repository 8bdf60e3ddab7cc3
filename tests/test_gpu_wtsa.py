"""The paper's index on the GPU: text + wavelet tree over the suffix array, searched lazily
(sdsl::vlg_index / vlg_iterator, include/sdsl/vlg_index.hpp:109-373), byte and integer alphabets -- against the reference's known
answers, the CPU oracle (the survey established vlg_index == merge join tuple for tuple) and brute force."""
import json
import os

import numpy as np
import pytest

from util import dna_text, naive_sa, skewed_text

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vlg_known_answers.json")))


@pytest.fixture(scope="module")
def V():
    import vlg_matching_amd as v
    v.lib()
    return v


def test_known_answers_byte_and_int(V):
    for case in GOLD["cases"]:
        idx = V.WtsaIndex(case["text"].encode())
        if "error" in case:
            with pytest.raises(V.VlgError):
                idx.search([case["query"]])
        else:
            assert idx.search([case["query"]]).tuples(0).tolist() == case["tuples"], case
    for case in GOLD["int_cases"]:
        idx = V.WtsaIndex(np.array(case["int_text"], dtype=np.uint32))
        r = idx.search([case["query"]])
        assert r.tuples(0).tolist() == case["tuples"], case
        assert int(r.counts[0]) == len(case["tuples"])
        if r.summary["n_matches"]:
            with pytest.raises(V.VlgError):                                 # the lazy index keeps 64-bit positions: no narrow fetch
                r.fetch32()


@pytest.mark.parametrize("name", ["abracadabra", "one_byte", "100a", "dna", "zipf", "empty"])
def test_suffix_array_access_and_ranges(V, oracle, name):
    import torch
    text = {"abracadabra": b"abracadabrasimsalabim", "one_byte": b"a", "100a": b"a" * 100, "dna": dna_text(3000, 4).tobytes(),
            "zipf": skewed_text(5000, 6).tobytes(), "empty": b""}[name]
    idx = V.WtsaIndex(text)
    info = idx.info()
    assert info["n"] == len(text) + 1 and info["symbol_bytes"] == 1
    tz = np.frombuffer(text + b"\0", dtype=np.uint8)
    sa = oracle.suffix_array(tz) if len(text) else np.zeros(1, np.uint64)
    # wt[i] == SA[i] for every i (wt_int::operator[]; csa_byte_test.cpp:136-147 checks csa[j] == SA[j] the same way)
    d_i = torch.arange(len(sa), dtype=torch.int64, device="cuda")
    d_o = torch.zeros_like(d_i)
    idx.sa_device(d_i.data_ptr(), d_o.data_ptr(), len(sa))
    torch.cuda.synchronize()
    assert (d_o.cpu().numpy().view(np.uint64) == sa).all()
    if not len(text):
        return
    # forward_search == backward_search of the FM-index oracle (same suffix array, same ranges)
    o = oracle.Index.from_text(text)
    rng = np.random.default_rng(3)
    pats = [text[s:s + int(rng.integers(1, 6))] for s in rng.integers(0, len(text), 60)] + [b"\xfe", text[:1] * 300, text[-3:]]
    qs = [p.decode("latin-1") for p in pats]
    sp, ep = idx.ranges(qs)
    for p, a, b in zip(pats, sp, ep):
        cnt, l, r = o.backward_search(p)
        assert int(b) + 1 - int(a) == cnt, p
        if cnt:
            assert (int(a), int(b)) == (l, r), p


def _random_queries(text, rng, nq, kmax=4, mmax=4, gapmax=60, gaplo=20):
    qs = []
    for _ in range(nq):
        k = int(rng.integers(1, kmax + 1))
        subs = [text[s:s + int(rng.integers(1, mmax + 1))] for s in rng.integers(0, max(len(text) - mmax - 1, 1), k)]
        q = subs[0].decode("latin-1")
        for sp in subs[1:]:
            a = int(rng.integers(0, gaplo))
            q += ".{%d,%d}?%s" % (a, a + int(rng.integers(0, gapmax)), sp.decode("latin-1"))
        qs.append(q)
    return qs


@pytest.mark.parametrize("name,seed", [("dna", 1), ("dna_skew", 2), ("zipf", 3), ("100a", 4), ("abab", 5)])
def test_lazy_search_equals_oracle_and_fm_index_path(V, oracle, monkeypatch, name, seed):
    text = {"dna": dna_text(20000, 1).tobytes(), "dna_skew": dna_text(15000, 2, (0.7, 0.1, 0.1, 0.1)).tobytes(),
            "zipf": skewed_text(20000, 3).tobytes(), "100a": b"a" * 100, "abab": (b"ab" * 1200) + b"aab" * 200}[name]      # (abab: matches by the ten thousand per query -- kept short)
    rng = np.random.default_rng(seed)
    qs = _random_queries(text, rng, 150)
    qs += ["\xfe.{0,5}?" + qs[0][:1], qs[1][:1] + ".{0,5}?\xfe", qs[2][:1], text[:2].decode() + ".{0,100000000}?" + text[5:7].decode()]
    w = V.WtsaIndex(text)
    o = oracle.Index.from_text(text)
    fm = V.VlgIndex.build(text).search(qs)
    res = w.search(qs)
    total, chk = 0, 0
    for i, q in enumerate(qs):
        want = o.search(q)
        assert res.tuples(i).tolist() == want.tolist(), q
        assert res.tuples(i).tolist() == fm.tuples(i).tolist()
        total += len(want)
        chk = (chk + int(want[:, 0].sum())) % (1 << 64) if len(want) else chk
    assert res.summary["n_matches"] == total and res.summary["checksum"] == chk
    # lazily: the first N matches of every query are a prefix of all of them (an iterator that is not run to its end)
    for cap in (1, 3, 70):
        part = w.search(qs, max_matches=cap)
        for i in range(len(qs)):
            assert part.tuples(i).tolist() == res.tuples(i).tolist()[:cap], (cap, qs[i])
    # one lane per query (the round-2 kernel) instead of one wavefront: same tuples
    monkeypatch.setenv("VLG_WTSA_LANE_PER_QUERY", "1")
    old = w.search(qs)
    for x, y in zip(old.fetch(), res.fetch()):
        assert (x == y).all()
    assert [w.search(qs, max_matches=2).tuples(i).tolist() for i in range(len(qs))] == [res.tuples(i).tolist()[:2] for i in range(len(qs))]
    monkeypatch.delenv("VLG_WTSA_LANE_PER_QUERY")
    # first positions only
    from vlg_matching_amd.index import Workspace
    ws = Workspace()
    ws.set_option("tuples", 0)
    fp = w.search(qs, workspace=ws)
    assert (fp.counts == res.counts).all() and fp.summary["checksum"] == chk and fp.summary["n_tuple_values"] == 0
    for i in (0, 7, len(qs) - 1):
        assert fp.positions(i).tolist() == res.tuples(i)[:, 0].tolist() if int(res.counts[i]) else len(fp.positions(i)) == 0


def _int_occurrences(text, pat):
    n, m = len(text), len(pat)
    if m > n:
        return np.zeros(0, np.uint64)
    ok = np.ones(n - m + 1, dtype=bool)
    for t in range(m):
        ok &= text[t:n - m + 1 + t] == pat[t]
    return np.nonzero(ok)[0].astype(np.uint64)


def test_integer_alphabet_index_vs_brute_force(V, oracle):
    """vlg_index<int_alphabet_tag>: symbols far beyond a byte, zero as a symbol, queries as whitespace-separated decimals."""
    rng = np.random.default_rng(11)
    vocab = np.array([0, 1, 2, 255, 256, 1000, 65535, 65536, 2 ** 31, 2 ** 32 - 1, 7, 8], dtype=np.uint64)
    text = vocab[rng.choice(len(vocab), 6000, p=np.array([5, 5, 4, 3, 3, 2, 2, 1, 1, 1, 4, 4]) / 35.0)].astype(np.uint32)
    idx = V.WtsaIndex(text)
    info = idx.info()
    assert info["symbol_bytes"] == 4 and info["n"] == len(text) + 1
    # the suffix array: integer symbols compare as numbers, the sentinel is the smallest
    import torch
    sa = np.array(sorted(range(len(text) + 1), key=lambda i: [int(x) + 1 for x in text[i:]] + [0]), dtype=np.uint64) if len(text) <= 6000 else None
    d_i = torch.arange(len(text) + 1, dtype=torch.int64, device="cuda")
    d_o = torch.zeros_like(d_i)
    idx.sa_device(d_i.data_ptr(), d_o.data_ptr(), len(text) + 1)
    torch.cuda.synchronize()
    assert (d_o.cpu().numpy().view(np.uint64) == sa).all()
    qs, parsed = [], []
    for _ in range(120):
        k = int(rng.integers(1, 4))
        subs = [text[s:s + int(rng.integers(1, 4))] for s in rng.integers(0, len(text) - 4, k)]
        gaps = [(a, a + int(rng.integers(0, 40))) for a in rng.integers(0, 10, k - 1)]
        q = " ".join(str(int(x)) for x in subs[0])
        for (a, b), sp in zip(gaps, subs[1:]):
            q += " .{%d,%d}? " % (a, b) + " ".join(str(int(x)) for x in sp)
        qs.append(q)
        parsed.append((subs, gaps))
    qs.append("4242 .{0,5}? 7")                                             # a symbol that does not occur
    parsed.append(([np.array([4242], np.uint32), np.array([7], np.uint32)], [(0, 5)]))
    res = idx.search(qs)
    for i, (subs, gaps) in enumerate(parsed):
        lists = [_int_occurrences(text, sp) for sp in subs]
        lo = [a + len(subs[j]) for j, (a, b) in enumerate(gaps)]                # vlg_index.hpp:95: gaps count symbols
        hi = [b + len(subs[j]) for j, (a, b) in enumerate(gaps)]
        m, want = oracle.join(lists, lo, hi, len(subs[-1])) if all(len(l) for l in lists) else (0, np.zeros((0, len(subs)), np.uint64))
        assert res.tuples(i).tolist() == want.tolist(), qs[i]
    part = idx.search(qs, max_matches=2)
    for i in range(len(qs)):
        assert part.tuples(i).tolist() == res.tuples(i).tolist()[:2]
    # a byte batch is refused by an integer index and the other way round; so is an integer batch by the FM-index entry points
    with pytest.raises(V.VlgError):
        V.capi.check(V.lib().vlg_wtsa_search_batch(idx._h, V.index.Queries(["a"])._h, 0, V.index.Workspace()._h, None)) if False else idx.search(V.index.Queries(["a"]))
    with pytest.raises(V.VlgError):
        V.VlgIndex.build(b"abcabc").search(idx.queries(["1 2"]))


@pytest.mark.parametrize("name", ["abracadabra", "one_byte", "100a", "dna", "zipf", "ints", "keeper"])
def test_tree_equals_reference_wt_int(V, oracle, refmod, name):
    """The device tree against the reference's OWN wt_int<bit_vector_il<>, rank_support_il<>> (oracle/_ref, built by its constructor
    from the same suffix array, as construct(wts, KEY_SA) does, vlg_index.hpp:386-387): number of levels, every level's bits ==
    wt_int::tree, wt[i] for every i (wt_int.hpp:339-361), and count_less / quantile on random suffix-array ranges == what the
    reference's expand(v) / expand(v, range) descent answers (wt_int.hpp:824-939)."""
    import torch
    if name in ("ints", "keeper"):
        rng = np.random.default_rng(8)
        if name == "keeper":                     # the reference's own integer fixture (test/test_cases/keeper.int; csa_int_test.config:7)
            import os
            itext = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keeper.int"), dtype="<u8").astype(np.uint32)
            assert len(itext) == 63 and int(itext.max()) == 21
        else:
            itext = rng.choice(np.array([3, 7, 7, 19, 1000, 70000, 2 ** 31 + 5], dtype=np.uint32), 1500)
        idx = V.WtsaIndex(itext)
        vals = np.concatenate([itext.astype(np.int64) + 1, [0]])                 # the sentinel is smaller than every symbol
        sa = np.array(sorted(range(len(vals)), key=lambda i: vals[i:].tolist()), dtype=np.uint64)
    else:
        text = {"abracadabra": b"abracadabrasimsalabim", "one_byte": b"a", "100a": b"a" * 100, "dna": dna_text(3000, 4).tobytes(),
                "zipf": skewed_text(5000, 6).tobytes()}[name]
        idx = V.WtsaIndex(text)
        sa = oracle.suffix_array(np.frombuffer(text + b"\0", dtype=np.uint8))
    n = len(sa)
    ref = oracle.RefWtInt(sa)
    info = idx.info()
    assert info["n"] == n and info["levels"] == ref.levels
    want_bits = ref.level_bits()
    for lvl in range(ref.levels):
        assert (idx.level_bits(lvl) == want_bits[lvl]).all(), lvl
    d_i = torch.arange(n, dtype=torch.int64, device="cuda")
    d_o = torch.zeros_like(d_i)
    idx.sa_device(d_i.data_ptr(), d_o.data_ptr(), n)
    torch.cuda.synchronize()
    got = d_o.cpu().numpy().view(np.uint64)
    assert [int(x) for x in got] == [ref[i] for i in range(n)]
    rng = np.random.default_rng(17)
    m = 400
    l = rng.integers(0, n, m).astype(np.uint64)
    ln = np.array([rng.integers(1, n - int(a) + 1) for a in l], dtype=np.uint64)
    x = rng.integers(0, n + 3, m).astype(np.uint64)
    q = np.array([rng.integers(0, int(b)) for b in ln], dtype=np.uint64)

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    d_l, d_n, d_x, d_q, d_out = dev(l), dev(ln), dev(x), dev(q), torch.zeros(m, dtype=torch.int64, device="cuda")
    idx.range_walk_device(d_l.data_ptr(), d_n.data_ptr(), d_x.data_ptr(), False, d_out.data_ptr(), m)
    torch.cuda.synchronize()
    assert [int(v) for v in d_out.cpu().numpy().view(np.uint64)] == [ref.count_less(a, b, c) for a, b, c in zip(l, ln, x)]
    idx.range_walk_device(d_l.data_ptr(), d_n.data_ptr(), d_q.data_ptr(), True, d_out.data_ptr(), m)
    torch.cuda.synchronize()
    assert [int(v) for v in d_out.cpu().numpy().view(np.uint64)] == [ref.quantile(a, b, c) for a, b, c in zip(l, ln, q)]
    # out-of-range requests are refused per element, not walked
    bad = dev(np.array([n + 5], dtype=np.uint64))
    one = dev(np.array([1], dtype=np.uint64))
    idx.range_walk_device(bad.data_ptr(), one.data_ptr(), one.data_ptr(), False, d_out.data_ptr(), 1)
    torch.cuda.synchronize()
    assert int(d_out.cpu().numpy().view(np.uint64)[0]) == (1 << 64) - 1
