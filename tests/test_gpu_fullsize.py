"""BASELINE-size workloads on the GPU, checked through size-independent properties (the oracle cannot build a 100 MB
index in seconds, but it can adopt the device-built parts and answer a bounded sample of the same queries):
  * every tuple obeys its gap bounds, tuples are ascending and non-overlapping;
  * every reported position really is an occurrence of its sub-pattern in the text;
  * interval sharing on/off, sorted-sweep vs random-access locate, the window filter on/off all give identical results;
  * a random sample of queries equals the CPU oracle tuple for tuple."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Lap:
    """VLG_TEST_TIMING=1: where a full-size test spends its time (stderr, run with -s)."""
    def __init__(self, name):
        import os, time
        self.on, self.name, self.t0, self.clock = bool(os.environ.get("VLG_TEST_TIMING")), name, time.time(), time.time

    def __call__(self, what):
        if self.on:
            import sys
            t1 = self.clock()
            print("[lap] %s: %-40s %7.1f s" % (self.name, what, t1 - self.t0), file=sys.stderr, flush=True)
            self.t0 = t1


@pytest.fixture(scope="module")
def c2():
    import vlg_matching_amd as V
    from vlg_matching_amd import workload
    cfg = workload.config("C2")                       # 100 MiB DNA-like text, 10 000 queries, k=2, m=10, gap <= 100
    text = workload.gen_text(cfg["kind"], cfg["n"], cfg["seed"])
    idx = V.VlgIndex.build(text)
    parts = workload.gen_query_parts(text, cfg["nq"], cfg["k"], cfg["m"], cfg["qseed"])
    # mix in short, frequent sub-patterns so that lists of 10^5..10^6 occurrences and real matches occur
    heavy = workload.gen_query_parts(text, 400, 3, 5, cfg["qseed"] + 1)
    g = ".{%d,%d}?" % cfg["gap"]
    queries = [g.join(s.decode() for s in subs) for subs in parts] + [".{0,300}?".join(s.decode() for s in subs) for subs in heavy]
    return V, text, idx, queries, parts + heavy, cfg


def test_c2_properties_and_mode_equivalence(c2):
    V, text, idx, queries, parts, cfg = c2
    from vlg_matching_amd.index import Workspace
    from vlg_matching_amd import workload
    workload.check_expected("C2", idx.search(queries[:cfg["nq"]]).summary)     # the config's own batch: the constants bench.py asserts too
    base = idx.search(queries)
    s = base.summary
    assert s["n_queries"] == len(queries) and s["n_matches"] > 1000
    counts, offsets, first, tuples = base.fetch()
    # --- structural properties of every match ------------------------------------------------------------
    ks = np.asarray(base._ks, dtype=np.int64)
    toff = np.concatenate([[0], np.cumsum(counts.astype(np.int64) * ks)])
    checked = 0
    for qi in np.nonzero(counts)[0]:
        k = int(ks[qi])
        t = tuples[toff[qi]: toff[qi + 1]].reshape(-1, k).astype(np.int64)
        subs = parts[qi]
        lo, hi = (0, 100) if qi < cfg["nq"] else (0, 300)
        for i in range(1, k):
            d = t[:, i] - t[:, i - 1] - len(subs[i - 1])
            assert (d >= lo).all() and (d <= hi).all()
        assert (t[1:, 0] >= t[:-1, -1] + len(subs[-1])).all()               # non-overlapping, left to right
        for i in range(k):                                                  # positions are real occurrences
            for p in t[:50, i]:
                assert text[p: p + len(subs[i])].tobytes() == subs[i]
        checked += len(t)
    assert checked == s["n_matches"]
    assert int(first.astype(np.uint64).sum()) % (1 << 64) == s["checksum"]
    # --- identical results whatever the execution strategy ------------------------------------------------
    # (sweep_tail: how many elements are left to the stragglers' walk -- none of the sort, a third of it, all of it)
    for opts in ({"dedup": 0}, {"sweep": 0}, {"filter_min": 0, "filter_stream_min": 0}, {"sweep_min": 1, "sweep_tail": 1000},
                 {"sweep_min": 1, "sweep_tail": 1 << 25}, {"sweep_min": 1, "sweep_tail": 1 << 40, "trail": 0}):
        ws = Workspace()
        for k_, v_ in opts.items():
            ws.set_option(k_, v_)
        r = idx.search(queries, workspace=ws)
        assert r.summary["n_matches"] == s["n_matches"] and r.summary["checksum"] == s["checksum"], opts
        f2 = r.fetch()
        assert (f2[0] == counts).all() and (f2[3] == tuples).all(), opts
        if "dedup" in opts:
            assert r.summary["located_occurrences"] == s["logical_occurrences"] >= s["located_occurrences"]


def test_c2_sample_equals_oracle(c2, oracle):
    V, text, idx, queries, parts, cfg = c2
    o = oracle.Index.from_parts(idx.export_parts())                         # device-built parts, CPU algorithm
    res = idx.search(queries)
    rng = np.random.default_rng(99)
    light = rng.choice(cfg["nq"], 300, replace=False)
    heavy = cfg["nq"] + rng.choice(len(queries) - cfg["nq"], 6, replace=False)
    for qi in list(light) + list(heavy):
        assert res.tuples(int(qi)).tolist() == o.search(queries[int(qi)]).tolist(), queries[int(qi)]
    # SA samples of the device build: csa[i] == true suffix position for sampled i (csa_byte_test.cpp:136-147 restated)
    p = idx.export_parts()
    n = p["n"]
    for j in rng.integers(0, len(p["samples"]), 200):
        sa = int(p["samples"][j])
        i = int(j) * 32
        # suffix sa must be lexicographically between its neighbours' samples is too weak a check;
        # instead verify the LF/BWT relation: rank of suffix (sa) in SA order is i  <=>  csa[i] == sa
        assert o.sa(i) == sa


def _check_device_suffix_array(text, sa, rng, samples=20000):
    """The suffix array the SASEARCH restatement runs on comes from the device sorter: it must be a permutation of [0, n] whose
    neighbouring suffixes ascend (sampled; csa_byte_test.cpp:136-147 checks csa[j] == SA[j] the same way)."""
    n = len(text) + 1
    assert len(sa) == n and int(sa[0]) == n - 1                              # the sentinel suffix sorts first
    seen = np.zeros(n, dtype=bool)
    seen[sa] = True
    assert seen.all()
    del seen
    for i in rng.integers(1, n - 1, samples):
        a, b = int(sa[i]), int(sa[i + 1])
        x, y = text[a: a + 96].tobytes(), text[b: b + 96].tobytes()          # a proper prefix (suffix at the text end) sorts first
        assert x < y or (x == y and len(x) == 96), (i, a, b)


def test_c3_headline_config_modes_and_oracle_sample(oracle):
    """The headline workload (1 GiB english-like text, 100 000 x 3 sub-patterns, gap <= 1000) at full size."""
    import torch
    import vlg_matching_amd as V
    from vlg_matching_amd import workload
    from vlg_matching_amd.index import Queries, Workspace
    cfg = workload.config("C3")
    text = workload.gen_text(cfg["kind"], cfg["n"], cfg["seed"])
    idx = V.VlgIndex.build(text)
    parts = workload.gen_query_parts(text, cfg["nq"], cfg["k"], cfg["m"], cfg["qseed"])
    g = ".{%d,%d}?" % cfg["gap"]
    queries = [g.join(sp.decode("latin-1") for sp in subs) for subs in parts]
    assert queries == workload.gen_queries(text, cfg["nq"], cfg["k"], cfg["m"], cfg["gap"], cfg["qseed"])
    q = Queries(queries)
    ws = Workspace(100 << 30)
    a = idx.search(q, workspace=ws)
    sa = a.summary
    workload.check_expected("C3", sa)                            # what bench.py's metric run returns (its checksum_rank0)
    assert sa["n_queries"] == cfg["nq"] and sa["logical_occurrences"] > 100 * sa["located_occurrences"] > 0
    ws.set_option("sweep", 0)                                    # random-access locate kernel instead of the sorted sweep
    b = idx.search(q, workspace=ws)
    ws.set_option("sweep", 1)
    ws.set_option("trail", 0)                                    # sorted sweep, every occurrence walks its own LF steps,
    ws.set_option("filter", 0)                                   # join over the full lists
    c = idx.search(q, workspace=ws)
    for k in ("n_matches", "checksum", "n_tuple_values", "located_occurrences"):
        assert sa[k] == b.summary[k] == c.summary[k], k
    for k in ("lf_steps", "wt_levels_locate"):
        assert c.summary[k] == b.summary[k] and sa[k] * 4 < b.summary[k], k     # shared trails: a fraction of the LF steps
    counts = a.counts
    assert (counts == b.counts).all() and (counts == c.counts).all()
    del b, c
    ws.set_option("trail", 1)
    ws.set_option("filter", 1)
    ws.set_option("sweep_tail", 1 << 26)                         # a tenth of the occurrences finish as stragglers, on shared trails
    d = idx.search(q, workspace=ws)
    for k in ("n_matches", "checksum", "n_tuple_values", "located_occurrences"):
        assert sa[k] == d.summary[k], k
    assert (counts == d.counts).all()
    del d
    ws.set_option("sweep_tail", 1 << 22)
    kst = ws.kernel_stats()
    assert kst["filter_compact"]["launches"] > 0
    # ---- structural properties of EVERY match of the batch (as C2 has them) ----------------------------------------------
    _, offsets, first, tuples = a.fetch()
    m = cfg["m"]
    t = tuples.reshape(-1, cfg["k"])
    assert len(t) == sa["n_matches"] and (t[:, 0] == first).all()
    for i in range(1, cfg["k"]):
        d = t[:, i] - t[:, i - 1]                                            # uint64: a negative distance wraps and fails the bound
        assert int(d.min()) >= m + cfg["gap"][0] and int(d.max()) <= m + cfg["gap"][1]
    ok = t[1:, 0] >= t[:-1, -1] + np.uint64(m)                               # non-overlapping, left to right, inside every query
    bnd = offsets[1:-1].astype(np.int64)
    ok[bnd[(bnd >= 1) & (bnd <= len(t) - 1)] - 1] = True                      # pairs that straddle two queries
    assert ok.all()
    del ok, d
    assert int(first.sum(dtype=np.uint64)) == sa["checksum"]
    rng = np.random.default_rng(5)
    for mi in rng.integers(0, len(t), 5000):                                # reported positions are real occurrences
        qi = int(np.searchsorted(offsets, np.uint64(mi), side="right")) - 1
        for i in range(cfg["k"]):
            p0 = int(t[mi, i])
            assert text[p0: p0 + m].tobytes() == parts[qi][i]
    # ---- oracle: light queries through the FM-index restatement -----------------------------------------------------------
    o = oracle.Index.from_parts(idx.export_parts())
    done = 0
    for qi in rng.permutation(cfg["nq"]):
        subs, _, _, _ = oracle.query_fields(oracle.parse(queries[qi]))
        occ = [o.backward_search(sp)[0] for sp in subs]
        if min(occ) == 0 or sum(occ) > 20000:
            continue
        want = o.search(queries[qi])
        assert a.tuples(int(qi)).tolist() == want.tolist()
        done += 1
        if done >= 40:
            break
    assert done >= 20
    del o
    # ---- oracle: HEAVY queries (the ones that make up the step) through the SASEARCH restatement (index_sasearch.hpp: text + plain
    #      suffix array, forward_search, sort, the same merge join), on the suffix array of the device sorter --------------------
    occ_sub, _ = idx.occurrences(q)
    occ_q = occ_sub.reshape(-1, cfg["k"]).astype(np.int64)
    assert (occ_q.sum(axis=1)[occ_q.min(axis=1) > 0]).sum() == sa["logical_occurrences"]
    d_text = torch.from_numpy(text).cuda()
    d_sa = torch.empty(len(text) + 1, dtype=torch.int32, device="cuda")
    V.capi.check(V.lib().vlg_suffix_array_device(d_text.data_ptr(), len(text), d_sa.data_ptr(), None))
    torch.cuda.synchronize()
    sarr = d_sa.cpu().numpy().view(np.uint32)
    del d_text, d_sa
    torch.cuda.empty_cache()
    _check_device_suffix_array(text, sarr, rng)
    sas = oracle.SaSearch(np.concatenate([text, np.zeros(1, dtype=np.uint8)]), sarr)
    heavy = [int(qi) for qi in rng.permutation(cfg["nq"]) if occ_q[qi].min() > 0 and 100000 < occ_q[qi].sum() <= 4000000][:220]
    assert len(heavy) >= 200
    n_heavy_matches = 0
    for qi in heavy:
        for i in range(cfg["k"]):
            assert sas.count(parts[qi][i]) == occ_q[qi, i]                   # forward_search on the SA == backward_search on the FM-index
        want = sas.search(queries[qi])
        assert int(counts[qi]) == len(want)
        assert a.tuples(qi).tolist() == want.tolist(), queries[qi]
        n_heavy_matches += len(want)
    assert n_heavy_matches > 10000


def test_c5_one_gib_dna_rrr_full_size(oracle):
    """BASELINE config 5 at full size: 2^30 DNA-like characters, csa_wt<wt_huff<rrr_vector<63>>>-equivalent index,
    100 000 queries x k=2, m=12, gap <= 100."""
    import vlg_matching_amd as V
    from vlg_matching_amd import workload
    from vlg_matching_amd.index import Queries, Workspace
    cfg = workload.config("C5")
    lap = _Lap("C5")
    text = workload.gen_text(cfg["kind"], cfg["n"], cfg["seed"])
    lap("text")
    plain = V.VlgIndex.build(text)
    rrr = plain.compress()
    lap("build + compress")
    ip, ir = plain.info(), rrr.info()
    assert ir["bv_kind"] == 1 and ip["bv_kind"] == 0 and ir["n"] == ip["n"] == cfg["n"] + 1
    parts = workload.gen_query_parts(text, cfg["nq"], cfg["k"], cfg["m"], cfg["qseed"])
    # the batch of SURVEY.md 8(d) (every 12-mer occurs ~64 times) plus 300 queries on short, frequent sub-patterns so that long
    # lists, the sorted sweep and the window filter run on the rrr index too
    heavy_parts = workload.gen_query_parts(text, 300, 2, 6, cfg["qseed"] + 1)
    g = ".{%d,%d}?" % cfg["gap"]
    queries = [g.join(sp.decode() for sp in subs) for subs in parts] + [g.join(sp.decode() for sp in subs) for subs in heavy_parts]
    parts = parts + heavy_parts
    q = Queries(queries)
    lap("queries")
    ws = Workspace(100 << 30)
    workload.check_expected("C5", rrr.search(queries[:cfg["nq"]], workspace=ws).summary)      # the config's own batch (bench.py: other_configs)
    base = rrr.search(q, workspace=ws)
    s = base.summary
    assert s["n_queries"] == len(queries) and s["n_matches"] > 1000 and s["located_occurrences"] > 10 ** 7
    counts, offsets, first, tuples = base.fetch()
    lap("base search + fetch")
    # ---- structural properties of every match --------------------------------------------------------------------------------
    t = tuples.reshape(-1, 2)
    lens = np.array([len(parts[qi][0]) for qi in range(len(queries))], dtype=np.uint64)
    len_of_match = np.repeat(lens, counts.astype(np.int64))
    d = t[:, 1] - t[:, 0] - len_of_match
    assert int(d.max()) <= cfg["gap"][1]                                     # (uint64: a distance below the minimum wraps to a huge value)
    ok = t[1:, 0] >= t[:-1, 1] + len_of_match[:-1]
    bnd = offsets[1:-1].astype(np.int64)
    ok[bnd[(bnd >= 1) & (bnd <= len(t) - 1)] - 1] = True                      # pairs that straddle two queries
    assert ok.all()
    assert int(first.sum(dtype=np.uint64)) == s["checksum"]
    rng = np.random.default_rng(15)
    qi_of = np.repeat(np.arange(len(queries)), counts.astype(np.int64))
    for mi in rng.integers(0, len(t), 4000):
        subs = parts[qi_of[mi]]
        for i in range(2):
            p0 = int(t[mi, i])
            assert text[p0: p0 + len(subs[i])].tobytes() == subs[i]
    lap("properties")
    # ---- identical results: plain index, and the rrr index under every execution strategy -------------------------------------
    pl = plain.search(q, workspace=ws)
    assert pl.summary["n_matches"] == s["n_matches"] and pl.summary["checksum"] == s["checksum"]
    for x, y in zip(pl.fetch(), base.fetch()):
        assert (x == y).all()
    for k in ("lf_steps", "wt_levels_locate", "located_occurrences", "wt_levels_bsearch"):
        assert pl.summary[k] == s[k], k                                      # same walks, only the rank primitive differs
    del pl
    for opts in ({"sweep": 0}, {"trail": 0, "filter": 0}, {"sweep_tail": 1 << 20}):
        w2 = Workspace(100 << 30)
        for k_, v_ in opts.items():
            w2.set_option(k_, v_)
        r = rrr.search(q, workspace=w2)
        assert r.summary["n_matches"] == s["n_matches"] and r.summary["checksum"] == s["checksum"], opts
        f2 = r.fetch()
        assert (f2[0] == counts).all() and (f2[3] == tuples).all(), opts
        lap("mode %r" % (opts,))
    # ---- >= 200 sampled queries (and some heavy ones) equal the CPU oracle tuple for tuple -----------------------------------------
    o = oracle.Index.from_parts(plain.export_parts())
    lap("oracle from parts")
    # (a heavy query -- two 6-mers of 2.6 * 10^5 occurrences each -- costs the one-core oracle ~17 s: two of them; the heavy lists are
    # checked in full by the plain-vs-rrr and mode comparisons above, and at C2 size against the oracle)
    sample = list(rng.choice(cfg["nq"], 260, replace=False)) + list(cfg["nq"] + rng.choice(300, 2, replace=False))
    nonempty = 0
    for qi in sample:
        want = o.search(queries[int(qi)])
        assert base.tuples(int(qi)).tolist() == want.tolist(), queries[int(qi)]
        nonempty += len(want) > 0
    assert nonempty >= 2                                                     # (the two heavy ones at least)
    lap("oracle sample")


def test_c4_four_gib_text_64bit_positions(oracle):
    """BASELINE config 4 at full text size on one GPU: 2^32 protein characters (n = 2^32 + 1, so positions are 64-bit
    inside the kernels), a 100 000-query slice of the 1M batch."""
    import vlg_matching_amd as V
    from vlg_matching_amd import workload
    from vlg_matching_amd.index import Workspace
    import torch
    cfg = workload.config("C4")
    text = workload.gen_text(cfg["kind"], cfg["n"], cfg["seed"])
    assert len(text) == 1 << 32
    d_text = torch.from_numpy(text).cuda()
    idx = V.VlgIndex.build_device(d_text.data_ptr(), len(text))
    del d_text
    torch.cuda.empty_cache()
    info = idx.info()
    assert info["n"] == (1 << 32) + 1 and info["pos_bytes"] == 8
    nq = 100000
    parts = workload.gen_query_parts(text, nq, cfg["k"], cfg["m"], cfg["qseed"])
    g = ".{%d,%d}?" % cfg["gap"]
    queries = [g.join(s.decode() for s in subs) for subs in parts]
    res = idx.search(queries, workspace=Workspace(100 << 30))
    s = res.summary
    assert s["n_queries"] == nq and s["located_occurrences"] > 10 ** 8 and s["n_matches"] > 1000
    counts, offsets, first, tuples = res.fetch()
    t = tuples.reshape(-1, 2).astype(np.int64)
    d = t[:, 1] - t[:, 0] - cfg["m"]
    assert (d >= cfg["gap"][0]).all() and (d <= cfg["gap"][1]).all()
    assert int(first.astype(np.uint64).sum()) % (1 << 64) == s["checksum"]
    assert int(t.max()) > (1 << 32) - (1 << 24)               # positions span the whole text: some match lies in its last 16 MiB
    qi_of = np.repeat(np.arange(nq), counts.astype(np.int64))
    rng = np.random.default_rng(4)
    for m in rng.choice(len(t), 3000, replace=False):           # reported positions are real occurrences
        subs = parts[qi_of[m]]
        for i in range(2):
            p = int(t[m, i])
            assert text[p: p + cfg["m"]].tobytes() == subs[i]
    # beyond-2^32 arithmetic: some occurrence must lie in the top half of the text
    assert (t[:, 0] >= (1 << 31)).any()
    # the config's whole batch (10^6 queries): the constants bench.py's other_configs asserts
    del res
    full = workload.gen_queries(text, cfg["nq"], cfg["k"], cfg["m"], cfg["gap"], cfg["qseed"])
    assert full[:nq] == queries
    workload.check_expected("C4", idx.search(full, workspace=Workspace(140 << 30)).summary)
    del full
    res = idx.search(queries, workspace=Workspace(100 << 30))
    # bounded oracle sample on the device-built parts (CPU algorithm, reference layout)
    o = oracle.Index.from_parts(idx.export_parts())
    done = 0
    for qi in rng.permutation(nq)[:200]:
        want = o.search(queries[qi])
        assert res.tuples(int(qi)).tolist() == want.tolist()
        done += 1
    assert done == 200
