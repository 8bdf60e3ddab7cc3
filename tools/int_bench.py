#!/usr/bin/env python3
"""f-4 on the clock: (1) a word-level integer text through the integer-alphabet FM-index (vlg_index_build_int: wavelet
matrix, one lane per occurrence in locate), (2) BASELINE config 3's batch on a text_order_sa_sampling index (vlg_index_resample).
Prints one JSON line per part; development / profiling tool, not the metric.

    python tools/int_bench.py [n_tokens_log2=27] [steps=3]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vlg_matching_amd as V
from vlg_matching_amd import workload
from vlg_matching_amd.index import Queries, Workspace

HBM_PEAK_GBS = 8000.0


def timed(idx, q, ws, steps):
    idx.search(q, workspace=ws)
    ws.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = idx.search(q, workspace=ws)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks = ws.kernel_stats()
    ws.profile(False)
    loc = ks["locate"]
    ach = loc["algorithmic_bytes"] / (loc["total_ms"] * 1e-3) / 1e9 if loc["total_ms"] > 0 else 0.0
    s = r.summary
    return {"ms_per_step": dt / steps * 1e3, "queries_per_sec": s["n_queries"] * steps / dt, "located_occ_per_step": s["located_occurrences"],
            "matches_per_step": s["n_matches"], "checksum": s["checksum"], "lf_steps_per_occ": s["lf_steps"] / max(s["located_occurrences"], 1),
            "levels_per_lf": s["wt_levels_locate"] / max(s["lf_steps"], 1), "locate_mode": s["locate_mode"],
            "kernels_ms_per_step": {k: v["total_ms"] / steps for k, v in ks.items() if v["total_ms"] > 0},
            "rank_kernel_roofline": {"bound": "hbm", "kernel_class": "locate", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                     "ms_per_step": loc["total_ms"] / steps, "launches": loc["launches"],
                                     "algorithmic_bytes_are": "32 B per wavelet level of every LF step + one SA sample per occurrence (SURVEY 8d K3)"}}


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    # ---- (1) word-level text: Zipf(1.0)-distributed word ids over a 50 000-word vocabulary (what tokenising C3's text would give) ----
    n = 1 << lg
    rng = np.random.default_rng(3)
    ranks = np.arange(1, 50001, dtype=np.float64)
    p = (1.0 / ranks) / (1.0 / ranks).sum()
    text = (rng.choice(50000, n, p=p) + 1).astype(np.uint32)
    t0 = time.perf_counter()
    idx = V.VlgIndex.build_int(text)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    nq, k, m, gap = 100000, 3, 2, (0, 100)
    pos = rng.integers(0, n - m, nq * k)
    qs = []
    for i in range(nq):
        subs = [" ".join(str(int(x)) for x in text[pos[i * k + j]: pos[i * k + j] + m]) for j in range(k)]
        qs.append((" .{%d,%d}? " % gap).join(subs))
    q = Queries.from_int(qs)
    ws = Workspace(120 << 30)
    ws.set_option("tuples", 0)
    out = timed(idx, q, ws, steps)
    out.update({"what": "integer-alphabet FM-index (csa_wt<wt_int<>, 32, ., ., ., int_alphabet<>>): word-level text of 2^%d tokens, Zipf(1.0) over 50 000 "
                        "words (seed 3), %d queries x k=%d, m=%d tokens, gap .{%d,%d}? tokens" % (lg, nq, k, m, gap[0], gap[1]),
                "index": idx.info(), "index_build_s": t_build})
    print(json.dumps(out), flush=True)
    del idx, q, ws, text
    torch.cuda.empty_cache()
    # ---- (2) C3 on text-order sampling ---------------------------------------------------------------------------------------------------
    cfg = workload.config("C3")
    text = workload.gen_text(cfg["kind"], cfg["n"], cfg["seed"])
    d_text = torch.from_numpy(text).cuda()
    base = V.VlgIndex.build_device(d_text.data_ptr(), len(text))
    del d_text
    t0 = time.perf_counter()
    idx = base.resample(text_order=True, dens=32)
    torch.cuda.synchronize()
    t_rs = time.perf_counter() - t0
    queries = workload.gen_queries(text, cfg["nq"], cfg["k"], cfg["m"], cfg["gap"], cfg["qseed"])
    q = Queries(queries)
    ws = Workspace(160 << 30)
    ws.set_option("tuples", 0)
    a = timed(base, q, ws, steps)
    b = timed(idx, q, ws, steps)
    workload.check_expected("C3", {"n_matches": b["matches_per_step"], "checksum": b["checksum"], "located_occurrences": b["located_occ_per_step"]})
    b.update({"what": "BASELINE config 3's batch on csa_wt<wt_huff<>, 32, ., text_order_sa_sampling> (vlg_index_resample): same matches and checksum "
                      "as the SA-order index", "resample_s": t_rs, "index": idx.info(), "sa_order_ms_per_step_same_run": a["ms_per_step"]})
    print(json.dumps(b), flush=True)


if __name__ == "__main__":
    main()
