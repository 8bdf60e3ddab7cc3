#!/usr/bin/env python3
"""The lazy WT-over-SA search (vlg_wtsa_*) beside the FM-index path on the same workload: build time, the
forward searches, and the search with a cap on the matches per query (what a caller that stops iterating early pays).
    python tools/wtsa_bench.py [--config C2] [--caps 1,10,100,0]   -> one JSON line"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--caps", default="1,10,100")
    ap.add_argument("--fm", action="store_true", help="also time the FM-index path (all matches) on the same batch")
    args = ap.parse_args()
    import vlg_matching_amd as V
    from vlg_matching_amd import workload
    from vlg_matching_amd.index import Queries, Workspace
    torch.zeros(1, device="cuda")
    cfg = workload.config(args.config, args.scale)
    text = workload.gen_text(cfg["kind"], cfg["n"], cfg["seed"])
    queries = workload.gen_queries(text, cfg["nq"], cfg["k"], cfg["m"], cfg["gap"], cfg["qseed"])
    t0 = time.perf_counter()
    w = V.WtsaIndex(text)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    q = Queries(queries)
    ws = Workspace(64 << 30)
    ws.set_option("tuples", 0)
    t0 = time.perf_counter()
    sp, ep = w.ranges(q)
    t_ranges = time.perf_counter() - t0
    out = {"config": args.config, "scale": args.scale, "n": cfg["n"], "queries": cfg["nq"], "k": cfg["k"], "wtsa_info": w.info(),
           "build_s": t_build, "forward_search_ms": t_ranges * 1e3, "logical_occurrences": int((ep + 1 - sp).sum()), "caps": {}}
    for cap in [int(c) for c in args.caps.split(",")]:
        w.search(q, max_matches=cap, workspace=ws)
        t0 = time.perf_counter()
        r = w.search(q, max_matches=cap, workspace=ws)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["caps"][str(cap)] = {"ms": dt * 1e3, "matches": r.summary["n_matches"], "checksum": r.summary["checksum"],
                                 "queries_per_sec": cfg["nq"] / dt}
    if args.fm:
        idx = V.VlgIndex.build(text)
        idx.search(q, workspace=ws)
        t0 = time.perf_counter()
        r = idx.search(q, workspace=ws)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["fm_index_all_matches"] = {"ms": dt * 1e3, "matches": r.summary["n_matches"], "checksum": r.summary["checksum"]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
