#!/usr/bin/env python3
"""K1 micro-benchmark: batched bit-rank (rank_support_v::rank equivalent) on 256-bit super-blocks.

Random positions over a bit-vector far larger than the 256 MiB Infinity Cache; algorithmic bytes = 32 B per rank
(SURVEY.md 8d).  Prints one JSON line; run under rocprofv3 to get the matching kernel-trace / PMC numbers.
"""
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
    ap.add_argument("--mbits", type=int, default=3500, help="bit-vector size in 10^6 bits (< 4295)")
    ap.add_argument("--queries", type=int, default=1 << 28)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--sorted", action="store_true", help="ascending positions (coalesced regime)")
    args = ap.parse_args()
    import vlg_matching_amd as V
    nbits = args.mbits * 1000000
    rng = np.random.default_rng(1)
    words = rng.integers(0, 1 << 63, (nbits + 63) // 64, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (nbits + 63) // 64, dtype=np.uint64)
    t0 = time.perf_counter()
    bv = V.BitVector(words, nbits)
    t_create = time.perf_counter() - t0
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    idx = torch.randint(0, nbits + 1, (args.queries,), generator=g, device="cuda", dtype=torch.int64)
    if args.sorted:
        idx = torch.sort(idx).values
    out = torch.zeros_like(idx)
    bv.rank_device(idx.data_ptr(), out.data_ptr(), args.queries)
    torch.cuda.synchronize()
    # spot-check against numpy
    sample = idx[:2000].cpu().numpy()
    bits = np.unpackbits(words[: (int(sample.max()) + 63) // 64 + 1].view(np.uint8), bitorder="little") if False else None
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = []
    for _ in range(args.iters):
        ev0.record()
        bv.rank_device(idx.data_ptr(), out.data_ptr(), args.queries, stream=None)
        ev1.record()
        torch.cuda.synchronize()
        times.append(ev0.elapsed_time(ev1))
    ms = float(np.median(times))
    alg = 32.0 * args.queries
    print(json.dumps({"kernel": "bitrank_kernel", "nbits": nbits, "blocks_bytes": bv.hbm_bytes(), "queries": args.queries,
                      "sorted": args.sorted, "ms": ms, "ranks_per_s": args.queries / (ms * 1e-3),
                      "algorithmic_GBps": alg / (ms * 1e-3) / 1e9, "frac_of_8TBps": alg / (ms * 1e-3) / 8e12,
                      "io_GBps_idx_and_out": 16.0 * args.queries / (ms * 1e-3) / 1e9, "create_s": t_create}))


if __name__ == "__main__":
    main()
