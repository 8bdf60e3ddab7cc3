#!/usr/bin/env python3
"""Differential fuzzing of the execution strategies (developer tool, needs an MI355X): random texts and query batches, every
batch searched once with all batch-level shortcuts off and once with a random combination of them on; results must be equal.
usage: fuzz_strategies.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlg_matching_amd as V  # noqa: E402
from vlg_matching_amd.index import Workspace  # noqa: E402


def make_text(rng):
    kind = int(rng.integers(0, 5))
    n = int(rng.integers(50, 120000)) if rng.random() < 0.85 else int(rng.integers(500000, 4000000))      # mostly small, sometimes MBs
    if kind == 0:
        return rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes()
    if kind == 1:
        return rng.choice(np.frombuffer(b"ab", np.uint8), n, p=[0.9, 0.1]).tobytes()
    if kind == 2:                                           # periodic with noise
        unit = rng.choice(np.frombuffer(b"xyzw ", np.uint8), int(rng.integers(2, 40))).tobytes()
        t = bytearray((unit * (n // len(unit) + 1))[:n])
        for p in rng.integers(0, n, n // 50):
            t[p] = 0x71
        return bytes(t)
    if kind == 3:                                           # words
        words = [bytes(rng.choice(np.arange(97, 123, dtype=np.uint8), int(rng.integers(1, 7)))) for _ in range(int(rng.integers(2, 60)))]
        out = bytearray()
        while len(out) < n:
            out += words[int(rng.zipf(1.5)) % len(words)] + b" "
        return bytes(out[:n])
    return bytes(rng.integers(1, 255, n, dtype=np.uint8))


def make_queries(rng, text, nq):
    qs = []
    for _ in range(nq):
        k = int(rng.integers(1, 7))
        subs = []
        for _ in range(k):
            m = int(rng.integers(1, 5))
            s = int(rng.integers(0, max(1, len(text) - m)))
            subs.append(text[s:s + m])
        q = subs[0].decode("latin-1")
        for sp in subs[1:]:
            a = int(rng.integers(0, 40))
            b = a + int(rng.choice([0, 3, 50, 700, 100000]))
            q += ".{%d,%d}?%s" % (a, b, sp.decode("latin-1"))
        if all(c not in q.replace(".{", "").replace("}?", "") for c in "{}"):      # sub-patterns must not look like gap syntax
            qs.append(q)
    return qs


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    rounds = 0
    t_say = time.time() + 30
    while time.time() < t_end:
        if time.time() > t_say:
            print("... %d batches" % rounds, flush=True)
            t_say = time.time() + 30
        text = make_text(rng)
        try:
            idx = V.VlgIndex.build(text)
        except V.capi.VlgError:
            continue
        qs = make_queries(rng, text, int(rng.integers(1, 400)))
        if not qs:
            continue
        base = Workspace()
        for k_, v_ in (("dedup", 1), ("sweep", 0), ("filter", 0), ("global_sort_min", 1 << 40)):
            base.set_option(k_, v_)
        a = idx.search(qs, workspace=base, strict=False)
        opts = {"sweep_min": int(rng.choice([1, 1 << 22])), "sweep_tail": int(rng.choice([1, 7, 1000, 1 << 40])), "trail": int(rng.integers(0, 2)),
                "global_sort_min": int(rng.choice([1, 1 << 40])), "list_sort": int(rng.integers(0, 2)), "filter_min": int(rng.choice([0, 1 << 12])), "filter_stream_min": int(rng.choice([0, 1 << 16])),
                "filter_pivot": int(rng.integers(0, 2)), "filter_pivot_ratio": int(rng.choice([1, 4, 12, 64])),
                "filter_group_bytes": int(rng.choice([0, 1 << 14, 1 << 18])), "dedup": int(rng.choice([1, 1, 0])), "pivot_rungs": int(rng.integers(0, 3)), "compact_dense_min": int(rng.choice([0, 16, 256, 4096]))}
        ws = Workspace(int(rng.choice([0, 64 << 20])))
        for k_, v_ in opts.items():
            ws.set_option(k_, v_)
        os.environ["VLG_FORCE_POS64"] = "0"
        other = idx.compress() if rng.random() < 0.3 else idx        # the rrr-63 variant of the index must answer the same
        mode = str(rng.choice(["0", "0", "0", "1", "2"]))              # 64-bit positions / wide SA indices with 32-bit positions on small texts
        variant = "plain"
        if mode != "0":
            os.environ["VLG_FORCE_POS64"] = mode
            d = int(rng.choice([32, 32, 1]))                         # (1: sa_dense_copy_kernel on 64-bit samples)
            other = V.VlgIndex.build(text, dens=d)
            variant = "pos64=%s/%d" % (mode, d)
        elif rng.random() < 0.3:                                     # text_order_sa_sampling on top (plain or rrr)
            other = other.resample(text_order=True, dens=int(rng.choice([1, 4, 32, 64])))
            variant = "text-order"
        elif rng.random() < 0.3:                                     # SA-order samples of another density; 1 = the whole suffix array resident
            d = int(rng.choice([1, 1, 2, 5, 64]))
            other = other.resample(text_order=False, dens=d)
            variant = "sa-order/%d" % d
        os.environ["VLG_NO_SPECULATIVE_COMPACT"] = str(rng.choice(["0", "0", "1"]))
        opts["variant"] = variant
        try:
            b = other.search(qs, workspace=ws, strict=False)
        except V.capi.VlgError as e:
            if "workspace" in str(e):
                continue
            raise
        finally:
            os.environ["VLG_FORCE_POS64"] = "0"
        for k_ in ("n_matches", "checksum", "n_tuple_values", "logical_occurrences"):
            assert a.summary[k_] == b.summary[k_], (seed, rounds, k_, opts, other is not idx)
        for x, y in zip(a.fetch(), b.fetch()):
            assert (x == y).all(), (seed, rounds, opts, other is not idx)
        rounds += 1
    os.environ["VLG_NO_SPECULATIVE_COMPACT"] = "0"
    os.environ["VLG_FORCE_POS64"] = "0"
    print("fuzz ok: %d batches, seed %d" % (rounds, seed))


if __name__ == "__main__":
    main()
