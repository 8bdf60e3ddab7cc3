#!/usr/bin/env python3
"""bench.py -- VLG queries/s + located occurrences/s of the MI355X hot path (BASELINE.json metric).

A "step" is one pass of the whole hot path (backward search -> locate -> sort -> gap join) over one batch
of synthetic queries, index and query batch already resident in HBM.  At N=1 the workload is SURVEY.md 8(d)
config C3 (1 GiB english-like text, 100k 3-sub-pattern queries, gap <= 1000).  With N GPUs the read-only
index is built on rank 0 and broadcast over RCCL, every rank runs its own batch (weak scaling, no data-path
collective); the timed region is bracketed by barrier + synchronize and the max over ranks is reported.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--scale 1.0]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s

# HBM bytes per launch from rocprofv3 PMC passes of this same command (profiles/*_pmc_*.csv; FETCH_SIZE + WRITE_SIZE,
# KiB -> bytes); filled in by tools/pmc_summary.py.  None = not measured for that kernel.
PMC_TRAFFIC, PMC_DETAIL = {}, {}
try:
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as _f:
        for _k, _v in json.load(_f).items():
            PMC_TRAFFIC[_k] = _v["bytes_per_launch"]
            # FETCH_SIZE is exact for the 64-byte requests of block reads (tools/k1_bench.py) and reads half of a wide
            # coalesced streaming read on gfx950 (MI355X_MICROARCH.md, HBM): both readings are kept next to the raw sum
            PMC_DETAIL[_k] = {"read_raw": _v["read_bytes_per_launch_raw"], "read_if_all_wide_streaming": _v["read_bytes_per_launch_x2_if_streaming"],
                              "write": _v["write_bytes_per_launch"], "source": "profiles/pmc_traffic.json (" + _v.get("tag", "") + ")"}
except Exception:
    pass


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baseline(idx, queries, occ_per_query_mean, budget_occ=450000, budget_s=25.0, threads=16):
    """The CPU oracle (restatement of the reference path, reference data layout) timed on this host, one thread,
    on a bounded seeded sample of the same query batch."""
    from oracle import oracle as O
    parts = idx.export_parts()
    o = O.Index.from_parts(parts)
    rng = np.random.default_rng(12345)
    order = rng.permutation(len(queries))
    stats = np.zeros(4, dtype=np.uint64)
    done, dt, remaining = 0, 0.0, budget_occ
    for qi in order:
        subs, _, _, _ = O.query_fields(O.parse(queries[qi]))
        occs = [o.backward_search(sp)[0] for sp in subs]
        need = 0 if min(occs) == 0 else sum(occs)
        if need > remaining:                      # keeps the sample bounded: heavy queries cost minutes on one core
            continue
        t0 = time.perf_counter()
        o.search(queries[qi], stats=stats)
        dt += time.perf_counter() - t0
        done += 1
        remaining -= need
        if dt > budget_s or remaining < 1000 or done >= 2000:
            break
    occ_rate = float(stats[0]) / dt if dt > 0 else 0.0
    # a sample's queries/s depends on which heavy queries it happened to draw; the stable figure is located
    # occurrences/s, converted with the exact mean occurrences per query of the full batch
    qps = occ_rate / occ_per_query_mean if occ_per_query_mean > 0 else 0.0
    out = {"value": qps, "unit": "queries/s", "cores": 1, "kind": "port",
           "located_occ_per_sec": occ_rate, "sample_queries": done, "sample_seconds": dt,
           "sample_located_occ": int(stats[0]), "sample_lf_steps": int(stats[1]),
           "sample": "%d queries of the same batch, drawn in random order (seed 12345) while their occurrence lists fit a "
                     "%d-occurrence budget; %.1f s on one core; queries/s = sample occurrences/s / mean occurrences "
                     "per query of the full batch (%.0f)" % (done, budget_occ, dt, occ_per_query_mean)}
    # disclosure: the same restatement on T threads over disjoint query shards
    # (the C library is re-entrant and ctypes releases the GIL); the reference itself is single-threaded.
    T = threads
    if T > 1:
        import concurrent.futures as cf
        light = []
        for qi in order:
            subs, _, _, _ = O.query_fields(O.parse(queries[qi]))
            occs = [o.backward_search(sp)[0] for sp in subs]
            if min(occs) and sum(occs) <= budget_occ // 8:
                light.append(queries[qi])
            if len(light) >= 64 * T:
                break
        shards = [light[i::T] for i in range(T)]

        def run(shard):
            st = np.zeros(4, dtype=np.uint64)
            t0 = time.perf_counter()
            for qq in shard:
                o.search(qq, stats=st)
                if time.perf_counter() - t0 > 8.0:
                    break
            return int(st[0])
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(T) as ex:
            occ_t = sum(ex.map(run, shards))
        dt_t = time.perf_counter() - t0
        out["threads_T"] = {"threads": T, "located_occ_per_sec": occ_t / dt_t if dt_t > 0 else 0.0,
                            "queries_per_sec": (occ_t / dt_t) / occ_per_query_mean if dt_t > 0 and occ_per_query_mean > 0 else 0.0,
                            "sample_seconds": dt_t, "sample_located_occ": occ_t}
    return out


def cpu_sasearch(text, queries, occ_per_query_mean, budget_occ=60000000, budget_s=12.0):
    """Disclosure (CPU baseline (3)): the benchmark's plain-suffix-array index -- text + SA, forward_search, sort,
    join (index_sasearch.hpp) -- restated in the oracle, one thread, same random query order.  The suffix array comes from the
    device sorter (vlg_suffix_array_device); building it is not part of the figure, as `load` is not in the reference's."""
    from oracle import oracle as O
    import vlg_matching_amd as V
    n_text = len(text)
    if n_text + 1 >= (1 << 32) - 1:
        return None
    d_text = torch.from_numpy(text).cuda()
    d_sa = torch.empty(n_text + 1, dtype=torch.int32, device="cuda")
    t0 = time.perf_counter()
    V.capi.check(V.lib().vlg_suffix_array_device(d_text.data_ptr(), n_text, d_sa.data_ptr(), None))
    torch.cuda.synchronize()
    t_sa = time.perf_counter() - t0
    sa = d_sa.cpu().numpy().view(np.uint32)
    del d_text, d_sa
    torch.cuda.empty_cache()
    tz = np.concatenate([text, np.zeros(1, dtype=np.uint8)])
    s = O.SaSearch(tz, sa)
    rng = np.random.default_rng(12345)
    stats = np.zeros(4, dtype=np.uint64)
    done, dt, remaining = 0, 0.0, budget_occ
    for qi in rng.permutation(len(queries)):
        subs, _, _, _ = O.query_fields(O.parse(queries[qi]))
        need = sum(s.count(sp) for sp in subs)
        if need > remaining:
            continue
        t0 = time.perf_counter()
        s.search(queries[qi], stats=stats)
        dt += time.perf_counter() - t0
        done += 1
        remaining -= need
        if dt > budget_s or remaining < 1000 or done >= 20000:
            break
    occ_rate = float(stats[0]) / dt if dt > 0 else 0.0
    return {"algorithm": "SASEARCH (plain suffix array + text, index_sasearch.hpp)", "cores": 1, "kind": "port",
            "queries_per_sec": occ_rate / occ_per_query_mean if occ_per_query_mean > 0 else 0.0, "located_occ_per_sec": occ_rate,
            "sample_queries": done, "sample_seconds": dt, "sample_located_occ": int(stats[0]), "suffix_array_on_device_s": t_sa,
            "note": "every sub-pattern's SA range is copied and sorted, query by query, as the reference does; "
                    "queries/s = sample occurrences/s / mean occurrences per query of the full batch"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink text and batch (development only)")
    ap.add_argument("--workspace-gb", type=float, default=160.0)
    ap.add_argument("--bv", choices=["plain", "rrr"], default=None, help="wavelet-tree bit-vectors (default: rrr for C5, else plain)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tuples", action="store_true", help="also materialise every sub-pattern position of every match (sdsl::locate output)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)

    import vlg_matching_amd as V
    from vlg_matching_amd import workload
    from vlg_matching_amd.index import Queries, Workspace
    V.capi.check(V.lib().vlg_set_device(local_rank))

    cfg = workload.config(args.config, args.scale)
    t_gen = t_build = 0.0
    # ---- index: built on rank 0, replicated over RCCL (one broadcast of the contiguous HBM image) ----------
    text = None
    if rank == 0:
        t0 = time.perf_counter()
        text = workload.gen_text(cfg["kind"], cfg["n"], cfg["seed"])
        t_gen = time.perf_counter() - t0
        log("text %s n=%d generated in %.1f s" % (cfg["kind"], cfg["n"], t_gen))
        d_text = torch.from_numpy(text).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx = V.VlgIndex.build_device(d_text.data_ptr(), len(text))
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        del d_text
        torch.cuda.empty_cache()
        log("index built on device in %.2f s: %s" % (t_build, idx.info()))
        if (args.bv or ("rrr" if args.config == "C5" else "plain")) == "rrr":
            t0 = time.perf_counter()
            plain_idx, idx = idx, idx.compress()
            log("rrr-63 re-encoding on device in %.2f s: %s" % (time.perf_counter() - t0, idx.info()))
            if args.no_cpu_baseline:
                del plain_idx
    if world > 1:
        nb = torch.tensor([idx.blob_bytes() if rank == 0 else 0], dtype=torch.int64, device="cuda")
        dist.broadcast(nb, 0)
        blob = torch.empty(int(nb.item()), dtype=torch.uint8, device="cuda")
        if rank == 0:
            idx.blob_export(blob.data_ptr(), blob.numel())
        t0 = time.perf_counter()
        dist.broadcast(blob, 0)
        torch.cuda.synchronize()
        log("rank %d: index image %.1f MB broadcast in %.3f s" % (rank, blob.numel() / 1e6, time.perf_counter() - t0))
        if rank != 0:
            idx = V.VlgIndex.attach_blob(blob.data_ptr(), blob.numel(), keep=blob)
    info = idx.info()

    # ---- query batches: one per rank (weak scaling), generated where the text is ---------------------------
    if rank == 0:
        batches = [workload.gen_queries(text, cfg["nq"], cfg["k"], cfg["m"], cfg["gap"], cfg["qseed"] + 1000 * r)
                   for r in range(world)]
    else:
        batches = None
    if world > 1:
        mine = [None]
        dist.scatter_object_list(mine, batches if rank == 0 else None, src=0)
        queries = mine[0]
    else:
        queries = batches[0]
    host_text = text if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    del text
    q = Queries(queries)                      # parsed + uploaded: resident in HBM before the timed region
    ws = Workspace(int(args.workspace_gb * (1 << 30)))
    ws.set_option("reserve", int(args.workspace_gb * (1 << 30)))     # scratch allocated before the timed region whatever --warmup is
    # the benchmark's result type (gapped_search_result, index_sasearch.hpp:58-118) holds the first position of every match and
    # nothing else; --tuples also materialises the other sub-pattern positions (what sdsl::locate returns)
    ws.set_option("tuples", 1 if args.tuples else 0)
    for kv in filter(None, os.environ.get("VLG_BENCH_OPTIONS", "").split(",")):      # development: name=value workspace options
        ws.set_option(kv.split("=")[0], int(kv.split("=")[1]))

    def step():
        return idx.search(q, workspace=ws)

    for _ in range(args.warmup):
        step()
    ws.profile(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    s = res.summary
    tot = torch.tensor([s["n_queries"], s["located_occurrences"], s["n_matches"], s["logical_occurrences"]],
                       dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    n_queries, n_occ, n_matches, n_logical = int(tot[0]), int(tot[1]), int(tot[2]), int(tot[3])
    kstats = ws.kernel_stats()

    if rank == 0:
        def roof(name, kernel):
            st = kstats[name]
            launches = max(st["launches"], 1)
            ms = st["total_ms"] / launches
            alg = st["algorithmic_bytes"] / launches
            ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"bound": "hbm", "kernel": kernel, "kernel_class": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": PMC_TRAFFIC.get(kernel), "traffic_detail": PMC_DETAIL.get(kernel),
                    "algorithmic_bytes_per_launch": alg,
                    "avg_launch_ms": ms, "launches": st["launches"], "ms_per_step": st["total_ms"] / args.steps}
        names = {"locate": "sweep_step_kernel" if kstats["locate_partition"]["launches"] else "locate_kernel",
                 "join_link": "join_link_kernel", "join_init": "join_init_kernel", "join_scan": "rocprim scan (reverse min)", "join_chain": "join_jump+chain_tiles/walk/emit",
                 "gather": "join_gather_kernel", "sort": "rocprim radix_sort_keys on (list, position) keys", "locate_partition": "rocprim radix_sort_pairs",
                 "backward_search": "backward_search_kernel", "expand": "expand_kernel", "filter_pass": "filter_pass_kernel",
                 "filter_pivot": "filter_pivot_kernel", "filter_compact": "filter_count_runs + scan + filter_compact_kernel",
                 "locate_resolve": "trail_resolve_kernel"}
        dominant = max(kstats, key=lambda k: kstats[k]["total_ms"])
        out = {
            "metric": "vlg_queries_per_sec",
            "value": n_queries * args.steps / dt,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32" if info["pos_bytes"] == 4 else "u64",
            "data": "synthetic",
            "config": {"workload": "%s%s: %s text n=%d (seed %d), %d queries/GPU x k=%d, m=%d, gap .{%d,%d}?, t_dens=32"
                                   % (args.config, "" if args.scale == 1.0 else " x%g" % args.scale, cfg["kind"], cfg["n"],
                                      cfg["seed"], cfg["nq"], cfg["k"], cfg["m"], cfg["gap"][0], cfg["gap"][1]),
                       "bit_vectors": "rrr_vector<63>" if info["bv_kind"] else "plain (256-bit super-blocks)",
                       "sigma": info["sigma"], "mean_code_len_bits": info["wt_bits"] / info["n"],
                       "index_hbm_bytes": info["hbm_bytes"],
                       "result": ("counts + first position and tuple of every match" if args.tuples else
                                  "counts + first position of every match (the benchmark's gapped_search_result)")},
            "located_occ_per_sec": n_occ * args.steps / dt,
            "located_occ_per_step": n_occ,
            "logical_occ_per_sec": n_logical * args.steps / dt,
            "logical_occ_per_step": n_logical,
            "matches_per_step": n_matches,
            "join_slots_per_step_rank0": s["join_slots"],   # list elements the join evaluated (after the window filter)
            "checksum_rank0": s["checksum"],
            "lf_steps_per_occ": s["lf_steps"] / max(s["located_occurrences"], 1),
            "wt_levels_per_lf": s["wt_levels_locate"] / max(s["lf_steps"], 1),
            "chunks_per_step": s["n_chunks"],
            "index_build_s": t_build, "text_gen_s": t_gen,
            "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in kstats.items()},
            "kernels_ms_sum_per_step": sum(v["total_ms"] for v in kstats.values()) / args.steps,
            "roofline": roof(dominant, names[dominant]),                 # the dominant kernel class of the step
            "rank_kernel_roofline": roof("locate", names["locate"]),     # the LF / bit-rank kernel the north star names
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(plain_idx if info["bv_kind"] else idx, queries, n_logical / max(n_queries, 1))
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
            sas = cpu_sasearch(host_text, queries, n_logical / max(n_queries, 1))
            if sas is not None:
                out["cpu_baseline"]["sasearch"] = sas
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
