#!/usr/bin/env python3
"""bench.py -- VLG queries/s + located occurrences/s of the MI355X hot path (BASELINE.json metric).

A "step" is one pass of the whole hot path (backward search -> locate -> sort -> window filter -> gap join) over one batch
of synthetic queries, index and parsed query batch already resident in HBM.  At N=1 the workload is SURVEY.md 8(d) config C3
(1 GiB english-like text, 100k 3-sub-pattern queries, gap <= 1000).

N GPUs = N processes, one per GPU: launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, or by
this script itself when it is started plainly with --gpus N > 1 (the parent then only spawns and waits: it never touches a
GPU).  The read-only index is built on rank 0 and broadcast over RCCL (vlg_matching_amd.dist.replicate_index), then
  * weak scaling (the printed `value`): every rank runs its own 100k-query batch;
  * strong scaling (reported beside it as `strong_scaling`): THE batch of rank 0 is cut into contiguous slices of equal
    work (sum of SA-interval sizes from one backward-search pass, vlg_matching_amd.dist.shard_by_work) -- the query loop of
    gm_search.cpp:91-121 sharded.
No collective on the data path; the timed regions are bracketed by barrier + synchronize and the max over ranks is reported.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--scale 1.0]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
RANDOM_REQUEST_WALL = 5.0e10   # random 64-byte read requests per second, measured with tools/k1_bench.py (profiles/r01_k1_bitrank_kernel.json)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def host_cpus():
    """Cores this process may actually use: the smaller of the scheduler affinity and the cgroup's CPU quota (a GPU box hands a
    container a share of its cores -- os.cpu_count() still reports all of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:                                          # noqa: BLE001
        pass
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:                                      # noqa: BLE001
            continue
    return n, quota


def wait_ranks(procs, poll_s=0.2, grace_s=10.0):
    """Wait for the rank processes TOGETHER: a rank that dies (OOM, RCCL init failure) leaves the others blocked in a
    collective, so the first non-zero exit ends the rest (terminate, then kill) and is returned."""
    rc = 0
    alive = list(procs)
    while alive and not rc:
        time.sleep(poll_s)
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r and not rc:
                rc = r
                log("rank process %d exited with code %d: ending the other ranks" % (p.pid, r))
    for p in alive:
        p.terminate()
    deadline = time.time() + grace_s
    for p in alive:
        try:
            p.wait(timeout=max(0.1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return rc


def self_launch(args):
    """--gpus N > 1 without a launcher: start N child processes (one per device) and wait.  Decided before this process has
    made any GPU call (it never makes one): a process that has initialised the GPU must not be replaced or forked."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    sys.exit(wait_ranks(procs))


def load_pmc():
    """HBM bytes per launch from rocprofv3 PMC passes of this same command (profiles/pmc_traffic.json, written by
    tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE passes).  Empty when the file is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)
    except Exception:
        return {}


# kernel class (vlg_workspace_kernel_stats) -> (display name, kernels of profiles/pmc_traffic.json that make it up,
#                                               what its algorithmic bytes are)
CLASSES = {
    "backward_search": ("backward_search_kernel", ["backward_search_kernel"], "32 B per wavelet-tree level of every rank (SURVEY 8d K2)"),
    "expand": ("expand_kernel", [], ""),
    "locate": ("sweep_step_kernel (rounds 1..; round 0 is sweep_first_kernel, the stragglers are finished by locate_kernel<.., kTail>: one launch each)",
               ["sweep_step_kernel", "sweep_first_kernel", "locate_kernel", "member_build_kernel", "sweep_chunk_lists_kernel"],
               "32 B per tree level actually walked + one SA sample per occurrence (SURVEY 8d K3)"),
    "locate_partition": ("rocprim radix_sort_pairs (u16 symbol key, u64 element): one 5-bit pass per round",
                         ["rocprim:radix_sort_onesweep<unsigned_short,unsigned_long>"],
                         "per element of the round: key + value read once, written once (2 x 10 B)"),
    "locate_resolve": ("trail_resolve_grouped_kernel (first pass: a workgroup's records regrouped by the symbol in front; trail_resolve_kernel for later passes)",
                       ["trail_resolve_grouped_kernel", "trail_resolve_kernel"], "8 B record read + 4..8 B position written per occurrence"),
    "sort": ("list_sort_* (32-bit positions inside every list: short lists bucket-sorted in LDS, long ones by two radix passes + windows)",
             ["list_sort_small_kernel", "list_sort_hist_kernel", "list_sort_chunk_kernel", "list_sort_scan_kernel", "list_sort_prefix_kernel",
              "list_sort_scatter_kernel", "list_sort_window_plan_kernel", "list_sort_window_desc_kernel", "list_sort_window_kernel"],
             "one read + one write of every position (2 x 4 B); the radix passes in between are overhead"),
    "filter_ladder": ("rung_build_kernel (4-ary search ladder over the sorted lists, for the pivot filter)", ["rung_build_kernel"],
                      "every list element read once (4 B); a third as many entries written"),
    "filter_pivot": ("filter_pivot_kernel", ["filter_pivot_kernel"], "16 B per (pivot element, level): two lower bounds"),
    "filter_pass": ("filter_pass_kernel", ["filter_pass_kernel"], "4 B per list element streamed"),
    "filter_compact": ("filter_compact_kernel", ["filter_compact_kernel", "filter_count_runs_kernel", "filter_gather_counts_kernel"],   # (+ a small rocPRIM scan: its
                       # trampoline is shared with the index builder's scans, so its counters are left out)
                       "activity bits read + every survivor read and written (2 x 4 B)"),
    "join_init": ("join_init_kernel", ["join_init_kernel"], ""),
    "join_link": ("join_link_kernel", ["join_link_kernel"], "8 B per join slot (SURVEY 8d K5: list element read + link state written)"),
    "join_scan": ("bits_summary_kernel", ["bits_summary_kernel"], ""),
    "join_chain": ("join_jump + chain_tiles/walk/emit", ["join_jump_kernel", "chain_tiles_kernel", "chain_walk_kernel", "chain_emit_kernel"],
                   "8 B per slot of every first list"),
    "gather": ("join_gather_kernel", ["join_gather_kernel"], "8 B per match (and per tuple value) written"),
    "exchange": ("vlg_comm_allgatherv (RCCL broadcasts of the ranks' sorted lists)", [], "4 B per occurrence located by another rank (received)"),
    "join": ("vlg_join_batch", [], ""),
}


def cpu_baseline(idx, queries, occ_per_query_mean, budget_occ=1600000, budget_s=30.0, min_queries=100, threads_budget_s=10.0):
    """The CPU oracle (restatement of the reference path, reference data layout) timed on this host, ONE thread (the
    reference is single-threaded, gm_search.cpp:91), on a bounded seeded sample of the same query batch; then the same
    code on nproc NATIVE threads (oracle/vlg_oracle.c: vlgo_search_many, a pthread pool drawing queries from a shared counter),
    for disclosure."""
    from oracle import oracle as O
    import numpy as np
    parts = idx.export_parts()
    o = O.Index.from_parts(parts)
    rng = np.random.default_rng(12345)
    order = rng.permutation(len(queries))
    occ_of = {}

    def need_of(qi):
        if qi not in occ_of:
            subs, _, _, _ = O.query_fields(O.parse(queries[qi]))
            occs = [o.backward_search(sp)[0] for sp in subs]
            occ_of[qi] = 0 if min(occs) == 0 else sum(occs)
        return occ_of[qi]
    # the sample: queries in random order whose lists fit what is left of the occurrence budget -- min_queries of them, each capped
    # so that no single heavy query eats the budget (a 10^6-occurrence query costs 40 s on one core); it stops at min_queries
    # queries or budget_s seconds, whichever comes first
    per_query_cap = budget_occ // min_queries * 2
    sample, remaining = [], budget_occ
    for qi in order:
        need = need_of(int(qi))
        if need > remaining or need > per_query_cap:
            continue
        sample.append(queries[qi])
        remaining -= need
        if len(sample) >= min_queries or remaining < 1000:
            break
    t0 = time.perf_counter()
    done, _, stats = O.search_many(o, sample, threads=1, budget_s=budget_s)
    dt = time.perf_counter() - t0
    occ_rate = float(stats[0]) / dt if dt > 0 else 0.0
    # a sample's queries/s depends on which heavy queries it happened to draw; the stable figure is located
    # occurrences/s, converted with the exact mean occurrences per query of the full batch
    qps = occ_rate / occ_per_query_mean if occ_per_query_mean > 0 else 0.0
    out = {"value": qps, "unit": "queries/s", "cores": 1, "kind": "port",
           "located_occ_per_sec": occ_rate, "sample_queries": done, "sample_seconds": dt,
           "sample_located_occ": int(stats[0]), "sample_lf_steps": int(stats[1]),
           "sample": "%d queries of the same batch, drawn in random order (seed 12345) while their occurrence lists fit a "
                     "%d-occurrence budget (at most %d per query; stops after %d queries or %.0f s); %.1f s on one core; queries/s = sample "
                     "occurrences/s / mean occurrences per query of the full batch (%.0f)"
                     % (done, budget_occ, per_query_cap, min_queries, budget_s, dt, occ_per_query_mean)}
    # disclosure: the same restatement on T = nproc native threads; the reference itself is single-threaded
    T = max(1, host_cpus()[0])
    if T > 1:
        light = []
        for qi in order:
            if 0 < need_of(int(qi)) <= 20000:
                light.append(queries[qi])
            if len(light) >= 400 * T:
                break
        t0 = time.perf_counter()
        done_t, _, st_t = O.search_many(o, light, threads=T, budget_s=threads_budget_s)
        dt_t = time.perf_counter() - t0
        rate_t = float(st_t[0]) / dt_t if dt_t > 0 else 0.0
        out["threads_T"] = {"threads": T, "located_occ_per_sec": rate_t,
                            "queries_per_sec": rate_t / occ_per_query_mean if occ_per_query_mean > 0 else 0.0,
                            "speedup_over_one_core": rate_t / occ_rate if occ_rate > 0 else None,
                            "sample_seconds": dt_t, "sample_located_occ": int(st_t[0]), "sample_queries": done_t,
                            "what": "vlgo_search_many: %d native threads draw queries (each <= 20 000 occurrences) from a shared counter for "
                                    "%.0f s; the index is only read" % (T, threads_budget_s)}
    return out


def cpu_sasearch(text, queries, occ_per_query_mean, budget_occ=60000000, budget_s=8.0, threads_budget_s=5.0):
    """Disclosure (CPU baseline (3)): the benchmark's plain-suffix-array index -- text + SA, forward_search, sort,
    join (index_sasearch.hpp) -- restated in the oracle, one thread and nproc native threads, same random query order.  The suffix
    array comes from the device sorter (vlg_suffix_array_device); building it is not part of the figure, as `load` is not in the
    reference's."""
    from oracle import oracle as O
    import numpy as np
    import torch
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
    sample, remaining = [], budget_occ
    order = rng.permutation(len(queries))
    for qi in order:
        subs, _, _, _ = O.query_fields(O.parse(queries[qi]))
        need = sum(s.count(sp) for sp in subs)
        if need > remaining:
            continue
        sample.append(queries[qi])
        remaining -= need
        if remaining < 1000 or len(sample) >= 20000:
            break
    t0 = time.perf_counter()
    done, _, stats = O.search_many(s, sample, threads=1, budget_s=budget_s)
    dt = time.perf_counter() - t0
    occ_rate = float(stats[0]) / dt if dt > 0 else 0.0
    out = {"algorithm": "SASEARCH (plain suffix array + text, index_sasearch.hpp)", "cores": 1, "kind": "port",
           "queries_per_sec": occ_rate / occ_per_query_mean if occ_per_query_mean > 0 else 0.0, "located_occ_per_sec": occ_rate,
           "sample_queries": done, "sample_seconds": dt, "sample_located_occ": int(stats[0]), "suffix_array_on_device_s": t_sa,
           "note": "every sub-pattern's SA range is copied and sorted, query by query, as the reference does; "
                   "queries/s = sample occurrences/s / mean occurrences per query of the full batch"}
    T = max(1, host_cpus()[0])
    if T > 1:
        many = [queries[qi] for qi in order[:min(len(order), 200 * T)]]
        t0 = time.perf_counter()
        done_t, _, st_t = O.search_many(s, many, threads=T, budget_s=threads_budget_s)
        dt_t = time.perf_counter() - t0
        rate_t = float(st_t[0]) / dt_t if dt_t > 0 else 0.0
        out["threads_T"] = {"threads": T, "located_occ_per_sec": rate_t,
                            "queries_per_sec": rate_t / occ_per_query_mean if occ_per_query_mean > 0 else 0.0,
                            "speedup_over_one_core": rate_t / occ_rate if occ_rate > 0 else None,
                            "sample_seconds": dt_t, "sample_located_occ": int(st_t[0]), "sample_queries": done_t,
                            "what": "vlgo_sasearch_many: the batch's queries in the same random order on %d native threads for %.0f s "
                                    "(queries of any weight: a thread that draws a heavy one sorts millions of positions)" % (T, threads_budget_s)}
    return out


STRONG_FAILED_EXIT = 3          # exit code of every rank when the strong-scaling region failed or stalled (the JSON line is printed first)


def guarded_region(region, store, rank, world, timeout_s, report, exit_fn=os._exit, poll_s=0.25):
    """region() on every rank, under a watchdog, with the outcome agreed through the process group's key-value STORE and not
    through a collective: a rank that raised must not enter an all-reduce while its peers sit inside another collective (on
    RCCL a mismatched collective is undefined, not a clean hang).  Every rank posts `ok` or `fail` for itself; when a failure
    is posted, or the region has not finished on all ranks within timeout_s, each rank calls report(message) -- rank 0 prints
    the weak-scaling line with the error there -- and leaves through exit_fn(STRONG_FAILED_EXIT): launchers and CI gating on the
    exit status see the failure, consumers of stdout still get the line.  Returns region()'s value when all ranks finished."""
    import threading
    finished = threading.Event()
    key_fail, key_ok = "vlg_strong_fail", "vlg_strong_ok_%d"

    def failed_message():
        try:
            return store.get(key_fail).decode("utf-8", "replace") if store.check([key_fail]) else None
        except Exception:                                      # noqa: BLE001 -- the store is gone with rank 0: the run is over
            return "the process group's store is unreachable"

    def leave(msg):
        try:
            report(msg)
        finally:
            sys.stdout.flush()
            sys.stderr.flush()
            exit_fn(STRONG_FAILED_EXIT)

    def watchdog():
        t0 = time.time()
        while not finished.wait(poll_s):
            msg = failed_message()
            if msg is None and time.time() - t0 > timeout_s:
                msg = "strong-scaling region did not finish within %d s" % timeout_s
            if msg is not None:
                leave(msg)
                finished.set()                                 # (only reached when exit_fn returns: tests)
                return
    threading.Thread(target=watchdog, daemon=True).start()
    value = None
    try:
        value = region()
        store.set(key_ok % rank, "1")
    except Exception as e:                                     # noqa: BLE001 -- reported in the line
        import traceback
        traceback.print_exc()
        store.set(key_fail, "rank %d: %s: %s" % (rank, type(e).__name__, e))
    # all ranks done?  (the watchdog ends the wait when somebody failed or stalled)
    keys = [key_ok % r for r in range(world)]
    while not finished.is_set():
        try:
            if store.check(keys):
                break
        except Exception:                                      # noqa: BLE001
            pass
        if failed_message() is not None:
            time.sleep(10 * poll_s)                            # the watchdog thread reports and exits; do not race it
            continue
        time.sleep(poll_s / 4)
    finished.set()
    return value


def claim_stdout():
    """ONE JSON line on stdout: libraries below this process write to file descriptor 1 on their own (RCCL prints a version banner when
    a communicator is made, gloo its connection lines), so from here on everything written to fd 1 -- by them or by print() -- goes to
    stderr, and the returned function writes an object as one line to the real stdout."""
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())
    return emit


def other_config(name, steps, warmup, workspace_gb, tuples):
    """One of BASELINE's other configs (C2 100 MiB DNA, C4 4 GiB protein with 10^6 queries, C5 1 GiB DNA on rrr-63
    bit-vectors) on this GPU: text, index and batch made here, `steps` batches timed as the metric's are (resident batch, synchronize
    on both sides), results checked against workload.EXPECTED -- the constants tests/test_gpu_fullsize.py asserts for the same batch."""
    import torch
    import vlg_matching_amd as V
    from vlg_matching_amd import workload
    from vlg_matching_amd.index import Queries, Workspace
    cfg = workload.config(name)
    t0 = time.perf_counter()
    text = workload.gen_text(cfg["kind"], cfg["n"], cfg["seed"])
    t_gen = time.perf_counter() - t0
    d_text = torch.from_numpy(text).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx = V.VlgIndex.build_device(d_text.data_ptr(), len(text))
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    del d_text
    if name == "C5":
        idx = idx.compress()                                   # csa_wt<wt_huff<rrr_vector<63>>>
    torch.cuda.empty_cache()
    queries = workload.gen_queries(text, cfg["nq"], cfg["k"], cfg["m"], cfg["gap"], cfg["qseed"])
    del text
    q = Queries(queries)
    ws = Workspace(int(workspace_gb * (1 << 30)))
    ws.set_option("tuples", 1 if tuples else 0)
    for _ in range(warmup):
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
    sm = r.summary
    info = idx.info()
    want = workload.EXPECTED.get(name)
    got = {k: sm[k] for k in ("n_matches", "checksum", "located_occurrences")}
    if want is not None and got != want:
        raise AssertionError("%s: %r, expected %r" % (name, got, want))
    loc = ks["locate"]
    launches = max(loc["launches"], 1)
    ach = loc["algorithmic_bytes"] / (loc["total_ms"] * 1e-3) / 1e9 if loc["total_ms"] > 0 else 0.0
    return {"workload": "%s: %s text n=%d (seed %d), %d queries x k=%d, m=%d, gap .{%d,%d}?, %s bit-vectors"
                        % (name, cfg["kind"], cfg["n"], cfg["seed"], cfg["nq"], cfg["k"], cfg["m"], cfg["gap"][0], cfg["gap"][1],
                           "rrr_vector<63>" if info["bv_kind"] else "plain"),
            "ms_per_step": dt / steps * 1e3, "value": sm["n_queries"] * steps / dt, "unit": "queries/s", "steps": steps,
            "located_occ_per_sec": sm["located_occurrences"] * steps / dt,
            "matches_per_step": sm["n_matches"], "checksum": sm["checksum"], "located_occ_per_step": sm["located_occurrences"],
            "checked_against_expected": want is not None,
            "lf_steps_per_occ": sm["lf_steps"] / max(sm["located_occurrences"], 1), "chunks_per_step": sm["n_chunks"],
            "index_hbm_bytes": info["hbm_bytes"], "index_build_s": t_build, "text_gen_s": t_gen,
            "kernels_ms_per_step": {k: v["total_ms"] / steps for k, v in ks.items() if v["total_ms"] > 0},
            "rank_kernel_roofline": {"bound": "hbm", "kernel_class": "locate", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": ach / HBM_PEAK_GBS, "ms_per_step": loc["total_ms"] / steps,
                                     "avg_launch_ms": loc["total_ms"] / launches, "launches": loc["launches"],
                                     "algorithmic_bytes_per_launch": loc["algorithmic_bytes"] / launches,
                                     "algorithmic_bytes_are": CLASSES["locate"][2]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink text and batch (development only)")
    ap.add_argument("--workspace-gb", type=float, default=160.0)
    ap.add_argument("--bv", choices=["plain", "rrr"], default=None, help="wavelet-tree bit-vectors (default: rrr for C5, else plain)")
    ap.add_argument("--sa-dens", type=int, default=32,
                    help="suffix-array sample density t_dens of csa_wt<> (default 32 = the benchmark's index; 1 keeps the whole suffix "
                         "array resident in HBM: 4 B x n, no LF walk in locate). Not the headline config unless 32.")
    ap.add_argument("--no-dense-sa", action="store_true", help="N = 1: skip the secondary run on the index that keeps the whole suffix array")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--other-configs", default="C2,C5,C4",
                    help="N = 1, default config only: BASELINE's other single-GPU configs measured behind the metric's regions and reported in "
                         "`other_configs` (each: ms per batch, matches / checksum checked against the constants the full-size tests assert, "
                         "roofline of its rank kernel); '' = none")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling region")
    ap.add_argument("--strong-timeout", type=float, default=120.0,
                    help="N > 1: seconds the strong-scaling region may take before the weak-scaling line is printed without it")
    ap.add_argument("--strong-sharding", choices=["lists", "affinity", "work"], default="lists",
                    help="strong-scaling region: 'lists' = collective search (distinct lists sharded for locate + sort, sorted lists "
                         "all-gathered, queries sharded for the joins); 'affinity' / 'work' = only the query loop is sharded")
    ap.add_argument("--exchange", choices=["needed", "allgather"], default="needed",
                    help="strong-scaling region with --strong-sharding lists: 'needed' = a sorted list travels only to the ranks whose queries "
                         "use it, pairwise (vlg_comm_alltoallv); 'allgather' = every list to every rank (vlg_comm_allgatherv)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (parse + H2D + search + D2H) region")
    ap.add_argument("--tuples", action="store_true", help="also materialise every sub-pattern position of every match (sdsl::locate output)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--dist-at-1", action="store_true",
                    help="rehearsal: with --gpus 1, still open a 1-rank process group and take the multi-rank code path (RCCL broadcast "
                         "of the index, scatter of the batches, reductions, strong-scaling region)")
    ap.add_argument("--index-broadcast", choices=["rccl", "torch"], default="torch",
                    help="N > 1: how the index gets to the ranks BEFORE the metric's region: torch.distributed.broadcast of its image "
                         "(default: RCCL through torch) or vlg_index_broadcast over a communicator made through the C-ABI (the product's "
                         "own RCCL binding).  With the default the product's broadcast runs -- and is timed and checked -- inside the "
                         "guarded strong-scaling region instead, where a failure of the so far one-rank-only path cannot cost the metric")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                     # does not return

    emit = claim_stdout()
    # (the host driver of this pool only supports dmabuf IPC: without this RCCL fails with hipIpcGetMemHandle: invalid argument;
    # set before anything initialises HIP, kept if the launcher already chose a value)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or plainly "
                 "without WORLD_SIZE set)" % (args.gpus, world, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.dist_at_1:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus

    import vlg_matching_amd as V
    from vlg_matching_amd import dist as vdist
    from vlg_matching_amd import workload
    from vlg_matching_amd.index import Queries, Workspace
    V.capi.check(V.lib().vlg_set_device(local_rank))
    dev = torch.device("cuda", local_rank)

    cfg = workload.config(args.config, args.scale)
    t_gen = t_build = t_bcast = 0.0
    # ---- index: built on rank 0, replicated over RCCL (one broadcast of the contiguous HBM image) ----------
    text = None
    idx = plain_idx = None
    if rank == 0:
        t0 = time.perf_counter()
        text = workload.gen_text(cfg["kind"], cfg["n"], cfg["seed"])
        t_gen = time.perf_counter() - t0
        log("text %s n=%d generated in %.1f s" % (cfg["kind"], cfg["n"], t_gen))
        d_text = torch.from_numpy(text).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx = V.VlgIndex.build_device(d_text.data_ptr(), len(text), dens=args.sa_dens)    # csa_wt<wt_huff<>, sa_dens, .>
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        del d_text
        torch.cuda.empty_cache()
        log("index built on device in %.2f s: %s" % (t_build, idx.info()))
        if (args.bv or ("rrr" if args.config == "C5" else "plain")) == "rrr":
            t0 = time.perf_counter()
            plain_idx, idx = idx, idx.compress()
            log("rrr-63 re-encoding on device in %.2f s: %s" % (time.perf_counter() - t0, idx.info()))
            if args.no_cpu_baseline or world > 1:
                plain_idx = None
    bcast_via = None
    if dist is not None:
        comm = None
        if args.backend == "nccl" and args.index_broadcast == "rccl":
            comm = vdist.Comm.from_torch_dist(dist)                # ncclCommInitRank through the C-ABI; the id goes through torch's store
            bcast_via = "vlg_index_broadcast (RCCL: %s)" % comm.library()
        else:
            bcast_via = "torch.distributed.broadcast (%s)" % args.backend
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx = vdist.replicate_index(idx, dist, dev, src=0, comm=comm)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
        log("rank %d: index image replicated in %.3f s via %s" % (rank, t_bcast, bcast_via))
        if comm is not None:
            comm.close()
    info = idx.info()

    # ---- query batches: one per rank (weak scaling), generated where the text is ---------------------------
    if rank == 0:
        batches = [workload.gen_queries(text, cfg["nq"], cfg["k"], cfg["m"], cfg["gap"], cfg["qseed"] + 1000 * r)
                   for r in range(world)]
    else:
        batches = None
    if dist is not None:
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

    def timed(fn, steps):
        """barrier + synchronize on both sides, max over ranks -> (seconds, last return value)"""
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = fn()
        torch.cuda.synchronize()
        mine_dt = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, mine_dt, out

    def reduce_sum(vals):
        t = torch.tensor(vals, dtype=torch.int64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [int(x) for x in t.tolist()]

    def gather_per_rank(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        if dist is None:
            return [t.tolist()]
        outl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outl, t)
        return [o.tolist() for o in outl]

    # ---- the timed region of the metric: K steps of the whole hot path over this rank's resident batch ------------
    for _ in range(args.warmup):
        idx.search(q, workspace=ws)
    ws.profile(True)
    dt, my_dt, res = timed(lambda: idx.search(q, workspace=ws), args.steps)
    s = res.summary
    n_queries, n_occ, n_matches, n_logical = reduce_sum([s["n_queries"], s["located_occurrences"], s["n_matches"], s["logical_occurrences"]])
    kstats = ws.kernel_stats()
    ws.profile(False)
    per_rank = gather_per_rank([my_dt / args.steps * 1e3, s["located_occurrences"], s["n_matches"]])

    # ---- end to end, as SURVEY.md 8(d) words it: parse + H2D of the query batch, search, D2H of counts and first positions
    #      (pinned host buffers; the index upload is separate, as `load` is in gm_search.cpp:68-83) ---------------------------
    e2e = r_e = r_p = None
    if not args.no_e2e:
        raws = [r.encode("latin-1") for r in queries]
        off = np.zeros(len(raws) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in raws])
        blob = b"".join(raws)                                  # the pattern file's bytes: read before the clock starts, like gm_search.cpp:174
        nq = len(raws)
        pin_counts = torch.empty(max(nq, 1), dtype=torch.int64).pin_memory()
        pin_off = torch.empty(nq + 1, dtype=torch.int64).pin_memory()
        pin_first = torch.empty(max(s["n_matches"], 1) + 1024, dtype=torch.int64).pin_memory()
        phases = [0.0, 0.0, 0.0]
        pos_width = 4 if info["n"] <= (1 << 32) + 1 and os.environ.get("VLG_FORCE_POS64", "0") != "1" else 8

        def e2e_step():
            t0 = time.perf_counter()
            qq = Queries.from_blob(blob, off)
            t1 = time.perf_counter()
            r = idx.search(qq, workspace=ws)
            t2 = time.perf_counter()
            assert r.summary["n_matches"] <= pin_first.numel()
            r.fetch_into(pin_counts.data_ptr(), pin_off.data_ptr(), pin_first.data_ptr(), None)
            t3 = time.perf_counter()
            phases[0] += t1 - t0; phases[1] += t2 - t1; phases[2] += t3 - t2
            return r
        e2e_step()                                             # pinned pages touched, result buffers recycled
        phases[:] = [0.0, 0.0, 0.0]
        dt_e, _, r_e = timed(e2e_step, args.steps)
        assert r_e.summary["checksum"] == s["checksum"] and r_e.summary["n_matches"] == s["n_matches"]
        assert int(pin_counts[:nq].sum()) == s["n_matches"]
        e2e = {"ms_per_step": dt_e / args.steps * 1e3, "queries_per_sec": n_queries * args.steps / dt_e,
               "parse_and_h2d_ms": phases[0] / args.steps * 1e3, "search_ms": phases[1] / args.steps * 1e3,
               "d2h_ms": phases[2] / args.steps * 1e3, "h2d_bytes": len(blob) + off.nbytes,
               "d2h_bytes": int(pos_width * s["n_matches"]),
               "what": "vlg_queries_parse (host parse + upload) + vlg_search_batch + vlg_result_fetch of counts, offsets and first "
                       "positions (64-bit values, as gapped_search_result holds them) into pinned host memory; d2h_bytes is what "
                       "crosses PCIe (results of a text <= 4 GiB are held 4 bytes wide in HBM and widened by host threads); "
                       "rank 0's figures, max over ranks for ms_per_step"}
        # the same batches as a stream: a second host thread fetches batch i while batch i + 1 is searched (a result owns its HBM
        # buffers, so nothing of the next batch touches them)
        import threading

        def fetch_of(r):
            torch.cuda.set_device(dev)                         # the current device is a per-thread setting
            r.fetch_into(pin_counts.data_ptr(), pin_off.data_ptr(), pin_first.data_ptr(), None)

        def e2e_stream():
            th = None
            for _ in range(args.steps):
                r = idx.search(Queries.from_blob(blob, off), workspace=ws)
                if th is not None:
                    th.join()
                th = threading.Thread(target=fetch_of, args=(r,))
                th.start()
            th.join()
            return r
        dt_p, _, r_p = timed(e2e_stream, 1)
        assert r_p.summary["checksum"] == s["checksum"] and int(pin_counts[:nq].sum()) == s["n_matches"]
        assert int(pin_first[: s["n_matches"]].sum().item()) % (1 << 64) == s["checksum"]      # (int64 sums wrap like the checksum)
        e2e["overlapped_ms_per_step"] = dt_p / args.steps * 1e3
        e2e["overlapped_what"] = ("%d batches back to back, vlg_result_fetch of batch i on a second host thread while batch i + 1 is "
                                  "parsed and searched (last fetch inside the clock)" % args.steps)
        del pin_first

    # ---- the same batch on an index that keeps the whole suffix array in HBM (t_dens = 1: 4 B x n more; what 288 GB afford) ----------
    #      Reported beside the metric, never as the metric: BASELINE's index is csa_wt<wt_huff<>, 32, 64>.
    dense = None
    if world == 1 and args.sa_dens == 32 and not args.no_dense_sa and info["n"] <= (1 << 32):
        t0 = time.perf_counter()
        idx_d = idx.resample(text_order=False, dens=1)
        torch.cuda.synchronize()
        t_rs = time.perf_counter() - t0
        idx_d.search(q, workspace=ws)                          # warm-up
        ws.profile(True)
        dt_d, _, r_d = timed(lambda: idx_d.search(q, workspace=ws), args.steps)
        ks_d = ws.kernel_stats()
        ws.profile(False)
        sd = r_d.summary
        assert sd["checksum"] == s["checksum"] and sd["n_matches"] == s["n_matches"] and sd["located_occurrences"] == s["located_occurrences"]
        assert sd["lf_steps"] == 0
        dense = {"ms_per_step": dt_d / args.steps * 1e3, "value": s["n_queries"] * args.steps / dt_d, "unit": "queries/s",
                 "located_occ_per_sec": sd["located_occurrences"] * args.steps / dt_d,
                 "index_hbm_bytes": idx_d.info()["hbm_bytes"], "resample_s": t_rs,
                 "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in ks_d.items() if v["total_ms"] > 0},
                 "what": "NOT the metric: the same batch, same matches and checksum, on csa_wt<wt_huff<>, 1, .> made by "
                         "vlg_index_resample(SA order, 1) -- every suffix-array value is a sample, locate copies SA intervals "
                         "(sa_dense_copy_kernel), no LF walk, no trails, no records"}
        del idx_d, r_d
        torch.cuda.empty_cache()

    # ---- strong scaling: THE batch (rank 0's) cut by work, every rank searches its slice ------------------------------------
    def strong_region():
        strong = None
        fault = os.environ.get("VLG_BENCH_STRONG_FAULT", "")       # rehearsal of the watchdog: the last rank fails or stalls here
        if fault and rank == world - 1:
            if fault == "raise":
                raise RuntimeError("VLG_BENCH_STRONG_FAULT=raise")
            time.sleep(1e6)
        if dist is not None and not args.no_strong and args.strong_sharding == "lists":
            # collective search: every rank gets THE batch (rank 0's), locates + sorts its share of the distinct lists, receives the other
            # shares (RCCL all-gather of the sorted lists; gloo rehearsals move them through the host), joins its piece of the queries
            box = [queries if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            qs = Queries(box[0])
            comm_x = None
            idx_s, t_bx = idx, None
            if args.backend == "nccl":
                comm_x = vdist.Comm.from_torch_dist(dist)
                # the product's own replication (vlg_index_broadcast, include/vlg_hip.h): the ranks other than 0 answer the collective
                # search from the image they receive here, so its content is checked by the totals below
                torch.cuda.synchronize()
                dist.barrier()
                t0x = time.perf_counter()
                idx_s = comm_x.broadcast_index(idx if rank == 0 else None, root=0)
                torch.cuda.synchronize()
                dist.barrier()
                t_bx = time.perf_counter() - t0x
                ws.set_comm(comm_x)
            elif args.exchange == "allgather":
                ws.set_exchange(world, rank, vdist.host_exchange(dist))
            else:
                ws.set_exchange_alltoall(world, rank, vdist.host_alltoall(dist))
            ws.set_option("exchange_all", 1 if args.exchange == "allgather" else 0)
            idx_s.search(qs, workspace=ws)                         # warm-up
            ws.profile(True)
            dt_s, my_dt_s, r_s = timed(lambda: idx_s.search(qs, workspace=ws), args.steps)
            x_stats = ws.kernel_stats()
            ws.profile(False)
            ss = r_s.summary
            tq_all = ss["n_queries"]
            owned = r_s.owned_queries()
            n_own = sum(b - a for a, b in owned) if owned else tq_all
            tq, tocc, tm = reduce_sum([n_own, ss["located_occurrences"], ss["n_matches"]])
            chk = vdist.reduce_checksum(ss["checksum"], dist, dev)
            pr = gather_per_rank([my_dt_s / args.steps * 1e3, n_own, ss["located_occurrences"], x_stats["exchange"]["total_ms"] / args.steps,
                                  x_stats["exchange"]["algorithmic_bytes"] / args.steps])
            strong = {"scaling": "strong", "value": tq * args.steps / dt_s, "unit": "queries/s", "ms_per_step": dt_s / args.steps * 1e3,
                      "queries": tq, "matches": tm, "checksum": chk, "located_occ_per_step_all_ranks": tocc,
                      "per_rank": [{"ms_per_step": p[0], "queries": int(p[1]), "located_occ": int(p[2]), "exchange_ms": p[3],
                                    "exchange_bytes_received": int(p[4])} for p in pr],
                      "sharding": "lists",
                      "exchange": ("%s, %s" % ("pairwise, needed lists only" if args.exchange == "needed" else "in-place all-gather of every list",
                                               ("vlg_comm_alltoallv (RCCL)" if args.exchange == "needed" else "vlg_comm_allgatherv (RCCL)")
                                               if comm_x is not None else "host (gloo rehearsal)")),
                      "exchange_bytes_all_gather_would_receive": [int(4 * (tocc - p[2])) for p in pr],
                      "index_broadcast_rccl_s": t_bx,
                      "index_broadcast_rccl": None if comm_x is None else
                      "vlg_index_broadcast of %d bytes over %s; ranks > 0 searched the image they received" % (info["hbm_bytes"], comm_x.library()),
                      "note": "the 1-GPU batch answered by all ranks together: every distinct occurrence list is located and sorted by exactly one "
                              "rank (located_occ summed over ranks = the 1-GPU figure), the sorted lists are all-gathered, every rank filters and "
                              "joins a contiguous piece of the queries of equal join work"}
            ws.set_comm(None)
            if comm_x is not None:
                comm_x.close()
            if rank == 0 and not (tq == s["n_queries"] and tm == s["n_matches"] and chk == s["checksum"] and tocc == s["located_occurrences"]):
                raise AssertionError("collective search disagrees with the 1-GPU batch: %r vs %r" % ((tq, tm, chk, tocc), s))
        elif dist is not None and not args.no_strong:
            if rank == 0:
                if args.strong_sharding == "affinity":             # queries that share their longest list stay together
                    l_, r_, q_all = idx.intervals(queries)
                    sets = vdist.shard_by_affinity(l_, r_, q_all.subpattern_range(), world)
                    slices = [[queries[i] for i in st_] for st_ in sets]
                else:
                    w = idx.query_weights(queries)                 # sum of SA-interval sizes per query: one backward-search pass
                    slices = [queries[b:e] for b, e in vdist.shard_by_work(w, world)]
            else:
                slices = None
            mine = [None]
            dist.scatter_object_list(mine, slices, src=0)
            qs = Queries(mine[0])
            idx.search(qs, workspace=ws)                           # warm-up of the slice's shapes
            dt_s, my_dt_s, r_s = timed(lambda: idx.search(qs, workspace=ws), args.steps)
            ss = r_s.summary
            tq, tocc, tm, tlog = reduce_sum([ss["n_queries"], ss["located_occurrences"], ss["n_matches"], ss["logical_occurrences"]])
            chk = vdist.reduce_checksum(ss["checksum"], dist, dev)
            pr = gather_per_rank([my_dt_s / args.steps * 1e3, ss["n_queries"], ss["located_occurrences"], ss["logical_occurrences"]])
            strong = {"scaling": "strong", "value": tq * args.steps / dt_s, "unit": "queries/s", "ms_per_step": dt_s / args.steps * 1e3,
                      "queries": tq, "matches": tm, "checksum": chk, "located_occ_per_step_all_ranks": tocc, "logical_occ_per_step": tlog,
                      "per_rank": [{"ms_per_step": p[0], "queries": int(p[1]), "located_occ": int(p[2]), "logical_occ": int(p[3])} for p in pr],
                      "sharding": args.strong_sharding,
                      "note": "the 1-GPU batch sharded; each rank locates the distinct intervals of its own queries, so a list that queries "
                              "on several ranks share is located once per such rank (located_occ summed over ranks >= the 1-GPU figure). "
                              "'affinity' keeps the queries that share their longest list on one rank (vlg_matching_amd.dist.shard_by_affinity); "
                              "'work' cuts contiguous slices of equal sum of SA-interval sizes"}
            if rank == 0 and not (tq == s["n_queries"] and tm == s["n_matches"] and chk == s["checksum"]):   # same batch as rank 0's weak region
                raise AssertionError("sharded batch disagrees with the 1-GPU batch: %r vs %r" % ((tq, tm, chk), s))
        return strong

    out = None
    if rank == 0:
        pmc = load_pmc()
        step_ms = dt / args.steps * 1e3

        def roof(name):
            st = kstats[name]
            disp, members, what = CLASSES.get(name, (name, [], ""))
            if name == "locate" and not kstats["locate_partition"]["launches"]:
                disp, members = "locate_kernel", ["locate_kernel"]
            launches = max(st["launches"], 1)
            ms = st["total_ms"] / launches
            alg = st["algorithmic_bytes"] / launches
            ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            # counter traffic of the class per step = sum over its kernels of bytes per launch x launches per step in the PMC run
            tr = rd = wr = None
            got = [pmc[m] for m in members if m in pmc]
            if got:
                steps_pmc = max(int(pmc.get("_meta", {}).get("steps_profiled", 1)), 1)
                rd = sum(g["read_bytes_per_launch_raw"] * g["launches"] for g in got) / steps_pmc
                wr = sum(g["write_bytes_per_launch"] * g["launches"] for g in got) / steps_pmc
                tr = rd + wr
            class_ms = st["total_ms"] / args.steps
            traffic_per_launch = tr / (st["launches"] / args.steps) if tr is not None and st["launches"] else None
            return {"bound": "hbm", "kernel": disp, "kernel_class": name, "kernels_in_class": members,   # (avg_launch_ms averages over the launches of all of them)
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic_per_launch,
                    "traffic_detail": None if tr is None else {
                        "read_raw_per_step": rd, "read_if_all_wide_streaming_per_step": 2 * rd, "write_per_step": wr,
                        "source": "profiles/pmc_traffic.json (%s; counter passes taken at commit %s -- a kernel changed since then "
                                  "makes its figure stale)" % (got[0].get("tag", ""), pmc.get("_meta", {}).get("commit", "unknown")),
                        "note": "FETCH_SIZE is exact for 64-byte random requests and reads half of a wide coalesced streaming read on "
                                "gfx950 (MI355X_MICROARCH.md, HBM): both readings are given"},
                    # the other wall of this workload: random 64-byte read requests (K1, profiles/r01_k1_bitrank_kernel.json: 5.0e10 per
                    # second chip-wide whatever the working set; FETCH_SIZE counts exactly 64 B per such request)
                    "read_requests_64B_per_s": None if rd is None or class_ms <= 0 else rd / 64.0 / (class_ms * 1e-3),
                    "frac_of_random_request_wall": None if rd is None or class_ms <= 0 else rd / 64.0 / (class_ms * 1e-3) / RANDOM_REQUEST_WALL,
                    "hbm_frac_counter": None if tr is None or class_ms <= 0 else tr / (class_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "hbm_frac_counter_if_streaming": None if tr is None or class_ms <= 0 else (2 * rd + wr) / (class_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic_over_algorithmic": None if tr is None or not st["algorithmic_bytes"] else tr / (st["algorithmic_bytes"] / args.steps),
                    "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_are": what,
                    "avg_launch_ms": ms, "launches": st["launches"], "ms_per_step": class_ms, "share_of_step": class_ms / step_ms}
        dominant = max(kstats, key=lambda k: kstats[k]["total_ms"])
        out = {
            "metric": "vlg_queries_per_sec",
            "value": n_queries * args.steps / dt,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # text positions are 32-bit inside the kernels up to n = 2^32 + 1 (C4: 33-bit SA indices, 32-bit positions); u64 at the boundary
            "dtype": "u32" if info["n"] <= (1 << 32) + 1 and os.environ.get("VLG_FORCE_POS64", "0") != "1" else "u64",
            "data": "synthetic",
            "config": {"workload": "%s%s: %s text n=%d (seed %d), %d queries/GPU x k=%d, m=%d, gap .{%d,%d}?, t_dens=%d%s"
                                   % (args.config, "" if args.scale == 1.0 else " x%g" % args.scale, cfg["kind"], cfg["n"],
                                      cfg["seed"], cfg["nq"], cfg["k"], cfg["m"], cfg["gap"][0], cfg["gap"][1], args.sa_dens,
                                      "" if args.sa_dens == 32 else " (NOT the benchmark's index: --sa-dens)"),
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
            "index_build_s": t_build, "text_gen_s": t_gen, "index_broadcast_s": t_bcast, "index_broadcast_via": bcast_via,
            "per_rank": [{"ms_per_step": p[0], "located_occ": int(p[1]), "matches": int(p[2])} for p in per_rank],
            "e2e_ms_per_step": e2e["ms_per_step"] if e2e else None,
            "e2e": e2e,
            "hbm_resident_sa": dense,
            "strong_scaling": None,
            "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in kstats.items()},
            "kernels_ms_sum_per_step": sum(v["total_ms"] for v in kstats.values()) / args.steps,
            "roofline": roof(dominant),                                  # the dominant kernel class of the step
            "rank_kernel_roofline": roof("locate"),                      # the LF / bit-rank kernel the north star names
            # every kernel class that takes at least 5 % of the step
            "rooflines": [roof(k) for k in sorted(kstats, key=lambda k: -kstats[k]["total_ms"])
                          if kstats[k]["total_ms"] / args.steps >= 0.05 * step_ms],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(plain_idx if info["bv_kind"] else idx, queries, n_logical / max(n_queries, 1))
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
            out["cpu_baseline"]["host_cores_usable"] = host_cpus()[0]
            out["cpu_baseline"]["host_cpu_quota_cores"] = host_cpus()[1]   # cgroup CPU quota (None: unlimited): what T threads can really get
            sas = cpu_sasearch(host_text, queries, n_logical / max(n_queries, 1))
            if sas is not None:
                out["cpu_baseline"]["sasearch"] = sas

    # The strong region is the only part of an N-GPU run that exchanges data between the ranks (RCCL all-gather of the sorted lists).
    # Whatever happens in it -- an exception on one rank, a collective that never completes -- the weak-scaling line measured above
    # is still printed, and then every rank leaves with STRONG_FAILED_EXIT (guarded_region below).
    if dist is not None and not args.no_strong:
        def report(msg):
            log("rank %d: %s -- leaving with exit code %d; the weak-scaling line stands" % (rank, msg, STRONG_FAILED_EXIT))
            if rank == 0:
                out["strong_scaling"] = {"error": msg}
                emit(out)
        strong = guarded_region(strong_region, dist.distributed_c10d._get_default_store(), rank, world, args.strong_timeout, report)
        if rank == 0:
            out["strong_scaling"] = strong
    # ---- BASELINE's other single-GPU configs, behind everything the metric needs (their failure must not cost the line) ----------
    if rank == 0 and world == 1 and args.config == "C3" and args.scale == 1.0 and args.sa_dens == 32 and args.other_configs:
        idx = q = ws = res = plain_idx = r_e = r_p = None          # (the closures above see None from here on: their work is done)
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out["other_configs"] = {}
        for name in [c for c in args.other_configs.split(",") if c]:
            t0 = time.perf_counter()
            try:
                out["other_configs"][name] = other_config(name, max(2, min(args.steps, 5)), 1, args.workspace_gb, bool(args.tuples))
            except Exception as e:                                 # noqa: BLE001 -- reported in the line
                import traceback
                traceback.print_exc()
                out["other_configs"][name] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["other_configs"][name]["wall_s"] = time.perf_counter() - t0
            torch.cuda.empty_cache()
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
