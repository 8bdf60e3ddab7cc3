"""N>1 path on CPU: world_size-2 gloo processes shard a query batch, search their slices (the oracle stands in for
the GPU matcher here -- tests may use it) and reduce the totals; the result must equal the single-process run."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, text, queries, weights, out_q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import oracle as O
    from vlg_matching_amd import dist as vdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = O.Index.from_text(text)

    def search_fn(qs):
        counts, chk, st = [], 0, np.zeros(4, dtype=np.uint64)
        for q in qs:
            t = idx.search(q, stats=st)
            counts.append(len(t))
            chk = (chk + int(t[:, 0].sum())) % (1 << 64) if len(t) else chk
        return counts, chk, int(st[0])

    r = vdist.run_sharded(search_fn, queries, dist, weights)
    out_q.put((rank, r["local_range"], r["num_results"], r["checksum"], r["located"], list(map(int, r["counts"]))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("by_work", [False, True])
def test_two_ranks_equal_one(oracle, by_work):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from util import dna_text
    from vlg_matching_amd import dist as vdist
    text = dna_text(30000, 5).tobytes()
    rng = np.random.default_rng(4)
    queries = []
    for _ in range(61):
        a, b = (int(x) for x in rng.integers(0, len(text) - 6, 2))
        queries.append("%s.{0,40}?%s" % (text[a:a + int(rng.integers(1, 5))].decode(), text[b:b + int(rng.integers(1, 5))].decode()))
    idx = oracle.Index.from_text(text)
    st = np.zeros(4, dtype=np.uint64)
    ref = [idx.search(q, stats=st) for q in queries]
    want_counts = [len(t) for t in ref]
    want_chk = sum(int(t[:, 0].sum()) for t in ref if len(t)) % (1 << 64)
    weights = None
    if by_work:
        weights = [sum(idx.backward_search(s)[0] for s in oracle.query_fields(oracle.parse(q))[0]) for q in queries]
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, text, queries, weights, out_q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(out_q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (b0, e0), (b1, e1) = got[0][1], got[1][1]
    assert b0 == 0 and e0 == b1 and e1 == len(queries)
    assert got[0][5] + got[1][5] == want_counts
    for g in got:
        assert g[2] == sum(want_counts) and g[3] == want_chk and g[4] == int(st[0])
    if by_work:
        assert vdist.shard_by_work(weights, 2) == [(b0, e0), (b1, e1)]


def test_shard_bounds_cover_exactly():
    from vlg_matching_amd import dist as vdist
    for n in (0, 1, 7, 8, 100001):
        for w in (1, 2, 3, 8):
            parts = [vdist.shard_bounds(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in parts) - min(e - b for b, e in parts) <= 1


def test_shard_by_work_takes_the_nearer_cut():
    from vlg_matching_amd import dist as vdist
    assert vdist.shard_by_work([10, 0, 0, 0], 2) == [(0, 1), (1, 4)]          # the heavy head goes to rank 0, not past it
    assert vdist.shard_by_work([1, 1, 1, 1], 2) == [(0, 2), (2, 4)]
    assert vdist.shard_by_work([0, 0, 0, 10], 2)[0][1] in (3, 4)
    for w, world in (([5, 1, 1, 1, 5], 3), ([1] * 7, 8), ([3, 0, 0, 9, 1, 1, 7, 2], 4)):
        parts = vdist.shard_by_work(w, world)
        assert parts[0][0] == 0 and parts[-1][1] == len(w) and all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))


def test_checksum_reduction_is_modulo_2_64():
    from vlg_matching_amd import dist as vdist
    big = (1 << 64) - 5
    assert vdist.reduce_checksum(big) == big                                  # single rank: unchanged, no 2^63 folding
    r = vdist.run_sharded(lambda qs: ([1] * len(qs), big, 7), ["a", "b", "c"])
    assert r["checksum"] == big and r["num_results"] == 3 and r["located"] == 7


def test_shard_by_affinity_keeps_sharers_together_and_covers_all():
    from vlg_matching_amd import dist as vdist
    rng = np.random.default_rng(3)
    # 400 queries x 3 sub-patterns drawn from 40 lists of very different lengths (interval start = list identity)
    starts = np.cumsum(rng.integers(1, 10 ** 6, 40)).astype(np.uint64)
    lens = (10 ** rng.uniform(0, 6, 40)).astype(np.uint64) + 1
    pick = rng.integers(0, 40, (400, 3))
    l = starts[pick].ravel()
    r = (starts[pick] + lens[pick] - 1).ravel()
    qsub = np.arange(0, 1201, 3)
    for world in (1, 2, 4, 8):
        parts = vdist.shard_by_affinity(l, r, qsub, world)
        allq = np.sort(np.concatenate(parts))
        assert len(parts) == world and (allq == np.arange(400)).all()              # every query exactly once
        heavy = np.array([pick[i][np.argmax(lens[pick[i]])] for i in range(400)])
        rank_of = np.zeros(400, dtype=int)
        for t, p in enumerate(parts):
            rank_of[p] = t
        spread = [len(set(rank_of[heavy == h])) for h in range(40) if (heavy == h).sum() and (heavy == h).sum() * world <= 400]
        assert all(x == 1 for x in spread)                                           # one rank per heavy list (unless it is too popular to fit one)
        # distinct lists per rank, summed, against contiguous slices: fewer located occurrences
        def located(sets):
            return sum(int(lens[np.unique(pick[list(s)].ravel())].sum()) for s in sets if len(s))
        if world > 1:
            contiguous = [np.arange(b, e) for b, e in (vdist.shard_bounds(400, t, world) for t in range(world))]
            assert located(parts) <= located(contiguous)


def test_bench_wait_ranks_ends_the_survivors_of_a_dead_rank():
    """bench.py --gpus N started plainly: when one rank dies the others sit in a collective; the parent must end them and
    return the dead rank's exit code instead of waiting for them one after the other."""
    import subprocess
    import time
    sys.path.insert(0, ROOT)
    import bench
    sleeper = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(120)"])
    dier = subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.3); sys.exit(3)"])
    t0 = time.time()
    rc = bench.wait_ranks([sleeper, dier], poll_s=0.05, grace_s=5.0)
    assert rc == 3
    assert sleeper.poll() is not None, "the surviving rank was left running"
    assert time.time() - t0 < 30
    ok = [subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(2)]
    assert bench.wait_ranks(ok, poll_s=0.05) == 0


def test_bench_stdout_carries_the_json_line_alone():
    """bench.py's contract is ONE JSON line on stdout, but RCCL (its version banner at communicator creation) and gloo (its connection
    lines) write to file descriptor 1 on their own: after claim_stdout() whatever a library or print() writes to fd 1 lands on stderr and
    only emit() reaches the real stdout."""
    import json
    import subprocess
    prog = ("import os, sys; sys.path.insert(0, %r); import bench; emit = bench.claim_stdout(); "
            "os.write(1, b'RCCL version : banner\\n'); print('a print'); sys.stdout.flush(); "
            "os.system('echo from a child process'); emit({'metric': 'm', 'value': 1})" % ROOT)
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("\n") == 1 and json.loads(r.stdout) == {"metric": "m", "value": 1}
    for noise in ("RCCL version : banner", "a print", "from a child process"):
        assert noise in r.stderr


_GUARD_SCRIPT = r'''
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
import bench
rank, fault = int(os.environ["RANK"]), os.environ["FAULT"]
emit = bench.claim_stdout()              # before the process group, as in bench.py: gloo prints its connection lines to fd 1
dist.init_process_group("gloo", rank=rank, world_size=2)
out = {"metric": "m", "value": 1, "strong_scaling": None}

def region():
    if rank == 1 and fault == "raise":
        raise RuntimeError("injected")
    if rank == 1 and fault == "hang":
        time.sleep(1e6)
    t = torch.ones(1)
    dist.all_reduce(t)                 # never completes when rank 1 is not in it
    return {"sum": float(t.item())}

def report(msg):
    if rank == 0:
        out["strong_scaling"] = {"error": msg}
        emit(out)

v = bench.guarded_region(region, dist.distributed_c10d._get_default_store(), rank, 2, float(os.environ["TIMEOUT"]), report)
if rank == 0:
    out["strong_scaling"] = v
    emit(out)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("fault", ["none", "raise", "hang"])
def test_bench_strong_region_watchdog_prints_the_line_and_exits_nonzero(fault):
    """bench.guarded_region: a rank that raises, or a collective that never completes, inside the strong-scaling region still
    yields the weak-scaling JSON line on rank 0 (with the error in it) -- and every rank then exits with STRONG_FAILED_EXIT, so
    a launcher gating on the status sees it.  The outcome is agreed through the store, never through a collective."""
    import json
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FAULT=fault, TIMEOUT="4")
        procs.append(subprocess.Popen([sys.executable, "-c", _GUARD_SCRIPT, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert outs[0][0].count("\n") == 1 and outs[1][0] == "", (outs[0][0], outs[1][0])
    assert line["metric"] == "m" and line["value"] == 1
    if fault == "none":
        assert [p.returncode for p in procs] == [0, 0], [o[1][-400:] for o in outs]
        assert line["strong_scaling"] == {"sum": 2.0}
    else:
        assert [p.returncode for p in procs] == [bench.STRONG_FAILED_EXIT] * 2, ([p.returncode for p in procs], [o[1][-400:] for o in outs])
        assert "error" in line["strong_scaling"]
        assert ("injected" in line["strong_scaling"]["error"]) == (fault == "raise")
