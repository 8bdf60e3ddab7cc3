#!/usr/bin/env python3
"""How much do two (three) batches in flight overlap?  The same C3 batch searched by T host threads, each with its own workspace and
stream, against T batches one after the other.  Development tool (tools/): prints one JSON line."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vlg_matching_amd as V
from vlg_matching_amd import workload
from vlg_matching_amd.index import Queries, Workspace

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
gb = float(sys.argv[2]) if len(sys.argv) > 2 else 90.0
cfg = workload.config("C3", scale)
text = workload.gen_text(cfg["kind"], cfg["n"], cfg["seed"])
d_text = torch.from_numpy(text).cuda()
idx = V.VlgIndex.build_device(d_text.data_ptr(), len(text))
del d_text
queries = workload.gen_queries(text, cfg["nq"], cfg["k"], cfg["m"], cfg["gap"], cfg["qseed"])
q = Queries(queries)
out = {"scale": scale, "workspace_gb": gb}
for T in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(T)]
    wss = []
    for t in range(T):
        ws = Workspace(int(gb * (1 << 30) / T * (1 if T == 1 else 1.0)) if T > 1 else int(gb * (1 << 30)), stream=streams[t].cuda_stream)
        ws.set_option("tuples", 0)
        wss.append(ws)
    for ws in wss:
        idx.search(q, workspace=ws)                      # warm-up (allocations)
    torch.cuda.synchronize()
    steps = 2

    def run(ws):
        torch.cuda.set_device(0)
        for _ in range(steps):
            r = idx.search(q, workspace=ws)
        return r
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(ws,)) for ws in wss]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["T%d" % T] = {"batches": T * steps, "seconds": dt, "ms_per_batch": dt / (T * steps) * 1e3}
    del wss
    V.lib().vlg_release_cached_memory() if hasattr(V.lib(), "vlg_release_cached_memory") else None
    torch.cuda.empty_cache()
print(json.dumps(out))
