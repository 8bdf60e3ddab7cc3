#!/usr/bin/env python3
"""Per-kernel ISA summary of a `hipcc -S --cuda-device-only` listing: registers, scratch, LDS, instruction mix.
usage: isa_stats.py file.s [substring-of-kernel-name ...]   (developer tool; not part of the product)"""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2:]
i = 0
kern = {}
cur = None
for ln in src:
    m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
    if m and not ln.startswith("\t"):
        cur = m.group(1); kern[cur] = []
        continue
    if cur is not None:
        if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            cur = None if ln.startswith(".Lfunc_end") else cur
        if cur: kern[cur].append(ln)
meta = {}
for ln in src:
    m = re.match(r"\s*\.set (\S+)\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)", ln)
    if m: meta.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
for name, body in kern.items():
    if want and not any(w in name for w in want): continue
    if "rocprim" in name and not want: continue
    ops = collections.Counter()
    for ln in body:
        m = re.match(r"\s+([a-z_0-9]+)\s", ln + " ")
        if m and not ln.strip().startswith((".", ";")):
            op = m.group(1)
            cls = ("vmem_ld" if op.startswith(("global_load", "flat_load", "buffer_load")) else
                   "vmem_st" if op.startswith(("global_store", "flat_store", "buffer_store", "global_atomic")) else
                   "lds" if op.startswith("ds_") else "smem" if op.startswith("s_load") else
                   "wait" if op.startswith("s_waitcnt") else "branch" if op.startswith(("s_cbranch", "s_branch")) else
                   "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "other")
            ops[cls] += 1
    print(name[:100], meta.get(name, {}), dict(ops), "total", sum(ops.values()))
