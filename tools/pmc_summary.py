#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into profiles/pmc_traffic.json (HBM bytes per launch per kernel).

usage: tools/pmc_summary.py <counter_collection.csv with FETCH_SIZE> <counter_collection.csv with WRITE_SIZE> [tag] [search steps in the run] [commit]
(commit: the tree the passes were taken on -- bench.py quotes it beside every traffic figure, which goes stale when a kernel changes)
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3).  gfx950 caveat (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reads half
the bytes of wide coalesced streaming reads; for the 64-byte random requests of this workload tools/k1_bench.py
calibrates it at 1.0 (FETCH_SIZE = TCC_EA0_RDREQ x 64 B, one request per 32-byte block read).  Both raw and
stream-corrected (x2) read figures are kept; `bytes_per_launch` uses the raw read figure + writes."""
import collections
import csv
import json
import os
import re
import sys


def kernel_key(full):
    """Short name of a kernel.  rocPRIM's kernels all share one trampoline: they are told apart by algorithm and key / value types,
    e.g. rocprim:radix_sort_onesweep<unsigned long,empty_type> (the (list, position) sort) vs <unsigned short,unsigned long> (the
    sweep's stable partition)."""
    name = re.sub(r"^void ", "", full)
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", name)
    m = re.search(r"wrapped_(\w+)_config<rocprim::default_config, ([^>]*?)>, \(rocprim::detail::target_arch\)", name)
    if m:
        types = m.group(2).replace("rocprim::", "").replace(" ", "_")
        return "rocprim:%s<%s>" % (m.group(1), types.replace(",_", ","))
    if name.startswith("rocprim::"):
        return "rocprim:" + name.split("<")[0].split("::")[-1]
    return re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].split("<")[0]


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = kernel_key(r["Kernel_Name"])
            agg[name].append(float(r["Counter_Value"]) * 1024.0)
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    tag = sys.argv[3] if len(sys.argv) > 3 else ""
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    commit = sys.argv[5] if len(sys.argv) > 5 else os.environ.get("VLG_COMMIT", "unknown")
    out = {"_meta": {"tag": tag, "steps_profiled": steps, "commit": commit,
                     "note": "launches = launches of the kernel in the whole profiled run (steps_profiled passes of the hot path, warm-up included)"}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        if not f and not w:
            continue
        rd = sum(f) / max(len(f), 1)
        wr = sum(w) / max(len(w), 1)
        out[k] = {"launches": max(len(f), len(w)), "read_bytes_per_launch_raw": rd, "read_bytes_per_launch_x2_if_streaming": 2 * rd,
                  "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr, "tag": tag}
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for k, v in out.items():
        if k == "_meta":
            continue
        print("%-40s launches %5d  read %.3e  write %.3e" % (k, v["launches"], v["read_bytes_per_launch_raw"], v["write_bytes_per_launch"]))


if __name__ == "__main__":
    main()
