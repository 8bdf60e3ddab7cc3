#!/usr/bin/env python3
"""GPU idle time between kernels of a rocprofv3 --kernel-trace run (developer tool).
usage: gap_analysis.py kernel_trace.csv [min_gap_us]   -- prints the largest gaps and what ran before / after them."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
busy_end = ev[0][1]
gaps = []
for i in range(1, len(ev)):
    s, e, name = ev[i]
    if s > busy_end:
        gaps.append(((s - busy_end) / 1e3, ev[i - 1][2][:60], name[:60], s))
    busy_end = max(busy_end, e)
span = (ev[-1][1] - ev[0][0]) / 1e6
idle = sum(g[0] for g in gaps) / 1e3
print("span %.1f ms, idle %.1f ms in %d gaps" % (span, idle, len(gaps)))
for g in sorted(gaps, reverse=True)[:60]:
    if g[0] >= min_gap:
        print("%10.1f us  after %-60s before %s" % (g[0], g[1], g[2]))

# per-launch durations of the kernels whose name contains argv[3] (e.g. sweep_step_kernel), in launch order
if len(sys.argv) > 3:
    for pat in sys.argv[3:]:
        d = [(s, (e - s) / 1e3) for s, e, n in ev if pat in n]
        print("%s: %d launches, us each: %s" % (pat, len(d), " ".join("%.0f" % x[1] for x in d)))
