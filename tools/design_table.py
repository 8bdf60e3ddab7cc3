#!/usr/bin/env python3
"""Markdown table of the per-class rooflines of a bench.py line (what DESIGN.md section 6 shows): design_table.py <bench json>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("step %.1f ms, %.0f queries/s, located %.3e/s, e2e %s ms" % (d["ms_per_step"], d["value"], d["located_occ_per_sec"], d.get("e2e_ms_per_step")))
print("| class | ms | frac | counter | traffic / alg. | requests / wall |")
print("|---|---|---|---|---|---|")
for r in d["rooflines"]:
    f = lambda x, fmt="%.2f": "-" if x is None else fmt % x
    print("| %s (%s) | %.1f | %s | %s ... %s | %s | %s |" % (r["kernel_class"], r["kernel"][:60], r["ms_per_step"], f(r["frac"]), f(r["hbm_frac_counter"]),
                                                        f(r["hbm_frac_counter_if_streaming"]), f(r["traffic_over_algorithmic"], "%.1f"), f(r["frac_of_random_request_wall"])))
print("kernel classes sum %.1f ms" % d["kernels_ms_sum_per_step"])
if d.get("e2e"):
    print("e2e:", {k: round(v, 1) for k, v in d["e2e"].items() if isinstance(v, float)})
if d.get("cpu_baseline"):
    c = d["cpu_baseline"]
    print("cpu: %.4f q/s on %d core; threads: %s; sasearch: %s" % (c["value"], c["cores"], c.get("threads_T", {}).get("queries_per_sec"), c.get("sasearch", {}).get("queries_per_sec")))
