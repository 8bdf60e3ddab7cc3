#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv (developer tool).  usage: sq_summary.py file.csv [topN]"""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"^void ", "", r["Kernel_Name"]); name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"([\w:]+)(<[^(]*>)?\(", name)
    short = (m.group(1) + (m.group(2) or "")) if m else name
    short = short.replace("unsigned int", "u32").replace("unsigned long", "u64")[:44]
    agg[short][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[short][r["Counter_Name"]] += 1
cols = sorted({c for k in agg for c in agg[k]})
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
names = sorted(agg, key=lambda k: -max(agg[k].values()))[:top]
print("%-44s %5s " % ("kernel", "n") + " ".join("%12s" % c.replace("SQ_", "")[:12] for c in cols))
for n in names:
    print("%-44s %5d " % (n, max(cnt[n].values())) + " ".join("%12.4g" % agg[n][c] for c in cols))
