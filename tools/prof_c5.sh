#!/bin/bash
# kernel statistics + a few SQ counters of a C5 batch (rrr-63 index): where the sweep's time goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_c5
rm -rf $O; mkdir -p $O
cd $R
B="python3 bench.py --config C5 --no-cpu-baseline --no-e2e --no-dense-sa --no-strong"
date +"%T start" >> $O/log
$B --steps 2 --warmup 1 > $O/bench_plain.json 2>$O/bench_plain.err || exit 1
date +"%T plain done" >> $O/log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5stats -o c5 -- $B --steps 2 --warmup 1 > $O/bench_stats.json 2>$O/bench_stats.err || exit 1
date +"%T stats done" >> $O/log
for f in $(find /tmp/c5stats -name "*kernel_stats.csv"); do cp $f $O/; done; ls -la /tmp/c5stats >> $O/log
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/c5pmc1 -o c5 -- $B --steps 1 --warmup 1 > $O/bench_pmc1.json 2>$O/bench_pmc1.err || exit 1
date +"%T pmc1 done" >> $O/log
ls -la /tmp/c5pmc1 >> $O/log
python3 - <<'PY' >> $O/pmc1_sweep.txt
import csv,glob,collections
for f in glob.glob("/tmp/c5pmc1/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:60]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in sorted(acc.items(), key=lambda kv:-kv[1].get('SQ_BUSY_CYCLES',0))[:12]:
        print(k, dict(v))
PY
date +"%T end" >> $O/log
