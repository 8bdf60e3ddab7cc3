#!/bin/bash
# The round's measurements of record, into gpurun_out/r02/ (copied into profiles/ afterwards):
#   default bench line (cpu baseline, end-to-end and strong-scaling regions as they apply at N = 1), kernel statistics, FETCH_SIZE and
#   WRITE_SIZE passes (separate runs, --pmc alone), the C5 line.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O
cd $R
B="python3 bench.py --no-cpu-baseline --no-e2e"
step() { date +"%T $1" >> $O/log; }
step start
python3 bench.py > $O/bench_C3_default.json 2> $O/bench_C3_default.err || exit 1
step "default line"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r02_stats -o r02 -- $B --steps 2 --warmup 1 > $O/bench_C3_under_rocprof.json 2> $O/stats.err || exit 1
for f in $(find /tmp/r02_stats -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_bench_C3.csv; done
for f in $(find /tmp/r02_stats -name "*kernel_trace.csv"); do python3 tools/gap_analysis.py $f 200 > $O/gaps_C3.txt 2>&1; done
step "kernel stats"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/r02_fetch -o r02 -- $B --steps 1 --warmup 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
for f in $(find /tmp/r02_fetch -name "*counter_collection.csv"); do cp $f $O/pmc_FETCH_SIZE_bench_C3.csv; done
step "FETCH_SIZE pass"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/r02_write -o r02 -- $B --steps 1 --warmup 1 > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
for f in $(find /tmp/r02_write -name "*counter_collection.csv"); do cp $f $O/pmc_WRITE_SIZE_bench_C3.csv; done
step "WRITE_SIZE pass"
python3 bench.py --config C5 --steps 3 --warmup 1 > $O/bench_C5.json 2> $O/bench_C5.err || exit 1
step "C5 line"
ls -la $O >> $O/log
