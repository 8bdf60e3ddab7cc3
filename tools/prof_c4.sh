#!/bin/bash
# kernel statistics of one configuration (default C4) into gpurun_out/<tag>_kernel_stats_<config>.csv.  usage: tools/prof_c4.sh [config] [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=${1:-C4}
T=${2:-r04}
cd $R
rm -rf /tmp/pc4
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc4 -o pc4 -- python3 bench.py --config $C --no-cpu-baseline --no-e2e --no-dense-sa --other-configs= --steps 2 --warmup 1 > gpurun_out/${T}_bench_${C}_under_rocprof.json 2> gpurun_out/${T}_prof_${C}.err || exit 1
for f in $(find /tmp/pc4 -name "*kernel_stats.csv"); do cp $f gpurun_out/${T}_kernel_stats_bench_${C}.csv; done
