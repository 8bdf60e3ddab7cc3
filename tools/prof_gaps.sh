#!/bin/bash
# kernel trace of one bench configuration + the idle-gap report (developer tool): prof_gaps.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
[ -n "$TAG" ] || { echo "usage: prof_gaps.sh <tag> [bench args...]"; exit 2; }
O=$R/gpurun_out/$TAG
rm -rf $O /tmp/$TAG; mkdir -p $O
cd $R
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$TAG -o $TAG -- python3 bench.py --no-cpu-baseline --no-e2e --no-dense-sa --steps 2 --warmup 1 "$@" > $O/bench.json 2> $O/bench.err || exit 1
for f in $(find /tmp/$TAG -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; done
for f in $(find /tmp/$TAG -name "*kernel_trace.csv"); do python3 tools/gap_analysis.py $f 150 sweep_step_kernel trail_resolve_kernel radix_sort_onesweep join_link_kernel > $O/gaps.txt 2>&1; done
