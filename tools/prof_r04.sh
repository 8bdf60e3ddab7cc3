#!/bin/bash
# Round 4's measurements of record, into gpurun_out/r04/ (copied into profiles/ afterwards):
#   default bench line (cpu baselines, end-to-end region, other_configs C2 / C5 / C4), kernel statistics + idle gaps, FETCH_SIZE and
#   WRITE_SIZE passes (separate runs, --pmc alone), the one-rank RCCL line, two- and four-rank rehearsals of the collective search on one
#   device (gloo; pairwise needed-only exchange and, for comparison, the all-gather), the integer-alphabet / text-order lines
#   (tools/int_bench.py), SQ / TCP counter passes per kernel (tools/prof_sq.sh).   usage: tools/prof_r04.sh <commit>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
COMMIT=${1:-unknown}
rm -rf $O; mkdir -p $O
cd $R
B="python3 bench.py --no-cpu-baseline --no-e2e --no-dense-sa --other-configs="
step() { date +"%T $1" >> $O/log; }
step start
python3 bench.py > $O/bench_C3_default.json 2> $O/bench_C3_default.err || exit 1
step "default line"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_stats -o r04 -- $B --steps 2 --warmup 1 > $O/bench_C3_under_rocprof.json 2> $O/stats.err || exit 1
for f in $(find /tmp/r04_stats -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_bench_C3.csv; done
for f in $(find /tmp/r04_stats -name "*kernel_trace.csv"); do python3 tools/gap_analysis.py $f 200 > $O/gaps_C3.txt 2>&1; done
step "kernel stats"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/r04_fetch -o r04 -- $B --steps 1 --warmup 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
for f in $(find /tmp/r04_fetch -name "*counter_collection.csv"); do cp $f $O/pmc_FETCH_SIZE_bench_C3.csv; done
step "FETCH_SIZE pass"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/r04_write -o r04 -- $B --steps 1 --warmup 1 > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
for f in $(find /tmp/r04_write -name "*counter_collection.csv"); do cp $f $O/pmc_WRITE_SIZE_bench_C3.csv; done
step "WRITE_SIZE pass"
python3 tools/pmc_summary.py $O/pmc_FETCH_SIZE_bench_C3.csv $O/pmc_WRITE_SIZE_bench_C3.csv r04 2 $COMMIT > $O/pmc_summary.txt 2>&1 && cp profiles/pmc_traffic.json $O/pmc_traffic.json
step "pmc summary"
python3 bench.py --gpus 1 --dist-at-1 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-dense-sa --other-configs= > $O/bench_C3_one_rank_rccl.json 2> $O/bench_C3_one_rank_rccl.err || exit 1
step "one-rank RCCL line"
for X in needed allgather; do
python3 bench.py --gpus 2 --single-device --backend gloo --scale 0.25 --workspace-gb 60 --steps 2 --warmup 1 --exchange $X --no-cpu-baseline --no-e2e --no-dense-sa > $O/rehearse_2ranks_one_device_C3x0.25_$X.json 2> $O/rehearse_2_$X.err || exit 1
step "two-rank rehearsal ($X)"
python3 bench.py --gpus 4 --single-device --backend gloo --scale 0.25 --workspace-gb 40 --steps 2 --warmup 1 --exchange $X --no-cpu-baseline --no-e2e --no-dense-sa > $O/rehearse_4ranks_one_device_C3x0.25_$X.json 2> $O/rehearse_4_$X.err || exit 1
step "four-rank rehearsal ($X)"
done
python3 tools/int_bench.py 27 3 > $O/int_bench.jsonl 2> $O/int_bench.err || exit 1
step "integer alphabet / text order"
VLG_RESOLVE_STATS=1 $B --steps 1 --warmup 0 > /dev/null 2> $O/resolve_stats.err; grep "vlg resolve" $O/resolve_stats.err > $O/resolve_stats.txt
step "resolve hop statistics"
bash tools/prof_sq.sh C3 sq sq2 tcp
for P in sq sq2 tcp; do cp gpurun_out/prof_sq_C3/$P.txt $O/pmc_${P}_bench_C3.txt; done
step "SQ / TCP counter passes"
bash tools/prof_c4.sh C4 r04 && cp gpurun_out/r04_kernel_stats_bench_C4.csv $O/kernel_stats_bench_C4.csv && cp gpurun_out/r04_bench_C4_under_rocprof.json $O/bench_C4_under_rocprof.json
step "C4 kernel statistics"
ls -la $O >> $O/log
