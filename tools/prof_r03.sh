#!/bin/bash
# Round 3's measurements of record, into gpurun_out/r03/ (copied into profiles/ afterwards):
#   default bench line (cpu baseline and end-to-end region), kernel statistics + idle gaps, FETCH_SIZE and WRITE_SIZE passes
#   (separate runs, --pmc alone), the C5 / C4 / C2 lines, the one-rank RCCL line, the two-rank rehearsal of the collective search,
#   SQ / TCP counter passes per kernel (tools/prof_sq.sh).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
rm -rf $O; mkdir -p $O
cd $R
B="python3 bench.py --no-cpu-baseline --no-e2e --no-dense-sa"
step() { date +"%T $1" >> $O/log; }
step start
python3 bench.py > $O/bench_C3_default.json 2> $O/bench_C3_default.err || exit 1
step "default line"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03_stats -o r03 -- $B --steps 2 --warmup 1 > $O/bench_C3_under_rocprof.json 2> $O/stats.err || exit 1
for f in $(find /tmp/r03_stats -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_bench_C3.csv; done
for f in $(find /tmp/r03_stats -name "*kernel_trace.csv"); do python3 tools/gap_analysis.py $f 200 > $O/gaps_C3.txt 2>&1; done
step "kernel stats"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/r03_fetch -o r03 -- $B --steps 1 --warmup 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
for f in $(find /tmp/r03_fetch -name "*counter_collection.csv"); do cp $f $O/pmc_FETCH_SIZE_bench_C3.csv; done
step "FETCH_SIZE pass"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/r03_write -o r03 -- $B --steps 1 --warmup 1 > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
for f in $(find /tmp/r03_write -name "*counter_collection.csv"); do cp $f $O/pmc_WRITE_SIZE_bench_C3.csv; done
step "WRITE_SIZE pass"
python3 bench.py --config C5 --steps 3 --warmup 1 > $O/bench_C5.json 2> $O/bench_C5.err || exit 1
step "C5 line"
python3 bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-dense-sa > $O/bench_C4.json 2> $O/bench_C4.err || exit 1
step "C4 line"
python3 bench.py --config C4 --sa-dens 1 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_C4_dense_sa.json 2> $O/bench_C4_dense_sa.err || exit 1
step "C4 line, suffix array resident"
python3 bench.py --config C2 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_C2.json 2> $O/bench_C2.err || exit 1
step "C2 line"
python3 bench.py --gpus 1 --dist-at-1 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-dense-sa > $O/bench_C3_one_rank_rccl.json 2> $O/bench_C3_one_rank_rccl.err || exit 1
step "one-rank RCCL line"
python3 bench.py --gpus 2 --single-device --backend gloo --scale 0.25 --workspace-gb 60 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-dense-sa > $O/rehearse_2ranks_one_device_C3x0.25.json 2> $O/rehearse_2.err || exit 1
step "two-rank rehearsal"
python3 bench.py --gpus 4 --single-device --backend gloo --scale 0.25 --workspace-gb 40 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-dense-sa > $O/rehearse_4ranks_one_device_C3x0.25.json 2> $O/rehearse_4.err || exit 1
step "four-rank rehearsal"
bash tools/prof_sq.sh C3 sq sq2 tcp
for P in sq sq2 tcp; do cp gpurun_out/prof_sq_C3/$P.txt $O/pmc_${P}_bench_C3.txt; done
step "SQ / TCP counter passes"
ls -la $O >> $O/log
