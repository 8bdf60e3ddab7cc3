#!/bin/bash
# SQ and TA/TCP counters per kernel for one bench config (developer tool): which kernels are issue-bound, which wait on the vector memory path.
# usage (on the GPU box): tools/prof_sq.sh [C3|C5|...] [sq|ta|tcp ...]   -> gpurun_out/prof_sq_<cfg>/<pass>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=${1:-C3}
shift
PASSES=${@:-sq ta}
O=$R/gpurun_out/prof_sq_$C
mkdir -p $O
cd $R
B="python3 bench.py --config $C --no-cpu-baseline --no-e2e --no-dense-sa --no-strong"
for P in $PASSES; do
  case $P in
    sq)  CTR="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY";;
    sq2) CTR="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM";;
    ta)  CTR="TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum";;
    tcp) CTR="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum";;
    *) echo "unknown pass $P"; exit 1;;
  esac
  date +"%T $P start" >> $O/log
  rm -rf /tmp/sq_$P
  timeout -k 10 400 rocprofv3 --pmc $CTR --output-format csv -d /tmp/sq_$P -o $P -- $B --steps 1 --warmup 1 > $O/bench_$P.json 2>$O/bench_$P.err || { grep -v "^W2026\|^    @" $O/bench_$P.err | tail -4; echo "$P failed" >> $O/log; continue; }
  for f in $(find /tmp/sq_$P -name "*counter_collection.csv"); do python3 tools/sq_summary.py $f 24 > $O/$P.txt; done
  date +"%T $P done" >> $O/log
done
