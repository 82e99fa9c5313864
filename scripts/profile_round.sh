#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate PMC passes of the default bench command.
# usage: scripts/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/...
set -o pipefail
TAG=${1:-x}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err || exit 1
PB="python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" ; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $C -d $OUT/pmc$i -o p --output-format csv -- $PB > $OUT/bench_pmc$i.json 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.err; exit 1; }
done
python3 $PWD/scripts/summarize_pmc.py $OUT > /dev/null
tail -40 $OUT/summary.txt
