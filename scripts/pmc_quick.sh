#!/bin/bash
# Runs on the GPU box: two PMC passes (instruction mix + LDS / waits) of a short bench run, summarised for the kernels whose
# name contains PATTERN.  usage: scripts/pmc_quick.sh TAG PATTERN [bench flags]   -> gpurun_out/pq_TAG/summary.txt
set -o pipefail
TAG=$1; PAT=$2; shift 2
OUT=$PWD/gpurun_out/pq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PB="python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VMEM" ; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $C -d $OUT/pmc$i -o p --output-format csv -- $PB > $OUT/bench_pmc$i.json 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.err; exit 1; }
done
python3 $PWD/scripts/summarize_pmc.py $OUT > /dev/null
awk -v pat="$PAT" '/^[a-zA-Z(]/ {show = index($0, pat) > 0} show' $OUT/summary.txt
