#!/bin/bash
# Runs on the GPU box: kernel trace of bench.py with extra flags; keeps the bench line and the per-kernel stats.
# usage: scripts/ktrace_bench.sh TAG [bench flags]   -> gpurun_out/kt_TAG/{out.log, **/t_kernel_stats.csv}
set -o pipefail
TAG=${1:-x}; shift
OUT=$PWD/gpurun_out/kt_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $PWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/out.log 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
tail -n 1 $OUT/out.log | cut -c1-300
