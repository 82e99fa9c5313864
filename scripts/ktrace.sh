#!/bin/bash
# Runs on the GPU box: kernel trace of a short bench run; prints the per-kernel stats.  usage: scripts/ktrace.sh TAG [bench args]
set -o pipefail
TAG=${1:-x}; shift
OUT=$PWD/gpurun_out/kt_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $PWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]:
    print('%-70s calls %5s avg_us %9.1f pct %5s' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
