#!/bin/bash
# Runs on the GPU box: kernel trace of an arbitrary python script; prints the per-kernel stats.  usage: scripts/ktrace_cmd.sh TAG script.py [args]
set -o pipefail
TAG=${1:-x}; shift
OUT=$PWD/gpurun_out/kt_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 "$@" > $OUT/out.log 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
tail -1 $OUT/out.log
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6
print('all kernels: %.1f ms' % tot)
for r in rows[:44]:
    print('%8.2f ms %6s calls %9.1f us  %s' % (float(r['TotalDurationNs'])/1e6, r['Calls'], float(r['AverageNs'])/1e3, r['Name'][:90]))
PY
