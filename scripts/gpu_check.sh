#!/bin/bash
# Standard GPU-box check: parity tests, then the bench line; every GPU step under its own time limit, chained.
# usage: scripts/gpu_check.sh TAG [pytest -k expression]
set -o pipefail
TAG=${1:-run}
KEXPR=${2:-}
mkdir -p gpurun_out
if [ -n "$KEXPR" ]; then
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "$KEXPR" > gpurun_out/test_$TAG.log 2>&1
else
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/test_$TAG.log 2>&1
fi
rc=$?
tail -15 gpurun_out/test_$TAG.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.log 2>&1 || { tail -20 gpurun_out/bench_$TAG.log; exit 1; }
tail -1 gpurun_out/bench_$TAG.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
for k,v in d['config']['kernels'].items(): print('  %-16s %.4f ms x%g' % (k, v['ms_per_step'], v['launches_per_step']))
print('roofline', d['roofline'])
print('cpu', d.get('cpu_baseline'))
"
