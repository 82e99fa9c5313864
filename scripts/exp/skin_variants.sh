#!/bin/bash
# skinning kernels on random vs mesh-like weight tables, sparse (default) vs dense-MFMA dA reduction
set -o pipefail
mkdir -p gpurun_out
for w in "" "--local-weights"; do for mf in 0 1; do
  if [ $mf = 1 ]; then export SAGA_DA_MFMA=1; else unset SAGA_DA_MFMA; fi
  timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $w > gpurun_out/skinvar.log 2>&1 || { tail -5 gpurun_out/skinvar.log; exit 1; }
  tail -1 gpurun_out/skinvar.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['config']['kernels']; print('weights[$w] mfma=$mf  ms/step', d['ms_per_step'], 'skin_bwd', k['skin_bwd']['ms_per_step'], 'skin_fwd', k['skin_fwd']['ms_per_step'])"
done; done
