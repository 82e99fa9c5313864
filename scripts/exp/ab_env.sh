#!/bin/bash
# Same-box A/B of bench.py under environment settings: scripts/exp/ab_env.sh "VAR=1" "" "VAR=1" ""  (each argument one run)
for E in "$@"; do
  env $E timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config']['kernels']
print('%-28s %.4f ms | gemm %.3f/%.3f skin %.3f/%.3f loss %.3f/%.3f' % (sys.argv[1] or '(default)', d['ms_per_step'], k['blend_gemm_fwd']['ms_per_step'], k['blend_gemm_bwd']['ms_per_step'], k['skin_fwd']['ms_per_step'], k['skin_bwd']['ms_per_step'], k['grasp_loss_fwd']['ms_per_step'], k['grasp_loss_bwd']['ms_per_step']))" "$E"
done
