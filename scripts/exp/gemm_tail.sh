# Runs on the GPU box: forward / backward blend GEMM launch time against the batch size (tiles), both forward launch forms: the intercept
# of the line is what a launch loses to its start-up and end-game
set -o pipefail
for xt in 3 0; do
  for b in 1024 2048 4096 8192; do
    SAGA_GEMM_XT=$xt timeout -k 10 300 python bench.py --steps 40 --warmup 5 --batch $b --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stages']; print('XT=$xt B=$b fwd', s['blend_gemm_fwd']['ms'], 'bwd', s['blend_gemm_bwd']['ms'], 'step', d['ms_per_step'])" || exit 1
  done
done
