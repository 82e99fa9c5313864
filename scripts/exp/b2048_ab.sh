# Runs on the GPU box: the persistent blend kernels from fewer tiles per block on (SAGA_GEMM_XT=n) at smaller batches
set -o pipefail
for cfg in "768 3" "768 2" "1024 3" "1024 2" "1024 1" "768 3" "768 2" "1024 3" "1024 2" "1024 1" "512 3" "512 1" "512 3" "512 1"; do
set -- $cfg
SAGA_GEMM_XT=$2 timeout -k 10 200 python bench.py --steps 200 --warmup 10 --batch $1 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stages']; print('B=$1 XT=$2', d['ms_per_step'], 'fwd', s['blend_gemm_fwd']['ms'], 'bwd', s['blend_gemm_bwd']['ms'])" || exit 1
done
