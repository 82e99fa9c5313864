# Runs on the GPU box: the overlapped-tile backward (default) against one tile per block (SAGA_GEMM_XT_BWD=0): bits, GPU tests, per-step time
set -o pipefail
timeout -k 10 300 python scripts/exp/step_bits.py > gpurun_out/bits_xt.txt 2>&1 || { tail -5 gpurun_out/bits_xt.txt; exit 1; }
diff scripts/exp/step_bits_reference.txt gpurun_out/bits_xt.txt && echo XT_BITS_IDENTICAL
timeout -k 10 900 python -u -m pytest tests/test_gpu_smplx.py tests/test_gpu_fullsize.py tests/test_gpu_fitting.py tests/test_zz_no_alloc.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/xt_gputest.log 2>&1 || { tail -20 gpurun_out/xt_gputest.log; exit 1; }
tail -2 gpurun_out/xt_gputest.log
for r in 1 2; do
  for v in 1 0; do
    SAGA_GEMM_XT_BWD=$v timeout -k 10 200 python bench.py --steps ${STEPS:-300} --warmup 20 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stages']; print('B=4096 XT_BWD=$v', d['ms_per_step'], 'fwd', s['blend_gemm_fwd']['ms'], 'bwd', s['blend_gemm_bwd']['ms'], 'frac', d['roofline']['frac'])" || exit 1
  done
done
