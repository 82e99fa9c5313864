#!/bin/bash
# Runs on the GPU box: the default bench line for 300 steps with the library in SAGA_HIP_LIB (or the tree's), while a background loop
# samples the card's power and shader clock (rocm-smi) every ~0.3 s: is the steady state of the loop power-limited?
set -o pipefail
TAG=${1:-x}
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks --csv 2>/dev/null | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/power_$TAG.txt &
SMI=$!
timeout -k 10 300 python bench.py --steps 1500 --warmup 20 --no-secondary --no-cpu-baseline > gpurun_out/bench_p_$TAG.json 2> gpurun_out/bench_p_$TAG.err || { tail -5 gpurun_out/bench_p_$TAG.err; kill $SMI; exit 1; }
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_p_$TAG.json").read().strip().splitlines()[-1])
print("$TAG", d["ms_per_step"], "ms/step over", d["steps"], "steps")
PY
