#!/bin/bash
# Runs on the GPU box: the default bench line (no secondary / cpu legs) with the tree's library, or SAGA_HIP_LIB=...; prints ms/step + per-stage ms
set -o pipefail
TAG=${1:-x}
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_$TAG.json").read().strip().splitlines()[-1])
print("$TAG", d["ms_per_step"], "ms/step  roofline frac", d["roofline"]["frac"])
st = d["config"].get("stages", {})
for k, v in st.items():
    print("   ", k, v if not isinstance(v, dict) else {a: b for a, b in v.items() if a in ("ms", "achieved", "frac")})
PY
