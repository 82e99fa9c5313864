#!/bin/bash
# Runs on the GPU box: the bench lines a round's DESIGN.md quotes, one JSON line per file.  usage: scripts/collect_round.sh TAG
set -o pipefail
TAG=${1:-x}; OUT=gpurun_out/lines_$TAG; mkdir -p $OUT
run() { local name=$1; shift; timeout -k 10 400 "$@" 2>$OUT/$name.err | tail -1 > $OUT/$name.json || { echo "$name failed"; tail -3 $OUT/$name.err; exit 1; }; echo "$name $(cut -c1-160 $OUT/$name.json)"; }
run bench python3 bench.py
run bench_batch1024 python3 bench.py --batch 1024 --no-cpu-baseline
run bench_sparse_backward python3 bench.py --sparse-backward --no-cpu-baseline
run bench_local_weights python3 bench.py --local-weights --no-cpu-baseline
run bench_mesh_faces python3 bench.py --mesh-faces --no-cpu-baseline
run bench_mesh_faces_local_weights python3 bench.py --mesh-faces --local-weights --no-cpu-baseline
run bench_autograd python3 bench.py --autograd --no-cpu-baseline
run config4_motion python3 scripts/bench_motion.py
run config4_motion_mesh_like python3 scripts/bench_motion.py --mesh-faces
run config4_motion_full_forward python3 scripts/bench_motion.py --full-forward
run config4_motion_whole_fit python3 scripts/bench_motion.py --whole-fit
run whole_fit python3 scripts/bench_fit.py
run config4_motion_autograd python3 scripts/bench_motion.py --autograd
run config4_motion_1x61 python3 scripts/bench_motion.py --sequences 1 --iters 200
run config5_dense_chamfer python3 scripts/bench_nn_dense.py --check
