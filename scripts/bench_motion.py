#!/usr/bin/env python3
"""BASELINE config 4: GraspMotion-Opt stage 2 at S sequences x T = 61 frames (S = 256 -> 15,616 SMPL-X bodies per
iteration): marker L1 + end-pose + skating + contact / collision / hand-angle on frames 30..60 + regularisers, one
Adam step.  Synthetic targets (random smooth motions of the synthetic body model).  Times whole stage-2 iterations and
prints one JSON line (frame-iterations / s = S * T * iterations / s)."""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from saga_amd.fitting import load_grasp_indices
from saga_amd.motion_fitting import FittingOP, convert_to_6D_all
from saga_amd.synthetic import make_object_cloud, write_body_model

ap = argparse.ArgumentParser()
ap.add_argument('--sequences', type=int, default=256)
ap.add_argument('--frames', type=int, default=61)
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--warmup', type=int, default=2)
ap.add_argument('--dense-backward', action='store_true', help='dense [S*T,V,3] vertex gradient (the form the reference has)')
ap.add_argument('--full-forward', action='store_true', help='evaluate the whole mesh in the forward (reduced_forward=False)')
ap.add_argument('--autograd', action='store_true', help='the autograd composition of the operators instead of saga_motion_step (native_stage2=False)')
ap.add_argument('--mesh-faces', action='store_true', help='mesh-like synthetic model (id-local faces and weights)')
ap.add_argument('--whole-fit', action='store_true', help='time a whole two-stage fit (T x 100 stage-1 iterations, 300 stage-2 iterations)')
a = ap.parse_args()
S, T, start_t = a.sequences, a.frames, 30
tmp = tempfile.TemporaryDirectory()
write_body_model(tmp.name, gender='male', seed=7, mesh_faces=a.mesh_faces, local_weights=a.mesh_faces)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from workloads import motion_stage2
op, run_n = motion_stage2(tmp.name, S, T, start_t, sparse_backward=not a.dense_backward, reduced_forward=not a.full_forward,
                           native_stage2=not a.autograd)
dd = op.bench_data
idx, markers_rec, markers_end, lbl, c_obj, c_mk, no, vo = (dd[k] for k in ('idx', 'markers_rec', 'markers_end', 'lbl', 'c_obj', 'c_mk', 'no', 'vo'))
if a.whole_fit:
    op.num_iter = [100, 300]
    op.init_param_t()
    out = {}
    for name in ('warm-up', 'timed'):
        op.init_param_t()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr, ro, sh, ot, mk = op.fitting_stage1(markers_rec)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        op.init_param_T(tr.detach(), ro.detach(), ot.detach(), sh.detach())
        op.fitting_stage2(mk, markers_end, idx['rhand'], lbl, c_obj, c_mk, no, vo)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        out = {'stage1_s': round(t1 - t0, 3), 'stage2_s': round(t2 - t1, 3), 'total_s': round(t2 - t0, 3)}
        if name == 'warm-up':
            op.num_iter = [100, 300]
    print(json.dumps({'config': 'GraspMotion-Opt whole fit, %d sequences x %d frames: %d x 100 stage-1 + 300 stage-2 iterations' % (S, T, T),
                      'model': 'mesh-like faces, local weights' if a.mesh_faces else 'SURVEY synthetic (random faces and joints)', **out,
                      'sequences_per_s': round(S / out['total_s'], 1), 'data': 'synthetic'}))
    sys.exit(0)
run = lambda n: (setattr(op, 'num_iter', [0, n]), op.fitting_stage2(markers_rec, markers_end, idx['rhand'], lbl, c_obj, c_mk, no, vo))
run(a.warmup)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(a.iters)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
print(json.dumps({'config': 'GraspMotion-Opt stage 2, %d sequences x %d frames' % (S, T), 'bodies_per_iteration': S * T,
                  'iteration': 'saga_motion_step' if op._stage2_native_ok() else 'autograd composition', 'backward': 'dense' if a.dense_backward else 'loss vertices only',
                  'vertices_evaluated': (op._loss_model['bm'].num_verts if op._loss_model is not None else op.smplx_model_batch.num_verts),
                  'model': 'mesh-like faces, local weights' if a.mesh_faces else 'SURVEY synthetic (random faces and joints)',
                  'ms_per_iteration': round(dt * 1e3, 2), 'frame_iterations_per_s': round(S * T / dt, 1),
                  'note': 'each timed call builds a fresh Adam (as the reference does per sequence); data synthetic',
                  'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
