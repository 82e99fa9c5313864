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
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--warmup', type=int, default=2)
ap.add_argument('--dense-backward', action='store_true', help='dense [S*T,V,3] vertex gradient (the form the reference has)')
ap.add_argument('--full-forward', action='store_true', help='evaluate the whole mesh in the forward (reduced_forward=False)')
ap.add_argument('--mesh-faces', action='store_true', help='mesh-like synthetic model (id-local faces and weights)')
ap.add_argument('--whole-fit', action='store_true', help='time a whole two-stage fit (T x 100 stage-1 iterations, 300 stage-2 iterations)')
a = ap.parse_args()
S, T, start_t = a.sequences, a.frames, 30
tmp = tempfile.TemporaryDirectory()
write_body_model(tmp.name, gender='male', seed=7, mesh_faces=a.mesh_faces, local_weights=a.mesh_faces)
idx = load_grasp_indices()
g = torch.Generator().manual_seed(0)
op = FittingOP(dict(T=T, n_seq=S, start_t=start_t, gender='male', device='cuda', smplx_beta=0.5 * torch.randn(S, 10, generator=g).numpy(),
                    num_iter=[0, 1], markers_ids=idx['markers_f0_p5'], markers_ids_143=idx['markers_143'], body_model_path=tmp.name,
                    indices=idx, sparse_backward=not a.dense_backward, reduced_forward=not a.full_forward))
lam = torch.linspace(0, 1, T)[None, :, None]
transl = torch.tensor([0.0, 0.4, 1.0]) + 0.3 * lam * torch.randn(S, 1, 3, generator=g)
rot6 = convert_to_6D_all(torch.tensor([[0.0, 1.6, 3.14]]))[None].expand(S, T, 6).contiguous()
other = 0.1 * torch.randn(S, 1, 80, generator=g) + 0.1 * lam * torch.randn(S, 1, 80, generator=g)
op.init_param_T(transl, rot6, other, np.repeat(np.asarray(op.shape_t.cpu())[:, None], T, axis=1))
with torch.no_grad():     # targets: the start motion's own markers, perturbed; the object next to the last frame's right hand
    from saga_amd.motion_fitting import convert_to_3D_rot, gen_body_mesh_v1
    p = convert_to_3D_rot(torch.cat([op.transl_opt_T, op.rot_6d_opt_T, op.shape_T, op.other_params_opt_T], -1).reshape(S * T, -1))
    v = gen_body_mesh_v1(p, op.smplx_model_batch, op.vposer_model)[0].reshape(S, T, -1, 3)
    markers_rec = v[:, :, op.markers_ids] + 0.01 * torch.randn(S, T, 79, 3, generator=g).cuda()
    markers_end = markers_rec[:, -1].clone()
    centre = v[:, -1, torch.from_numpy(idx['rhand']).cuda()].mean(1)
    del v, p
vo, no, _, _ = make_object_cloud(S, seed=3)
vo = torch.from_numpy(vo).float().cuda()
vo = (vo - vo.mean(1, keepdim=True) + centre[:, None])[:, None].expand(S, T - start_t, -1, 3).contiguous()
no = torch.from_numpy(no).float().cuda()[:, None].expand(S, T - start_t, -1, 3).contiguous()
lbl = (torch.rand(S, T, 4, generator=g) > 0.4).float()
c_obj, c_mk = torch.rand(S, 2048, generator=g), torch.rand(S, 143, generator=g)
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
print(json.dumps({'config': 'GraspMotion-Opt stage 2, %d sequences x %d frames' % (S, T), 'bodies_per_iteration': S * T, 'backward': 'dense' if a.dense_backward else 'loss vertices only',
                  'vertices_evaluated': (op._loss_model['bm'].num_verts if op._loss_model is not None else op.smplx_model_batch.num_verts),
                  'model': 'mesh-like faces, local weights' if a.mesh_faces else 'SURVEY synthetic (random faces and joints)',
                  'ms_per_iteration': round(dt * 1e3, 2), 'frame_iterations_per_s': round(S * T / dt, 1),
                  'note': 'each timed call builds a fresh Adam (as the reference does per sequence); data synthetic',
                  'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
