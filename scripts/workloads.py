"""Synthetic workloads of the BASELINE configs other than the headline, shared by bench.py (config.secondary: every number measured
live in the driver-run process) and the stand-alone scripts (scripts/bench_motion.py, scripts/bench_nn_dense.py, scripts/bench_fit.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def timed(fn, iters, warmup=1):
    """Mean wall time of fn() in ms over ``iters`` calls after ``warmup`` calls (device synchronised on both sides)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def motion_stage2(model_root, S=256, T=61, start_t=30, device='cuda', **op_kw):
    """BASELINE config 4: GraspMotion-Opt stage 2 for S sequences x T frames (S = 256 -> 15,616 SMPL-X bodies per iteration).
    -> (op, run) with run(n) = n stage-2 iterations (fresh Adam per call, as the reference builds one per sequence)."""
    from saga_amd.fitting import load_grasp_indices
    from saga_amd.motion_fitting import FittingOP, convert_to_3D_rot, convert_to_6D_all, gen_body_mesh_v1
    from saga_amd.synthetic import make_object_cloud
    idx = load_grasp_indices()
    g = torch.Generator().manual_seed(0)
    op = FittingOP(dict(T=T, n_seq=S, start_t=start_t, gender='male', device=device, smplx_beta=0.5 * torch.randn(S, 10, generator=g).numpy(),
                        num_iter=[0, 1], markers_ids=idx['markers_f0_p5'], markers_ids_143=idx['markers_143'], body_model_path=model_root,
                        indices=idx, **op_kw))
    lam = torch.linspace(0, 1, T)[None, :, None]
    transl = torch.tensor([0.0, 0.4, 1.0]) + 0.3 * lam * torch.randn(S, 1, 3, generator=g)
    rot6 = convert_to_6D_all(torch.tensor([[0.0, 1.6, 3.14]]))[None].expand(S, T, 6).contiguous()
    other = 0.1 * torch.randn(S, 1, 80, generator=g) + 0.1 * lam * torch.randn(S, 1, 80, generator=g)
    op.init_param_T(transl, rot6, other, np.repeat(np.asarray(op.shape_t.cpu())[:, None], T, axis=1))
    with torch.no_grad():     # targets: the start motion's own markers, perturbed; the object next to the last frame's right hand
        p = convert_to_3D_rot(torch.cat([op.transl_opt_T, op.rot_6d_opt_T, op.shape_T, op.other_params_opt_T], -1).reshape(S * T, -1))
        v = gen_body_mesh_v1(p, op.smplx_model_batch, op.vposer_model)[0].reshape(S, T, -1, 3)
        markers_rec = v[:, :, op.markers_ids] + 0.01 * torch.randn(S, T, 79, 3, generator=g).to(v.device)
        markers_end = markers_rec[:, -1].clone()
        centre = v[:, -1, torch.from_numpy(idx['rhand']).to(v.device)].mean(1)
        del v, p
    vo, no, _, _ = make_object_cloud(S, seed=3)
    vo = torch.from_numpy(vo).float().to(centre.device)
    vo = (vo - vo.mean(1, keepdim=True) + centre[:, None])[:, None].expand(S, T - start_t, -1, 3).contiguous()
    no = torch.from_numpy(no).float().to(centre.device)[:, None].expand(S, T - start_t, -1, 3).contiguous()
    lbl = (torch.rand(S, T, 4, generator=g) > 0.4).float()
    c_obj, c_mk = torch.rand(S, 2048, generator=g), torch.rand(S, 143, generator=g)
    data = dict(markers_rec=markers_rec, markers_end=markers_end, rhand=idx['rhand'], lbl=lbl, c_obj=c_obj, c_mk=c_mk, no=no, vo=vo, idx=idx)

    def run(n):
        op.num_iter = [0, n]
        return op.fitting_stage2(markers_rec, markers_end, idx['rhand'], lbl, c_obj, c_mk, no, vo)

    op.bench_data = data
    return op, run


def dense_chamfer_clouds(batch=512, verts=10475, points=100000, device='cuda', seed=0):
    """BASELINE config 5: a body-sized cloud of 10,475 points against a dense scene cloud of 100,000 points per batch element."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn(batch, verts, 3, device=device, generator=g) * torch.tensor([0.3, 0.15, 0.85], device=device) + torch.tensor([0., 0., 0.9], device=device)
    y = torch.randn(batch, points, 3, device=device, generator=g) * 0.25 + torch.tensor([0.3, 0.1, 1.0], device=device)
    return x, y


def whole_fit(model_root, B=4096, iters=(300, 400, 500), device='cuda', **kw):
    """A whole GraspPose-Opt fit (opt_grasppose.py:180: 300 + 400 + 500 iterations) of B candidates -> (seconds, FittingOP)."""
    from saga_amd.fitting import FittingOP
    from saga_amd.synthetic import make_grasp_targets, make_object_cloud
    from saga_amd.vposer import VPoserDecoder
    vo, no, _, _ = make_object_cloud(B, n_points=2048, seed=0)
    mk, mc, oc = make_grasp_targets(B, vo, seed=4815)
    cu = lambda x: torch.from_numpy(x).to(device)
    args = (cu(mk), cu(oc), cu(mc), cu(vo), cu(no), 'male')
    op = FittingOP({'batch_size': B, 'device': device, 'body_model_path': model_root, 'vposer': VPoserDecoder(seed=0),
                    'genders': ('male',), 'num_iter': [2, 2, 2], 'record_history': False, **kw})
    op.fitting(*args)                                   # warm-up: models, workspace
    op.num_iter = list(iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, params, _ = op.fitting(*args)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, op, params
