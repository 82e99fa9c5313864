"""Parity at BASELINE.json's full size (B = 4096 candidates x 2,048 object points), where the CPU oracle is too slow:
size-independent properties instead -- a candidate's iteration does not depend on the batch it sits in (bit for bit),
and the fused loss finds the same neighbours as the stand-alone exact chamfer scan on every one of the ~12 M queries."""
import numpy as np
import pytest
import torch

from saga_amd.synthetic import make_candidate_params, make_grasp_targets, make_object_cloud

pytestmark = pytest.mark.gpu
B_FULL, O = 4096, 2048


def _op(body_model_dir, B):
    from saga_amd.fitting import FittingOP
    from saga_amd.vposer import VPoserDecoder
    op = FittingOP({'batch_size': B, 'device': 'cuda', 'body_model_path': body_model_dir, 'vposer': VPoserDecoder(seed=0),
                    'genders': ('male',), 'verbose': False})
    op.bm = op.bm_male
    return op


def _inputs(B):
    vo, no, _, _ = make_object_cloud(B, n_points=O, seed=0)
    mk, mc, oc = make_grasp_targets(B, vo, seed=4815)
    return make_candidate_params(B, seed=1234, scale=0.3), dict(mk=mk, mc=mc, oc=oc, vo=vo, no=no)


@pytest.mark.parametrize('B_FULL', [4096, 1024])          # the metric's batch, and BASELINE config 2
def test_candidates_are_independent_at_full_batch(body_model_dir, B_FULL):
    var, d = _inputs(B_FULL)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    big = _op(body_model_dir, B_FULL)
    big.set_variables(var)
    prob = big.make_problem(cu(d['mk']), cu(d['vo']), cu(d['no']), cu(d['oc']), cu(d['mc']))
    terms = torch.zeros(B_FULL, 16, device='cuda')
    for _ in range(2):
        loss_big, mk_big = big.native_step(2, prob, terms)
    sel = np.array([0, 3, 130, 1001, 2047, 2048, 4000, 4095]) % B_FULL
    sel = np.unique(sel)
    small = _op(body_model_dir, len(sel))
    small.set_variables({k: v[sel] for k, v in var.items()})
    ps = small.make_problem(*(cu(d[k][sel]) for k in ('mk', 'vo', 'no', 'oc', 'mc')))
    ts = torch.zeros(len(sel), 16, device='cuda')
    for _ in range(2):
        loss_s, mk_s = small.native_step(2, ps, ts)
    idx = torch.from_numpy(sel).cuda()
    assert torch.equal(loss_big[idx], loss_s)
    assert torch.equal(mk_big[idx], mk_s)
    assert torch.equal(terms[idx], ts)
    assert torch.equal(big.vars[idx], small.vars)             # the Adam update too
    assert torch.isfinite(loss_big).all()


def test_fused_loss_neighbours_equal_exact_chamfer_at_full_batch(body_model_dir):
    import saga_amd.chamfer_distance as chd
    from saga_amd import _lib
    var, d = _inputs(B_FULL)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    op = _op(body_model_dir, B_FULL)
    op.set_variables(var)
    prob = op.make_problem(cu(d['mk']), cu(d['vo']), cu(d['no']), cu(d['oc']), cu(d['mc']))
    op.native_step(2, prob)
    H, M = op.tables.n_hand, op.tables.n_markers
    verts = op.workspace_view('vertices', O, (B_FULL, op.bm.num_verts, 3))
    hand = verts[:, op.rhand_verts].contiguous()
    markers = verts[:, op.markers_143].contiguous()
    obj = prob['obj']
    d1, d2, i1, i2 = chd.ChamferDistance(hand, obj)            # two-level scan (single-tile database)
    assert torch.equal(op.workspace_view('idx_h2o', O, (B_FULL, H), torch.int32), i1)
    assert torch.equal(op.workspace_view('idx_o2h', O, (B_FULL, O), torch.int32), i2)
    _, _, im, _ = chd.ChamferDistance(markers, obj)
    assert torch.equal(op.workspace_view('idx_m2o', O, (B_FULL, M), torch.int32), im)
    # and through the other code paths: a second LDS tile (pad the object cloud with far-away points), and the plain
    # 6-op exact scan
    far = torch.full((B_FULL, 64, 3), 1.0e3, device='cuda')
    _, _, i1x, _ = chd.ChamferDistance(hand, torch.cat([obj, far], 1))          # two tiles, minimum carried across
    assert torch.equal(i1x, i1)
    import os
    os.environ['SAGA_NN_EXACT_SCAN'] = '1'                                       # the plain 6-op exact scan kernel
    try:
        d1e, d2e, i1e, i2e = chd.ChamferDistance(hand, obj)
    finally:
        del os.environ['SAGA_NN_EXACT_SCAN']
    assert torch.equal(i1e, i1) and torch.equal(i2e, i2) and torch.equal(d1e, d1) and torch.equal(d2e, d2)
    # distances are the exact fp32 chain of the winner
    g = obj.gather(1, i1.long()[..., None].expand(-1, -1, 3))
    dx, dy, dz = (hand - g).unbind(-1)
    ref = torch.addcmul(torch.addcmul(dx * dx, dy, dy), dz, dz)      # fma(dz,dz,fma(dy,dy,dx*dx)) up to fusion
    assert torch.allclose(d1, ref, rtol=1e-6, atol=0)


def test_pose_opt_writes_reference_schema(body_model_dir, tmp_path):
    from saga_amd.fitting import FittingOP
    from saga_amd.opt_grasppose import pose_opt
    from saga_amd.vposer import VPoserDecoder
    n, k = 2, 3
    vo, no, go, tt = make_object_cloud(n, n_points=O, seed=1)
    mk, mc, oc = make_grasp_targets(n * k, np.repeat(vo, k, axis=0), seed=7)
    samples = dict(verts_object=vo, normal_object=no, global_orient_object=go, transf_transl=tt, markers_gen=mk,
                   object_contact_gen=oc, markers_contact_gen=mc)
    op = FittingOP({'batch_size': n * k, 'device': 'cuda', 'body_model_path': body_model_dir, 'vposer': VPoserDecoder(seed=0),
                    'genders': ('male',), 'num_iter': [2, 2, 2], 'verbose': False})
    out = pose_opt(op, samples, k, 'camera', 'male', save_dir=str(tmp_path))
    back = np.load(tmp_path / 'fitting_results.npz', allow_pickle=True)
    assert back['markers_fit'].shape == (n * k, 143, 3) and np.isfinite(back['markers_fit']).all()
    assert back['body'][()]['body_pose'].shape == (n * k, 63)
    assert np.array_equal(back['object'][()]['verts_object'][k], vo[1])          # candidate k belongs to object pose 1
    assert np.array_equal(out['markers'], mk)
    # gather=True (one rank: no process group needed): same rows + the per-candidate loss and the best random sample per object pose
    op2 = FittingOP({'batch_size': n * k, 'device': 'cuda', 'body_model_path': body_model_dir, 'vposer': VPoserDecoder(seed=0),
                     'genders': ('male',), 'num_iter': [2, 2, 2], 'verbose': False})
    out2 = pose_opt(op2, samples, k, 'camera', 'male', gather=True)
    for key in out['body']:
        assert np.array_equal(out2['body'][key], out['body'][key]), key
    assert np.array_equal(out2['markers_fit'], out['markers_fit'])
    assert out2['loss'].shape == (n * k,) and np.isfinite(out2['loss']).all()
    assert out2['best_of_n'].tolist() == [int(out2['loss'][:k].argmin()), k + int(out2['loss'][k:].argmin())]


def test_config5_full_mesh_chamfer_at_full_size():
    """BASELINE config 5 at its full size under pytest: X = 10,475 body vertices, Y = 100,000 object points, B = 512
    (5.36e11 point pairs per direction, ~56 ms).  The CPU oracle cannot follow at this size, so: (i) a slice of batch elements
    against the plain 6-operation exact scan kernel (a different code path, itself oracle-checked bit for bit at small sizes),
    (ii) size-independent properties on ALL outputs: every returned distance IS the fp32 fma-chain distance to the returned
    index, and no other point of a sampled subset is closer (or equally close with a lower index)."""
    import os
    import saga_amd.chamfer_distance as chd
    B, V, P = 512, 10475, 100000
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(B, V, 3, device='cuda', generator=g) * torch.tensor([0.3, 0.15, 0.85], device='cuda') + torch.tensor([0., 0., 0.9], device='cuda')
    y = torch.randn(B, P, 3, device='cuda', generator=g) * 0.25 + torch.tensor([0.3, 0.1, 1.0], device='cuda')
    d1, d2, i1, i2 = chd.ChamferDistance(x, y)
    torch.cuda.synchronize()
    assert d1.shape == (B, V) and d2.shape == (B, P) and i1.dtype == torch.int32
    assert int(i1.min()) >= 0 and int(i1.max()) < P and int(i2.min()) >= 0 and int(i2.max()) < V
    sl = slice(100, 104)
    os.environ['SAGA_NN_EXACT_SCAN'] = '1'
    try:
        r1, r2, j1, j2 = chd.ChamferDistance(x[sl].contiguous(), y[sl].contiguous())
    finally:
        del os.environ['SAGA_NN_EXACT_SCAN']
    assert torch.equal(j1, i1[sl]) and torch.equal(j2, i2[sl]) and torch.equal(r1, d1[sl]) and torch.equal(r2, d2[sl])

    def chain(a, b):                       # fma(dz,dz, fma(dy,dy, dx*dx)) up to fusion of the first product
        dx, dy, dz = (a - b).unbind(-1)
        return torch.addcmul(torch.addcmul(dx * dx, dy, dy), dz, dz)

    for b0 in range(0, B, 64):             # (ii) in slabs of 64 batch elements (memory)
        bs = slice(b0, b0 + 64)
        near = y[bs].gather(1, i1[bs].long()[..., None].expand(-1, -1, 3))
        assert torch.allclose(d1[bs], chain(x[bs], near), rtol=1e-6, atol=0)
        near2 = x[bs].gather(1, i2[bs].long()[..., None].expand(-1, -1, 3))
        assert torch.allclose(d2[bs], chain(y[bs], near2), rtol=1e-6, atol=0)
    # no sampled database point beats the returned neighbour: 256 random queries x 4,096 random database points per element
    qs = torch.randint(0, V, (256,), device='cuda', generator=g)
    ps = torch.randint(0, P, (4096,), device='cuda', generator=g)
    for b0 in range(0, B, 128):
        bs = slice(b0, b0 + 128)
        dd = ((x[bs][:, qs, None, :] - y[bs][:, None, ps, :]) ** 2).sum(-1)              # [128, 256, 4096]
        assert (dd.min(dim=2).values >= d1[bs][:, qs] * (1 - 1e-6)).all()


def test_config4_motion_stage2_iteration_at_full_size(body_model_dir):
    """BASELINE config 4 at its full size under pytest: ONE GraspMotion-Opt stage-2 iteration (saga_motion_step: objective,
    backward, Adam) of S = 256 sequences x T = 61 frames = 15,616 bodies.  The fp64 oracle cannot follow at this size (it is
    checked against this path at S = 2 in test_gpu_motion.py); here the size-independent property: a sequence's result does not
    depend on the batch it is in -- sequences 0, 100 and 255 of the S = 256 run are BIT-EQUAL (objective, every term, updated
    variables, Adam moments) to an S = 3 run on those three sequences alone."""
    import os
    import sys
    from saga_amd import _lib
    from saga_amd.motion_fitting import FittingOP
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts'))
    import workloads as wl
    S, T, start_t, pick = 256, 61, 30, [0, 100, 255]
    op, _ = wl.motion_stage2(body_model_dir, S=S, T=T, start_t=start_t)
    d = op.bench_data
    rh = torch.as_tensor(d['rhand'], dtype=torch.long, device='cuda')

    def one_iteration(o, sel):
        n = len(sel)
        args = (d['markers_rec'][sel].cuda().float(), d['markers_end'][sel].cuda().float(), rh, d['lbl'][sel].cuda().float(),
                d['c_obj'][sel].cuda().float(), d['c_mk'][sel].cuda().float(), d['no'][sel].cuda().float(), d['vo'][sel].cuda().float())
        o._make_picker(rh)
        ctx = o.stage2_native_setup(*args)
        packed = o.pack_stage2_vars()
        m, v = torch.zeros_like(packed), torch.zeros_like(packed)
        loss, terms = o.stage2_native_step(ctx, packed, m, v, adam_step=1, lr=0.01)
        torch.cuda.synchronize()
        r = lambda t: t.reshape(n, T, -1)
        return loss.clone(), terms.clone(), r(packed), r(m), r(v)

    full = one_iteration(op, list(range(S)))
    assert torch.isfinite(full[0]).all() and (full[1][:, 1:10] > 0).any(dim=0).all()          # every term is active at this size
    op3 = FittingOP(dict(T=T, n_seq=len(pick), start_t=start_t, gender='male', device='cuda', smplx_beta=op.shape_t[pick].cpu().numpy(),
                         num_iter=[0, 1], markers_ids=d['idx']['markers_f0_p5'], markers_ids_143=d['idx']['markers_143'],
                         body_model_path=body_model_dir, indices=d['idx']))
    op3.init_param_T(op.transl_opt_T.detach()[pick], op.rot_6d_opt_T.detach()[pick], op.other_params_opt_T.detach()[pick],
                     op.shape_T[pick].cpu().numpy())
    small = one_iteration(op3, pick)
    for name, a, b in zip(('loss', 'terms', 'variables', 'exp_avg', 'exp_avg_sq'), full, small):
        assert torch.equal(a[pick], b), name
    assert not torch.equal(full[2][pick], op3.pack_stage2_vars().reshape(len(pick), T, -1))      # the update moved the variables
