"""GraspMotion-Opt stage-2 geometry losses (MotionFill/models/fittingop.py:132-280) on the HIP operators, frames as the
batch (BASELINE config 4: 61-frame sequences), against the CPU restatement oracle/motion_ref.py.  Bar: losses within
1e-4 relative; gradients w.r.t. the SMPL-X pose parameters within 2e-3."""
import numpy as np
import pytest
import torch

from oracle import motion_ref
from saga_amd import motion_losses as ml
from saga_amd.fitting import load_grasp_indices
from saga_amd.synthetic import make_object_cloud
from tests.helpers import load_oracle_model, make_hip_model, random_body_inputs, rel_err, to_cuda

pytestmark = pytest.mark.gpu
T, START = 61, 30


def _sequence(seed=0):
    """A smooth 61-frame motion: linear blend between two random poses, ending next to the object."""
    a, b = random_body_inputs(1, seed=seed, scale=0.3), random_body_inputs(1, seed=seed + 1, scale=0.3)
    s = torch.linspace(0, 1, T, dtype=torch.float64)[:, None]
    inp = {k: (1 - s) * a[k] + s * b[k] for k in a}
    inp['transl'] = inp['transl'] * 0.05 + torch.tensor([0.0, 0.0, 0.05])
    return inp


def test_motion_losses_match_oracle(body_model_dir):
    idx = load_grasp_indices()
    groups = ml.foot_groups(idx)
    mk = torch.from_numpy(idx['markers_143']).long()
    rh = torch.from_numpy(idx['rhand']).long()
    inp = _sequence()
    bm_ref = load_oracle_model(body_model_dir, T, dtype=torch.float64)
    bm = make_hip_model(body_model_dir, T)
    vo, no, _, _ = make_object_cloud(1, seed=3)
    g = torch.Generator().manual_seed(5)
    contact_object = torch.rand(1, 2048, generator=g, dtype=torch.float64)
    contact_markers = torch.rand(1, 143, generator=g, dtype=torch.float64)
    lbl = (torch.rand(T, 4, generator=g) > 0.4).double()

    # ---- oracle (float64, CPU) ----
    pr = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    vr = bm_ref(**pr)['vertices']
    # put the object where the right hand is at the end of the motion so that contact / collision terms are active
    centre = vr[-1, rh].mean(0).detach()
    obj = torch.from_numpy(vo[0]).double() - torch.from_numpy(vo[0]).double().mean(0) + centre
    obj[:22] = vr[-1, mk[-22:]].detach() + 0.004          # object points 7 mm from the palm markers of the last frame (angle term)
    vo_T = obj[None].expand(T - START, -1, -1).contiguous()
    no_T = torch.from_numpy(no[0]).double()[None].expand(T - START, -1, -1).contiguous()
    lc_r, st_r = motion_ref.contact_loss(torch.from_numpy(bm_ref.faces.astype(np.int64)), START, vr, mk, rh, contact_object,
                                         contact_markers, no_T, vo_T, n_frames=T)
    lcol_r = motion_ref.collision_loss(st_r)
    lsk_r = motion_ref.skating_loss(vr, lbl, groups)
    lang_r = motion_ref.angle_loss(st_r, START, n_frames=T)
    (lc_r + lcol_r + lsk_r + lang_r).backward()

    # ---- HIP ----
    ph = to_cuda(inp, requires_grad=True)
    vh = bm(return_verts=True, **ph).vertices
    c = lambda t: t.float().cuda()
    lc, st = ml.contact_loss(bm, START, vh, mk.cuda(), rh.cuda(), c(contact_object), c(contact_markers), c(no_T), c(vo_T), n_frames=T)
    lcol = ml.collision_loss(st)
    lsk = ml.skating_loss(vh, c(lbl), groups)
    lang = ml.angle_loss(st, START, T)
    (lc + lcol + lsk + lang).backward()

    assert lc_r.item() > 0 and lsk_r.item() > 0 and lang_r.item() > 0          # the terms are exercised
    for name, a, b in [('contact', lc, lc_r), ('collision', lcol, lcol_r), ('skating', lsk, lsk_r), ('angle', lang, lang_r)]:
        np.testing.assert_allclose(a.item(), b.item(), rtol=1e-4, atol=1e-9, err_msg=name)
    for k in ('body_pose', 'global_orient', 'transl', 'right_hand_pose'):
        assert rel_err(ph[k].grad.cpu().numpy(), pr[k].grad.numpy()) < 2e-3, k


def test_frames_as_batch_at_config4_size(body_model_dir):
    """B = 256 sequences x 61 frames = 15,616 SMPL-X evaluations in one call (config 4): equal to 61-frame calls."""
    n = 256 * T
    inp = random_body_inputs(n, seed=11, scale=0.3)
    bm_big = make_hip_model(body_model_dir, n)
    bm_small = make_hip_model(body_model_dir, T)
    ph = to_cuda(inp)
    with torch.no_grad():
        big = bm_big(return_verts=True, **ph)
        small = bm_small(return_verts=True, **{k: v[5 * T:6 * T] for k, v in ph.items()})
    assert torch.equal(big.vertices[5 * T:6 * T], small.vertices) and torch.equal(big.joints[5 * T:6 * T], small.joints)


def test_row_picker_matches_index_select():
    """saga_pick_sets_fwd/bwd: overlapping and repeated sets out of one cloud, some on a window of the frame axis, == one
    index_select (+ slice) per set; the backward is the sum of their backwards in the documented order (sets in order, picks of
    a row in ascending position), bit for bit."""
    from saga_amd.geometry import RowPicker
    g = torch.Generator().manual_seed(3)
    U, S, T = 301, 3, 4
    sets = {'a': torch.randperm(U, generator=g)[:79], 'b': (torch.arange(40, 200), 2), 'c': torch.tensor([5, 5, 7, 300, 0]),
            'd': (torch.arange(U), 3)}
    picker = RowPicker(sets, U)
    src = torch.randn(S, T, U, 3, generator=g).cuda().requires_grad_(True)
    ref_src = src.detach().clone().requires_grad_(True)
    out = picker(src)
    plain = {k: (v if isinstance(v, tuple) else (v, 0)) for k, v in sets.items()}
    wts = {k: torch.randn(S, T - t0, len(v), 3, generator=g).cuda() for k, (v, t0) in plain.items()}
    loss = sum((out[k] * wts[k]).sum() for k in sets)
    loss_ref = 0
    for k, (v, t0) in plain.items():
        ref = ref_src[:, t0:].index_select(2, v.cuda())
        assert out[k].is_contiguous() and torch.equal(out[k], ref), k
        loss_ref = loss_ref + (ref * wts[k]).sum()
    loss.backward()
    loss_ref.backward()
    np.testing.assert_allclose(src.grad.cpu().numpy(), ref_src.grad.cpu().numpy(), rtol=1e-6, atol=1e-6)
    out2 = picker(src)
    g2 = torch.autograd.grad(sum((out2[k] * wts[k]).sum() for k in sets), src)[0]
    assert torch.equal(g2, src.grad)                                   # deterministic
    exp = torch.zeros_like(g2)
    for k, (v, t0) in plain.items():                                   # the documented order
        for i in range(len(v)):
            exp[:, t0:, v[i]] += wts[k][:, :, i]
    assert torch.equal(g2, exp)
    out3 = picker(src)                                                 # a set nobody differentiates gets no gradient work
    g3 = torch.autograd.grad((out3['c'] * wts['c']).sum(), src)[0]
    exp_c = torch.zeros_like(g3)
    for i, v in enumerate(sets['c']):
        exp_c[:, :, v] += wts['c'][:, :, i]
    assert torch.equal(g3, exp_c)
    with pytest.raises(ValueError):
        RowPicker({'x': torch.tensor([U])}, U)
    with pytest.raises(ValueError):
        picker(torch.zeros(2, 4, U + 1, 3).cuda())
    with pytest.raises(ValueError):
        picker(torch.zeros(2, 3, U, 3).cuda())                        # set d starts at frame 3


def test_round2_operators_accept_empty_batches(body_model_dir):
    """Zero sequences / bodies are a no-op for the row picks, the stage-2 terms, the parameter maps and the windowed normals."""
    from saga_amd import _lib
    from saga_amd.fitting import create_decoder, param_maps
    from saga_amd.geometry import RowPicker, point2point_signed, vertex_normals_subset
    from saga_amd.vposer import VPoserDecoder
    dev = torch.device('cuda', 0)
    out = RowPicker({'a': torch.arange(5), 'b': (torch.tensor([1, 1, 3]), 2)}, 7)(torch.zeros(0, 4, 7, 3, device=dev, requires_grad=True))
    assert out['a'].shape == (0, 4, 5, 3) and out['b'].shape == (0, 2, 3, 3)
    (out['a'].sum() + out['b'].sum()).backward()
    z = lambda *sh: torch.zeros(*sh, device=dev)
    t = ml.motion_terms(z(0, 5, 79, 3), z(0, 5, 8, 3), [2, 2, 2, 2], z(0, 3, 778, 3), z(0, 2048), z(0, 143), z(0, 143, 3), z(0, 5, 79, 3),
                        torch.ones(79, device=dev), z(0, 79, 3), z(0, 5, 4), z(0, 2048), z(0, 143), z(0, 22, 3))
    assert t.shape == (0, 7)
    bm = make_hip_model(body_model_dir, 1)
    dec = create_decoder(VPoserDecoder(seed=0).cuda(), bm, dev)
    fp, sh, tr = param_maps(z(0, _lib.VAR_STRIDE), dec, 20)
    assert fp.shape == (0, 165) and sh.shape == (0, 20) and tr.shape == (0, 3)
    _lib.load().saga_decoder_destroy(dec)
    n = vertex_normals_subset(bm, z(0, 3, 10475, 3), torch.arange(4, dtype=torch.int32, device=dev), first_frame=1)
    assert n.shape == (0, 4, 3)
    o = point2point_signed(z(0, 5, 3), z(0, 9, 3), directions='y2x')
    assert o[0].shape == (0, 9) and o[1] is None


# ---- the two-stage driver (MotionFill/models/fittingop.py:283-428), S sequences per launch ------------------------------
def _fit_problem(body_model_dir, S, Tn, start_t, seed=0):
    """Targets from smooth synthetic motions: 79 markers per frame, an object next to the right hand of the last frame."""
    from saga_amd.vposer import VPoserDecoder
    idx = load_grasp_indices()
    mk79 = torch.from_numpy(idx['markers_f0_p5']).long()
    rh = torch.from_numpy(idx['rhand']).long()
    bm_ref = load_oracle_model(body_model_dir, Tn, dtype=torch.float64)
    g = torch.Generator().manual_seed(100 + seed)
    prob = dict(markers_rec=[], markers_end=[], lbl=[], c_obj=[], c_mk=[], vo=[], no=[], beta=[])
    vo, no, _, _ = make_object_cloud(S, seed=3 + seed)
    for s in range(S):
        a, b = random_body_inputs(1, seed=seed + 2 * s, scale=0.3), random_body_inputs(1, seed=seed + 2 * s + 1, scale=0.3)
        lam = torch.linspace(0, 1, Tn, dtype=torch.float64)[:, None]
        inp = {k: (1 - lam) * a[k] + lam * b[k] for k in a}
        inp['transl'] = inp['transl'] * 0.05 + torch.tensor([0.0, 0.4, 1.0])
        with torch.no_grad():
            v = bm_ref(**inp)['vertices']
        prob['markers_rec'].append(v[:, mk79])
        prob['markers_end'].append(v[-1, mk79] + 0.01 * torch.randn(79, 3, generator=g, dtype=torch.float64))
        prob['lbl'].append((torch.rand(Tn, 4, generator=g) > 0.4).double())
        prob['c_obj'].append(torch.rand(2048, generator=g, dtype=torch.float64))
        prob['c_mk'].append(torch.rand(143, generator=g, dtype=torch.float64))
        o = torch.from_numpy(vo[s]).double()
        o = o - o.mean(0) + v[-1, rh].mean(0)
        prob['vo'].append(o[None].expand(Tn - start_t, -1, -1).contiguous())
        prob['no'].append(torch.from_numpy(no[s]).double()[None].expand(Tn - start_t, -1, -1).contiguous())
        prob['beta'].append(0.5 * torch.randn(10, generator=g, dtype=torch.float64))
    prob = {k: torch.stack(v) for k, v in prob.items()}
    vposer = VPoserDecoder(seed=0)
    return idx, bm_ref, vposer, prob


def _hip_op(body_model_dir, idx, vposer, prob, S, Tn, start_t, num_iter):
    from saga_amd.motion_fitting import FittingOP
    return FittingOP(dict(T=Tn, n_seq=S, start_t=start_t, gender='male', device='cuda', smplx_beta=prob['beta'].numpy(),
                          num_iter=num_iter, markers_ids=idx['markers_f0_p5'], markers_ids_143=idx['markers_143'],
                          body_model_path=body_model_dir, vposer=vposer, indices=idx))


def test_motion_stage1_matches_oracle(body_model_dir):
    """Per-frame marker fit, warm-started frame to frame: two sequences in one batch == two B=1 oracle runs."""
    from oracle import motion_fit_ref as mfr
    S, Tn, n1 = 2, 4, 4
    idx, _, vposer, prob = _fit_problem(body_model_dir, S, Tn, 2)
    op = _hip_op(body_model_dir, idx, vposer, prob, S, Tn, 2, [n1, 0])
    transl, rot, shape, other, rec = op.fitting_stage1(prob['markers_rec'])
    assert transl.shape == (S, Tn, 3) and rot.shape == (S, Tn, 6) and shape.shape == (S, Tn, 10) and other.shape == (S, Tn, 80)
    bm1 = load_oracle_model(body_model_dir, 1, dtype=torch.float64)
    w = vposer.weights_dict()
    mk79 = torch.from_numpy(idx['markers_f0_p5']).long()
    for s in range(S):
        t_r, r_r, s_r, o_r = mfr.fitting_stage1(bm1, w, prob['beta'][s], mk79, prob['markers_rec'][s], n1)
        # Adam's first steps are +-lr per element whatever the gradient's size: float32 vs float64 gradients agree in sign
        # for every element here, so the trajectories stay within rounding of each other
        for name, a, b in [('transl', transl[s], t_r), ('rot6d', rot[s], r_r), ('other', other[s], o_r)]:
            np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=2e-4, rtol=0, err_msg='%s seq %d' % (name, s))
        np.testing.assert_allclose(shape[s].cpu().numpy(), s_r.numpy(), atol=1e-7)


def test_motion_stage1_graph_replay_and_marker_subset(body_model_dir):
    """Stage 1 three ways against the eager autograd loop: every iteration as one ``saga_fit_step`` call with the stage-1
    weights (the default: the library's own decoder GEMMs and Adam kernel), and the autograd iteration replayed as a HIP
    graph (the same kernels in the same order; only Adam's capturable form differs: device-side step counter and bias correction).  The body model over the 79
    markers gives the full mesh's rows bit for bit; its backward sums fewer (all non-zero) terms.  Both differences are
    rounding (amplified by Adam, whose first steps are +-lr whatever the gradient size): after 5 iterations per frame the three forms agree to 5e-4; over 84 iterations per frame (through the
    learning-rate switches at steps 61 and 81, moments carried from frame to frame) graph and eager loop reach the same marker error."""
    from saga_amd.motion_fitting import convert_to_3D_rot, gen_body_mesh_v1
    S, Tn = 2, 3
    idx, _, vposer, prob = _fit_problem(body_model_dir, S, Tn, 1)
    off = dict(native_stage1=False)
    modes = [('native', {}), ('graph', off), ('eager', dict(off, graph_stage1=False)),
             ('eager full mesh', dict(off, graph_stage1=False, reduced_forward=False))]

    def marker_error(op):          # mean |markers - target| of the last frame, per sequence, at the variables the fit left
        with torch.no_grad():
            p = convert_to_3D_rot(torch.cat([op.transl_opt_t, op.rot_6d_opt_t, op.shape_t, op.other_params_opt_t], dim=-1))
            mk = gen_body_mesh_v1(p, op.smplx_model, op.vposer_model)[0].index_select(1, op.markers_ids)
            return (mk - prob['markers_rec'][:, -1].float().cuda()).abs().mean(dim=(1, 2)).cpu().numpy()

    for n1, which in [(5, modes), (84, modes[:3])]:
        res, err = {}, {}
        for mode, kw in which:
            op = _hip_op(body_model_dir, idx, vposer, prob, S, Tn, 1, [n1, 0])
            for k, v in kw.items():
                setattr(op, k, v)
            if n1 == 84 and mode == 'native':
                err['start'] = marker_error(op)
            res[mode] = [t.clone() for t in op.fitting_stage1(prob['markers_rec'])]
            err[mode] = marker_error(op)
        assert all(torch.isfinite(t).all() for r in res.values() for t in r)
        if n1 == 5:
            for mode in res:
                for name, a, b in zip(('transl', 'rot6d', 'shape', 'other'), res[mode], res['eager']):
                    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=5e-4, rtol=0, err_msg='%s %s' % (mode, name))
        else:       # 252 Adam steps amplify the rounding differences in the variables; the fits are equally good
            for mode in ('native', 'graph'):
                assert (err[mode] < 0.5 * err['start']).all(), err
                np.testing.assert_allclose(err[mode], err['eager'], rtol=0.05, err_msg=mode)
        assert (res['native'][0][:, 1] - res['native'][0][:, 0]).abs().max() > 0     # frames continue from the previous result


def test_motion_stage2_matches_oracle(body_model_dir):
    """Whole-sequence stage: every loss term of the first iteration within 1e-4 of the oracle's, totals of the following
    iterations within 5e-3 (the collision / skating / angle masks are discontinuous), for 2 sequences in one batch."""
    from oracle import motion_fit_ref as mfr
    S, Tn, start_t, n2 = 2, 13, 6, 4
    idx, bm_ref, vposer, prob = _fit_problem(body_model_dir, S, Tn, start_t, seed=20)
    groups = ml.foot_groups(idx)
    mk79 = torch.from_numpy(idx['markers_f0_p5']).long()
    mk143 = torch.from_numpy(idx['markers_143']).long()
    rh = torch.from_numpy(idx['rhand']).long()
    g = torch.Generator().manual_seed(9)
    start = []
    for s in range(S):     # a smooth start: transl near the targets, a constant orientation, small latent / hand values
        lam = torch.linspace(0, 1, Tn, dtype=torch.float64)[:, None]
        transl = torch.tensor([0.0, 0.4, 1.0], dtype=torch.float64) + 0.02 * lam
        rot6 = mfr.convert_to_6D_all(torch.tensor([[0.0, 1.6, 3.14]], dtype=torch.float64)).expand(Tn, -1).contiguous()
        other = 0.1 * torch.randn(1, 80, generator=g, dtype=torch.float64) + 0.05 * lam * torch.randn(1, 80, generator=g, dtype=torch.float64)
        start.append((transl, rot6, prob['beta'][s][None].expand(Tn, -1).contiguous(), other))
    op = _hip_op(body_model_dir, idx, vposer, prob, S, Tn, start_t, [0, n2])
    st = lambda i: torch.stack([v[i] for v in start])
    op.init_param_T(st(0).float(), st(1).float(), st(3).float(), st(2).float().numpy())
    hist = []
    out = op.fitting_stage2(prob['markers_rec'], prob['markers_end'], idx['rhand'], prob['lbl'], prob['c_obj'], prob['c_mk'],
                            prob['no'], prob['vo'], record=lambda i, tot, terms: hist.append((tot.cpu(), {k: v.cpu() for k, v in terms.items()})))
    faces = torch.from_numpy(bm_ref.faces.astype(np.int64))
    w = vposer.weights_dict()
    for s in range(S):
        ref_hist = []
        res = mfr.fitting_stage2(bm_ref, faces, w, groups, mk79, mk143, start[s], prob['markers_rec'][s], prob['markers_end'][s], rh,
                                 prob['lbl'][s], prob['c_obj'][s][None], prob['c_mk'][s][None], prob['no'][s], prob['vo'][s], n2,
                                 start_t=start_t, record=lambda i, tot, terms: ref_hist.append((tot, terms)))
        for k, v in ref_hist[0][1].items():
            np.testing.assert_allclose(hist[0][1][k][s].item(), v.item(), rtol=1e-4, atol=1e-9, err_msg='%s seq %d' % (k, s))
        assert ref_hist[0][1]['contact'].item() > 0 and ref_hist[0][1]['skating'].item() > 0
        for i in range(n2):
            np.testing.assert_allclose(hist[i][0][s].item(), ref_hist[i][0].item(), rtol=1e-4 if i == 0 else 5e-3, err_msg='total it %d' % i)
        for name, a, b in [('transl', out[0][s], res[0]), ('rot6d', out[1][s], res[1]), ('other', out[3][s], res[3])]:
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.numpy(), atol=2e-3, rtol=0, err_msg='%s seq %d' % (name, s))


@pytest.mark.parametrize('mesh', ['random_faces', 'mesh_like'])
def test_motion_stage2_sparse_backward_matches_dense(body_model_dir, mesh_model_dir, mesh):
    """Stage 2 with ``sparse_backward`` (smplx forward(vertex_grad_ids=...): only the vertices the losses read carry a
    gradient, backward on the subset model) against the dense autograd path: equal losses, gradients equal to rounding.
    ``reduced_forward`` (the forward also runs over the loss vertices and the hand's face ring only) changes no bit of
    either."""
    model_dir = body_model_dir if mesh == 'random_faces' else mesh_model_dir
    S, Tn, start_t = 2, 9, 4
    idx, _, vposer, prob = _fit_problem(model_dir, S, Tn, start_t, seed=60)
    g = torch.Generator().manual_seed(5)
    transl = torch.tensor([0.0, 0.4, 1.0]) + 0.02 * torch.rand(S, Tn, 3, generator=g)
    rot6 = torch.tensor([-1., 0, 0, 0, 0, 1]).expand(S, Tn, 6) + 0.05 * torch.randn(S, Tn, 6, generator=g)
    other = 0.1 * torch.randn(S, Tn, 80, generator=g)
    shape = prob['beta'][:, None].expand(S, Tn, 10).float().contiguous()
    f = lambda a, *sh: torch.as_tensor(a).cuda().float().reshape(*sh)
    res, n_eval = {}, {}
    for mode in ('dense', 'sparse', 'reduced'):
        op = _hip_op(model_dir, idx, vposer, prob, S, Tn, start_t, [0, 1])
        op.sparse_backward, op.reduced_forward = mode != 'dense', mode == 'reduced'
        op.init_param_T(transl, rot6, other, shape.numpy())
        rh = torch.as_tensor(idx['rhand'], dtype=torch.long, device='cuda')
        total, terms = op.stage2_losses(f(prob['markers_rec'], S, Tn, -1, 3), f(prob['markers_end'], S, -1, 3), rh,
                                        f(prob['lbl'], S, Tn, 4), f(prob['c_obj'], S, -1), f(prob['c_mk'], S, -1),
                                        f(prob['no'], S, Tn - start_t, 2048, 3), f(prob['vo'], S, Tn - start_t, 2048, 3))
        total.sum().backward()
        res[mode] = (total.detach(), {k: v.detach() for k, v in terms.items()},
                     [t.grad.clone() for t in (op.transl_opt_T, op.rot_6d_opt_T, op.other_params_opt_T)])
        n_eval[mode] = op._loss_model['bm'].num_verts if op._loss_model is not None else op.smplx_model_batch.num_verts
    assert n_eval['reduced'] < n_eval['sparse'] == n_eval['dense'] == 10475
    if mesh == 'mesh_like':
        assert n_eval['reduced'] < 3000, n_eval           # ~1,100 loss vertices + the hand's ring
    for mode in ('sparse', 'reduced'):
        assert torch.equal(res[mode][0], res['dense'][0])
        for k in res['dense'][1]:
            assert torch.equal(res[mode][1][k], res['dense'][1][k]), k
        for a, b in zip(res[mode][2], res['dense'][2]):
            assert b.abs().max() > 0
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-5
    for a, b in zip(res['reduced'][2], res['sparse'][2]):
        assert torch.equal(a, b)


def test_motion_fused_terms_match_composition(body_model_dir):
    """saga_motion_terms_fwd/bwd (the seven geometric stage-2 terms as one operator) against their autograd composition in
    motion_losses (itself checked against the oracle above): every term within 2e-6 relative (float32 sums in another order),
    gradients of the optimisation variables within 1e-4, run-to-run bit-equal; skating / contact / collision / angle active."""
    S, Tn, start_t = 2, 13, 6
    idx, _, vposer, prob = _fit_problem(body_model_dir, S, Tn, start_t, seed=20)
    g = torch.Generator().manual_seed(9)
    lam = torch.linspace(0, 1, Tn)[None, :, None]
    transl = torch.tensor([0.0, 0.4, 1.0]) + 0.02 * lam + 0.002 * torch.randn(S, Tn, 3, generator=g)
    rot6 = torch.tensor([-1., 0, 0, 0, 0, 1]).expand(S, Tn, 6) + 0.05 * torch.randn(S, Tn, 6, generator=g)
    other = 0.1 * torch.randn(S, 1, 80, generator=g) + 0.05 * lam * torch.randn(S, 1, 80, generator=g)
    shape = prob['beta'][:, None].expand(S, Tn, 10).float().contiguous()
    f = lambda a, *sh: torch.as_tensor(a).cuda().float().reshape(*sh)
    rh = torch.as_tensor(idx['rhand'], dtype=torch.long, device='cuda')
    vo = f(prob['vo'], S, Tn - start_t, 2048, 3).clone()
    res = {}
    for mode in ('composition', 'fused', 'fused again'):
        op = _hip_op(body_model_dir, idx, vposer, prob, S, Tn, start_t, [0, 1])
        op.fused_terms = mode != 'composition'
        op.init_param_T(transl, rot6, other, shape.numpy())
        if mode == 'composition':      # object points 4-7 mm from the palm markers of the last frames: the angle term is active
            from saga_amd.motion_fitting import convert_to_3D_rot, gen_body_mesh_v1
            with torch.no_grad():
                p = convert_to_3D_rot(torch.cat([op.transl_opt_T, op.rot_6d_opt_T, op.shape_T, op.other_params_opt_T], -1).reshape(S * Tn, -1))
                v = gen_body_mesh_v1(p, op.smplx_model_batch, op.vposer_model)[0].reshape(S, Tn, -1, 3)
                vo[:, :, :22] = v[:, start_t:, op.markers_ids_143[-22:]] + 0.004
        total, terms = op.stage2_losses(f(prob['markers_rec'], S, Tn, -1, 3), f(prob['markers_end'], S, -1, 3), rh,
                                        f(prob['lbl'], S, Tn, 4), f(prob['c_obj'], S, -1), f(prob['c_mk'], S, -1),
                                        f(prob['no'], S, Tn - start_t, 2048, 3), vo)
        total.sum().backward()
        res[mode] = (total.detach(), {k: v.detach() for k, v in terms.items()},
                     [t.grad.clone() for t in (op.transl_opt_T, op.rot_6d_opt_T, op.other_params_opt_T)])
    ref, got = res['composition'], res['fused']
    assert set(ref[1]) == set(got[1])
    for k in ref[1]:
        np.testing.assert_allclose(got[1][k].cpu().numpy(), ref[1][k].cpu().numpy(), rtol=2e-6, atol=1e-9, err_msg=k)
    for k in ('skating', 'contact', 'collision', 'hand_angle', 'hand_smooth'):
        assert (ref[1][k] > 0).all(), k
    np.testing.assert_allclose(got[0].cpu().numpy(), ref[0].cpu().numpy(), rtol=2e-6)
    for a, b in zip(got[2], ref[2]):
        assert b.abs().max() > 0 and rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-4
    assert torch.equal(res['fused again'][0], got[0])
    for a, b in zip(res['fused again'][2], got[2]):
        assert torch.equal(a, b)


def test_param_maps_operator_matches_torch_maps(body_model_dir):
    """fitting.param_maps (saga_params_fwd/bwd: VPoser MLP + 22 x 6D->aa + hand PCA + pose mean as one operator) against the
    torch composition GraspMotion-Opt uses (convert_to_3D_rot + VPoser decode + the hand PCA inside the body model): SMPL-X
    arguments within 5e-5, gradients of a random linear loss within 1e-4; and stage 2 with and without it: first-iteration
    terms within 1e-5, gradients within 1e-4."""
    from saga_amd import _lib
    from saga_amd.fitting import create_decoder, param_maps
    from saga_amd.motion_fitting import convert_to_3D_rot
    from saga_amd.vposer import VPoserDecoder
    B = 37
    bm = make_hip_model(body_model_dir, B)
    vp = VPoserDecoder(seed=0).cuda()
    dec = create_decoder(vp, bm, torch.device('cuda', 0))
    g = torch.Generator().manual_seed(4)
    other = (0.5 * torch.randn(B, 80, generator=g)).cuda().requires_grad_(True)
    transl = torch.randn(B, 3, generator=g).cuda().requires_grad_(True)
    rot6 = torch.randn(B, 6, generator=g).cuda().requires_grad_(True)
    betas = torch.randn(B, 10, generator=g).cuda()
    w_pose, w_tr = torch.randn(B, 165, generator=g).cuda(), torch.randn(B, 3, generator=g).cuda()
    packed = torch.cat([other, other.new_zeros(B, 6), betas, transl, rot6, other.new_zeros(B, _lib.VAR_STRIDE - 105)], dim=1)
    full_pose, shape, tr = param_maps(packed, dec, 20)
    objective = (full_pose * w_pose).sum() + (tr * w_tr).sum()
    objective.backward(retain_graph=True)       # the form the reference calls every iteration
    got = [t.grad.clone() for t in (other, transl, rot6)]
    for t in (other, transl, rot6):
        t.grad = None
    # a second backward through the same forward (retain_graph, or two losses differentiated separately) used to
    # read decoder activations the first one had overwritten -- the backward now leaves the saved workspace intact
    objective.backward()
    for a, t in zip(got, (other, transl, rot6)):
        assert torch.equal(a, t.grad), 'the second backward through param_maps differs from the first'
        t.grad = None
    p = convert_to_3D_rot(torch.cat([transl, rot6, betas, other], dim=-1))
    body_pose = vp.decode(p[:, 16:48], output_type='aa').view(B, -1)
    lh = torch.einsum('bi,ij->bj', p[:, 48:72], bm.left_hand_components)
    rh = torch.einsum('bi,ij->bj', p[:, 72:], bm.right_hand_components)
    ref_pose = torch.cat([p[:, 3:6], body_pose, p.new_zeros(B, 9), lh, rh], dim=1) + bm.pose_mean
    ((ref_pose * w_pose).sum() + (p[:, :3] * w_tr).sum()).backward()
    np.testing.assert_allclose(full_pose.detach().cpu().numpy(), ref_pose.detach().cpu().numpy(), atol=5e-5, rtol=0)    # (aa near pi)
    assert torch.equal(shape[:, :10], betas) and shape[:, 10:].abs().max() == 0 and torch.equal(tr.detach(), transl.detach())
    for a, t in zip(got, (other, transl, rot6)):
        assert rel_err(a.cpu().numpy(), t.grad.cpu().numpy()) < 1e-4
    _lib.load().saga_decoder_destroy(dec)
    # inside stage 2
    S, Tn, start_t = 2, 9, 4
    idx, _, vposer, prob = _fit_problem(body_model_dir, S, Tn, start_t, seed=60)
    transl = torch.tensor([0.0, 0.4, 1.0]) + 0.02 * torch.rand(S, Tn, 3, generator=g)
    rot6 = torch.tensor([-1., 0, 0, 0, 0, 1]).expand(S, Tn, 6) + 0.05 * torch.randn(S, Tn, 6, generator=g)
    other = 0.1 * torch.randn(S, Tn, 80, generator=g)
    shape = prob['beta'][:, None].expand(S, Tn, 10).float().contiguous()
    f = lambda a, *sh: torch.as_tensor(a).cuda().float().reshape(*sh)
    res = {}
    for native in (True, False):
        op = _hip_op(body_model_dir, idx, vposer, prob, S, Tn, start_t, [0, 1])
        op.native_params = native
        op.init_param_T(transl, rot6, other, shape.numpy())
        rh = torch.as_tensor(idx['rhand'], dtype=torch.long, device='cuda')
        total, terms = op.stage2_losses(f(prob['markers_rec'], S, Tn, -1, 3), f(prob['markers_end'], S, -1, 3), rh,
                                        f(prob['lbl'], S, Tn, 4), f(prob['c_obj'], S, -1), f(prob['c_mk'], S, -1),
                                        f(prob['no'], S, Tn - start_t, 2048, 3), f(prob['vo'], S, Tn - start_t, 2048, 3))
        total.sum().backward()
        res[native] = (total.detach(), {k: v.detach() for k, v in terms.items()},
                       [t.grad.clone() for t in (op.transl_opt_T, op.rot_6d_opt_T, op.other_params_opt_T)])
    for k in res[False][1]:
        np.testing.assert_allclose(res[True][1][k].cpu().numpy(), res[False][1][k].cpu().numpy(), rtol=1e-5, atol=1e-9, err_msg=k)
    for a, b in zip(res[True][2], res[False][2]):
        assert b.abs().max() > 0 and rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-4


def test_motion_stage2_graph_replay_matches_eager(body_model_dir):
    """Stage 2 as a replayed HIP graph of one iteration (the default for small batches) against the eager loop: the same
    kernels; Adam in its capturable form.  First iteration equal to 1e-6, later totals within 5e-3, variables within 2e-3 after 6
    iterations; the recorded totals and terms are those of the iteration they are reported for."""
    S, Tn, start_t, n2 = 2, 9, 4, 6
    idx, _, vposer, prob = _fit_problem(body_model_dir, S, Tn, start_t, seed=60)
    g = torch.Generator().manual_seed(5)
    transl = torch.tensor([0.0, 0.4, 1.0]) + 0.02 * torch.rand(S, Tn, 3, generator=g)
    rot6 = torch.tensor([-1., 0, 0, 0, 0, 1]).expand(S, Tn, 6) + 0.05 * torch.randn(S, Tn, 6, generator=g)
    other = 0.1 * torch.randn(S, Tn, 80, generator=g)
    shape = prob['beta'][:, None].expand(S, Tn, 10).float().contiguous()
    res = {}
    for mode, flag in [('graph', True), ('eager', False)]:
        op = _hip_op(body_model_dir, idx, vposer, prob, S, Tn, start_t, [0, n2])
        op.graph_stage2, op.native_stage2 = flag, False          # (the autograd composition; saga_motion_step is checked below)
        op.init_param_T(transl, rot6, other, shape.numpy())
        hist = []
        out = op.fitting_stage2(prob['markers_rec'], prob['markers_end'], idx['rhand'], prob['lbl'], prob['c_obj'], prob['c_mk'],
                                prob['no'], prob['vo'], record=lambda i, tot, terms: hist.append((tot.cpu(), {k: v.cpu() for k, v in terms.items()})))
        res[mode] = (hist, [t.detach().clone() for t in out])
    assert len(res['graph'][0]) == n2
    for i, ((tg, dg), (te, de)) in enumerate(zip(res['graph'][0], res['eager'][0])):
        # the first iteration is the same arithmetic; later ones differ by Adam's two forms, amplified by the discontinuous
        # collision / skating / angle masks (the bar of the oracle test above)
        np.testing.assert_allclose(tg.numpy(), te.numpy(), rtol=1e-6 if i == 0 else 5e-3, err_msg='iteration %d' % i)
        if i == 0:
            for k in de:
                np.testing.assert_allclose(dg[k].numpy(), de[k].numpy(), rtol=1e-6, atol=1e-9, err_msg=k)
    assert not torch.equal(res['graph'][0][0][0], res['graph'][0][-1][0])          # the totals move from iteration to iteration
    for a, b in zip(res['graph'][1], res['eager'][1]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-3, rtol=0)


def _stage2_start(S, Tn, prob, seed=9):
    g = torch.Generator().manual_seed(seed)
    lam = torch.linspace(0, 1, Tn)[None, :, None]
    transl = torch.tensor([0.0, 0.4, 1.0]) + 0.02 * lam + 0.002 * torch.randn(S, Tn, 3, generator=g)
    rot6 = torch.tensor([-1., 0, 0, 0, 0, 1]).expand(S, Tn, 6) + 0.05 * torch.randn(S, Tn, 6, generator=g)
    other = 0.1 * torch.randn(S, 1, 80, generator=g) + 0.05 * lam * torch.randn(S, 1, 80, generator=g)
    shape = prob['beta'][:, None].expand(S, Tn, 10).float().contiguous()
    return transl, rot6, other, shape


def _motion_step_case(body_model_dir, S, Tn, start_t, reduced, seed, strict=True):
    """One saga_motion_step evaluation against the autograd composition (see the test below); strict: the activity asserts of
    the hand-picked case (every term positive, every gradient block non-zero) and the Adam / repeatability checks as well."""
    from saga_amd import _lib
    idx, _, vposer, prob = _fit_problem(body_model_dir, S, Tn, start_t, seed=seed)
    transl, rot6, other, shape = _stage2_start(S, Tn, prob)
    f = lambda a, *sh: torch.as_tensor(a).cuda().float().reshape(*sh)
    rh = torch.as_tensor(idx['rhand'], dtype=torch.long, device='cuda')
    vo = f(prob['vo'], S, Tn - start_t, 2048, 3).clone()
    op = _hip_op(body_model_dir, idx, vposer, prob, S, Tn, start_t, [0, 1])
    op.reduced_forward = reduced
    op.init_param_T(transl, rot6, other, shape.numpy())
    from saga_amd.motion_fitting import convert_to_3D_rot, gen_body_mesh_v1
    with torch.no_grad():       # object points 4 mm from the palm markers of the last frames: the angle term is active
        p = convert_to_3D_rot(torch.cat([op.transl_opt_T, op.rot_6d_opt_T, op.shape_T, op.other_params_opt_T], -1).reshape(S * Tn, -1))
        v = gen_body_mesh_v1(p, op.smplx_model_batch, op.vposer_model)[0].reshape(S, Tn, -1, 3)
        vo[:, :, :22] = v[:, start_t:, op.markers_ids_143[-22:]] + 0.004
    args = (f(prob['markers_rec'], S, Tn, -1, 3), f(prob['markers_end'], S, -1, 3), rh, f(prob['lbl'], S, Tn, 4), f(prob['c_obj'], S, -1),
            f(prob['c_mk'], S, -1), f(prob['no'], S, Tn - start_t, 2048, 3), vo)
    total, terms = op.stage2_losses(*args)
    total.sum().backward()
    assert (op._loss_model is not None) == reduced
    ctx = op.stage2_native_setup(*args)
    packed = op.pack_stage2_vars()
    before = packed.clone()
    loss, out = op.stage2_native_step(ctx, packed)                  # adam_step 0: objective and gradient only
    assert torch.equal(packed, before)
    tag = (S, Tn, start_t, reduced, seed)
    col = {k: out[:, i] for i, k in enumerate(_lib.MOTION_STEP_TERMS)}
    for k in ('rec_markers', 'end_markers_fit', 'skating', 'hand_smooth', 'contact', 'collision', 'hand_angle'):
        assert torch.equal(col[k], terms[k].detach()), (k, tag)
        assert not strict or (col[k] > 0).all(), k
    for k in ('vposer', 'hand'):
        np.testing.assert_allclose(col[k].cpu().numpy(), terms[k].detach().cpu().numpy(), rtol=1e-6, err_msg='%s %s' % (k, tag))
    np.testing.assert_allclose(loss.cpu().numpy(), total.detach().cpu().numpy(), rtol=1e-6, err_msg=str(tag))
    assert torch.equal(loss, col['total']) and torch.isfinite(loss).all()
    grad = op.stage2_native_buffer(ctx, 'grad', (S, Tn, _lib.VAR_STRIDE))
    sl = _lib.VAR_SLICES
    for name, got, ref in [('transl', grad[..., sl['transl_rec'][0]:sl['transl_rec'][1]], op.transl_opt_T.grad),
                           ('rot6d', grad[..., sl['glo_rot_rec'][0]:sl['glo_rot_rec'][1]], op.rot_6d_opt_T.grad),
                           ('other', grad[..., :80], op.other_params_opt_T.grad)]:
        assert ref.abs().max() > 0
        assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < 1e-6, (name, tag)
        if name != 'other':       # no regulariser on these: the same kernels on the same numbers
            assert torch.equal(got, ref), (name, tag)
    assert grad[..., 105:].abs().max() == 0
    if not strict:
        return
    assert grad[..., sl['betas'][0]:sl['betas'][1]].abs().max() > 0
    # run-to-run bit-equal, and an update moves exactly the variables the reference's optimiser owns (:344-350)
    g0 = grad.clone()
    loss2, _ = op.stage2_native_step(ctx, packed)
    assert torch.equal(loss2, loss) and torch.equal(op.stage2_native_buffer(ctx, 'grad', (S, Tn, _lib.VAR_STRIDE)), g0)
    m, v2 = torch.zeros_like(packed), torch.zeros_like(packed)
    op.stage2_native_step(ctx, packed, m, v2, adam_step=1, lr=0.01)
    moved = (packed != before).reshape(S * Tn, -1)
    assert moved[:, :80].any() and moved[:, 96:105].all() and not moved[:, 80:96].any() and not moved[:, 105:].any()
    # Adam's first step is -lr * g / (|g| + eps) per element
    g_rows = g0.reshape(S * Tn, -1)
    expect = before - 0.01 * g_rows / (g_rows.abs() + 1e-8)
    np.testing.assert_allclose(packed[:, 96:105].cpu().numpy(), expect[:, 96:105].cpu().numpy(), atol=1e-6)


@pytest.mark.parametrize('reduced', [True, False])
def test_motion_step_terms_bit_equal_autograd_composition(body_model_dir, reduced):
    """saga_motion_step (the whole stage-2 iteration as ONE library call) against the autograd composition of the same operators
    (``stage2_losses``, itself checked against the oracle): the seven geometric terms BIT-EQUAL (same kernels, same order, same
    inputs), the two regularisers and the total within 1e-6 relative (torch's mean against the library's fixed-order sum), the
    gradient rows within 1e-6 of autograd's (bit-equal except where the regulariser gradient is added), every named intermediate
    equal; on the loss-vertex body model (default) and on the full mesh.  skating / contact / collision / angle active."""
    _motion_step_case(body_model_dir, 2, 13, 6, reduced, seed=20)


def test_motion_step_size_fuzz_against_autograd_composition(body_model_dir):
    """The same comparison at sequence counts, lengths and grasp-window starts nobody picked (one sequence, two frames, a window
    of one frame ...): the seven geometric terms bit-equal, totals and gradient rows to 1e-6."""
    rng = np.random.default_rng(77)
    for case in range(8):
        S, Tn = int(rng.integers(1, 6)), int(rng.integers(3, 18))
        start_t = int(rng.integers(1, Tn - 1))
        _motion_step_case(body_model_dir, S, Tn, start_t, bool(rng.integers(0, 2)), seed=300 + case, strict=False)


def test_motion_stage2_native_loop_matches_autograd_loop(body_model_dir):
    """``fitting_stage2`` through saga_motion_step (the default) against the eager autograd loop with torch.optim.Adam: first
    iteration equal to 1e-6, later totals within 5e-3 and variables within 2e-3 after 6 iterations (the bars of the graph-replay
    test: the collision / skating / angle masks are discontinuous), through the learning-rate switch logic."""
    S, Tn, start_t, n2 = 2, 9, 4, 6
    idx, _, vposer, prob = _fit_problem(body_model_dir, S, Tn, start_t, seed=60)
    g = torch.Generator().manual_seed(5)
    transl = torch.tensor([0.0, 0.4, 1.0]) + 0.02 * torch.rand(S, Tn, 3, generator=g)
    rot6 = torch.tensor([-1., 0, 0, 0, 0, 1]).expand(S, Tn, 6) + 0.05 * torch.randn(S, Tn, 6, generator=g)
    other = 0.1 * torch.randn(S, Tn, 80, generator=g)
    shape = prob['beta'][:, None].expand(S, Tn, 10).float().contiguous()
    res = {}
    for mode in ('native', 'eager'):
        op = _hip_op(body_model_dir, idx, vposer, prob, S, Tn, start_t, [0, n2])
        op.native_stage2, op.graph_stage2 = mode == 'native', False
        op.init_param_T(transl, rot6, other, shape.numpy())
        hist = []
        out = op.fitting_stage2(prob['markers_rec'], prob['markers_end'], idx['rhand'], prob['lbl'], prob['c_obj'], prob['c_mk'],
                                prob['no'], prob['vo'], record=lambda i, tot, terms: hist.append((tot.cpu(), {k: v.cpu() for k, v in terms.items()})))
        assert (mode == 'native') == hasattr(op, 'last_stage2')
        res[mode] = (hist, [t.detach().clone() for t in out])
    assert len(res['native'][0]) == n2 and set(res['native'][0][0][1]) == set(res['eager'][0][0][1])
    for i, ((tn, dn), (te, de)) in enumerate(zip(res['native'][0], res['eager'][0])):
        np.testing.assert_allclose(tn.numpy(), te.numpy(), rtol=1e-6 if i == 0 else 5e-3, err_msg='iteration %d' % i)
        if i == 0:
            for k in de:
                np.testing.assert_allclose(dn[k].numpy(), de[k].numpy(), rtol=1e-6, atol=1e-9, err_msg=k)
    assert not torch.equal(res['native'][0][0][0], res['native'][0][-1][0])
    for a, b in zip(res['native'][1], res['eager'][1]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-3, rtol=0)
    assert not torch.equal(res['native'][1][0], transl.cuda())


def test_motion_fitting_sequences_are_independent(body_model_dir):
    """The batched driver gives each sequence the result of its own run: S = 2 equals two S = 1 runs up to the rounding
    of the torch / rocBLAS pieces whose reduction order depends on the batch shape (VPoser linears, means)."""
    S, Tn, start_t = 2, 7, 3
    idx, _, vposer, prob = _fit_problem(body_model_dir, S, Tn, start_t, seed=40)
    args = lambda sl: (prob['markers_rec'][sl], prob['markers_end'][sl], idx['rhand'], prob['lbl'][sl], prob['c_obj'][sl],
                       prob['c_mk'][sl], prob['no'][sl], prob['vo'][sl])
    both = _hip_op(body_model_dir, idx, vposer, prob, S, Tn, start_t, [2, 2]).fitting(*args(slice(0, 2)))
    for s in range(S):
        one_prob = {k: v[s:s + 1] for k, v in prob.items()}
        one = _hip_op(body_model_dir, idx, vposer, one_prob, 1, Tn, start_t, [2, 2]).fitting(*args(slice(s, s + 1)))
        for a, b in zip(both, one):
            np.testing.assert_allclose(a[s].detach().cpu().numpy(), b[0].detach().cpu().numpy(), atol=2e-5, rtol=0)


def test_motion_opt_result_schema(body_model_dir, tmp_path):
    """``saved_results`` of opt_graspmotion.py:549-561: keys, nesting and [S,T,.] shapes; the file loads back."""
    from saga_amd.opt_graspmotion import BODY_KEYS, motion_opt
    S, Tn, start_t = 2, 5, 2
    idx, _, vposer, prob = _fit_problem(body_model_dir, S, Tn, start_t, seed=60)
    op = _hip_op(body_model_dir, idx, vposer, prob, S, Tn, start_t, [1, 1])
    out = motion_opt(op, prob['markers_rec'], prob['markers_end'], idx['rhand'], prob['lbl'], prob['c_obj'], prob['c_mk'],
                     prob['no'], prob['vo'], 'camera', np.zeros((S, Tn, 3), np.float32), np.zeros((S, Tn, 3, 3), np.float32),
                     save_dir=str(tmp_path))
    assert set(out) == {'body_orig', 'body_opt', 'object_name', 'object'} and set(out['object']) == {'transl', 'global_orient'}
    dims = {'transl': 3, 'global_orient': 3, 'betas': 10, 'body_pose': 63, 'left_hand_pose': 24, 'right_hand_pose': 24}
    for part in ('body_orig', 'body_opt'):
        assert list(out[part]) == BODY_KEYS
        for k, n in dims.items():
            assert out[part][k].shape == (S, Tn, n) and np.isfinite(out[part][k]).all()
    np.testing.assert_allclose(out['body_opt']['betas'], np.broadcast_to(prob['beta'].numpy()[:, None], (S, Tn, 10)), rtol=1e-6)
    back = np.load(str(tmp_path / 'fitting_results.npy'), allow_pickle=True).item()
    assert back['object_name'] == 'camera' and back['body_opt']['body_pose'].shape == (S, Tn, 63)


def test_motion_terms_operator_fuzz_against_fp64_restatement():
    """saga_motion_terms_fwd/bwd on random tensors of random sizes (S 1-3, T 3-20, tails 2..T, ragged set sizes, values that cross
    every mask threshold: foot speeds around 0.1, object -> hand distances around -3 mm, marker distances around 1 cm, contact
    labels of every pattern incl. none / all) against a float64 restatement of MotionFill/models/fittingop.py:159-176 (contact),
    :178-191 (collision), :193-263 (skating), :265-280 (angle), :360-368 (marker / end fit), :386-387 (hand smoothness) under
    autograd: terms within 2e-5, gradients of the six differentiable inputs within 1e-4."""
    g = torch.Generator().manual_seed(31)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for case in range(40):
        S, Tn = ri(1, 3), ri(3, 20)
        nt = ri(2, Tn)
        M, H, O, CM, NP = ri(1, 90), ri(1, 300), ri(1, 400), ri(22, 150), 22
        fs = [ri(1, 40) for _ in range(4)]
        R = lambda *s, sc=1.0: torch.randn(*s, generator=g, dtype=torch.float64) * sc
        markers, markers_rec, markers_end = R(S, Tn, M, 3), R(S, Tn, M, 3), R(S, M, 3)
        mask = (torch.rand(M, generator=g) > 0.3).double()
        feet = torch.cumsum(R(S, Tn, sum(fs), 3, sc=0.003), dim=1) + R(S, 1, sum(fs), 3)
        lbl_mode = ri(0, 2)
        lbl = torch.zeros(S, Tn, 4, dtype=torch.float64) if lbl_mode == 0 else (torch.ones(S, Tn, 4, dtype=torch.float64) if lbl_mode == 1
                                                                               else (torch.rand(S, Tn, 4, generator=g) > 0.5).double())
        hand_tail = torch.cumsum(R(S, nt, H, 3, sc=0.01), dim=1) + R(S, 1, H, 3)
        o2h, h2o_mk = R(S * nt, O, sc=0.01), torch.rand(S * nt, CM, generator=g, dtype=torch.float64) * 0.03
        h2o_vec, palm_n = R(S * nt, CM, 3, sc=0.02), torch.nn.functional.normalize(R(S * nt, NP, 3), dim=-1) * (0.5 + torch.rand(S * nt, NP, 1, generator=g, dtype=torch.float64))
        c_obj, c_mk = torch.rand(S, O, generator=g, dtype=torch.float64), torch.rand(S, CM, generator=g, dtype=torch.float64)
        leaves = [t.clone().requires_grad_(True) for t in (markers, feet, hand_tail, o2h, h2o_mk, h2o_vec)]
        mk_, ft_, ht_, o2h_, hm_, hv_ = leaves

        # ---- float64 restatement --------------------------------------------------------------------------------------
        m4 = mask.view(1, 1, M, 1)
        t_rec = (mk_ * m4 - markers_rec * m4).abs().mean(dim=(1, 2, 3))
        t_end = (mk_[:, -1] - markers_end).abs().mean(dim=(1, 2))
        t_sk, o = torch.zeros(S, dtype=torch.float64), 0
        for col in range(4):
            foot = ft_[:, :, o:o + fs[col]]
            o += fs[col]
            speed = ((foot[:, 1:] - foot[:, :-1]) * 30).norm(dim=-1)
            on = (lbl[:, :-1, col] == 1)[:, :, None] & (speed > 0.1)
            cnt = on.sum(dim=(1, 2))
            t_sk = t_sk + torch.where(cnt > 0, (speed * on).sum(dim=(1, 2)) / cnt.clamp(min=1), torch.zeros(S, dtype=torch.float64))
        t_hs = (ht_[:, 1:] - ht_[:, :-1]).norm(dim=-1).mean(dim=(1, 2)) if nt > 1 else torch.zeros(S, dtype=torch.float64)
        w = torch.zeros(nt, 1, dtype=torch.float64)
        w[-5:] = 1
        o2h_s, hm_s = o2h_.view(S, nt, O), hm_.view(S, nt, CM)
        t_con = (o2h_s.abs() * c_obj.view(S, 1, O) * w).mean(dim=(1, 2)) + (hm_s.abs() * c_mk.view(S, 1, CM) * w).mean(dim=(1, 2))
        t_col = 5 * (o2h_s.abs() * torch.where(o2h_s < -0.003, 20.0, 0.0)).mean(dim=-1).mean(dim=-1)
        vec = -hv_.view(S, nt, CM, 3)[:, :, -NP:]
        pn = palm_n.view(S, nt, NP, 3)
        cos = (pn * vec).sum(-1) / (pn.norm(dim=-1) * vec.norm(dim=-1))
        wt = 1 - ((torch.arange(nt, dtype=torch.float64) + 1) / nt).view(-1, 1) ** 2
        t_ang = ((hm_s[:, :, -NP:] < 0.01).double() * (cos - 1) ** 2 * wt).mean(dim=(1, 2))
        ref = torch.stack([t_rec, t_end, t_sk, t_hs, t_con, t_col, t_ang], dim=1)
        gw = R(S, 7).abs() + 0.1
        (ref * gw).sum().backward()

        # ---- the operator ---------------------------------------------------------------------------------------------
        c = lambda t: t.detach().float().cuda()
        hl = [c(t).requires_grad_(True) for t in leaves]
        out = ml.motion_terms(hl[0], hl[1], fs, hl[2], hl[3], hl[4], hl[5], c(markers_rec), c(mask), c(markers_end), c(lbl), c(c_obj),
                              c(c_mk), c(palm_n))
        (out * c(gw)).sum().backward()
        info = (case, S, Tn, nt, M, H, O, CM, fs, lbl_mode)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=1e-9, err_msg=str(info))
        for name, a, b in zip(('markers', 'feet', 'hand_tail', 'o2h', 'h2o_marker', 'h2o_vec'), hl, leaves):
            if b.grad.abs().max() == 0:
                assert a.grad.abs().max() == 0, (name, info)
                continue
            assert rel_err(a.grad.cpu().numpy().reshape(-1), b.grad.numpy().reshape(-1)) < 1e-4, (name, info)
