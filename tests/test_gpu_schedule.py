"""Row a15 / BASELINE config 1 on the GPU: the stage schedule, the best-of-iteration rule
(WholeGraspPose/models/fittingop.py:429-447,464-488) and the config-1 inputs (tests/golden/vae_samples.npz).

* saga_fit_best_update driven directly through the C-ABI with crafted loss sequences, against a literal torch
  restatement of the rule;
* FittingOP.fitting's use of it (final-state fallback for a candidate that never went below 10000, :464);
* an end-to-end fit that reaches the rule (stage 3 longer than 201 iterations) against the CPU oracle, with the returned
  parameters shown to be the SNAPSHOT (variables after the best iteration's step, markers from before it) by a
  bit-exact replay;
* the 10 VAE-sampled objects of config 1 through pose_opt with the reference's full 300/400/500 schedule.
"""
import os

import numpy as np
import pytest
import torch

from oracle import fitting_ref
from tests.helpers import load_oracle_model

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _best_rule_reference(losses, vars_seq, markers_seq):
    """fittingop.py:354,430-443 for B independent candidates: strict '<' against the running best, which starts at 10000."""
    T, B = losses.shape
    best = torch.full((B,), 10000.0)
    bv, bm = torch.zeros_like(vars_seq[0]), torch.zeros_like(markers_seq[0])
    for t in range(T):
        better = losses[t] < best
        bv[better], bm[better] = vars_seq[t][better], markers_seq[t][better]
        best = torch.where(better, losses[t], best)
    return best, bv, bm


@pytest.mark.parametrize('M', [143, 7])
def test_best_update_rule_through_the_c_abi(M):
    from saga_amd import _lib
    lib = _lib.load()
    T, B, VS = 9, 8, _lib.VAR_STRIDE
    g = torch.Generator().manual_seed(17)
    losses = torch.empty(T, B)
    losses[:, 0] = torch.linspace(5.0, 1.0, T)                         # strictly improving: the last call wins
    losses[:, 1] = torch.tensor([3., 3., 2., 2., 2., 4., 2., 5., 2.])    # ties: the FIRST of equal losses is kept
    losses[:, 2] = 10000.0                                              # never below the initial best: nothing is taken
    losses[:, 3] = torch.tensor([20000.] * (T - 1) + [9999.0])           # improves only on the last call
    losses[:, 4] = torch.linspace(1.0, 5.0, T)                         # best on the first call
    losses[:, 5] = torch.tensor([7., float('nan'), 6., float('nan'), 6., 8., 5.5, float('nan'), 9.])   # NaN never wins
    losses[:, 6] = torch.tensor([float('inf')] * T)
    losses[:, 7] = torch.rand(T, generator=g)
    vars_seq = torch.randn(T, B, VS, generator=g)
    markers_seq = torch.randn(T, B, M, 3, generator=g)
    want_best, want_v, want_m = _best_rule_reference(losses, vars_seq, markers_seq)

    best = torch.full((B,), 10000.0, device='cuda')
    bv = torch.zeros(B, VS, device='cuda')
    bm = torch.zeros(B, M, 3, device='cuda')
    P = _lib.ptr
    for t in range(T):
        l, v, m = losses[t].cuda(), vars_seq[t].cuda(), markers_seq[t].cuda().contiguous()
        _lib.call('saga_fit_best_update', l.device, P(l), P(v), P(m), P(best), P(bv), P(bm), B, M)
    torch.cuda.synchronize()
    assert torch.equal(best.cpu(), want_best)
    assert torch.equal(bv.cpu(), want_v) and torch.equal(bm.cpu(), want_m)
    assert float(best[2]) == 10000.0 and not bv[2].any() and float(best[3]) == 9999.0
    # argument errors come back as codes, not crashes
    assert lib.saga_fit_best_update(None, P(bv), P(bm), P(best), P(bv), P(bm), B, M, None) == -1
    assert lib.saga_fit_best_update(P(best), P(bv), P(bm), P(best), P(bv), P(bm), B, 0, None) == -1


def test_fitting_returns_snapshot_or_final_state(body_model_dir):
    """FittingOP.fitting around the kernel: with crafted per-iteration losses (native_step replaced), candidate 0 returns
    the variables of its best iteration after ii > 200, candidate 1 (never below 10000) its final state (tmp_info None,
    fittingop.py:464), candidate 2 ties -> the first; iterations ii <= 200 are never candidates even when lower."""
    from saga_amd import _lib
    from saga_amd.fitting import FittingOP
    from saga_amd.vposer import VPoserDecoder
    B, n3 = 3, 206
    op = FittingOP({'batch_size': B, 'device': 'cuda', 'body_model_path': body_model_dir, 'vposer': VPoserDecoder(seed=0),
                    'genders': ('male',), 'num_iter': [1, 1, n3], 'record_history': False})
    g = torch.Generator().manual_seed(5)
    n_total = 2 + n3
    losses = torch.full((n_total, B), 50.0)
    losses[2 + 100, :] = 0.001                       # lower than anything later, but ii = 100 is not a candidate
    losses[2 + 201:, 0] = torch.tensor([3., 2., 1., 2.5, 4.])           # best at ii = 203
    losses[2 + 201:, 1] = 20000.0
    losses[2 + 201:, 2] = torch.tensor([2., 2., 2., 2., 2.])            # first of the ties: ii = 201
    vars_after = 0.1 * torch.randn(n_total, B, _lib.VAR_STRIDE, generator=g)
    vars_after[..., _lib.VAR_SLICES['glo_rot_rec'][0]:_lib.VAR_SLICES['glo_rot_rec'][1]] += torch.tensor([-1., 0, 0, 0, 0, 1])
    markers = torch.randn(n_total, B, 143, 3, generator=g)
    state = {'it': 0}

    def fake_step(ss, prob, terms=None):
        it = state['it']
        state['it'] += 1
        prob['loss'].copy_(losses[it])
        prob['markers_rec'].copy_(markers[it])
        op.vars.copy_(vars_after[it])             # "after the step"
        return prob['loss'], prob['markers_rec']

    op.native_step = fake_step
    z = lambda *s: torch.zeros(*s, device='cuda')
    mfit, params, _ = op.fitting(z(B, 143, 3), z(B, 2048), z(B, 143), z(B, 2048, 3), z(B, 2048, 3), 'male')
    assert state['it'] == n_total
    pick = [2 + 203, n_total - 1, 2 + 201]
    want_vars = torch.stack([vars_after[pick[b], b] for b in range(B)]).cuda()
    op.bm = op.bm_male
    want = op.decode_vars(want_vars)
    for k in want:
        assert torch.equal(params[-1][k], want[k]), k
    for b in range(B):
        assert np.array_equal(mfit[-1][b], markers[pick[b], b].numpy()), b
    # stages 1 and 2 return their final states
    for s, it in ((0, 0), (1, 1)):
        w = op.decode_vars(vars_after[it].cuda())
        for k in w:
            assert torch.equal(params[s][k], w[k]), (s, k)


def _problem(B, seed=0):
    from saga_amd.synthetic import make_grasp_targets, make_object_cloud
    vo, no, _, _ = make_object_cloud(B, seed=seed)
    mk, mc, oc = make_grasp_targets(B, vo, seed=4815)
    return dict(markers=mk, mc=mc, oc=oc, verts_o=vo, normals_o=no)


def _fit_gpu(body_model_dir, d, num_iter, lr=0.008, **kw):
    from saga_amd.fitting import FittingOP
    from saga_amd.vposer import VPoserDecoder
    vp = VPoserDecoder(seed=0)
    B = d['markers'].shape[0]
    op = FittingOP(dict({'batch_size': B, 'device': 'cuda', 'body_model_path': body_model_dir, 'vposer': vp,
                         'genders': ('male',), 'num_iter': list(num_iter), 'init_lr_h': lr}, **kw))
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    final, reset = {}, op.reset

    def reset_and_keep():                # fitting() ends with reset(): keep the variables' FINAL state (decoded) from before it
        final.update(op.decode_vars())
        reset()

    op.reset = reset_and_keep
    out = op.fitting(cu(d['markers']), cu(d['oc']), cu(d['mc']), cu(d['verts_o']), cu(d['normals_o']), 'male')
    op.final_params = final
    return vp, out + (final,)


def _first_argmin_after_200(total3):
    """Index ii of the iteration the rule keeps: first strict minimum over ii > 200 (below 10000)."""
    best, pick = 10000.0, None
    for ii in range(201, total3.shape[0]):
        if total3[ii] < best:
            best, pick = total3[ii], ii
    return pick


def test_fit_reaching_the_best_of_iteration_rule_matches_oracle(body_model_dir):
    """num_iter = (40, 60, 230): stage 3 runs past ii = 200, so saga_fit_best_update fires 29 times.
    (i) the RETURNED stage-3 result is the snapshot of the best iteration -- variables after that iteration's step, markers
    from before it -- shown bit-exactly by replaying the (deterministic) fit up to that iteration;
    (ii) loss curves and stage results against the oracle's loop where such a comparison exists: stages 1 and 2 (100 Adam
    steps) .  In stage 3 the synthetic problem is chaotic -- the oracle's OWN float32
    and float64 runs are > 10 % apart there (scripts/exp/traj_drift.py) -- so no tolerance exists to test;
    (iii) the oracle's returned result obeys the same rule on its own loss history (pins the oracle's restatement of :429-447)."""
    from saga_amd.fitting import load_grasp_indices
    B, num_iter = 2, (40, 60, 230)
    n12 = num_iter[0] + num_iter[1]
    d = _problem(B)
    vp, (mfit, params, save_loss, final) = _fit_gpu(body_model_dir, d, num_iter)
    total = save_loss['total']
    assert total.shape == (sum(num_iter), B)

    # (i) replay: a fit stopped right after iteration ii* ends in exactly the snapshot state
    picks = [_first_argmin_after_200(total[n12:, b]) for b in range(B)]
    assert all(p is not None for p in picks)
    for b in range(B):
        _, (mfit_r, params_r, sl_r, _) = _fit_gpu(body_model_dir, d, (num_iter[0], num_iter[1], picks[b] + 1))
        assert np.array_equal(sl_r['total'], total[:n12 + picks[b] + 1]), 'the fit is not bit-reproducible'
        for k in params[-1]:
            assert torch.equal(params[-1][k][b], params_r[-1][k][b]), (b, k)
        assert np.array_equal(mfit[-1][b], mfit_r[-1][b]), b
    # the snapshot of at least one candidate is NOT the final state (else this test would not tell them apart)
    differs = [any(not torch.equal(params[-1][k][b], final[k][b]) for k in params[-1]) for b in range(B)]
    assert [p < num_iter[2] - 1 for p in picks] == differs and any(differs), (picks, differs)

    # (ii) + (iii) oracle
    idx_np = load_grasp_indices()
    idx = {'markers_143': torch.from_numpy(idx_np['markers_143']), 'foot': torch.from_numpy(idx_np['foot']),
           'rhand': torch.from_numpy(idx_np['rhand']), 'v_weights': torch.from_numpy(idx_np['rhand_weight'])}
    bm = load_oracle_model(body_model_dir, B, dtype=torch.float32)
    t = lambda a: torch.from_numpy(a)
    snaps = {}

    def record(ss, ii, ld, cnt, snapshot, markers_rec):
        if ss == 2 and ii > 200:
            snaps[ii] = (snapshot(), markers_rec.clone())

    p_ref, m_ref, hist = fitting_ref.fitting(bm, vp.weights_dict(torch.float32), idx, t(d['markers']), t(d['oc']), t(d['mc']),
                                             t(d['verts_o']), t(d['normals_o']), num_iter=num_iter, record=record)
    hist = torch.stack(hist).numpy()
    np.testing.assert_allclose(total[0], hist[0], rtol=1e-4)
    n1 = num_iter[0]
    np.testing.assert_allclose(total[:n1], hist[:n1], rtol=2e-3)               # stage 1 (40 Adam steps): measured 3e-7
    np.testing.assert_allclose(total[n1:n12], hist[n1:n12], rtol=2e-2)         # stage 2 (betas + vpose free): measured 5e-3
    # (no stage-3 bar: on the synthetic mesh -- random vertices, random faces -- the collision terms flip sign under mm-level
    # moves, so the 5e-3 difference the two runs have accumulated by then is already a 20 % difference in the first stage-3
    # loss; from IDENTICAL states the stage-3 loss agrees to 1e-4, test_gpu_fitting.py::test_loss_and_gradients_match_oracle)
    for k in params[0]:
        np.testing.assert_allclose(params[0][k].cpu().numpy(), p_ref[0][k].numpy(), atol=5e-3, err_msg='stage 1 %s' % k)
    np.testing.assert_allclose(mfit[0], m_ref[0].numpy(), atol=5e-3)
    # stage 2 frees betas and the body pose: Adam's sign-like steps on near-zero gradient components send the two float32
    # runs along different flat directions of the synthetic problem (measured after 60 steps: 1.4 cm in transl, up to 10 cm
    # in single markers, at losses 5e-3 apart) -- the loss curve above is the stage-2 bar
    picks_ref = [_first_argmin_after_200(hist[n12:, b]) for b in range(B)]
    for b in range(B):
        for k in p_ref[-1]:
            assert torch.equal(p_ref[-1][k][b], snaps[picks_ref[b]][0][k][b]), k
        assert torch.equal(m_ref[-1][b], snaps[picks_ref[b]][1][b])


def _config1_inputs():
    z = np.load(os.path.join(GOLDEN, 'vae_samples.npz'), allow_pickle=False)
    n = z['markers_gen'].shape[0]
    go = np.zeros((n, 3), np.float32)
    go[:, 2] = z['yaw']                           # object yaw as axis-angle (opt_grasppose.py:31-38)
    return {'verts_object': z['verts_object'], 'normal_object': z['normal_object'], 'global_orient_object': go,
            'transf_transl': z['transf_transl'], 'markers_gen': z['markers_gen'], 'object_contact_gen': z['object_contact_gen'],
            'markers_contact_gen': z['markers_contact_gen']}


def test_config1_inputs_full_schedule_through_pose_opt(body_model_dir, tmp_path):
    """BASELINE config 1: the 10 VAE-sampled objects (tests/golden/vae_samples.npz, generated by the reference's own
    FullBodyGraspNet) through pose_opt with the reference's schedule num_iter = [300, 400, 500] (opt_grasppose.py:180) as ONE
    batch of 10 -- 1,200 iterations, the best-of-iteration rule live for ii = 201..499.  Checks: reference .npz schema, the
    returned stage-3 state is each candidate's best iteration (replayed for one candidate), losses fall, nothing is NaN."""
    from saga_amd.fitting import FittingOP
    from saga_amd.opt_grasppose import BODY_KEYS, pose_opt
    from saga_amd.vposer import VPoserDecoder
    res = _config1_inputs()
    n = res['markers_gen'].shape[0]
    assert n == 10
    num_iter = [300, 400, 500]
    mk_op = lambda ni: FittingOP({'batch_size': n, 'device': 'cuda', 'body_model_path': body_model_dir, 'vposer': VPoserDecoder(seed=0),
                                  'genders': ('male',), 'num_iter': ni})
    op = mk_op(num_iter)
    hist = {}
    fit = op.fitting

    def keep(*a, **k):
        out = fit(*a, **k)
        hist['loss'] = out[2]
        return out

    op.fitting = keep
    out = pose_opt(op, res, 1, 'camera', 'male', save_dir=str(tmp_path))
    total = hist['loss']['total']
    assert total.shape == (1200, n) and np.isfinite(total).all()
    # stage 3 adds the contact / collision terms to the total, so each stage is compared with its own start
    assert (hist['loss']['rec'][699] < hist['loss']['rec'][0]).all() and (total[700:].min(axis=0) < total[700]).all()
    z = np.load(os.path.join(str(tmp_path), 'fitting_results.npz'), allow_pickle=True)
    assert z['markers'].shape == (n, 143, 3) and z['markers_fit'].shape == (n, 143, 3)
    body = z['body'].item()
    assert list(body.keys()) == BODY_KEYS and body['body_pose'].shape == (n, 63) and body['betas'].shape == (n, 10)
    assert all(np.isfinite(v).all() for v in body.values())
    # the saved state is the best stage-3 iteration of every candidate: replay candidate b's fit up to its pick
    picks = [_first_argmin_after_200(total[700:, b]) for b in range(n)]
    assert all(p is not None for p in picks)
    b = int(np.argmin(picks))                          # the earliest pick: a snapshot that differs most from the final state
    op_r = mk_op([300, 400, picks[b] + 1])
    out_r = pose_opt(op_r, res, 1, 'camera', 'male')
    for k in BODY_KEYS:
        assert np.array_equal(out['body'][k][b], out_r['body'][k][b]), k
    assert np.array_equal(out['markers_fit'][b], out_r['markers_fit'][b])
    assert total[700 + picks[b], b] == total[700 + 201:, b].min()


@pytest.mark.skipif(bool(os.environ.get('SAGA_SKIP_FULL_ORACLE')), reason='~45 s of CPU oracle time on the GPU box (16 threads); SAGA_SKIP_FULL_ORACLE=1 skips it')
def test_config1_full_schedule_against_oracle(body_model_dir):
    """Two of the config-1 objects, full 300/400/500 schedule, HIP path against the oracle's loop (the oracle needs
    ~45 s on the GPU box's host cores).  Bars: first iteration 1e-4, stage-1 curve 5e-3, stage-2 curve 1e-1, stage 3 by the level it settles at."""
    from saga_amd.fitting import load_grasp_indices
    res = _config1_inputs()
    B = 2
    d = dict(markers=res['markers_gen'][:B], mc=res['markers_contact_gen'][:B], oc=res['object_contact_gen'][:B],
             verts_o=res['verts_object'][:B], normals_o=res['normal_object'][:B])
    num_iter = (300, 400, 500)
    vp, (mfit, params, save_loss, _) = _fit_gpu(body_model_dir, d, num_iter)
    idx_np = load_grasp_indices()
    idx = {'markers_143': torch.from_numpy(idx_np['markers_143']), 'foot': torch.from_numpy(idx_np['foot']),
           'rhand': torch.from_numpy(idx_np['rhand']), 'v_weights': torch.from_numpy(idx_np['rhand_weight'])}
    bm = load_oracle_model(body_model_dir, B, dtype=torch.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    p_ref, m_ref, hist = fitting_ref.fitting(bm, vp.weights_dict(torch.float32), idx, t(d['markers']), t(d['oc']), t(d['mc']),
                                             t(d['verts_o']), t(d['normals_o']), num_iter=num_iter)
    hist = torch.stack(hist).numpy()
    total = save_loss['total']
    np.testing.assert_allclose(total[0], hist[0], rtol=1e-4)
    np.testing.assert_allclose(total[:300], hist[:300], rtol=5e-3)          # stage 1: measured 2e-3
    np.testing.assert_allclose(total[300:700], hist[300:700], rtol=1e-1)     # stage 2 (betas + vpose free): measured 5e-2
    drift = np.abs(total - hist) / np.abs(hist)
    print('full-schedule drift vs oracle: stage 1 %.3g, stage 2 %.3g, stage 3 %.3g (max rel)' % (drift[:300].max(), drift[300:700].max(), drift[700:].max()))
    # stage 3 is chaotic on the synthetic inputs (oracle float32 vs float64: > 50 % apart): compare what the fit is judged by,
    # the loss level it settles at
    np.testing.assert_allclose(total[-100:].mean(axis=0), hist[-100:].mean(axis=0), rtol=0.35)


def _oracle_from_state(bm64, vp, idx, d, packed_vars, stage, **kw):
    """The fp64 oracle's loss, counts, aux and variable gradients evaluated FROM a packed variable state of the HIP path."""
    from oracle import loss_ref
    from saga_amd import _lib
    var = {k: packed_vars[:, a:b].detach().cpu().double().clone().requires_grad_(True) for k, (a, b) in _lib.VAR_SLICES.items()}
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()
    total, ld, cnt, aux = loss_ref.loss_batched(bm64, var, vp.weights_dict(torch.float64), idx, t(d['markers']), t(d['verts_o']),
                                                t(d['normals_o']), t(d['oc']), t(d['mc']), stage, **kw)
    total.sum().backward()
    return total, ld, cnt, aux, var


def _check_nn_indices(name, got_idx, q32, db32, ref_idx, q64, db64):
    """NN indices of the fused kernel at a visited state.  (a) On the HIP path's OWN float32 points they equal the C oracle's
    sequential strict-'<' scan bit for bit.  (b) Against the fp64 oracle (whose points differ by the 1e-5 m vertex tolerance) they
    are equal except at near-ties: the oracle's two candidate distances then differ by less than the vertex tolerance allows."""
    from oracle import chamfer_ref
    _, want = chamfer_ref._scan(np.ascontiguousarray(q32), np.ascontiguousarray(db32))
    assert np.array_equal(got_idx, want), '%s: %d of %d indices differ from the exact float32 scan' % (name, (got_idx != want).sum(), want.size)
    diff = np.argwhere(got_idx != ref_idx)
    for b, p in diff:
        da = np.linalg.norm(q64[b, p] - db64[b, got_idx[b, p]])
        dr = np.linalg.norm(q64[b, p] - db64[b, ref_idx[b, p]])
        assert abs(da - dr) <= 4e-5, '%s: candidate %d point %d: index %d (d=%.9g) vs oracle %d (d=%.9g) is not a near-tie' % (
            name, b, p, got_idx[b, p], da, ref_idx[b, p], dr)
    return len(diff)


def test_config1_teacher_forced_late_states_against_oracle(body_model_dir):
    """the states a fit actually VISITS -- large poses, hand in contact, NN near-ties, non-zero collision
    counts -- compared with the oracle at iteration-0 bars.  Two config-1 objects run the full 300/400/500 schedule on the HIP
    path; at iterations 299 (stage 1), 699 (stage 2), 950 and 1199 (stage 3) the variables the HIP path holds BEFORE the
    iteration are copied into the fp64 oracle, which evaluates THAT iteration: no trajectory, so no chaos.  Bars: the 10
    loss_dict keys 1e-4 relative, the 4 counts +-2, NN indices exact (see _check_nn_indices), gradient rows 2e-3."""
    from oracle import loss_ref
    from saga_amd import _lib
    from saga_amd.fitting import FittingOP, load_grasp_indices
    from saga_amd.vposer import VPoserDecoder
    res = _config1_inputs()
    B, O = 2, res['verts_object'].shape[1]
    d = dict(markers=res['markers_gen'][:B], mc=res['markers_contact_gen'][:B], oc=res['object_contact_gen'][:B],
             verts_o=res['verts_object'][:B], normals_o=res['normal_object'][:B])
    num_iter = (300, 400, 500)
    checks = {299: 0, 699: 1, 950: 2, 1199: 2}
    vp = VPoserDecoder(seed=0)
    op = FittingOP({'batch_size': B, 'device': 'cuda', 'body_model_path': body_model_dir, 'vposer': vp, 'genders': ('male',),
                    'num_iter': list(num_iter)})
    real, state, cap = op.native_step, {'it': 0}, {}
    H, M = op.tables.n_hand, op.tables.n_markers

    def step(ss, prob, terms=None):
        it = state['it']
        state['it'] += 1
        if it not in checks:
            return real(ss, prob, terms)
        assert checks[it] == ss
        before = op.vars.clone()
        out = real(ss, prob, terms)
        w = lambda n, s, dt=torch.float32: op.workspace_view(n, O, s, dt).clone().cpu().numpy()
        cap[it] = dict(vars=before, terms=terms.clone().cpu().numpy(), grad=w('grad', (B, _lib.VAR_STRIDE)), g_z=w('g_z', (B, 128)),
                       idx_m2o=w('idx_m2o', (B, M), torch.int32), idx_h2o=w('idx_h2o', (B, H), torch.int32),
                       idx_o2h=w('idx_o2h', (B, O), torch.int32), hand=w('hand_rec', (B, H, 3)), markers=out[1].clone().cpu().numpy(),
                       h2o=w('h2o_signed', (B, H)), o2h=w('o2h_signed', (B, O)))
        return out

    op.native_step = step
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    op.fitting(cu(d['markers']), cu(d['oc']), cu(d['mc']), cu(d['verts_o']), cu(d['normals_o']), 'male')
    assert sorted(cap) == sorted(checks)

    idx_np = load_grasp_indices()
    idx = {'markers_143': torch.from_numpy(idx_np['markers_143']), 'foot': torch.from_numpy(idx_np['foot']),
           'rhand': torch.from_numpy(idx_np['rhand']), 'v_weights': torch.from_numpy(idx_np['rhand_weight'])}
    bm64 = load_oracle_model(body_model_dir, B, dtype=torch.float64)
    owned = {0: ['transl_rec', 'glo_rot_rec'], 1: ['betas', 'transl_rec', 'glo_rot_rec', 'vpose_rec'],
             2: ['vpose_rec', 'hand_pose', 'eye_pose']}
    obj64 = d['verts_o'].astype(np.float64)
    seen_collisions, errs = 0, []
    for it, ss in sorted(checks.items()):
        c = cap[it]
        total, ld, cnt, aux, var = _oracle_from_state(bm64, vp, idx, d, c['vars'], ss)
        # the state is a visited one, not the seeded start: the body has moved and (stage 3) touches the object
        assert np.abs(c['vars'][:, 96:99].cpu().numpy()).max() > 0.05
        for i, k in enumerate(loss_ref.LOSS_KEYS):
            np.testing.assert_allclose(c['terms'][:, i], ld[k].detach().numpy(), rtol=1e-4, atol=1e-9, err_msg='%s at iteration %d' % (k, it))
        for i, k in enumerate(loss_ref.COUNT_KEYS):
            assert np.abs(c['terms'][:, 10 + i] - cnt[k].numpy()).max() <= 2, (k, it, c['terms'][:, 10 + i], cnt[k])
        seen_collisions += int(c['terms'][:, 10].sum() + c['terms'][:, 11].sum())
        assert np.abs(c['markers'] - aux['markers_rec'].detach().numpy()).max() < 1e-5
        assert np.abs(c['hand'] - aux['rhand_rec'].detach().numpy()).max() < 1e-5
        hand64, mk64 = aux['rhand_rec'].detach().numpy(), aux['markers_rec'].detach().numpy()
        n_ties = _check_nn_indices('markers->object @%d' % it, c['idx_m2o'], c['markers'], d['verts_o'], aux['idx_m2o'].numpy(), mk64, obj64)
        n_ties += _check_nn_indices('hand->object @%d' % it, c['idx_h2o'], c['hand'], d['verts_o'], aux['idx_h2o'].numpy(), hand64, obj64)
        n_ties += _check_nn_indices('object->hand @%d' % it, c['idx_o2h'], d['verts_o'], c['hand'], aux['idx_o2h'].numpy(), obj64, hand64)
        # signed distances: equal to 1e-5 m except where the sign itself is decided by a dot product at rounding level
        n_flips = 0
        for key, ref in (('h2o', aux['h2o']), ('o2h', aux['o2h'])):
            a, r = c[key], ref.detach().numpy()
            bad = np.abs(a - r) > 2e-5
            assert bad.sum() <= 2 and np.all(np.abs(np.abs(a[bad]) - np.abs(r[bad])) <= 2e-5), (key, it, int(bad.sum()))
            n_flips += int(bad.sum())
        if n_ties:
            # a neighbour that differs at a verified near-tie sends its gradient along another unit vector (and, object -> hand,
            # into another hand vertex): gradients are compared downstream of the SAME tie decisions, the HIP path's
            total, ld, cnt, aux, var = _oracle_from_state(bm64, vp, idx, d, c['vars'], ss, force_idx={k_: c[k_] for k_ in ('idx_m2o', 'idx_h2o', 'idx_o2h')})
            for i, k in enumerate(loss_ref.LOSS_KEYS):
                np.testing.assert_allclose(c['terms'][:, i], ld[k].detach().numpy(), rtol=1e-4, atol=1e-9, err_msg='%s at iteration %d (forced ties)' % (k, it))
        for k in owned[ss]:
            a, b_ = _lib.VAR_SLICES[k]
            got = c['g_z'][:, :32] if k == 'vpose_rec' else c['grad'][:, a:b_]
            if k == 'vpose_rec':
                got = got + 2e-4 / 32 * c['vars'][:, a:b_].cpu().numpy()       # the regulariser's gradient is added in the Adam kernel
            e = np.abs(got - var[k].grad.numpy()).max() / np.abs(var[k].grad.numpy()).max()
            print('   gradient of %s @%d: rel err %.3g (|g|max %.3g, sign flips %d)' % (k, it, e, np.abs(var[k].grad.numpy()).max(), n_flips))
            errs.append(('gradient of %s at iteration %d' % (k, it), e))
        print('teacher-forced @%d (stage %d): losses 1e-4, counts %s, %d near-tie indices' % (it, ss, c['terms'][:, 10:14].tolist(), n_ties))
    for name, e in errs:
        assert e < 2e-3, (name, e)
    assert seen_collisions > 0, 'no visited state had a vertex in collision: the late-state check is not exercising the collision terms'
