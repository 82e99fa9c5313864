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
