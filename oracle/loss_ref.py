"""CPU restatement of the per-iteration loss ``FittingOP.calc_loss_contact_map``
(``WholeGraspPose/models/fittingop.py:137-244``).  TEST INFRASTRUCTURE.

The reference is written for batch_size = 1 (``view(1,-1)`` broadcasts ``:190-191``, normals of
element ``[0]`` only ``:159``).  ``loss_b1`` restates it literally for one candidate;
``loss_batched`` evaluates B independent B=1 problems at once (per-candidate means), which is
the semantics the HIP path implements (SURVEY.md §7 'Reference is B=1-shaped').
Depends on the unpinned smplx / chamfer / normals restatements: PARITY UNPINNED.
"""
import torch

from .geometry_ref import point2point_signed, vertex_normals
from .rotations_ref import cont2rotmat, rotmat2aa
from .vposer_ref import vposer_decode

LOSS_KEYS = ['total', 'rec', 'body_rec', 'marker contact', 'object contact', 'prior contact',
             'hand collision', 'object collision', 'foot', 'reg']
COUNT_KEYS = ['hand colli', 'obj colli', 'contact', 'hand markers colli']


def body_params(var, vposer_w, hand_ncomps=24):
    """fittingop.py:138-147: optimiser variables -> SMPL-X keyword arguments."""
    B = var['transl_rec'].shape[0]
    return {
        'transl': var['transl_rec'],
        'global_orient': rotmat2aa(cont2rotmat(var['glo_rot_rec'])),
        'betas': var['betas'],
        'body_pose': vposer_decode(vposer_w, var['vpose_rec']).reshape(B, -1),
        'left_hand_pose': var['hand_pose'][:, :hand_ncomps],
        'right_hand_pose': var['hand_pose'][:, hand_ncomps:],
        'leye_pose': var['eye_pose'][:, :3],
        'reye_pose': var['eye_pose'][:, 3:],
    }


def loss_batched(bm, var, vposer_w, idx, body_markers, verts_object, normal_object,
                 contacts_object, contacts_markers, stage, only_rec=False, contact_loss='contact',
                 hand_ncomps=24, force_idx=None):
    """Per-candidate losses [B] and loss_dict/counts as [B] tensors.

    bm: oracle.smplx_ref.SMPLXRef; var: dict of optimiser variables [B,.];
    idx: dict markers_143 / foot / rhand (long), v_weights [778];
    body_markers [B,143,3]; verts_object, normal_object [B,O,3];
    contacts_object [B,O]; contacts_markers [B,143].
    force_idx: None, or {'idx_m2o', 'idx_h2o', 'idx_o2h'} -> neighbour indices to use instead of the search's
    (geometry_ref.point2point_signed: tie decisions taken from the implementation under test).
    """
    bp = body_params(var, vposer_w, hand_ncomps)
    out = bm(**bp)
    verts = out['vertices']
    B = verts.shape[0]
    markers_rec = verts[:, idx['markers_143']]
    foot_rec = verts[:, idx['foot']]
    rhand_rec = verts[:, idx['rhand']]
    faces = torch.as_tensor(bm.faces)
    # :158-164  normals are detached, computed in double on the CPU, cast to float32
    nrm = vertex_normals(verts.detach().double(), faces)[:, idx['rhand']].to(torch.float32).to(verts.dtype)

    loss_rec = (markers_rec - body_markers.detach()).abs().mean(dim=(1, 2))            # :168
    loss_body_rec = (markers_rec[:, :49] - body_markers.detach()[:, :49]).abs().mean(dim=(1, 2))  # :169
    loss_foot = 0.1 * foot_rec[:, :, -1].abs().mean(dim=1)                             # :173
    loss_vpose_reg = 0.0001 * (var['vpose_rec'] ** 2).mean(dim=1)                      # :177
    loss_hand_reg = 0.0005 * (var['hand_pose'] ** 2).mean(dim=1)                       # :178
    loss_eye_reg = 0.0001 * (var['eye_pose'] ** 2).mean(dim=1)                         # :179

    vo = verts_object.to(verts.dtype)
    no = normal_object.to(verts.dtype)
    # :183 -- normal_object lands in the x_normals slot (quirk a13q-i), y_normals=None
    fi = force_idx or {}
    _o2h_marker, h2o_marker, _, idx_m2o = point2point_signed(markers_rec, vo, no, force_idx=(None, fi.get('idx_m2o')))
    o2h, h2o, idx_o2h, idx_h2o = point2point_signed(rhand_rec, vo, nrm, no,
                                                    force_idx=(fi.get('idx_o2h'), fi.get('idx_h2o')))   # :184

    loss_marker_contact = (h2o_marker.abs() * contacts_markers.view(B, -1)).mean(dim=1)   # :190
    loss_object_contact = (o2h.abs() * contacts_object.view(B, -1)).mean(dim=1)           # :191
    vw = idx['v_weights'].to(verts.dtype)
    vw2 = torch.pow(vw, 1.0 / 2.5)                                                      # :87
    v_dist = (h2o < 0.02) & (h2o > 0) & (vw2[None] > 0.7)                               # :195
    v_dist_neg = h2o < 0                                                                # :196
    v_contact = torch.where(v_dist, vw[None].expand_as(h2o), torch.zeros_like(h2o))     # :199
    v_collision = torch.where(v_dist_neg, torch.full_like(h2o, 10.0), torch.zeros_like(h2o))  # :200
    w_dist_neg = o2h < 0                                                                # :204
    w = torch.where(w_dist_neg, torch.full_like(o2h, 20.0), torch.zeros_like(o2h))      # :205-206
    loss_prior_contact = (h2o.abs() * v_contact).mean(dim=1)                            # :210
    h_collision = (h2o.abs() * v_collision).mean(dim=1)                                 # :211
    loss_dist_o = (o2h.abs() * w).mean(dim=1)                                           # :212

    on = 0.0 if only_rec else 1.0
    total = (loss_rec
             + on * (contact_loss == 'contact') * (stage > 1) * 15 * (loss_marker_contact + loss_object_contact)
             + on * (contact_loss == 'prior') * (stage > 1) * 15 * loss_prior_contact
             + on * (stage > 1) * 10 * (h_collision + loss_dist_o)
             + (stage > 1) * loss_hand_reg
             + (stage > 1) * loss_eye_reg
             + (stage > 0) * loss_vpose_reg
             + 0.1 * loss_foot)                                                          # :215-223
    loss_dict = {
        'total': total, 'rec': loss_rec, 'body_rec': loss_body_rec,
        'marker contact': loss_marker_contact, 'object contact': loss_object_contact,
        'prior contact': loss_prior_contact, 'hand collision': h_collision,
        'object collision': loss_dist_o, 'foot': loss_foot,
        'reg': loss_vpose_reg + loss_hand_reg + loss_eye_reg,
    }
    counts = {
        'hand colli': v_dist_neg.sum(dim=1), 'obj colli': w_dist_neg.sum(dim=1),
        'contact': ((h2o < 0.001) & (h2o > -0.001)).sum(dim=1),
        'hand markers colli': (h2o_marker < 0).sum(dim=1),
    }
    aux = {'markers_rec': markers_rec, 'rhand_rec': rhand_rec, 'rh_normals': nrm, 'h2o': h2o,
           'o2h': o2h, 'h2o_marker': h2o_marker, 'body_param': bp, 'vertices': verts,
           'joints': out['joints'], 'idx_m2o': idx_m2o, 'idx_h2o': idx_h2o, 'idx_o2h': idx_o2h}
    return total, loss_dict, counts, aux


def loss_b1(bm, var, vposer_w, idx, body_markers, verts_object, normal_object,
            contacts_object, contacts_markers, stage, **kw):
    """Literal batch_size=1 evaluation (global means, [1,.] masks) for cross-checking."""
    assert var['transl_rec'].shape[0] == 1
    total, d, c, aux = loss_batched(bm, var, vposer_w, idx, body_markers, verts_object, normal_object,
                                    contacts_object, contacts_markers, stage, **kw)
    return total[0], {k: v[0] for k, v in d.items()}, {k: int(v[0]) for k, v in c.items()}, aux
