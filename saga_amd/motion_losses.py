"""Geometry losses of GraspMotion-Opt stage 2 (``MotionFill/models/fittingop.py:132-280,342-414``) on the HIP
operators: frames are the batch dimension of the SMPL-X forward, of the vertex-normal kernel (Open3D per frame on the
host in the reference, ``:137-146``) and of the nearest-neighbour search.  One sequence = [T, ...] tensors, as in the
reference; nothing leaves the device and nothing synchronises (the reference's ``.item()`` tests in the skating loss
become masked means).  The learned smoothness prior (``smooth_loss`` / ``body_smooth_loss``: a convolutional
auto-encoder) is a network, not part of this path.
"""
import torch

from .geometry import point2point_signed, vertex_normals


def foot_groups(idx):
    """{'right_heel','right_toe','left_heel','left_toe'} vertex ids from ``load_grasp_indices()`` output."""
    foot, sizes = idx['foot'], [int(s) for s in idx['foot_sizes']]
    out, o = {}, 0
    for name, n in zip(['right_heel', 'right_toe', 'left_heel', 'left_toe'], sizes):
        out[name] = torch.as_tensor(foot[o:o + n], dtype=torch.long)
        o += n
    return out


def contact_loss(bm, start_t, body_verts_T, markers_143, rhand_verts, contact_object, contact_markers, normal_object,
                 verts_object, n_frames=None):
    """``contact_loss`` (:132-176).  body_verts_T [T,V,3]; markers_143 / rhand_verts: long index tensors;
    verts_object, normal_object [T-start_t,O,3]; contact_object [.,O], contact_markers [.,143] (broadcast over frames).
    -> (loss_contact, state) where state holds what ``collision_loss`` / ``angle_loss`` read from ``self``."""
    T = body_verts_T.shape[0]
    n_frames = T if n_frames is None else n_frames            # the reference hard-codes 61
    tail = body_verts_T[start_t:]
    normals = vertex_normals(bm, tail)                        # all vertices of every frame, on the device
    rh_normals_full = normals.index_select(1, rhand_verts)
    rh_normals = normals.index_select(1, markers_143[-22:])   # right-hand palm markers (:133)
    body_markers = tail.index_select(1, markers_143)
    rh_verts = tail.index_select(1, rhand_verts)
    vo, no = verts_object.float(), normal_object.float()
    _o2h_m, h2o_signed_marker, _, _, _o2h_vec, h2o_marker_vector = point2point_signed(body_markers, vo, None, no, return_vector=True)
    o2h_signed, h2o_signed, _, _ = point2point_signed(rh_verts, vo, rh_normals_full, no)
    w = torch.zeros(n_frames - start_t, 1, device=tail.device)
    w[-5:] = 1                                                # attraction only over the last five frames (:163-165)
    loss_obj = (o2h_signed.abs() * contact_object * w).mean()
    loss_mk = (h2o_signed_marker.abs() * contact_markers * w).mean()
    state = dict(o2h_signed=o2h_signed, h2o_signed=h2o_signed, h2o_signed_marker=h2o_signed_marker,
                 h2o_marker_vector=h2o_marker_vector, rh_normals=rh_normals, rh_normals_norm=rh_normals.norm(dim=-1))
    return loss_obj + loss_mk, state


def collision_loss(state):
    """``collision_loss`` (:178-191): only the object -> hand penetration term is used (weight 20 below -3 mm)."""
    o2h = state['o2h_signed']
    mask = torch.where(o2h < -0.003, torch.full_like(o2h, 20.0), torch.zeros_like(o2h))
    return 5 * (o2h.abs() * mask).mean(dim=-1).mean()


def skating_loss(body_verts_T, contact_lbl_rec, groups, vel_thres=0.1):
    """``skating_loss`` (:193-263): mean speed (30 fps) above ``vel_thres`` of the heel / toe vertices on the frames whose
    contact label is 1; a group with no such vertex contributes 0.  contact_lbl_rec [T,4] = (left heel, right heel,
    left toe, right toe)."""
    vel = (body_verts_T[1:] - body_verts_T[:-1]) * 30
    total = body_verts_T.new_zeros(())
    for col, name in enumerate(['left_heel', 'right_heel', 'left_toe', 'right_toe']):
        ids = groups[name].to(body_verts_T.device)
        speed = vel.index_select(1, ids).norm(dim=-1)                                   # [T-1, n]
        on = (contact_lbl_rec[:-1, col] == 1)[:, None] & (speed > vel_thres)
        cnt = on.sum()
        total = total + torch.where(cnt > 0, (speed.abs() * on).sum() / cnt.clamp(min=1), speed.new_zeros(()))
    return total


def angle_loss(state, start_t, n_frames):
    """``angle_loss`` (:265-280): the palm normals of the right hand should face the object where a palm marker is
    within 1 cm of it; weight decays quadratically over the optimised frames."""
    vec = -state['h2o_marker_vector'][:, -22:]
    cos = (state['rh_normals'] * vec).sum(-1) / (state['rh_normals_norm'] * vec.norm(dim=-1))
    n = n_frames - start_t
    weight = 1 - ((torch.arange(n, device=vec.device) + 1).true_divide(n).view(-1, 1)) ** 2
    mask = (state['h2o_signed_marker'] < 0.01).int()
    return (mask[:, -22:] * (cos - 1) ** 2 * weight).mean()


def hand_smooth_loss(body_verts_T, start_t, rhand_verts):
    """``loss_rh_smooth`` (:386-387)."""
    rh = body_verts_T[start_t:].index_select(1, rhand_verts)
    return (rh[1:] - rh[:-1]).norm(dim=-1).mean()
