"""CPU restatement of GraspMotion-Opt's two-stage fit for ONE sequence, after
``MotionFill/models/fittingop.py:40-61,283-428`` with ``convert_to_3D_rot`` / ``convert_to_6D_all``
(``utils/como/como_utils.py:295-314``) and ``gen_body_mesh_v1`` (``utils/utils_body.py:93-106``).  TEST INFRASTRUCTURE.

PARITY UNPINNED: rests on the unpinned smplx / VPoser / chamfer / Open3D-normal restatements and on
torchgeometry's ``angle_axis_to_rotation_matrix`` (not in the reference tree; restated from its published source, used
only for the initial orientation).  The learned smoothness prior (``:70-130``) is left out on both sides.
"""
import torch

from . import motion_ref
from .rotations_ref import cont2rotmat, rotmat2aa
from .vposer_ref import vposer_decode

WEIGHTS = dict(stage1_weight_loss_rec_markers=1.0, stage1_weight_loss_vposer=0.02, stage1_weight_loss_shape=0.01,
               stage1_weight_loss_hand=0.01,
               stage2_weight_loss_rec_markers=0.1, stage2_weight_loss_end_markers_fit=10, stage2_weight_loss_vposer=0.02,
               stage2_weight_loss_hand=0.02, stage2_weight_loss_skating=0.05, stage2_weight_loss_collision=10,
               stage2_weight_loss_contact=60, stage2_weight_loss_hand_smooth=1, stage2_weight_loss_hand_angle=1)  # opt_graspmotion.py:377-396


def aa2matrot(aa, eps=1e-6):
    """torchgeometry.angle_axis_to_rotation_matrix, 3x3 part (como_utils.py:219-226)."""
    aa = aa.reshape(-1, 3)
    theta2 = (aa * aa).sum(-1, keepdim=True)
    theta = torch.sqrt(theta2)
    w = aa / (theta + eps)
    wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
    c, s = torch.cos(theta), torch.sin(theta)
    r = [c + wx * wx * (1 - c), wx * wy * (1 - c) - wz * s, wy * s + wx * wz * (1 - c),
         wz * s + wx * wy * (1 - c), c + wy * wy * (1 - c), -wx * s + wy * wz * (1 - c),
         -wy * s + wx * wz * (1 - c), wx * s + wy * wz * (1 - c), c + wz * wz * (1 - c)]
    normal = torch.cat(r, dim=1)
    rx, ry, rz = aa[:, 0:1], aa[:, 1:2], aa[:, 2:3]
    one = torch.ones_like(rx)
    taylor = torch.cat([one, -rz, ry, rz, one, -rx, -ry, rx, one], dim=1)
    return torch.where(theta2 > eps, normal, taylor).reshape(-1, 3, 3)


def convert_to_6D_all(aa):                                                                  # como_utils.py:311-314
    return aa2matrot(aa)[:, :, :-1].reshape(-1, 6)


def convert_to_3D_rot(x):                                                                   # como_utils.py:295-307
    return torch.cat([x[:, :3], rotmat2aa(cont2rotmat(x[:, 3:9])), x[:, 9:]], dim=-1)


def gen_body_mesh_v1(p, bm, vposer_w):                                                      # utils_body.py:93-106
    bs = p.shape[0]
    return bm(transl=p[:, 0:3], global_orient=p[:, 3:6], betas=p[:, 6:16],
              body_pose=vposer_decode(vposer_w, p[:, 16:48]).reshape(bs, -1),
              left_hand_pose=p[:, 48:72], right_hand_pose=p[:, 72:])['vertices']


def set_lr(opt, lr):
    for g in opt.param_groups:
        g['lr'] = lr


def fitting_stage1(bm1, vposer_w, smplx_beta, markers_ids, body_markers_rec, num_iter, init_lr=0.1, W=WEIGHTS,
                   dtype=torch.float64):
    """:40-55 + :283-340.  bm1: SMPLXRef(batch_size=1); body_markers_rec [T,79,3]."""
    transl = torch.zeros(1, 3, dtype=dtype)
    transl[:, 1] = 0.4
    transl[:, 2] = 1.0
    rot = torch.zeros(1, 3, dtype=dtype)
    rot[:, 1] = 1.6
    rot[:, 2] = 3.14
    rot6 = convert_to_6D_all(rot)
    other = torch.zeros(1, 56 + 24, dtype=dtype)
    shape = torch.as_tensor(smplx_beta, dtype=dtype).view(1, -1)
    for v in (transl, rot6, other):
        v.requires_grad_(True)
    opt = torch.optim.Adam([transl, rot6, other], lr=init_lr)
    T = body_markers_rec.shape[0]
    out = [[], [], [], []]
    for t in range(T):
        rec_t = body_markers_rec[t:t + 1].to(dtype)
        if t > 0:
            set_lr(opt, 0.005 * 1)
        for step in range(num_iter):
            if step > 60:
                set_lr(opt, 0.01)
            if step > 80:
                set_lr(opt, 0.003)
            opt.zero_grad()
            p = convert_to_3D_rot(torch.cat([transl, rot6, shape, other], dim=-1))
            verts = gen_body_mesh_v1(p, bm1, vposer_w)
            markers = verts[:, markers_ids, :]
            loss = (W['stage1_weight_loss_rec_markers'] * torch.nn.functional.l1_loss(markers, rec_t)
                    + W['stage1_weight_loss_vposer'] * torch.mean(p[:, 16:48] ** 2)
                    + W['stage1_weight_loss_shape'] * torch.mean(p[:, 6:16] ** 2)
                    + W['stage1_weight_loss_hand'] * torch.mean(p[:, 48:] ** 2))
            loss.backward()
            opt.step()
        for lst, v in zip(out, (transl, rot6, shape, other)):
            lst.append(v.clone().detach())
    return [torch.stack(o).squeeze(1).detach() for o in out]


def fitting_stage2(bmT, faces, vposer_w, groups, markers_ids, markers_ids_143, start_vars, markers_rec_T, markers_end_new,
                   rhand_verts, contact_lbl_rec, contact_object, contact_markers, normal_object, verts_object, num_iter,
                   start_t=30, init_lr=0.01, W=WEIGHTS, record=None):
    """:57-61 + :342-414 (without ``body_smooth_loss``).  bmT: SMPLXRef(batch_size=T)."""
    transl, rot6, shape, other = [v.clone() for v in start_vars]
    for v in (transl, rot6, other):
        v.requires_grad_(True)
    T = transl.shape[0]
    opt = torch.optim.Adam([transl, rot6, other], lr=init_lr)
    for step in range(num_iter):
        if step > 150:
            set_lr(opt, 0.005)
        opt.zero_grad()
        p = convert_to_3D_rot(torch.cat([transl, rot6, shape, other], dim=-1))
        verts = gen_body_mesh_v1(p, bmT, vposer_w)
        markers = verts[:, markers_ids, :]
        mask = torch.ones(1, markers.shape[1], 1, dtype=verts.dtype)
        mask[:, 49:49 + 6] = 0
        mask[:, 33:33 + 6] = 0
        mask[:, 69:] = 0
        l1 = torch.nn.functional.l1_loss
        terms = {'rec_markers': l1(markers * mask, markers_rec_T * mask), 'end_markers_fit': l1(markers[-1], markers_end_new)}
        terms['skating'] = motion_ref.skating_loss(verts, contact_lbl_rec, groups)
        terms['contact'], state = motion_ref.contact_loss(faces, start_t, verts, markers_ids_143, rhand_verts, contact_object,
                                                          contact_markers, normal_object, verts_object, n_frames=T)
        rh = verts[start_t:, rhand_verts, :]
        terms['hand_smooth'] = torch.mean(torch.norm(rh[1:] - rh[:-1], dim=-1))
        terms['collision'] = motion_ref.collision_loss(state)
        terms['hand_angle'] = motion_ref.angle_loss(state, start_t, n_frames=T)
        terms['vposer'] = torch.mean(p[:, 16:48] ** 2)
        terms['hand'] = torch.mean(p[:, 48:] ** 2)
        total = sum(W['stage2_weight_loss_' + k] * v for k, v in terms.items())
        total.backward()
        opt.step()
        if record is not None:
            record(step, total.detach(), {k: v.detach() for k, v in terms.items()})
    return transl.detach(), rot6.detach(), shape, other.detach()
