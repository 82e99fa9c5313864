"""CPU restatement of the 3-stage Adam loop ``FittingOP.fitting``
(``WholeGraspPose/models/fittingop.py:71-84,129-135,343-499``).  TEST INFRASTRUCTURE.

Batched as B independent problems: the objective handed to ``backward`` is the SUM of the
per-candidate losses, so every candidate sees exactly the gradient (and, Adam being
element-wise, the update sequence) it would see at batch_size = 1.
"""
import copy

import torch

from .loss_ref import body_params, loss_batched


def init_vars(B, hand_ncomps=24, dtype=torch.float32):
    """fittingop.py:71-76 / reset() :129-135."""
    v = {
        'betas': torch.zeros(B, 10, dtype=dtype),
        'transl_rec': torch.zeros(B, 3, dtype=dtype),
        'glo_rot_rec': torch.tensor([[-1, 0, 0, 0, 0, 1]], dtype=dtype).repeat(B, 1),
        'vpose_rec': torch.zeros(B, 32, dtype=dtype),
        'hand_pose': torch.zeros(B, 2 * hand_ncomps, dtype=dtype),
        'eye_pose': torch.zeros(B, 6, dtype=dtype),
    }
    for t in v.values():
        t.requires_grad_(True)
    return v


def fitting(bm, vposer_w, idx, body_markers, object_contact, markers_contact, verts_object,
            normal_object, num_iter=(300, 400, 500), init_lr_h=0.008, var=None,
            only_rec=False, contact_loss='contact', hand_ncomps=24, record=None):
    """-> (smplxparams_list[3], markers_fit_list[3], loss_history[list of [B] tensors]).

    ``record(ss, ii, loss_dict, counts, snapshot_fn, markers_rec)`` is called after every ``step()``: ``snapshot_fn()`` gives
    the SMPL-X parameter dict of the variables as they are then (after the update); ``markers_rec`` is from before it."""
    B = body_markers.shape[0]
    var = init_vars(B, hand_ncomps, body_markers.dtype) if var is None else var
    opts = [
        torch.optim.Adam([var['transl_rec'], var['glo_rot_rec']], lr=init_lr_h * 2.0),                       # :77
        torch.optim.Adam([var['betas'], var['transl_rec'], var['glo_rot_rec'], var['vpose_rec']], lr=init_lr_h * 1.5),  # :78
        torch.optim.Adam([var['vpose_rec'], var['hand_pose'], var['eye_pose']], lr=init_lr_h),               # :79
    ]
    history = []
    params_list, markers_list = [], []
    best = torch.full((B,), 10000.0, dtype=body_markers.dtype)                          # :354
    best_params, best_markers = None, None

    def snapshot():
        with torch.no_grad():
            d = {k: v.detach().clone() for k, v in var.items()}
            return {k: v.detach().clone() for k, v in body_params(d, vposer_w, hand_ncomps).items()}

    for ss, opt in enumerate(opts):
        for ii in range(num_iter[ss]):
            opt.zero_grad()
            total, ld, cnt, aux = loss_batched(bm, var, vposer_w, idx, body_markers, verts_object,
                                               normal_object, object_contact, markers_contact, ss,
                                               only_rec=only_rec, contact_loss=contact_loss,
                                               hand_ncomps=hand_ncomps)
            total.sum().backward()
            opt.step()
            history.append(total.detach().clone())
            if record is not None:
                record(ss, ii, ld, cnt, snapshot, aux['markers_rec'].detach())
            if not only_rec and ss == 2 and ii > 200:                                   # :430
                better = total.detach() < best
                if better.any():
                    snap = snapshot()            # params AFTER step(), markers from BEFORE (:433-443)
                    mk = aux['markers_rec'].detach()
                    if best_params is None:
                        best_params = {k: v.clone() for k, v in snap.items()}
                        best_markers = mk.clone()
                    for k in snap:
                        best_params[k][better] = snap[k][better]
                    best_markers[better] = mk[better]
                    best = torch.where(better, total.detach(), best)
        last_of_stage3 = (ss == 2 and not only_rec and best_params is not None and num_iter[2] > 0)
        if last_of_stage3:
            # early_stop is set on the last iteration of stage 3 (:458-460) -> tmp params kept (:486); a candidate whose
            # loss never went below 10000 has tmp_info None at batch size 1 and keeps its final state (:464)
            never = best >= 10000.0
            final, mk_final = snapshot(), aux['markers_rec'].detach()
            params_list.append({k: torch.where(never.view(-1, *[1] * (v.dim() - 1)), final[k], v) for k, v in best_params.items()})
            markers_list.append(torch.where(never[:, None, None], mk_final, best_markers))
        else:
            params_list.append(snapshot())                                              # :466-475
            markers_list.append(aux['markers_rec'].detach().clone())
    return params_list, markers_list, history
