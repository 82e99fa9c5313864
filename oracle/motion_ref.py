"""CPU restatement of the geometry losses of GraspMotion-Opt stage 2, line by line after
``MotionFill/models/fittingop.py:132-280`` (per-frame normals loop included).  TEST INFRASTRUCTURE.
PARITY UNPINNED: depends on the unpinned chamfer / Open3D-normal restatements."""
import torch

from .geometry_ref import point2point_signed, vertex_normals


def contact_loss(faces, start_t, body_verts_T, markers_143, rhand_verts, contact_object, contact_markers, normal_object,
                 verts_object, n_frames=61):
    palm = markers_143[-22:]                                                             # :133
    rh_normals, rh_normals_full = [], []
    for i in range(start_t, len(body_verts_T)):                                          # :137-146, one mesh per frame
        n = vertex_normals(body_verts_T[i:i + 1].detach().double(), faces)[0]
        rh_normals_full.append(n[rhand_verts])
        rh_normals.append(n[palm])
    dt = body_verts_T.dtype
    rh_normals_full = torch.stack(rh_normals_full).to(torch.float32).to(dt)              # :148
    rh_normals = torch.stack(rh_normals).to(torch.float32).to(dt)
    body_markers = body_verts_T[start_t:, markers_143]
    rh_verts = body_verts_T[start_t:, rhand_verts]
    _, h2o_signed_marker, _, _, _, h2o_marker_vector = point2point_signed(body_markers, verts_object, None, normal_object,
                                                                           return_vector=True)      # :154-155
    o2h_signed, h2o_signed, _, _ = point2point_signed(rh_verts, verts_object, rh_normals_full, normal_object)   # :156-157
    w = torch.zeros((n_frames - start_t, 1), dtype=dt)
    w[-5:] = 1                                                                           # :161-163
    loss = torch.mean(torch.abs(o2h_signed) * contact_object * w) + torch.mean(torch.abs(h2o_signed_marker) * contact_markers * w)
    state = dict(o2h_signed=o2h_signed, h2o_signed=h2o_signed, h2o_signed_marker=h2o_signed_marker,
                 h2o_marker_vector=h2o_marker_vector, rh_normals=rh_normals, rh_normals_norm=torch.norm(rh_normals, dim=-1))
    return loss, state


def collision_loss(state):                                                               # :178-191
    o2h = state['o2h_signed']
    mask = torch.zeros_like(o2h)
    mask[o2h < -0.003] = 20
    return 5 * torch.mean(torch.mean(torch.abs(o2h) * mask, dim=-1))


def skating_loss(body_verts_T, contact_lbl_rec, groups, vel_thres=0.1):                  # :193-263
    vel = (body_verts_T[1:] - body_verts_T[0:-1]) * 30
    total = torch.zeros((), dtype=body_verts_T.dtype)
    for col, name in enumerate(['left_heel', 'right_heel', 'left_toe', 'right_toe']):
        v = vel[:, groups[name], :][contact_lbl_rec[0:-1, col] == 1]
        v = torch.norm(v, dim=-1)
        if (v - vel_thres).gt(0).sum().item() >= 1:
            total = total + v[v > vel_thres].abs().mean()
    return total


def angle_loss(state, start_t, n_frames=61):                                             # :265-280
    vec = -state['h2o_marker_vector'][:, -22:]
    vec_norm = torch.norm(vec, dim=-1)
    T = vec.shape[0]
    cos = torch.bmm(state['rh_normals'].reshape(-1, 3).unsqueeze(1), vec.reshape(-1, 3).unsqueeze(2)).view(T, -1) / (
        state['rh_normals_norm'] * vec_norm)
    weight = 1 - torch.true_divide(torch.arange(n_frames - start_t) + 1, n_frames - start_t).view(-1, 1) ** 2
    mask = (state['h2o_signed_marker'] < 0.01).int()
    return torch.mean(mask[:, -22:] * ((cos - 1) ** 2) * weight)
