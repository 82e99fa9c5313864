"""Geometry losses of GraspMotion-Opt stage 2 (``MotionFill/models/fittingop.py:132-280,342-414``) on the HIP
operators: frames are the batch dimension of the SMPL-X forward, of the vertex-normal kernel (Open3D per frame on the
host in the reference, ``:137-146``) and of the nearest-neighbour search.  Nothing leaves the device and nothing
synchronises (the reference's ``.item()`` tests in the skating loss become masked means).

Every loss exists in two forms sharing one implementation: ``*_seq`` takes S independent sequences ``[S, T, ...]`` and
returns one value per sequence ``[S]`` (what the batched driver ``motion_fitting.FittingOP`` sums, so each sequence sees
the gradient of its own B=1 problem); the un-suffixed form takes one sequence ``[T, ...]`` with the reference's
argument order and returns the reference's scalar.  The learned smoothness prior (``body_smooth_loss`` ``:70-130``: a
convolutional auto-encoder) is a network, not part of this path.
"""
import ctypes

import torch

from . import _lib
from .geometry import RowPicker, point2point_signed, vertex_normals_subset


def foot_groups(idx):
    """{'right_heel','right_toe','left_heel','left_toe'} vertex ids from ``load_grasp_indices()`` output."""
    foot, sizes = idx['foot'], [int(s) for s in idx['foot_sizes']]
    out, o = {}, 0
    for name, n in zip(['right_heel', 'right_toe', 'left_heel', 'left_toe'], sizes):
        out[name] = torch.as_tensor(foot[o:o + n], dtype=torch.long)
        o += n
    return out


class VertexPicker:
    """One gather of the union of several vertex-id sets out of [S,T,V,3] vertices, then cheap picks out of the small
    result: the backward pass scatters into (and zero-fills, and accumulates) the full-size gradient once instead of
    once per set -- at 256 x 61 bodies each of those is a 2 GB round trip."""

    def __init__(self, sets, device, windows=None):
        ids = {k: torch.as_tensor(v, dtype=torch.long, device=device) for k, v in sets.items()}
        self.union = torch.unique(torch.cat(list(ids.values())))                       # sorted
        self.pos = {k: torch.searchsorted(self.union, v) for k, v in ids.items()}      # position of each id in the union
        self.ids = ids
        # every set out of the gathered union in one kernel; ``windows`` = {output name: (set name, first frame)} adds picks
        # of a set on the frames first.. only (contiguous [S, T - first, n, 3]: what the contact terms read)
        replaced = {src for src, _ in (windows or {}).values()}
        picks = {k: v for k, v in self.pos.items() if k not in replaced}      # (a windowed set is not also picked on every frame)
        self.foot_sizes = None
        if all(k in picks for k in FOOT_ORDER):                               # the heel / toe groups side by side, one pick
            self.foot_sizes = [int(picks[k].shape[0]) for k in FOOT_ORDER]
            picks['feet'] = torch.cat([picks.pop(k) for k in FOOT_ORDER])
        for name, (src, t0) in (windows or {}).items():
            picks[name] = (self.pos[src], int(t0))
        self.rows = RowPicker(picks, int(self.union.shape[0]))

    def gather(self, body_verts):
        return body_verts.index_select(2, self.union)

    def pick(self, union_verts):
        """{set name: [S,T,n,3]} out of ``gather``'s result (one kernel forward, one deterministic gather backward)."""
        out = self.rows(union_verts)
        if self.foot_sizes is not None:
            out.update(zip(FOOT_ORDER, out['feet'].split(self.foot_sizes, dim=-2)))
        return out


def contact_loss_seq(bm, start_t, body_verts, markers_143, rhand_verts, contact_object, contact_markers, normal_object,
                     verts_object, n_frames=None, full_verts=None, full_ids=None, picked=None):
    """body_verts [S,T,V,3]; verts_object / normal_object [S,T-start_t,O,3]; contact_object [S,O];
    contact_markers [S,143] -> (loss [S], state).  With ``full_verts`` [S,T,V,3] and ``full_ids`` = (rhand ids, 143
    marker ids) in the full mesh, ``body_verts`` may be a :class:`VertexPicker` gather and ``markers_143`` /
    ``rhand_verts`` positions in it: the normals are then computed from ``full_verts``.  ``picked`` (optional):
    {'markers_143': [S,T,143,3], 'rhand': [S,T,n,3]} already picked out of ``body_verts`` (``VertexPicker.pick``)."""
    S, T = body_verts.shape[:2]
    n_frames = T if n_frames is None else n_frames            # the reference hard-codes 61
    n_tail = T - start_t
    if full_verts is None:
        full_verts, full_ids = body_verts, (rhand_verts, markers_143)
    sel = torch.cat([full_ids[0], full_ids[1][-22:]]).to(torch.int32)          # right hand + its 22 palm markers (:133)
    normals = vertex_normals_subset(bm, full_verts, sel, first_frame=start_t)   # only the rows the loss reads, frames start_t..
    n_rh = rhand_verts.shape[0]
    rh_normals_full, rh_normals = normals[:, :n_rh], normals[:, n_rh:]
    if picked is not None:
        tail = lambda k: picked[k + '_tail'] if k + '_tail' in picked else picked[k][:, start_t:]
        body_markers = tail('markers_143').reshape(S * n_tail, -1, 3)
        rh_verts = tail('rhand').reshape(S * n_tail, -1, 3)
    else:
        tail = body_verts[:, start_t:].reshape(S * n_tail, -1, 3)
        body_markers = tail.index_select(1, markers_143)
        rh_verts = tail.index_select(1, rhand_verts)
    O = verts_object.shape[-2]
    vo, no = verts_object.float().reshape(S * n_tail, O, 3), normal_object.float().reshape(S * n_tail, O, 3)
    # (the reference discards the object -> marker direction of this call, :154: it is not searched)
    _o2h_m, h2o_signed_marker, _, _, _o2h_vec, h2o_marker_vector = point2point_signed(body_markers, vo, None, no, return_vector=True,
                                                                                        directions='x2y')
    # (no loss reads the hand -> object direction of this call -- h2o_signed_min :158 and mask_pulse_h2o :179-181 are dead code in
    # the reference -- so it is not searched; state['h2o_signed'] is None)
    o2h_signed, h2o_signed, _, _ = point2point_signed(rh_verts, vo, rh_normals_full, no, directions='y2x')
    w = torch.zeros(n_frames - start_t, 1, device=body_verts.device)
    w[-5:] = 1                                                 # attraction only over the last five frames (:163-165)
    seq = lambda t: t.reshape((S, n_tail) + t.shape[1:])
    o2h_signed, h2o_signed_marker = seq(o2h_signed), seq(h2o_signed_marker)
    loss_obj = (o2h_signed.abs() * contact_object.reshape(S, 1, -1) * w).mean(dim=(1, 2))
    loss_mk = (h2o_signed_marker.abs() * contact_markers.reshape(S, 1, -1) * w).mean(dim=(1, 2))
    rh_normals = seq(rh_normals)
    state = dict(o2h_signed=o2h_signed, h2o_signed=h2o_signed, h2o_signed_marker=h2o_signed_marker,
                 h2o_marker_vector=seq(h2o_marker_vector), rh_normals=rh_normals, rh_normals_norm=rh_normals.norm(dim=-1))
    return loss_obj + loss_mk, state


def collision_loss_seq(state):
    """``collision_loss`` (:178-191): only the object -> hand penetration term is used (weight 20 below -3 mm)."""
    o2h = state['o2h_signed']
    mask = torch.where(o2h < -0.003, torch.full_like(o2h, 20.0), torch.zeros_like(o2h))
    return 5 * (o2h.abs() * mask).mean(dim=-1).mean(dim=-1)


def skating_loss_seq(body_verts, contact_lbl_rec, groups, vel_thres=0.1, picked=None):
    """``skating_loss`` (:193-263): mean speed (30 fps) above ``vel_thres`` of the heel / toe vertices on the frames whose
    contact label is 1; a group with no such vertex contributes 0.  body_verts [S,T,V,3]; contact_lbl_rec [S,T,4] =
    (left heel, right heel, left toe, right toe)."""
    S = body_verts.shape[0]
    total = body_verts.new_zeros(S)
    for col, name in enumerate(['left_heel', 'right_heel', 'left_toe', 'right_toe']):
        foot = picked[name] if picked is not None else body_verts.index_select(2, groups[name].to(body_verts.device))
        speed = ((foot[:, 1:] - foot[:, :-1]) * 30).norm(dim=-1)                        # [S, T-1, n]
        on = (contact_lbl_rec[:, :-1, col] == 1)[:, :, None] & (speed > vel_thres)
        cnt = on.sum(dim=(1, 2))
        total = total + torch.where(cnt > 0, (speed.abs() * on).sum(dim=(1, 2)) / cnt.clamp(min=1), speed.new_zeros(S))
    return total


def angle_loss_seq(state, start_t, n_frames):
    """``angle_loss`` (:265-280): the palm normals of the right hand should face the object where a palm marker is
    within 1 cm of it; weight decays quadratically over the optimised frames."""
    vec = -state['h2o_marker_vector'][:, :, -22:]
    cos = (state['rh_normals'] * vec).sum(-1) / (state['rh_normals_norm'] * vec.norm(dim=-1))
    n = n_frames - start_t
    weight = 1 - ((torch.arange(n, device=vec.device) + 1).true_divide(n).view(-1, 1)) ** 2
    mask = (state['h2o_signed_marker'] < 0.01).int()
    return (mask[:, :, -22:] * (cos - 1) ** 2 * weight).mean(dim=(1, 2))


def hand_smooth_loss_seq(body_verts, start_t, rhand_verts, picked=None):
    """``loss_rh_smooth`` (:386-387)."""
    if picked is not None:
        rh = picked['rhand_tail'] if 'rhand_tail' in picked else picked['rhand'][:, start_t:]
    else:
        rh = body_verts[:, start_t:].index_select(2, rhand_verts)
    return (rh[:, 1:] - rh[:, :-1]).norm(dim=-1).mean(dim=(1, 2))


# ---- the same terms as ONE operator (saga_motion_terms_fwd/bwd) -------------------------------------------------------------
TERM_NAMES = ('rec_markers', 'end_markers_fit', 'skating', 'hand_smooth', 'contact', 'collision', 'hand_angle')
FOOT_ORDER = ('left_heel', 'right_heel', 'left_toe', 'right_toe')          # the columns of contact_lbl_rec (:218-258)


class _MotionTerms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, markers, feet, hand_tail, o2h_signed, h2o_marker, h2o_marker_vec, markers_rec, marker_mask,
                markers_end, contact_lbl, contact_object, contact_markers, palm_normals):
        args = [markers, markers_rec, marker_mask, markers_end, feet, contact_lbl, hand_tail, o2h_signed, contact_object,
                h2o_marker, contact_markers, h2o_marker_vec, palm_normals]
        _lib.require_device(*args)
        args = [a.detach().contiguous().float() for a in args]
        S = cfg.n_seq
        expect = [(S, cfg.n_frames, cfg.n_markers, 3), (S, cfg.n_frames, cfg.n_markers, 3), (cfg.n_markers,), (S, cfg.n_markers, 3),
                  (S, cfg.n_frames, sum(cfg.foot_group_size), 3), (S, cfg.n_frames, 4), (S, cfg.n_tail, cfg.n_hand, 3),
                  (S * cfg.n_tail * cfg.n_obj,), (S, cfg.n_obj), (S * cfg.n_tail * cfg.n_contact_markers,), (S, cfg.n_contact_markers),
                  (S * cfg.n_tail * cfg.n_contact_markers * 3,), (S * cfg.n_tail * cfg.n_palm * 3,)]
        for i, (a, e) in enumerate(zip(args, expect)):
            n = 1
            for x in e:
                n *= x
            if a.numel() != n:
                raise ValueError('motion_terms: argument %d has shape %s, expected %s' % (i, tuple(a.shape), e))
        terms = torch.empty(S, _lib.MOTION_NTERMS, device=args[0].device, dtype=torch.float32)
        _lib.call('saga_motion_terms_fwd', terms.device, ctypes.byref(cfg), *[_lib.ptr(a) for a in args], _lib.ptr(terms))
        ctx.cfg = cfg
        ctx.save_for_backward(terms, *args)
        ctx.shapes = [t.shape for t in (markers, feet, hand_tail, o2h_signed, h2o_marker, h2o_marker_vec)]
        return terms[:, :7].clone()

    @staticmethod
    def backward(ctx, g_terms):
        terms, *args = ctx.saved_tensors
        f32 = dict(device=terms.device, dtype=torch.float32)
        outs = [torch.empty(s, **f32) for s in ctx.shapes]
        g = g_terms.contiguous().float()
        _lib.call('saga_motion_terms_bwd', terms.device, ctypes.byref(ctx.cfg), *[_lib.ptr(a) for a in args], _lib.ptr(terms), _lib.ptr(g),
                  *[_lib.ptr(o) for o in outs])
        return (None,) + tuple(outs) + (None,) * 7


def motion_terms(markers, feet, foot_sizes, hand_tail, o2h_signed, h2o_marker, h2o_marker_vec, markers_rec, marker_mask,
                 markers_end, contact_lbl, contact_object, contact_markers, palm_normals, vel_thres=0.1, n_last_frames=5):
    """The seven geometric stage-2 terms per sequence -> [S,7] in ``TERM_NAMES`` order.  markers [S,T,M,3]; feet [S,T,NF,3] =
    the four groups of ``FOOT_ORDER`` side by side (``foot_sizes``); hand_tail [S,n_tail,H,3]; o2h_signed [S*n_tail,O],
    h2o_marker [S*n_tail,CM], h2o_marker_vec [S*n_tail,CM,3] as ``point2point_signed`` returns them; palm_normals
    [S*n_tail,NP,3].  Differentiable w.r.t. the first six tensors."""
    S, T, M, _ = markers.shape
    n_tail, H = hand_tail.shape[1], hand_tail.shape[2]
    cfg = _lib.MotionTermsCfg(n_seq=S, n_frames=T, n_tail=n_tail, n_markers=M, foot_group_size=(ctypes.c_int32 * 4)(*foot_sizes),
                              n_hand=H, n_obj=contact_object.shape[-1], n_contact_markers=contact_markers.shape[-1],
                              n_palm=palm_normals.shape[-2], n_last_frames=n_last_frames, vel_thres=vel_thres)
    return _MotionTerms.apply(cfg, markers, feet, hand_tail, o2h_signed, h2o_marker, h2o_marker_vec, markers_rec, marker_mask,
                              markers_end, contact_lbl, contact_object, contact_markers, palm_normals)


# ---- one sequence, the reference's argument order and scalar results ------------------------------------------------------

def contact_loss(bm, start_t, body_verts_T, markers_143, rhand_verts, contact_object, contact_markers, normal_object,
                 verts_object, n_frames=None):
    """``contact_loss`` (:132-176).  body_verts_T [T,V,3]; markers_143 / rhand_verts: long index tensors;
    verts_object, normal_object [T-start_t,O,3]; contact_object [1,O], contact_markers [1,143] (broadcast over frames).
    -> (loss_contact, state) where state holds what ``collision_loss`` / ``angle_loss`` read from ``self``."""
    loss, st = contact_loss_seq(bm, start_t, body_verts_T[None], markers_143, rhand_verts, contact_object.reshape(1, -1),
                                contact_markers.reshape(1, -1), normal_object[None], verts_object[None], n_frames)
    return loss[0], st


def collision_loss(state):
    return collision_loss_seq(state)[0]


def skating_loss(body_verts_T, contact_lbl_rec, groups, vel_thres=0.1):
    return skating_loss_seq(body_verts_T[None], contact_lbl_rec[None], groups, vel_thres)[0]


def angle_loss(state, start_t, n_frames):
    return angle_loss_seq(state, start_t, n_frames)[0]


def hand_smooth_loss(body_verts_T, start_t, rhand_verts):
    return hand_smooth_loss_seq(body_verts_T[None], start_t, rhand_verts)[0]
