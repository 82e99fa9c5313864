"""Fused contact / penetration loss terms on the HIP kernels (``saga_grasp_loss_fwd/bwd``).

One call replaces ``WholeGraspPose/models/fittingop.py:152-212`` for B independent candidates: marker /
foot / right-hand gathers, hand vertex normals, both ``point2point_signed`` calls and the per-candidate
means.  The returned ``terms`` tensor is differentiable w.r.t. ``verts``; the weighted sum over terms
(stage gates, ``:215-223``) stays in Python on [B]-sized tensors.
"""
import ctypes

import numpy as np
import torch

from . import _lib

TERM_NAMES = ['rec', 'body_rec', 'foot', 'marker contact', 'object contact', 'prior contact', 'hand collision',
              'object collision', 'hand colli', 'obj colli', 'contact', 'hand markers colli']


class GraspTables:
    """Device index tables of the loss (``fittingop.py:86-124``) + the struct handed to the C-ABI."""

    def __init__(self, markers, hand, foot, hand_weights, device, n_body_markers=49):
        markers, hand, foot = (np.asarray(a, dtype=np.int64) for a in (markers, hand, foot))
        uniq = np.unique(np.concatenate([markers, hand, foot]))
        slot = lambda ids: np.searchsorted(uniq, ids).astype(np.int32)
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=device)
        self.marker_ids, self.hand_ids, self.foot_ids = i32(markers), i32(hand), i32(foot)
        self.unique_ids = i32(uniq)
        self.unique_ids_host = np.ascontiguousarray(uniq, dtype=np.int32)
        self.marker_slot, self.hand_slot, self.foot_slot = i32(slot(markers)), i32(slot(hand)), i32(slot(foot))
        self.hand_weights = torch.as_tensor(np.asarray(hand_weights), dtype=torch.float32, device=device).contiguous()
        self.n_markers, self.n_hand, self.n_foot, self.n_unique = len(markers), len(hand), len(foot), len(uniq)
        self.max_id = int(uniq.max())
        # the prior-contact vertex mask (fittingop.py:195: rhand_weight ** (1 / 2.5) > 0.7) is a constant of the table: precomputed
        # unless an entry sits so close to the threshold that the host's and the device's power could disagree
        w32 = np.asarray(hand_weights, dtype=np.float32).reshape(-1)
        pw = np.power(w32.astype(np.float64), 1.0 / 2.5)
        self.hand_prior_w = None
        if w32.size and np.abs(pw - 0.7).min() > 1e-5:
            self.hand_prior_w = torch.as_tensor(np.where(pw > 0.7, w32, np.float32(0)).astype(np.float32), device=device).contiguous()
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        self._fields = dict(
            n_markers=self.n_markers, n_hand=self.n_hand, n_foot=self.n_foot,
            n_body_markers=min(n_body_markers, self.n_markers), n_unique=self.n_unique,
            marker_ids=p(self.marker_ids), hand_ids=p(self.hand_ids), foot_ids=p(self.foot_ids),
            hand_weights=p(self.hand_weights), unique_ids=p(self.unique_ids), marker_slot=p(self.marker_slot),
            hand_slot=p(self.hand_slot), foot_slot=p(self.foot_slot),
            hand_prior_w=p(self.hand_prior_w) if self.hand_prior_w is not None else None)
        self.struct = _lib.GraspTables(**self._fields)
        self._hand_host, self._uniq_host, self._device = hand, uniq, device
        self._mesh_structs = {}

    def struct_for_mesh(self, faces, n_verts):
        """The table struct with the optional mesh-dependent fields filled in (``include/saga_hip.h``: ``ring_ids``,
        ``hand_tri_ptr``, ``hand_tri``) for a mesh with these ``faces`` [F,3]: what lets ``saga_fit_step`` stage
        the hand's face ring in LDS and take the hand normals from there.  Built once per mesh (cached by the identity of ``faces``)."""
        key = (id(faces), int(n_verts))
        hit = self._mesh_structs.get(key)
        if hit is not None:
            return hit[0]
        faces = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
        V = int(n_verts)
        is_hand = np.zeros(V, dtype=bool)
        is_hand[self._hand_host] = True
        touching = is_hand[faces].any(axis=1)
        ring = np.unique(faces[touching].reshape(-1)) if touching.any() else np.zeros(0, dtype=np.int64)
        ring_slot = np.full(V, -1, dtype=np.int32)
        ring_slot[ring] = np.arange(ring.size, dtype=np.int32)
        # incident faces of every hand vertex in ascending face order (the order saga_vertex_normal sums in)
        fid = np.nonzero(touching)[0]
        ptr = np.zeros(self.n_hand + 1, dtype=np.int32)
        rows = []
        inc = {}
        for f in fid:
            for v in faces[f]:
                if is_hand[v]:
                    inc.setdefault(int(v), []).append(int(f))
        for i, v in enumerate(self._hand_host):
            fl = sorted(inc.get(int(v), []))      # one entry per incidence (corner), like the model's vertex -> face CSR
            ptr[i + 1] = ptr[i] + len(fl)
            rows.extend(fl)
        tri = np.zeros((max(len(rows), 1), 2), dtype=np.int64)
        if rows:
            c = ring_slot[faces[np.asarray(rows)]].astype(np.int64)
            tri[:len(rows), 0] = c[:, 0] | (c[:, 1] << 16)
            tri[:len(rows), 1] = c[:, 2]
        tri = tri.astype(np.uint32).view(np.int32)
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=self._device)
        if ring.size >= 65536:
            raise ValueError('the hand ring has %d vertices: too many for the packed corner table' % ring.size)
        keep = [i32(ring), i32(ptr), i32(tri)]
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        st = _lib.GraspTables(**dict(self._fields, n_ring=int(ring.size), ring_ids=p(keep[0]),
                                     hand_tri_ptr=p(keep[1]), hand_tri=p(keep[2])))
        self._mesh_structs[key] = (st, keep, faces)
        return st


class _GraspLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, model, tables, obj, obj_normals, markers_target, c_markers, c_object):
        _lib.require_device(verts, obj, obj_normals, markers_target, c_markers, c_object)
        lib = _lib.load()
        B, V, _ = verts.shape
        O = obj.shape[1]
        if V != model.num_verts or tables.max_id >= V:
            raise ValueError('vertex tables do not fit a mesh of %d vertices' % V)
        if obj.shape != (B, O, 3) or obj_normals.shape != (B, O, 3) or markers_target.shape != (B, tables.n_markers, 3):
            raise ValueError('grasp loss: inconsistent shapes')
        c = lambda t: t.detach().contiguous()
        verts_c, obj, obj_normals, markers_target = c(verts), c(obj), c(obj_normals), c(markers_target)
        c_markers, c_object = c(c_markers).view(B, -1), c(c_object).view(B, -1)
        dev = verts.device
        f32 = dict(device=dev, dtype=torch.float32)
        i32 = dict(device=dev, dtype=torch.int32)
        M, H = tables.n_markers, tables.n_hand
        terms = torch.empty(B, _lib.GRASP_NTERMS, **f32)
        markers_rec = torch.empty(B, M, 3, **f32)
        hand_rec = torch.empty(B, H, 3, **f32)
        hand_normals = torch.empty(B, H, 3, **f32)
        h2o_marker = torch.empty(B, M, **f32)
        h2o_signed = torch.empty(B, H, **f32)
        o2h_signed = torch.empty(B, O, **f32)
        idx_m2o = torch.empty(B, M, **i32)
        idx_h2o = torch.empty(B, H, **i32)
        idx_o2h = torch.empty(B, O, **i32)
        P = _lib.ptr
        _lib.call('saga_grasp_loss_fwd', dev, model.handle(dev), ctypes.byref(tables.struct), P(verts_c), P(obj), P(obj_normals),
                                           P(markers_target), P(c_markers), P(c_object), P(terms), P(markers_rec), P(hand_rec),
                                           P(hand_normals), P(h2o_marker), P(h2o_signed), P(o2h_signed), P(idx_m2o), P(idx_h2o),
                                           P(idx_o2h), B, O)
        ctx.model, ctx.tables = model, tables
        ctx.save_for_backward(verts_c, obj, markers_target, c_markers, c_object, markers_rec, hand_rec, h2o_signed, o2h_signed,
                              idx_m2o, idx_h2o, idx_o2h)
        aux = (markers_rec, hand_rec, hand_normals, h2o_marker, h2o_signed, o2h_signed, idx_m2o, idx_h2o, idx_o2h)
        ctx.mark_non_differentiable(*aux)
        return (terms,) + aux

    @staticmethod
    def backward(ctx, g_terms, *_unused):
        (verts, obj, markers_target, c_markers, c_object, markers_rec, hand_rec, h2o_signed, o2h_signed, idx_m2o, idx_h2o,
         idx_o2h) = ctx.saved_tensors
        lib = _lib.load()
        B, V, _ = verts.shape
        O = obj.shape[1]
        g_terms = g_terms.contiguous()
        g_verts = torch.empty_like(verts)
        P = _lib.ptr
        _lib.call('saga_grasp_loss_bwd', verts.device, ctx.model.handle(verts.device), ctypes.byref(ctx.tables.struct), P(g_terms), P(verts),
                                           P(obj), P(markers_target), P(c_markers), P(c_object), P(markers_rec), P(hand_rec),
                                           P(h2o_signed), P(o2h_signed), P(idx_m2o), P(idx_h2o), P(idx_o2h), P(g_verts), B, O)
        return g_verts, None, None, None, None, None, None, None


def grasp_loss_terms(verts, model, tables, obj, obj_normals, markers_target, c_markers, c_object):
    """-> dict with ``terms`` [B,16] (differentiable w.r.t. ``verts``) and the auxiliary outputs."""
    out = _GraspLossFn.apply(verts, model, tables, obj, obj_normals, markers_target, c_markers, c_object)
    keys = ['terms', 'markers_rec', 'hand_rec', 'hand_normals', 'h2o_marker', 'h2o_signed', 'o2h_signed', 'idx_m2o',
            'idx_h2o', 'idx_o2h']
    return dict(zip(keys, out))
