"""Geometry / loss helpers of the hot path on the HIP kernels.

* :func:`point2point_signed`  mirrors ``utils/train_helper.py:17-86`` (same arguments, return order
  and ``ValueError``) as ONE HIP operator with its own deterministic backward (``saga_p2p_signed_fwd/bwd``).
* :func:`vertex_normals` / :func:`vertex_normals_subset` replace Open3D's
  ``compute_vertex_normals`` (``WholeGraspPose/models/fittingop.py:158-164``) on the device.
"""
import torch

from . import _lib


class _P2PSigned(torch.autograd.Function):
    """One HIP operator for ``point2point_signed`` (``saga_p2p_signed_fwd/bwd``): NN search + gathered difference vectors +
    norms + normal-dot signs forward; a deterministic (fixed-order gather, no atomics) backward to both clouds."""

    @staticmethod
    def forward(ctx, x, y, x_normals, y_normals, want_vec):
        _lib.require_device(x, y, x_normals, y_normals)
        xc, yc = x.detach().contiguous(), y.detach().contiguous()
        xn = None if x_normals is None else x_normals.detach().contiguous()
        yn = None if y_normals is None else y_normals.detach().contiguous()
        N, P1, _ = xc.shape
        P2 = yc.shape[1]
        f32, i32 = dict(device=xc.device, dtype=torch.float32), dict(device=xc.device, dtype=torch.int32)
        y2x_s, x2y_s = torch.empty(N, P2, **f32), torch.empty(N, P1, **f32)
        yidx, xidx = torch.empty(N, P2, **i32), torch.empty(N, P1, **i32)
        y2x_v = torch.empty(N, P2, 3, **f32) if want_vec else None
        x2y_v = torch.empty(N, P1, 3, **f32) if want_vec else None
        P = _lib.ptr
        _lib.call('saga_p2p_signed_fwd', xc.device, P(xc), P(yc), P(xn), P(yn), P(y2x_s), P(x2y_s), P(yidx), P(xidx), P(y2x_v),
                  P(x2y_v), N, P1, P2)
        ctx.save_for_backward(xc, yc, yidx, xidx, y2x_s, x2y_s)
        ctx.mark_non_differentiable(yidx, xidx)
        ctx.set_materialize_grads(False)
        if want_vec:
            return y2x_s, x2y_s, yidx, xidx, y2x_v, x2y_v
        return y2x_s, x2y_s, yidx, xidx

    @staticmethod
    def backward(ctx, g_y2x_s, g_x2y_s, _gi1, _gi2, g_y2x_v=None, g_x2y_v=None):
        xc, yc, yidx, xidx, y2x_s, x2y_s = ctx.saved_tensors
        need_x, need_y = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_x or need_y) or all(g is None for g in (g_y2x_s, g_x2y_s, g_y2x_v, g_x2y_v)):
            return None, None, None, None, None
        c = lambda g: None if g is None else g.contiguous()
        gx = torch.empty_like(xc) if need_x else None
        gy = torch.empty_like(yc) if need_y else None
        N, P1, _ = xc.shape
        P = _lib.ptr
        _lib.call('saga_p2p_signed_bwd', xc.device, P(xc), P(yc), P(yidx), P(xidx), P(y2x_s), P(x2y_s), P(c(g_y2x_s)), P(c(g_x2y_s)),
                  P(c(g_y2x_v)), P(c(g_x2y_v)), P(gx), P(gy), N, P1, yc.shape[1])
        return gx, gy, None, None, None


def point2point_signed(x, y, x_normals=None, y_normals=None, return_vector=False):
    """Signed distance between two point clouds (``utils/train_helper.py:17-86``: same arguments, return order and
    ``ValueError``).  x [N,P1,3], y [N,P2,3] -> y2x_signed [N,P2], x2y_signed [N,P1], yidx_near [N,P2] int32,
    xidx_near [N,P1] int32 (, y2x [N,P2,3], x2y [N,P1,3]).  Differentiable w.r.t. ``x`` and ``y`` (not the normals: only
    the sign of their dot product is used)."""
    N, P1, D = x.shape
    if y.shape[0] != N or y.shape[2] != D:
        raise ValueError("y does not have the correct shape.")
    if D != 3:
        raise NotImplementedError('point2point_signed on the HIP path takes 3-D points (got D=%d)' % D)
    return _P2PSigned.apply(x, y, x_normals, y_normals, bool(return_vector))


def vertex_normals(model, verts):
    """All vertex normals of the model's mesh: verts [B,V,3] -> [B,V,3] (detached)."""
    _lib.require_device(verts)
    verts = verts.detach().contiguous()
    B, V, _ = verts.shape
    if V != model.num_verts:
        raise ValueError('verts has %d vertices, the model %d' % (V, model.num_verts))
    out = torch.empty_like(verts)
    lib = _lib.load()
    _lib.call('saga_vertex_normals', verts.device, model.handle(verts.device), _lib.ptr(verts), _lib.ptr(out), B)
    return out


def vertex_normals_subset(model, verts, sel):
    """Normals of the listed vertices only: verts [B,V,3], sel int32 [n] (device) -> [B,n,3] (detached)."""
    _lib.require_device(verts, sel)
    verts = verts.detach().contiguous()
    B, V, _ = verts.shape
    if V != model.num_verts:
        raise ValueError('verts has %d vertices, the model %d' % (V, model.num_verts))
    out = torch.empty(B, sel.shape[0], 3, device=verts.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.call('saga_vertex_normals_subset', verts.device, model.handle(verts.device), _lib.ptr(verts), _lib.ptr(sel), sel.shape[0],
                                              _lib.ptr(out), B)
    return out
