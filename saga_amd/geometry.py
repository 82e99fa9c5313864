"""Geometry / loss helpers of the hot path on the HIP kernels.

* :func:`point2point_signed`  mirrors ``utils/train_helper.py:17-86`` (same arguments, return order
  and ``ValueError``), with the nearest-neighbour search on ``saga_chamfer_fwd``.
* :func:`vertex_normals` / :func:`vertex_normals_subset` replace Open3D's
  ``compute_vertex_normals`` (``WholeGraspPose/models/fittingop.py:158-164``) on the device.
"""
import torch

from . import _lib
from .chamfer_distance import chamfer_nn


def point2point_signed(x, y, x_normals=None, y_normals=None, return_vector=False):
    """Signed distance between two point clouds.  x [N,P1,D], y [N,P2,D] ->
    y2x_signed [N,P2], x2y_signed [N,P1], yidx_near [N,P2] int32, xidx_near [N,P1] int32 (, y2x, x2y)."""
    N, P1, D = x.shape
    P2 = y.shape[1]
    if y.shape[0] != N or y.shape[2] != D:
        raise ValueError("y does not have the correct shape.")
    xidx_near, yidx_near = chamfer_nn(x, y)             # the reference discards the distances (:57-63)
    xi = xidx_near.view(N, P1, 1).expand(N, P1, D).to(torch.long)
    x_near = y.gather(1, xi)
    yi = yidx_near.view(N, P2, 1).expand(N, P2, D).to(torch.long)
    y_near = x.gather(1, yi)
    x2y = x - x_near
    y2x = y - y_near
    if x_normals is not None:
        y_nn = x_normals.gather(1, yi)
        in_out = (y_nn * y2x).detach().sum(-1).sign()     # only the sign is used (no gradient); a batched 1x3.3x1 bmm is 40 ms here
        y2x_signed = y2x.norm(dim=2) * in_out
    else:
        y2x_signed = y2x.norm(dim=2)
    if y_normals is not None:
        x_nn = y_normals.gather(1, xi)
        in_out_x = (x_nn * x2y).detach().sum(-1).sign()
        x2y_signed = x2y.norm(dim=2) * in_out_x
    else:
        x2y_signed = x2y.norm(dim=2)
    if not return_vector:
        return y2x_signed, x2y_signed, yidx_near, xidx_near
    return y2x_signed, x2y_signed, yidx_near, xidx_near, y2x, x2y


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
