"""CPU restatements of the geometry helpers on the hot path.  TEST INFRASTRUCTURE.

* ``point2point_signed``  follows ``utils/train_helper.py:17-86`` (PINNED by
  ``tests/golden/p2p_signed.npz``, generated from the imported reference function).
* ``vertex_normals``      follows Open3D ``TriangleMesh.compute_vertex_normals`` as used at
  ``WholeGraspPose/models/fittingop.py:158-164``: un-normalised face cross products summed
  onto their three corners (area weighting), then each row normalised; a zero row stays
  zero and a NaN row becomes (0,0,1).  Open3D is absent: PARITY UNPINNED.
"""
import torch

from .chamfer_ref import chamfer_distance


def point2point_signed(x, y, x_normals=None, y_normals=None, return_vector=False, force_idx=None):
    """train_helper.py:17-86.  x [N,P1,D], y [N,P2,D] -> y2x_signed [N,P2], x2y_signed [N,P1],
    yidx_near [N,P2] int32, xidx_near [N,P1] int32 (, y2x, x2y).
    ``force_idx`` = (yidx_near or None, xidx_near or None): neighbours given by the caller instead of searched -- for tests that
    have shown another implementation's indices to differ from this search only at distance near-ties and want everything
    downstream of the tie decision compared (no counterpart in the reference)."""
    N, P1, D = x.shape
    P2 = y.shape[1]
    if y.shape[0] != N or y.shape[2] != D:
        raise ValueError("y does not have the correct shape.")          # :52-53
    _, _, xidx_near, yidx_near = chamfer_distance(x.detach(), y.detach())  # :57 (distances discarded)
    if force_idx is not None:
        yidx_near = yidx_near if force_idx[0] is None else torch.as_tensor(force_idx[0]).to(yidx_near.dtype).reshape(yidx_near.shape)
        xidx_near = xidx_near if force_idx[1] is None else torch.as_tensor(force_idx[1]).to(xidx_near.dtype).reshape(xidx_near.shape)
    xi = xidx_near.view(N, P1, 1).expand(N, P1, D).to(torch.long)        # :59
    x_near = y.gather(1, xi)                                             # :60
    yi = yidx_near.view(N, P2, 1).expand(N, P2, D).to(torch.long)        # :62
    y_near = x.gather(1, yi)                                             # :63
    x2y = x - x_near                                                     # :65
    y2x = y - y_near                                                     # :66
    if x_normals is not None:                                            # :68-71
        y_nn = x_normals.gather(1, yi)
        in_out = (y_nn * y2x).sum(-1).sign()
        y2x_signed = y2x.norm(dim=2) * in_out
    else:
        y2x_signed = y2x.norm(dim=2)                                     # :74
    if y_normals is not None:                                            # :76-79
        x_nn = y_normals.gather(1, xi)
        in_out_x = (x_nn * x2y).sum(-1).sign()
        x2y_signed = x2y.norm(dim=2) * in_out_x
    else:
        x2y_signed = x2y.norm(dim=2)                                     # :81
    if not return_vector:
        return y2x_signed, x2y_signed, yidx_near, xidx_near              # :84
    return y2x_signed, x2y_signed, yidx_near, xidx_near, y2x, x2y        # :86


def vertex_normals(verts, faces):
    """verts [B,V,3] (any float dtype), faces [F,3] long -> unit normals [B,V,3]."""
    B, V, _ = verts.shape
    v0 = verts[:, faces[:, 0]]
    v1 = verts[:, faces[:, 1]]
    v2 = verts[:, faces[:, 2]]
    fn = torch.cross(v1 - v0, v2 - v0, dim=-1)                           # [B,F,3], |fn| = 2*area
    n = torch.zeros_like(verts)
    for c in range(3):
        n.index_add_(1, faces[:, c], fn)
    sq = (n * n).sum(-1, keepdim=True)
    out = torch.where(sq > 0, n / torch.sqrt(torch.where(sq > 0, sq, torch.ones_like(sq))), n)
    nan = torch.isnan(out[..., :1])
    fallback = torch.tensor([0.0, 0.0, 1.0], dtype=verts.dtype).expand_as(out)
    return torch.where(nan, fallback, out)
