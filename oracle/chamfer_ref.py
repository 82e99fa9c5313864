"""CPU restatement of ``chamfer_distance.ChamferDistance`` (brute-force bidirectional NN).

TEST INFRASTRUCTURE.  PARITY UNPINNED: the extension (github otaheri/chamfer_distance, no
commit pinned, reference ``README.md:59``) is not in the container.  Contract taken from the
call site ``utils/train_helper.py:57-63`` and SURVEY.md §2-D2: for every point the *squared*
euclidean distance to, and the int32 index of, its nearest neighbour in the other cloud;
candidates are scanned in index order with a strict ``<`` so the lowest index wins ties.

The squared distance is evaluated as ``(dx*dx + dy*dy) + dz*dz`` with one rounding per
operation (no fused multiply-add), in the dtype of the inputs.
"""
import numpy as np
import torch


def _sqdist(x, y):
    # x [N,P1,3], y [N,P2,3] -> [N,P1,P2]; explicit per-coordinate form keeps the rounding
    # sequence fixed (a matmul expansion would not).
    d = x[:, :, None, :] - y[:, None, :, :]
    d2 = d * d
    return (d2[..., 0] + d2[..., 1]) + d2[..., 2]


def chamfer_distance(x, y, chunk=256):
    """-> dist1 [N,P1], dist2 [N,P2], idx1 [N,P1] int32, idx2 [N,P2] int32."""
    N, P1, _ = x.shape
    P2 = y.shape[1]
    dist1 = torch.empty(N, P1, dtype=x.dtype)
    idx1 = torch.empty(N, P1, dtype=torch.int32)
    best2 = torch.full((N, P2), float('inf'), dtype=x.dtype)
    idx2 = torch.zeros(N, P2, dtype=torch.int32)
    for s in range(0, P1, chunk):
        d = _sqdist(x[:, s:s + chunk], y)                 # [N,c,P2]
        m = d.min(dim=2).values
        i = _first_argmin(d, m, dim=2)                    # lowest index among exact ties
        dist1[:, s:s + chunk] = m
        idx1[:, s:s + chunk] = i.to(torch.int32)
        m2 = d.min(dim=1).values                          # [N,P2] over this chunk of x
        i2 = _first_argmin(d, m2, dim=1) + s
        upd = m2 < best2                                  # strict: earlier chunk keeps ties
        best2 = torch.where(upd, m2, best2)
        idx2 = torch.where(upd, i2.to(torch.int32), idx2)
    return dist1, best2, idx1, idx2


def _first_argmin(d, m, dim):
    eq = d == m.unsqueeze(dim)
    n = d.shape[dim]
    shape = [1, 1, 1]
    shape[dim] = n
    ar = torch.arange(n).view(shape)
    big = torch.where(eq, ar, torch.full_like(ar, n))
    return big.min(dim=dim).values


def chamfer_distance_loop(x, y):
    """Literal scan (numpy, small inputs only): the definition the vectorised form is checked against."""
    x = np.asarray(x)
    y = np.asarray(y)
    N, P1, _ = x.shape
    P2 = y.shape[1]
    out = [np.zeros((N, P1), x.dtype), np.zeros((N, P2), x.dtype),
           np.zeros((N, P1), np.int32), np.zeros((N, P2), np.int32)]
    for (a, b, dd, ii) in ((x, y, out[0], out[2]), (y, x, out[1], out[3])):
        for n in range(N):
            for p in range(a.shape[1]):
                best, bi = None, 0
                for k in range(b.shape[1]):
                    dx, dy, dz = (a[n, p] - b[n, k]).astype(a.dtype)
                    d = a.dtype.type(a.dtype.type(dx * dx) + a.dtype.type(dy * dy)) + a.dtype.type(dz * dz)
                    d = a.dtype.type(d)
                    if best is None or d < best:
                        best, bi = d, k
                dd[n, p], ii[n, p] = best, bi
    return tuple(out)
