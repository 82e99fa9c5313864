"""CPU restatement of ``chamfer_distance.ChamferDistance`` (brute-force bidirectional NN).

TEST INFRASTRUCTURE.  PARITY UNPINNED: the extension (github otaheri/chamfer_distance, no
commit pinned, reference ``README.md:59``) is not in the container.  Contract taken from the
call site ``utils/train_helper.py:57-63`` and SURVEY.md §2-D2: for every point the *squared*
euclidean distance to, and the int32 index of, its nearest neighbour in the other cloud;
candidates are scanned in index order with a strict ``<`` so the lowest index wins ties.

The arithmetic lives in plain C (``oracle/csrc/nn_ref.c`` -> ``oracle/lib/liboracle_nn.so``, built by
``__graft_entry__.build()`` or on first use with gcc): the squared distance is the extension's
``(x2-x1)*(x2-x1)+(y2-y1)*(y2-y1)+(z2-z1)*(z2-z1)`` under the multiply-add contraction a CUDA
compiler applies by default, ``fma(dz,dz, fma(dy,dy, dx*dx))``, evaluated with ``fmaf`` (one rounding
per fused step) in the dtype of the inputs.  ``chamfer_distance_loop`` is the same definition in
pure Python (exact rational arithmetic per step) for small cases; the C scan is checked against it.
"""
import ctypes
import os
import subprocess
from fractions import Fraction

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'lib', 'liboracle_nn.so')
_lib = None


def build():
    subprocess.run(['make', '-C', os.path.join(_HERE, 'csrc')], check=True, stdout=subprocess.DEVNULL)


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        for name in ('oracle_nn_scan_f32', 'oracle_nn_scan_f64'):
            fn = getattr(lib, name)
            fn.restype = None
            fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3
        _lib = lib
    return _lib


def _scan(q, db):
    """q [N,Pq,3], db [N,Pd,3] (numpy, float32 or float64) -> (dist [N,Pq], idx [N,Pq] int32)."""
    assert q.dtype == db.dtype and q.dtype in (np.float32, np.float64)
    q = np.ascontiguousarray(q)
    db = np.ascontiguousarray(db)
    N, Pq, _ = q.shape
    Pd = db.shape[1]
    dist = np.empty((N, Pq), q.dtype)
    idx = np.empty((N, Pq), np.int32)
    fn = _load().oracle_nn_scan_f32 if q.dtype == np.float32 else _load().oracle_nn_scan_f64
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    fn(p(q), p(db), p(dist), p(idx), N, Pq, Pd)
    return dist, idx


def chamfer_distance(x, y, chunk=None):
    """-> dist1 [N,P1], dist2 [N,P2], idx1 [N,P1] int32, idx2 [N,P2] int32 (torch CPU tensors)."""
    xn, yn = x.detach().cpu().numpy(), y.detach().cpu().numpy()
    d1, i1 = _scan(xn, yn)
    d2, i2 = _scan(yn, xn)
    t = torch.from_numpy
    return t(d1), t(d2), t(i1), t(i2)


def _round(v, dtype):
    """Round an exact rational to the nearest value of ``dtype`` (float32 via float64 is exact here: a
    Fraction -> float64 conversion rounds once, and the second rounding to float32 is only taken for f32)."""
    if dtype == np.float64:
        return float(v)
    # round-to-nearest-even of an exact rational to float32: go through numpy's exact integer ratio
    f = np.float32(float(v))
    # float(v) may double-round; correct it by comparing the neighbours exactly
    cands = [f, np.nextafter(f, np.float32(np.inf), dtype=np.float32), np.nextafter(f, np.float32(-np.inf), dtype=np.float32)]
    best = min(cands, key=lambda c: (abs(Fraction(float(c)) - v), int(np.float32(c).view(np.uint32)) & 1))
    return float(best)


def chamfer_distance_loop(x, y):
    """Literal scan in exact arithmetic with one rounding per (fused) step -- small inputs only: the
    definition the C scan is checked against."""
    x = np.asarray(x)
    y = np.asarray(y)
    dt = x.dtype.type
    N, P1, _ = x.shape
    P2 = y.shape[1]
    out = [np.zeros((N, P1), x.dtype), np.zeros((N, P2), x.dtype),
           np.zeros((N, P1), np.int32), np.zeros((N, P2), np.int32)]
    F = lambda v: Fraction(float(v))
    for (a, b, dd, ii) in ((x, y, out[0], out[2]), (y, x, out[1], out[3])):
        for n in range(N):
            for p in range(a.shape[1]):
                best, bi = None, 0
                for k in range(b.shape[1]):
                    dx, dy, dz = (dt(a[n, p, c] - b[n, k, c]) for c in range(3))
                    t = _round(F(dx) * F(dx), x.dtype)                   # dx*dx
                    t = _round(F(dy) * F(dy) + Fraction(t), x.dtype)     # fma(dy,dy,.)
                    d = _round(F(dz) * F(dz) + Fraction(t), x.dtype)     # fma(dz,dz,.)
                    if best is None or d < best:
                        best, bi = d, k
                dd[n, p], ii[n, p] = best, bi
    return tuple(out)
