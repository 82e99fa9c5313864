"""Runs on the GPU box: saga_cont6d_to_aa at extreme input scales and near-degenerate 6D inputs against the float64 oracle
(oracle/rotations_ref, restating utils/utils.py:385-417).  Prints the worst deviations; exits 1 on a non-finite output where the
oracle's is finite, or on a deviation beyond the test bar on well-conditioned inputs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import rotations_ref  # noqa: E402
from saga_amd.rotations import cont6d_to_aa  # noqa: E402

g = torch.Generator().manual_seed(11)
bad = 0
for scale in (1e-18, 1e-12, 1e-6, 1e-3, 1.0, 1e3, 1e6, 1e12, 1e18):
    x = torch.randn(5000, 6, generator=g, dtype=torch.float64)
    xs = (x * scale).float()                      # what the kernel sees
    ref = rotations_ref.rotmat2aa(rotations_ref.cont2rotmat(xs.double()))
    got = cont6d_to_aa(xs.cuda()).cpu().double()
    fin_ref = torch.isfinite(ref).all(dim=1)
    fin_got = torch.isfinite(got).all(dim=1)
    ang = ref.norm(dim=1)
    ok = fin_ref & (ang < 3.0)
    err = (got - ref).abs().amax(dim=1)
    worst = float(err[ok & fin_got].max()) if (ok & fin_got).any() else float('nan')
    nonfinite = int((fin_ref & ~fin_got).sum())
    print('scale %8.0e  finite ref %5d  kernel non-finite where ref finite %4d  worst |d aa| (angle < 3) %.3e' % (scale, int(fin_ref.sum()), nonfinite, worst), flush=True)
    if nonfinite or (1e-12 <= scale <= 1e12 and not worst < 2e-5):
        bad += 1
# near-colinear pairs: a2 = a1 * s + eps * noise
for eps in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5):
    a1 = torch.randn(5000, 3, generator=g, dtype=torch.float64)
    a2 = a1 * torch.randn(5000, 1, generator=g, dtype=torch.float64) + eps * torch.randn(5000, 3, generator=g, dtype=torch.float64)
    xs = torch.cat([a1, a2], dim=1).float()
    ref = rotations_ref.rotmat2aa(rotations_ref.cont2rotmat(xs.double()))
    got = cont6d_to_aa(xs.cuda()).cpu().double()
    fin = torch.isfinite(ref).all(dim=1) & (ref.norm(dim=1) < 3.0)
    err = (got - ref).abs().amax(dim=1)[fin]
    print('colinear eps %.0e  kernel finite %5d / %d  worst |d aa| %.3e  (float32 conditioning ~ %.0e)' % (eps, int(torch.isfinite(got).all(dim=1).sum()), len(got), float(err.max()), 1e-7 / eps), flush=True)
    if not torch.isfinite(got).all():
        bad += 1
print('ROT_FUZZ_DONE bad', bad)
sys.exit(1 if bad else 0)
