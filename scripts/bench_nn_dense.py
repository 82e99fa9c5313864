#!/usr/bin/env python3
"""BASELINE config 5: full-mesh chamfer, 10,475 body vertices x 100,000 dense object points per candidate, B = 512
(5.36e11 point pairs per direction).  Times ChamferDistance(x, y) (both directions, distances + indices) and, with
--check, compares a slice with the plain exact scan kernel.  Prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saga_amd.chamfer_distance as chd

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=512)
ap.add_argument('--verts', type=int, default=10475)
ap.add_argument('--points', type=int, default=100000)
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--check', action='store_true')
a = ap.parse_args()
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(a.batch, a.verts, 3, device='cuda', generator=g) * torch.tensor([0.3, 0.15, 0.85], device='cuda') + torch.tensor([0., 0., 0.9], device='cuda')
y = torch.randn(a.batch, a.points, 3, device='cuda', generator=g) * 0.25 + torch.tensor([0.3, 0.1, 1.0], device='cuda')
out = chd.ChamferDistance(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    out = chd.ChamferDistance(x, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
pairs = 2.0 * a.batch * a.verts * a.points
res = {'config': 'full-mesh chamfer %d x %d, B=%d' % (a.verts, a.points, a.batch), 'ms': round(dt * 1e3, 2),
       'pairs_per_s': pairs / dt, 'equivalent_scan_tflops_8flop_per_pair': round(pairs * 8 / dt / 1e12, 2), 'peak_tflops': 157.3}
if a.check:
    os.environ['SAGA_NN_EXACT_SCAN'] = '1'
    ref = chd.ChamferDistance(x[:8], y[:8])
    del os.environ['SAGA_NN_EXACT_SCAN']
    res['slice_matches_exact_scan'] = all(torch.equal(r, o[:8]) for r, o in zip(ref, out))
print(json.dumps(res))
