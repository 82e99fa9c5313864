"""Blend GEMM forward time per tile as a function of the batch (M) on a vertex-subset model: is the large-M launch slower per tile?"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from saga_amd import _lib, smplx
from saga_amd.synthetic import write_body_model
tmp = tempfile.TemporaryDirectory(); write_body_model(tmp.name, gender='male', seed=7)
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 6604
keep = np.sort(np.random.default_rng(0).choice(10475, nv, replace=False)) if nv < 10475 else None
for B in [int(x) for x in os.environ.get("SWEEP_B", "1024,4096,8192,15616").split(",")]:
    bm = smplx.create(tmp.name, model_type='smplx', gender='male', ext='npz', num_pca_comps=24, batch_size=B,
                      **({'vertex_subset': keep} if keep is not None else {})).cuda().eval()
    g = torch.Generator().manual_seed(1)
    args = dict(body_pose=0.3 * torch.randn(B, 63, generator=g).cuda(), global_orient=torch.randn(B, 3, generator=g).cuda())
    with torch.no_grad():
        for _ in range(3): bm(**args)
        torch.cuda.synchronize()
        _lib.profile_enable(1)
        for _ in range(10): bm(**args)
        torch.cuda.synchronize()
        _lib.profile_enable(False)
    ms, n = _lib.profile_read()['blend_gemm_fwd']
    npad = bm.npad(torch.device('cuda', 0))
    tiles = ((B + 127) // 128) * (npad // 128)
    print('V=%d B=%5d: %.3f ms per launch, %6d tiles, %.1f ns per tile, %.1f TFLOP/s' % (nv, B, ms / n, tiles, ms / n * 1e6 / tiles, 2.0 * B * npad * 512 / (ms / n * 1e-3) / 1e12))
    del bm
