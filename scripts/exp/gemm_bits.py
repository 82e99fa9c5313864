"""Hashes of the SMPL-X forward vertices and parameter gradients at several batch sizes: run under two library settings (e.g.
SAGA_GEMM_SMALL=0 / 1, or two builds through SAGA_HIP_LIB) and compare the lines -- blend GEMM variants must agree bit for bit."""
import os, sys, tempfile, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from saga_amd import smplx
from saga_amd.synthetic import write_body_model
tmp = tempfile.TemporaryDirectory(); write_body_model(tmp.name, gender='male', seed=7)
for B in [int(x) for x in os.environ.get('BITS_B', '4096,300,1111,129').split(',')]:
    bm = smplx.create(tmp.name, model_type='smplx', gender='male', ext='npz', num_pca_comps=24, batch_size=B).cuda()
    g = torch.Generator().manual_seed(B)
    bp = (0.3 * torch.randn(B, 63, generator=g)).cuda().requires_grad_(True)
    go = torch.randn(B, 3, generator=g).cuda().requires_grad_(True)
    be = torch.randn(B, 10, generator=g).cuda().requires_grad_(True)
    out = bm(body_pose=bp, global_orient=go, betas=be)
    w = torch.randn(out.vertices.shape, generator=g).cuda()
    (out.vertices * w).sum().backward()
    torch.cuda.synchronize()
    h = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]
    print('B=%d vertices %s g_body_pose %s g_orient %s g_betas %s' % (B, h(out.vertices), h(bp.grad), h(go.grad), h(be.grad)))
    del bm
