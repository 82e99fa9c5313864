"""Hashes of what one fitting iteration produces (loss terms, NN indices, signed distances, variables after the update) and of
stand-alone ChamferDistance outputs: run under two builds (SAGA_HIP_LIB=build/variants/libsaga_X.so vs the tree's library) and
compare the lines -- register-allocation / live-range rewrites of the loss and NN kernels must agree bit for bit."""
import hashlib
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from saga_amd.chamfer_distance import ChamferDistance
from saga_amd.fitting import FittingOP
from saga_amd.synthetic import make_candidate_params, make_grasp_targets, make_object_cloud, write_body_model
from saga_amd.vposer import VPoserDecoder

h = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]
tmp = tempfile.TemporaryDirectory()
write_body_model(tmp.name, gender='male', seed=7)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for B, O, env in ((64, 2048, {}), (9, 2048, {'SAGA_FIT_NO_PRE': '1'}), (5, 1000, {}), (4, 37, {'SAGA_FIT_NO_PRE': '1'})):
    os.environ.update(env)
    op = FittingOP({'batch_size': B, 'device': 'cuda', 'body_model_path': tmp.name, 'vposer': VPoserDecoder(seed=0), 'genders': ('male',)})
    op.bm = op.bm_male
    op.set_variables(make_candidate_params(B, seed=1234, scale=0.3))
    vo, no, _, _ = make_object_cloud(B, n_points=O, seed=0)
    mk, mc, oc = make_grasp_targets(B, vo, seed=4815)
    prob = op.make_problem(cu(mk), cu(vo), cu(no), cu(oc), cu(mc))
    terms = torch.zeros(B, 16, device='cuda')
    for _ in range(3):
        loss, mrec = op.native_step(2, prob, terms)
    H = op.tables.n_hand
    w = lambda n, s, dt=torch.float32: h(op.workspace_view(n, O, s, dt))
    print('step B=%d O=%d %s loss %s terms %s vars %s h2o %s o2h %s idx %s %s %s nrm %s' % (
        B, O, env, h(loss), h(terms), h(op.vars), w('h2o_signed', (B, H)), w('o2h_signed', (B, O)),
        w('idx_m2o', (B, 143), torch.int32), w('idx_h2o', (B, H), torch.int32), w('idx_o2h', (B, O), torch.int32),
        w('hand_normals', (B, H, 3))))
    for k in env:
        os.environ.pop(k)
    op.close()
g = torch.Generator().manual_seed(5)
for N, P1, P2 in ((3, 778, 2048), (2, 143, 2048), (2, 3000, 9000), (1, 10475, 20000), (4, 100, 64)):
    x = torch.randn(N, P1, 3, generator=g).cuda()
    y = (0.5 * torch.randn(N, P2, 3, generator=g) + 0.3).cuda()
    for env in ({}, {'SAGA_NN_VALU_PASS': '1'}):
        # (the VALU-pass switch is read once per process: only the first setting of a process takes effect)
        d1, d2, i1, i2 = ChamferDistance(x, y)
    print('chamfer N=%d %dx%d d1 %s d2 %s i1 %s i2 %s' % (N, P1, P2, h(d1), h(d2), h(i1), h(i2)))
