"""Runs on the GPU box: one case of the step fuzz with per-candidate differences printed.  usage: step_fuzz_debug.py B O stage objective_index seed"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from saga_amd.synthetic import write_body_model  # noqa: E402
import tests.test_gpu_fitting as t  # noqa: E402
from oracle import loss_ref  # noqa: E402

B, O, stage, oi, seed = (int(a) for a in sys.argv[1:6])
tmp = tempfile.TemporaryDirectory()
write_body_model(tmp.name, gender='male', seed=7)
contact_loss, only_rec = t.OBJECTIVES[oi]
op_n, vp, idx, var_np, d = t._setup(tmp.name, B, contact_loss=contact_loss, only_rec=only_rec, n_points=O, seed=seed)
op_a, *_ = t._setup(tmp.name, B, contact_loss=contact_loss, only_rec=only_rec, n_points=O, seed=seed)
op_a.native = False
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for op in (op_n, op_a):
    op.bm = op.bm_male
    op.set_variables(var_np)
prob = op_n.make_problem(cu(d['markers']), cu(d['verts_o']), cu(d['normals_o']), cu(d['oc']), cu(d['mc']))
terms = torch.zeros(B, 16, device='cuda')
loss_n, mk_n = op_n.native_step(stage, prob, terms)
loss_a, ld, mk_a, *_ = op_a.calc_loss_contact_map(cu(d['markers']), cu(d['verts_o']), cu(d['normals_o']), cu(d['oc']), cu(d['mc']),
                                                  'male', None, stage, 1.0)
dm = (mk_n - mk_a).abs().amax(dim=(1, 2)).cpu().numpy()
print('marker diff per candidate: max %.3e at candidate %d; median %.3e' % (dm.max(), int(dm.argmax()), float(np.median(dm))))
w = int(dm.argmax())
with torch.no_grad():
    bp = op_a.body_params()
    for k in ('global_orient', 'body_pose'):
        a = bp[k].detach().reshape(B, -1, 3)[w].norm(dim=-1).cpu().numpy()
        print('  worst candidate', k, 'joint angles (rad): max %.4f' % a.max(), 'pi - max = %.2e' % (np.pi - a.max()))
bp_n = op_n.body_params()
with torch.no_grad():
    bp_a = op_a.body_params()
for k in bp_n:
    dd = (bp_n[k].detach().reshape(B, -1) - bp_a[k].detach().reshape(B, -1)).abs().amax(dim=1).cpu().numpy()
    if dd.max() > 0:
        print('body param', k, 'differs: rows', np.nonzero(dd)[0].tolist()[:20], 'max', dd.max())
t_ = terms.cpu().numpy()
for i, k in enumerate(loss_ref.LOSS_KEYS):
    a = ld[k].detach().cpu().numpy()
    rel = np.abs(t_[:, i] - a) / np.maximum(np.abs(a), 1e-7)
    if rel.max() > 1e-4:
        print('term', k, 'rows', np.nonzero(rel > 1e-4)[0].tolist(), 'native', t_[rel > 1e-4, i], 'autograd', a[rel > 1e-4])

# both paths' markers against the float64 oracle on the same variables: which one carries the difference?
from tests.helpers import load_oracle_model  # noqa: E402
bm64 = load_oracle_model(tmp.name, B, dtype=torch.float64)
var = {k: torch.from_numpy(v).double() for k, v in var_np.items()}
t64 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()
with torch.no_grad():
    total, ld64, cnt, aux = loss_ref.loss_batched(bm64, var, vp.weights_dict(torch.float64), idx, t64(d['markers']), t64(d['verts_o']),
                                                  t64(d['normals_o']), t64(d['oc']), t64(d['mc']), stage, only_rec=only_rec,
                                                  contact_loss=contact_loss)
ref = aux['markers_rec'].numpy()
dn = np.abs(mk_n.detach().cpu().numpy() - ref).max(axis=(1, 2))
da = np.abs(mk_a.detach().cpu().numpy() - ref).max(axis=(1, 2))
print('against the float64 oracle: native max %.3e (candidate %d), autograd composition max %.3e (candidate %d); at the worst candidate of the pair: native %.3e autograd %.3e'
      % (dn.max(), int(dn.argmax()), da.max(), int(da.argmax()), dn[w], da[w]))
