#!/usr/bin/env python3
"""Whole GraspPose-Opt fit (opt_grasppose.py:179-191 iteration counts: 300 + 400 + 500 Adam iterations over three stages) for B
candidates x 2,048 object points on one GPU: wall time of FittingOP.fitting in the reference's behaviour and with the two opt-in
product switches (sparse_backward, log_gated_terms=False).  Synthetic body model and targets.  Prints one JSON line."""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saga_amd.fitting import FittingOP
from saga_amd.synthetic import make_grasp_targets, make_object_cloud, write_body_model
from saga_amd.vposer import VPoserDecoder

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=4096)
ap.add_argument('--iters', type=int, nargs=3, default=[300, 400, 500])
a = ap.parse_args()
tmp = tempfile.TemporaryDirectory()
write_body_model(tmp.name, gender='male', seed=7)
B = a.batch
vo, no, _, _ = make_object_cloud(B, n_points=2048, seed=0)
mk, mc, oc = make_grasp_targets(B, vo, seed=4815)
cu = lambda x: torch.from_numpy(x).cuda()
args = (cu(mk), cu(oc), cu(mc), cu(vo), cu(no), 'male')
res = {}
for name, kw in (('reference behaviour', {}), ('log_gated_terms=False', {'log_gated_terms': False}),
                 ('log_gated_terms=False + sparse_backward', {'log_gated_terms': False, 'sparse_backward': True})):
    op = FittingOP({'batch_size': B, 'device': 'cuda', 'body_model_path': tmp.name, 'vposer': VPoserDecoder(seed=0),
                    'genders': ('male',), 'num_iter': [2, 2, 2], 'record_history': False, **kw})
    op.fitting(*args)                                   # warm-up: models, workspace
    op.num_iter = list(a.iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, params, _ = op.fitting(*args)
    torch.cuda.synchronize()
    res[name] = {'seconds': round(time.perf_counter() - t0, 3), 'mean_transl_z': round(float(params[-1]['transl'][:, 2].mean()), 5)}
print(json.dumps({'config': 'GraspPose-Opt fit, B=%d candidates x 2048 object points, %d+%d+%d iterations' % (B, *a.iters), 'fits': res}))
