"""Time per iteration of GraspMotion-Opt stage 1 (per-frame marker fit) at S sequences."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from saga_amd.fitting import load_grasp_indices
from saga_amd.motion_fitting import FittingOP
from saga_amd.synthetic import write_body_model
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, n_it = 3, 40
tmp = tempfile.TemporaryDirectory(); write_body_model(tmp.name, gender='male', seed=7)
idx = load_grasp_indices()
g = torch.Generator().manual_seed(0)
op = FittingOP(dict(T=T, n_seq=S, gender='male', device='cuda', smplx_beta=0.5 * torch.randn(S, 10, generator=g).numpy(), num_iter=[n_it, 0],
                    markers_ids=idx['markers_f0_p5'], markers_ids_143=idx['markers_143'], body_model_path=tmp.name, indices=idx))
rec = torch.randn(S, T, 79, 3, generator=g) * 0.3 + torch.tensor([0.0, 0.4, 1.0])
op.fitting_stage1(rec); torch.cuda.synchronize()
t0 = time.perf_counter(); op.fitting_stage1(rec); torch.cuda.synchronize()
print('S=%d: %.3f ms per stage-1 iteration' % (S, (time.perf_counter() - t0) / (T * n_it) * 1e3))
