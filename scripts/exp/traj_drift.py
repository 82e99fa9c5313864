"""Where does the HIP fitting trajectory leave the oracle's?  Prints per-iteration relative loss difference for a schedule."""
import sys, os, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import fitting_ref
from tests.helpers import load_oracle_model
from saga_amd.synthetic import write_body_model, make_object_cloud, make_grasp_targets
from saga_amd.fitting import FittingOP, load_grasp_indices
from saga_amd.vposer import VPoserDecoder
num_iter = tuple(int(x) for x in sys.argv[1].split(',')) if len(sys.argv) > 1 else (2, 2, 206)
B = 2
root = tempfile.mkdtemp(); write_body_model(root, gender='male', seed=7)
vp = VPoserDecoder(seed=0)
vo, no, _, _ = make_object_cloud(B, seed=0); mk, mc, oc = make_grasp_targets(B, vo, seed=4815)
op = FittingOP({'batch_size': B, 'device': 'cuda', 'body_model_path': root, 'vposer': vp, 'genders': ('male',), 'num_iter': list(num_iter)})
cu = lambda a: torch.from_numpy(a).cuda()
mfit, params, sl = op.fitting(cu(mk), cu(oc), cu(mc), cu(vo), cu(no), 'male')
idx_np = load_grasp_indices()
idx = {'markers_143': torch.from_numpy(idx_np['markers_143']), 'foot': torch.from_numpy(idx_np['foot']), 'rhand': torch.from_numpy(idx_np['rhand']), 'v_weights': torch.from_numpy(idx_np['rhand_weight'])}
t = torch.from_numpy
for dt in (torch.float32, torch.float64):
    bm = load_oracle_model(root, B, dtype=dt)
    c = lambda a: t(a).to(dt)
    rec = []
    p, m, hist = fitting_ref.fitting(bm, vp.weights_dict(dt), idx, c(mk), c(oc), c(mc), c(vo), c(no), num_iter=num_iter,
                                     record=lambda ss, ii, ld, cnt, snap, mr: rec.append({k: v.detach().clone() for k, v in ld.items()}))
    hist = torch.stack(hist).double().numpy()
    rel = np.abs(sl['total'] - hist) / np.abs(hist)
    print(str(dt), 'rel diff by iteration (max over candidates):')
    for i in list(range(0, min(30, len(rel)))) + list(range(30, len(rel), 10)):
        print('  it %4d  hip %s  ref %s  rel %.3g' % (i, sl['total'][i], hist[i], rel[i].max()))
    if dt == torch.float32:
        h32 = hist
    else:
        r = np.abs(h32 - hist) / np.abs(hist)
        print('oracle f32 vs f64: first it with rel > 1e-3:', int(np.argmax(r.max(1) > 1e-3)), ' max', r.max())
