"""Non-finite / huge coordinates in a cloud: the pruned search, the tiled two-level scan and the plain exact scan must agree on every
query that is finite itself (a bad point never wins and must not disturb its neighbours)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import saga_amd.chamfer_distance as chd
g = torch.Generator().manual_seed(3)
x = (torch.randn(2, 5000, 3, generator=g) * 0.3).cuda()
y = (torch.randn(2, 6000, 3, generator=g) * 0.3).cuda()


def run(env, a, b):
    os.environ.update(env)
    try:
        out = chd.ChamferDistance(a, b)
        torch.cuda.synchronize()
    finally:
        for k in env:
            del os.environ[k]
    return out


for name, bad in (('nan', float('nan')), ('inf', float('inf')), ('-inf', float('-inf')), ('huge', 3e38), ('1e10', 1e10)):
    xb, yb = x.clone(), y.clone()
    yb[0, 17, 1] = bad; yb[1, 4000] = bad; xb[0, 100, 0] = bad
    pr = run({}, xb, yb)
    ti = run({'SAGA_NN_PRUNE': '0'}, xb, yb)
    ex = run({'SAGA_NN_EXACT_SCAN': '1'}, xb, yb)
    okq = (torch.isfinite(xb).all(-1) & (xb.abs().amax(-1) < 1e15), torch.isfinite(yb).all(-1) & (yb.abs().amax(-1) < 1e15))
    for lab, o in (('pruned', pr), ('tiled', ti)):
        bad_n = [int((o[2 + d][okq[d]] != ex[2 + d][okq[d]]).sum()) for d in range(2)]
        bad_d = [int((o[d][okq[d]] != ex[d][okq[d]]).sum()) for d in range(2)]
        print('%-5s %-6s vs exact scan: index mismatches (x->y, y->x) %s, distance mismatches %s' % (name, lab, bad_n, bad_d), flush=True)
