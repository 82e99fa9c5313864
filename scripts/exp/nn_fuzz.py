"""Random clouds of many shapes / scales / sizes through ChamferDistance: the pruned search and the two-level scan against the plain
exact scan (SAGA_NN_EXACT_SCAN=1), bit for bit.  usage: python scripts/exp/nn_fuzz.py [cases] [seed]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import saga_amd.chamfer_distance as chd

g = torch.Generator()
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
rf = lambda: float(torch.rand(1, generator=g))


def cloud(N, P):
    kind = ri(0, 7)
    if kind == 0:
        p = torch.randn(N, P, 3, generator=g) * torch.tensor([rf() + 0.01, rf() + 0.01, rf() + 0.01])
    elif kind == 1:
        p = torch.rand(N, P, 3, generator=g)
    elif kind == 2:                                    # clusters
        c = torch.randn(N, 8, 3, generator=g)
        p = c[:, torch.randint(0, 8, (P,), generator=g)] + 0.01 * torch.randn(N, P, 3, generator=g)
    elif kind == 3:                                    # a line
        t = torch.rand(N, P, 1, generator=g)
        p = t * torch.randn(N, 1, 3, generator=g) + 1e-4 * torch.randn(N, P, 3, generator=g)
    elif kind == 4:                                    # lattice: many exact ties
        p = torch.randint(0, 12, (N, P, 3), generator=g).float() * 0.125
    elif kind == 5:                                    # duplicates of a few points
        base = torch.randn(N, max(P // 50, 1), 3, generator=g)
        p = base[:, torch.randint(0, base.shape[1], (P,), generator=g)]
    elif kind == 6:                                    # a plane + outliers
        p = torch.randn(N, P, 3, generator=g) * torch.tensor([1.0, 1.0, 1e-3])
        p[:, :max(P // 100, 1)] *= 50
    else:                                              # a sphere shell
        p = torch.nn.functional.normalize(torch.randn(N, P, 3, generator=g), dim=-1)
    scale = 10.0 ** ri(-3, 3)
    return (p * scale + torch.randn(N, 1, 3, generator=g) * scale * ri(0, 3) * 10).contiguous()


def run(env, a, b):
    os.environ.update(env)
    try:
        out = chd.ChamferDistance(a, b)
        torch.cuda.synchronize()
    finally:
        for k in env:
            del os.environ[k]
    return out


def fuzz(n_cases=100, seed=0):
    g.manual_seed(seed)
    bad, t0 = 0, time.time()
    for case in range(n_cases):
        N = ri(1, 3)
        sizes = [1, 2, 31, 33, 255, 257, 777, 2047, 2049, 4095, 4096, 4097, 6000, 9001, 12345, 20000]
        P1, P2 = sizes[ri(0, len(sizes) - 1)], sizes[ri(0, len(sizes) - 1)]
        x, y = cloud(N, P1).cuda(), cloud(N, P2).cuda()
        if rf() < 0.3:
            y = (y - y.mean(1, keepdim=True) + x.mean(1, keepdim=True)).contiguous()          # overlapping clouds
        ex = run({'SAGA_NN_EXACT_SCAN': '1'}, x, y)
        for name, env in (('pruned', {}), ('two-level', {'SAGA_NN_PRUNE': '0'})):
            out = run(env, x, y)
            okk = all(torch.equal(a, b) for a, b in zip(out, ex))
            if not okk:
                bad += 1
                mm = [int((a != b).sum()) for a, b in zip(out, ex)]
                print('MISMATCH case %d %s N=%d P1=%d P2=%d mismatches (d1,d2,i1,i2) %s' % (case, name, N, P1, P2, mm), flush=True)
        if case % 20 == 19:
            print('case %d done, %d mismatching runs, %.0f s' % (case + 1, bad, time.time() - t0), flush=True)
    print('FUZZ %s: %d cases, %d mismatching runs' % ('OK' if bad == 0 else 'FAILED', n_cases, bad))
    return bad


if __name__ == '__main__':
    sys.exit(1 if fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
