"""Random clouds through point2point_signed (saga_p2p_signed_fwd/bwd): forward against the fp64 oracle restatement of
utils/train_helper.py:17-86 evaluated with the HIP indices (ties aside, the indices are checked against the exact scan), backward in
its two forms (fixed-point LDS accumulation / fixed-order gather, SAGA_P2P_GATHER=1) against the oracle's autograd."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nn_fuzz
from oracle import geometry_ref
from saga_amd.geometry import point2point_signed
import saga_amd.chamfer_distance as chd


def fuzz(n_cases=40, seed=0):
    nn_fuzz.g.manual_seed(seed)
    g, ri, rf = nn_fuzz.g, nn_fuzz.ri, nn_fuzz.rf
    bad = 0
    for case in range(n_cases):
        N = ri(1, 3)
        sizes = [1, 2, 33, 143, 257, 778, 1000, 2048, 2049, 3000]
        P1, P2 = sizes[ri(0, 9)], sizes[ri(0, 9)]
        x, y = nn_fuzz.cloud(N, P1), nn_fuzz.cloud(N, P2)
        if rf() < 0.6:
            y = (y - y.mean(1, keepdim=True)) / y.std().clamp(min=1e-9) * x.std() + x.mean(1, keepdim=True)
        xn = torch.nn.functional.normalize(torch.randn(N, P1, 3, generator=g), dim=-1) if rf() < 0.7 else None
        yn = torch.nn.functional.normalize(torch.randn(N, P2, 3, generator=g), dim=-1) if rf() < 0.7 else None
        wy, wx = torch.randn(N, P2, generator=g), torch.randn(N, P1, generator=g)
        vy, vx = torch.randn(N, P2, 3, generator=g), torch.randn(N, P1, 3, generator=g)
        cu = lambda t: None if t is None else t.cuda()
        res = {}
        for form in ('fixed', 'gather'):
            if form == 'gather':
                os.environ['SAGA_P2P_GATHER'] = '1'
            try:
                xx, yy = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
                out = point2point_signed(xx, yy, cu(xn), cu(yn), return_vector=True)
                ((out[0] * cu(wy)).sum() + (out[1] * cu(wx)).sum() + (out[4] * cu(vy)).sum() + (out[5] * cu(vx)).sum()).backward()
                res[form] = [o.detach().cpu() for o in out] + [xx.grad.cpu(), yy.grad.cpu()]
            finally:
                os.environ.pop('SAGA_P2P_GATHER', None)
        # indices against the exact scan
        os.environ['SAGA_NN_EXACT_SCAN'] = '1'
        try:
            _, _, i1, i2 = chd.ChamferDistance(x.cuda(), y.cuda())
        finally:
            del os.environ['SAGA_NN_EXACT_SCAN']
        got = res['fixed']
        ok = torch.equal(got[3], i1.cpu()) and torch.equal(got[2], i2.cpu())
        # oracle (fp64) with the same indices
        xd, yd = x.double().requires_grad_(True), y.double().requires_grad_(True)
        dn = lambda t: None if t is None else t.double()
        ref = geometry_ref.point2point_signed(xd, yd, dn(xn), dn(yn), return_vector=True, force_idx=(got[2], got[3]))
        ((ref[0] * wy.double()).sum() + (ref[1] * wx.double()).sum() + (ref[4] * vy.double()).sum() + (ref[5] * vx.double()).sum()).backward()
        scale = float(max(x.abs().max(), y.abs().max()))
        err = lambda a, b: float((a.double() - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        e_fwd = max(err(got[0], ref[0].detach()), err(got[1], ref[1].detach()), err(got[4], ref[4].detach()), err(got[5], ref[5].detach()))
        # points that coincide with their neighbour get no distance gradient here (torch: NaN) -- compare where the oracle is finite
        fin_x, fin_y = torch.isfinite(xd.grad).all(-1), torch.isfinite(yd.grad).all(-1)
        e_bwd = {}
        for form in res:
            gx, gy = res[form][6], res[form][7]
            ex = float((gx.double() - xd.grad)[fin_x].abs().max()) if fin_x.any() else 0.0
            ey = float((gy.double() - yd.grad)[fin_y].abs().max()) if fin_y.any() else 0.0
            ref_mag = max(float(xd.grad[fin_x].abs().max()) if fin_x.any() else 0.0, float(yd.grad[fin_y].abs().max()) if fin_y.any() else 0.0, 1e-30)
            e_bwd[form] = max(ex, ey) / ref_mag
        if not ok or e_fwd > 2e-5 or max(e_bwd.values()) > 2e-4:
            bad += 1
            print('MISMATCH case %d N=%d P1=%d P2=%d scale %.3g: indices %s fwd %.2e bwd %s' % (case, N, P1, P2, scale, ok, e_fwd, e_bwd), flush=True)
    print('P2P FUZZ %s: %d cases, %d bad' % ('OK' if bad == 0 else 'FAILED', n_cases, bad), flush=True)
    return bad


if __name__ == '__main__':
    sys.exit(1 if fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
