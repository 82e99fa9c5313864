"""Random object clouds (sizes 1 .. 4,100, shapes, scales, far sentinel points, duplicates, NaN points) through the fused loss kernel:
its three neighbour-index outputs and signed distances against the stand-alone ChamferDistance run as the plain exact scan."""
import os, sys, tempfile, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import saga_amd.chamfer_distance as chd
from saga_amd.grasp_loss import grasp_loss_terms
from saga_amd.synthetic import write_body_model
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nn_fuzz


def fuzz(n_cases=50, seed=0, B=6):
    nn_fuzz.g.manual_seed(seed)
    g, ri, rf = nn_fuzz.g, nn_fuzz.ri, nn_fuzz.rf
    tmp = tempfile.TemporaryDirectory()
    write_body_model(tmp.name, gender='male', seed=7)
    op, vp, d = bench.build_problem(B, 2048, 0, tmp.name, 'cuda:0')
    with torch.no_grad():
        verts = op.bm(return_verts=True, **op.body_params()).vertices
    hand, mk = verts[:, op.rhand_verts].contiguous(), verts[:, op.markers_143].contiguous()
    bad = 0
    for case in range(n_cases):
        O = [1, 2, 31, 33, 100, 777, 1000, 2047, 2048, 2049, 3000, 4100][ri(0, 11)]
        vo = nn_fuzz.cloud(B, O).cuda()
        if rf() < 0.7:          # around the hand, at the hand's scale
            vo = (vo - vo.mean(1, keepdim=True)) / vo.std().clamp(min=1e-6) * 0.1 * (rf() + 0.1) + hand.mean(1, keepdim=True)
        mode = ri(0, 3)
        if mode == 1 and O > 3:
            vo[:, ri(0, O - 1)] = 10.0 ** ri(3, 9)            # a far sentinel point
        elif mode == 2 and O > 3:
            vo[:, ri(0, O - 1), ri(0, 2)] = float('nan')     # a NaN coordinate
        elif mode == 3 and O > 3:
            vo[:, O // 2:] = vo[:, :O - O // 2]               # duplicates: exact ties
        vo = vo.contiguous()
        no = torch.nn.functional.normalize(torch.randn(B, O, 3, generator=g), dim=-1).cuda()
        oc = torch.rand(B, O, generator=g).cuda()
        r = grasp_loss_terms(verts, op.bm, op.tables, vo, no, d['markers'], d['mc'], oc)
        os.environ['SAGA_NN_EXACT_SCAN'] = '1'
        try:
            _, _, i_h2o, i_o2h = chd.ChamferDistance(hand, vo)
            _, _, i_m2o, _ = chd.ChamferDistance(mk, vo)
            torch.cuda.synchronize()
        finally:
            del os.environ['SAGA_NN_EXACT_SCAN']
        okq = torch.isfinite(vo).all(-1)                     # object points that are queries of the object -> hand direction
        same = [torch.equal(r['idx_h2o'], i_h2o), torch.equal(r['idx_m2o'], i_m2o), torch.equal(r['idx_o2h'][okq], i_o2h[okq])]
        if not all(same):
            bad += 1
            print('MISMATCH case %d O=%d mode=%d (h2o, m2o, o2h) %s: %d %d %d' % (case, O, mode, same, int((r['idx_h2o'] != i_h2o).sum()),
                  int((r['idx_m2o'] != i_m2o).sum()), int((r['idx_o2h'][okq] != i_o2h[okq]).sum())), flush=True)
    print('LOSS FUZZ %s: %d cases, %d mismatching' % ('OK' if bad == 0 else 'FAILED', n_cases, bad), flush=True)
    return bad


if __name__ == '__main__':
    sys.exit(1 if fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 50, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
