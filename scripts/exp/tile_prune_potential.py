"""VERDICT r2 item 4, second half: would a spatially SORTED object cloud let the fused loss kernel skip matrix tiles?

The kernel's approximate pass works in units of one matrix instruction = 32 queries x 32 database points.  A unit can be skipped only
if, for EVERY query of the 32, the 32 points' bounding box is farther than that query's nearest neighbour.  This script measures the
fraction of units that an IDEAL pruning rule (it knows every query's true nearest distance in advance) would still have to run, on the
bench workload: database tiles in the given order against a Morton-sorted order, for both directions of the kernel (hand + markers ->
object; object -> hand), queries in the kernel's order and in sorted order.  Anything a real rule achieves is above these numbers.
Prints one JSON line per body model (SURVEY synthetic: random geometry; --mesh-faces: id-local faces and weights)."""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from saga_amd.synthetic import write_body_model


def morton_order(p):
    """p [B,N,3] -> permutation [B,N] by 30-bit Morton code of each batch element's own bounding box."""
    lo, hi = p.min(1, keepdim=True).values, p.max(1, keepdim=True).values
    q = ((p - lo) / (hi - lo).clamp(min=1e-12) * 1023).long().clamp(0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    code = spread(q[..., 0]) | (spread(q[..., 1]) << 1) | (spread(q[..., 2]) << 2)
    return code.argsort(dim=1)


def needed_fraction(q, db):
    """q [B,Nq,3], db [B,Nd,3] (both padded by repeating their last point to multiples of 32) -> fraction of (query tile, database
    tile) units in which at least one query's nearest neighbour could lie: box distance <= true nearest distance."""
    pad = lambda t: torch.cat([t, t[:, -1:].expand(-1, (-t.shape[1]) % 32, -1)], 1)
    q, db = pad(q), pad(db)
    B = q.shape[0]
    d_nn = torch.cdist(q, db).min(dim=2).values                          # [B,Nq]
    t = db.reshape(B, -1, 32, 3)
    lo, hi = t.min(2).values, t.max(2).values                            # [B,Td,3]
    gap = torch.maximum(torch.maximum(lo[:, None] - q[:, :, None], q[:, :, None] - hi[:, None]), torch.zeros((), device=q.device))
    lb = gap.norm(dim=-1)                                                # [B,Nq,Td]
    need = (lb <= d_nn[:, :, None] * (1 + 1e-6)).reshape(B, -1, 32, lb.shape[-1]).any(dim=2)
    return float(need.float().mean())


def main():
    out = []
    for mesh in (False, True):
        tmp = tempfile.TemporaryDirectory()
        write_body_model(tmp.name, gender='male', seed=7, mesh_faces=mesh, local_weights=mesh)
        op, vp, d = bench.build_problem(64, 2048, 0, tmp.name, 'cuda:0')
        for _ in range(20):                      # a few iterations in: the candidates have moved towards their targets
            bench.step(op, d)
        with torch.no_grad():
            verts = op.bm(return_verts=True, **op.body_params()).vertices
            hand = verts[:, op.rhand_verts]
            q_xy = torch.cat([hand, verts[:, op.markers_143]], 1)        # the kernel's query order: hand, then markers
            obj = d['vo']
            g = lambda t, perm: torch.gather(t, 1, perm[..., None].expand(-1, -1, 3))
            obj_s, hand_s, q_s = g(obj, morton_order(obj)), g(hand, morton_order(hand)), g(q_xy, morton_order(q_xy))
            res = {'model': 'mesh-like faces, local weights' if mesh else 'SURVEY synthetic (random geometry)',
                   'hand_to_object': {'as_is': needed_fraction(q_xy, obj), 'object_sorted': needed_fraction(q_xy, obj_s),
                                      'object_and_queries_sorted': needed_fraction(q_s, obj_s)},
                   'object_to_hand': {'as_is': needed_fraction(obj, hand), 'hand_sorted': needed_fraction(obj, hand_s),
                                      'hand_and_queries_sorted': needed_fraction(obj_s, hand_s)},
                   'hand_box_m': [round(float(x), 3) for x in (hand.max(1).values - hand.min(1).values).mean(0)],
                   'object_box_m': [round(float(x), 3) for x in (obj.max(1).values - obj.min(1).values).mean(0)]}
        out.append(res)
        print(json.dumps(res))
    return out


if __name__ == '__main__':
    main()
