"""How much nn_prune.h's bounds prune on config 5 (library built with scripts/exp/build_variant.sh nnpstat -DNNP_EXP_STATS)."""
import os
os.environ.setdefault('SAGA_ALLOW_EXPERIMENT_LIB', '1')
os.environ.setdefault('SAGA_HIP_LIB', 'build/variants/libsaga_nnpstat.so')
import ctypes, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saga_amd import _lib
from saga_amd.chamfer_distance import chamfer_nn
from workloads import dense_chamfer_clouds
B = int(os.environ.get('NNP_B', 64))
x, y = dense_chamfer_clouds(B, 10475, 100000)
lib = _lib.load()
out = (ctypes.c_ulonglong * 16)()
names = ['blocks', 'tiles staged', 'wave-tile visits', 'wave-tile skips', 'active (group, query tile) pairs', 'groups with any']
P = _lib.ptr
N, P1, P2 = x.shape[0], x.shape[1], y.shape[1]
nb = lib.saga_chamfer_workspace_bytes(N, P1, P2)
ws = torch.empty(nb, device='cuda', dtype=torch.uint8)
i1 = torch.empty(N, P1, device='cuda', dtype=torch.int32)
i2 = torch.empty(N, P2, device='cuda', dtype=torch.int32)
for label, a1, a2 in (('x->y (10475 queries, 100k db)', i1, None), ('y->x (100k queries, 10475 db)', None, i2)):
    lib.saga_debug_nnp_stats(out, 1)
    _lib.call('saga_chamfer_fwd_ws', x.device, P(x), P(y), None, None, P(a1), P(a2), N, P1, P2, P(ws), nb)
    torch.cuda.synchronize()
    lib.saga_debug_nnp_stats(out, 1)
    v = list(out)
    print(label, {n: v[i] for i, n in enumerate(names)})
    print('  tiles staged per block %.2f; visited wave-tiles %.1f%%; active pairs per visited wave-tile %.1f of 512; groups %.1f of 64'
          % (v[1] / max(v[0], 1), 100.0 * v[2] / max(v[2] + v[3], 1), v[4] / max(v[2], 1), v[5] / max(v[2], 1)))
    print('  after the per-query refinement: %.1f pairs in %.1f groups per visited wave-tile' % (v[6] / max(v[2], 1), v[7] / max(v[2], 1)))
    tot = max(v[13], 1)
    print('  wave cycles: top..B1 %.1f%%, staging+tests %.1f%%, B2 wait %.1f%%, matrix-core pass %.1f%%, exact pass %.1f%% of %.0f k cycles per wave'
          % (100 * v[8] / tot, 100 * v[9] / tot, 100 * v[10] / tot, 100 * v[11] / tot, 100 * v[12] / tot, tot / max(v[0], 1) / 1e3))
    print('  slowest wave %.0f k cycles; %d waves above 4 M cycles' % (v[14] / 1e3, v[15]))
