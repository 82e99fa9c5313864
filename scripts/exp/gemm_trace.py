"""Per-block timeline of the persistent forward blend GEMM (library built with -DGEMM_EXP_TRACE: scripts/exp/build_variant.sh tr
-DGEMM_EXP_TRACE; run with SAGA_HIP_LIB=build/variants/libsaga_tr.so): which blocks share a CU, and whether their prologues /
epilogues coincide."""
import os
os.environ.setdefault('SAGA_ALLOW_EXPERIMENT_LIB', '1')      # these scripts load -DSAGA_EXPERIMENT_BUILD libraries (saga_amd/_lib.py refuses them otherwise)
import os, sys, tempfile, ctypes, collections
os.environ.setdefault('SAGA_GEMM_PERSIST', '-1')     # the stamps are taken in the persistent form of the launch (two blocks per CU walking the tiles)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from saga_amd import _lib, smplx
from saga_amd.synthetic import write_body_model
tmp = tempfile.TemporaryDirectory(); write_body_model(tmp.name, gender='male', seed=7)
B = 4096
bm = smplx.create(tmp.name, model_type='smplx', gender='male', ext='npz', num_pca_comps=24, batch_size=B).cuda().eval()
g = torch.Generator().manual_seed(1)
args = dict(body_pose=0.3 * torch.randn(B, 63, generator=g).cuda(), global_orient=torch.randn(B, 3, generator=g).cuda())
with torch.no_grad():
    for _ in range(3): bm(**args)
    torch.cuda.synchronize()
lib = _lib.load()
STRIDE, NB = 96, 512
buf = (ctypes.c_ulonglong * (NB * STRIDE))()
lib.saga_debug_gemm_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.saga_debug_gemm_trace(buf, NB * STRIDE); assert rc == 0, rc
a = np.frombuffer(buf, dtype=np.uint64).reshape(NB, STRIDE).astype(np.int64)
hw = a[:, 0] & 0xffffffff; xcc = a[:, 0] >> 32
wave_id = hw & 15; simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
groups = collections.defaultdict(list)
for b in range(NB): groups[(int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b]))].append(b)
print('CUs seen: %d; blocks per CU: %s' % (len(groups), collections.Counter(len(v) for v in groups.values())))
print('wave slot of wave 0, by block half:', collections.Counter((int(b >= 256), int(wave_id[b])) for b in range(NB)))
ntile = 15
ev6 = a[:, 1:1 + 6 * ntile].reshape(NB, ntile, 6)     # tile start, loop start (stage 0 landed), loop end, epilogue reads issued, landed, stores issued
ev = ev6[:, :, [0, 1, 2, 5]]
pro = (ev[:, :, 1] - ev[:, :, 0]); loop = ev[:, :, 2] - ev[:, :, 1]; epi = ev[:, :, 3] - ev[:, :, 2]
print('cycles (clock64 ticks) per tile, median over blocks and tiles 1..: prologue %d, k-loop %d, epilogue %d' %
      (np.median(pro[:, 1:]), np.median(loop[:, 1:]), np.median(epi[:, 1:])))
print('epilogue parts (median, tiles 1..): issue reads %d, reads land %d, accumulators + stores issued %d' % tuple(np.median((ev6[:, 1:, k + 1] - ev6[:, 1:, k])) for k in (2, 3, 4)))
print('first tile: prologue %d, k-loop %d, epilogue %d' % (np.median(pro[:, 0]), np.median(loop[:, 0]), np.median(epi[:, 0])))
shown = 0
for key, bl in sorted(groups.items()):
    if len(bl) != 2: continue
    b0, b1 = bl
    t0 = min(ev[b0, 0, 0], ev[b1, 0, 0])
    print('CU', key, 'blocks', bl, 'slots', int(wave_id[b0]), int(wave_id[b1]))
    for b in bl:
        print('   block %3d:' % b, ' '.join('%d/%d/%d/%d' % tuple((ev[b, i] - t0) // 100) for i in range(6)), ' (x100 ticks: start/loop/epi/end)')
    # overlap of the no-MFMA windows (epilogue + next prologue) of the two blocks over tiles 1..13
    def windows(b): return [(ev[b, i, 2], ev[b, i + 1, 1]) for i in range(ntile - 1)]
    ov = 0; tot = 0
    for (s0, e0) in windows(b0):
        tot += e0 - s0
        for (s1, e1) in windows(b1): ov += max(0, min(e0, e1) - max(s0, s1))
    print('   no-MFMA window per tile %d ticks; overlap with the partner\'s windows: %.0f %%' % (tot / (ntile - 1), 100.0 * ov / tot))
    shown += 1
    if shown >= 6: break
# are the epilogues of ALL blocks clustered in time (a chip-wide write burst)?  per XCD (its own clock): spread of the epilogue starts
for x in sorted(set(xcc.tolist()))[:2]:
    sel = np.where(xcc == x)[0]
    t0 = ev[sel, 0, 0].min()
    for i in (0, 1, 4, 8, 12):
        st = np.sort(ev[sel, i, 2] - t0); en = np.sort(ev[sel, i, 3] - t0)
        print('XCD %d tile %2d: epilogue starts p5 %7d p50 %7d p95 %7d | ends p5 %7d p50 %7d p95 %7d' %
              (x, i, st[len(st) // 20], st[len(st) // 2], st[-len(st) // 20], en[len(en) // 20], en[len(en) // 2], en[-len(en) // 20]))
    # occupancy of the epilogue phase over time: how many of the XCD's 64 blocks are in an epilogue at once
    grid = np.arange(0, int(ev[sel, ntile - 1, 3].max() - t0), 2000)
    cnt = np.zeros(len(grid), int)
    for b in sel:
        for i in range(ntile):
            cnt += ((grid >= ev[b, i, 2] - t0) & (grid < ev[b, i, 3] - t0))
    print('XCD %d: blocks in an epilogue at once: mean %.1f, p50 %d, p90 %d, max %d (of %d)' % (x, cnt.mean(), np.percentile(cnt, 50), np.percentile(cnt, 90), cnt.max(), len(sel)))
