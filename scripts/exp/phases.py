"""Phase timing of the fused loss kernel: needs a build of the library with -DGL_PHASE_TIMING loaded through SAGA_HIP_LIB."""
import os
os.environ.setdefault('SAGA_ALLOW_EXPERIMENT_LIB', '1')      # these scripts load -DSAGA_EXPERIMENT_BUILD libraries (saga_amd/_lib.py refuses them otherwise)
import ctypes, os, sys, json, subprocess
sys.path.insert(0, os.getcwd())
import torch, tempfile
import bench
from saga_amd import _lib
from saga_amd.synthetic import write_body_model
tmp = tempfile.TemporaryDirectory(); write_body_model(tmp.name, gender='male', seed=7)
op, vp, d = bench.build_problem(4096, 2048, 0, tmp.name, 'cuda:0')
for _ in range(3): bench.step(op, d)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
lib.saga_debug_read_phases(buf)
for _ in range(5): bench.step(op, d)
torch.cuda.synchronize()
lib.saga_debug_read_phases(buf)
tot = sum(buf)
names = ['gather queries', 'frames+stage', 'x->y scan', 'x->y resolve', 'x->y terms+foot', 'normals', 'y->x all', 'unused']
for n, v in zip(names, buf): print('%-18s %6.1f%%  %8.0f cycles/block' % (n, 100.0 * v / tot, v / (5 * 4096)))
