"""Phase timing of skin_bwd_dA_kernel (s_memtime stamps, cycles per wave per step).  Needs a build of the library with
-DDA_PHASE_TIMING (every .hip of saga_amd/csrc compiled with the Makefile's flags plus that define, linked to e.g.
scripts/exp/libsaga_DAT.so) and  SAGA_HIP_LIB=scripts/exp/libsaga_DAT.so python scripts/exp/da_phases.py"""
import os
os.environ.setdefault('SAGA_ALLOW_EXPERIMENT_LIB', '1')      # these scripts load -DSAGA_EXPERIMENT_BUILD libraries (saga_amd/_lib.py refuses them otherwise)
import ctypes, os, sys, tempfile
sys.path.insert(0, os.getcwd())
import torch
import bench
from saga_amd import _lib
from saga_amd.synthetic import write_body_model
tmp = tempfile.TemporaryDirectory(); write_body_model(tmp.name, gender='male', seed=7)
op, vp, d = bench.build_problem(4096, 2048, 0, tmp.name, 'cuda:0')
for _ in range(3): bench.step(op, d)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
lib.saga_debug_read_da_phases(buf)
N = 5
for _ in range(N): bench.step(op, d)
torch.cuda.synchronize()
lib.saga_debug_read_da_phases(buf)
names = ['issue Wd loads', 'wait rows + stage to LDS', 'g_v_posed (T, LDS gathers)', 'store + prefetch issue', 'A products + 48 MFMA', '-']
waves = N * 4096
for n, v in zip(names, buf): print('%-30s %8.0f cycles per wave per step' % (n, v / waves / 164.0))
