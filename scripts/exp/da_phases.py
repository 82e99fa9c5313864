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
