"""Experiment: two independent half-batches (2 x 2048 candidates) iterating on two HIP streams against one batch of 4096
on one stream -- does the MFMA-bound part of one half overlap the VALU-bound part of the other?"""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from saga_amd.synthetic import write_body_model
tmp = tempfile.TemporaryDirectory(); write_body_model(tmp.name, gender='male', seed=7)
K = 30
def run(ops, streams):
    for _ in range(5):
        for (op, d), s in zip(ops, streams):
            with torch.cuda.stream(s): bench.step(op, d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        for (op, d), s in zip(ops, streams):
            with torch.cuda.stream(s): bench.step(op, d)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
op, vp, d = bench.build_problem(4096, 2048, 0, tmp.name, 'cuda:0')
print('1 stream  x 4096: %.3f ms/iter' % run([(op, d)], [torch.cuda.current_stream()]))
del op, d
for n in (2, 4):
    ops = []
    for r in range(n):
        o, _, dd = bench.build_problem(4096 // n, 2048, r, tmp.name, 'cuda:0')
        ops.append((o, dd))
    same = [torch.cuda.current_stream()] * n
    print('%d batches x %d, one stream : %.3f ms/iter' % (n, 4096 // n, run(ops, same)))
    streams = [torch.cuda.Stream() for _ in range(n)]
    print('%d batches x %d, %d streams  : %.3f ms/iter' % (n, 4096 // n, n, run(ops, streams)))
    del ops
