import sys, os, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from saga_amd.synthetic import write_body_model
from torch.profiler import profile, ProfilerActivity
tmp = tempfile.TemporaryDirectory(); write_body_model(tmp.name, gender='male', seed=7)
op, vp, d = bench.build_problem(4096, 2048, 0, tmp.name, 'cuda:0')
for _ in range(3): bench.step(op, d, op.optimizer_s3)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): bench.step(op, d, op.optimizer_s3)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=35, max_name_column_width=60))
