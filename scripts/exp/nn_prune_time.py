"""config 5 through the pruned search a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saga_amd.chamfer_distance import ChamferDistance
from workloads import dense_chamfer_clouds, timed
B = int(os.environ.get('NNP_B', 512))
x, y = dense_chamfer_clouds(B, 10475, 100000)
print('ms', timed(lambda: ChamferDistance(x, y), 3, 1))
