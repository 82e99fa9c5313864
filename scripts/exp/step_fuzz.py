"""Runs on the GPU box: more cases of tests/test_gpu_fitting.py::test_fit_step_size_fuzz_against_autograd_composition (random batch and
object-cloud sizes, stages, objectives; iteration 0 of saga_fit_step against the autograd composition).  usage: step_fuzz.py [n_cases] [seed]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from saga_amd.synthetic import write_body_model  # noqa: E402
import tests.test_gpu_fitting as t  # noqa: E402

n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 60), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.TemporaryDirectory()
write_body_model(tmp.name, gender='male', seed=7)
rng = np.random.default_rng(seed)
bad = 0
for case in range(n):
    B = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 31, 33, 64, 65, 127, 129]))
    O = int(rng.choice([1, 2, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 500, 777, 1023, 1025, 2047, 2048, 2049, 2500, 4097]))
    stage, objective = int(rng.integers(0, 3)), t.OBJECTIVES[int(rng.integers(0, 4))]
    try:
        t._step_fuzz_case(tmp.name, B, O, stage, objective, seed=seed * 1000 + case)
    except AssertionError as e:
        bad += 1
        print('MISMATCH case', case, 'seed', seed * 1000 + case, 'args', B, O, stage, t.OBJECTIVES.index(objective), str(e)[:300].replace('\n', ' '), flush=True)
    if case % 10 == 9:
        print('case', case + 1, 'mismatches', bad, flush=True)
print('STEP_FUZZ_DONE cases', n, 'mismatches', bad)
sys.exit(1 if bad else 0)
