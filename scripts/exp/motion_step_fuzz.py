"""Runs on the GPU box: more cases of tests/test_gpu_motion.py::test_motion_step_size_fuzz_against_autograd_composition.
usage: motion_step_fuzz.py [n_cases] [seed]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from saga_amd.synthetic import write_body_model  # noqa: E402
import tests.test_gpu_motion as t  # noqa: E402

n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 40), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.TemporaryDirectory()
write_body_model(tmp.name, gender='male', seed=7)
rng = np.random.default_rng(seed)
bad = 0
for case in range(n):
    S, Tn = int(rng.integers(1, 9)), int(rng.integers(3, 40))
    start_t = int(rng.integers(1, Tn - 1))
    reduced = bool(rng.integers(0, 2))
    try:
        t._motion_step_case(tmp.name, S, Tn, start_t, reduced, seed=seed * 1000 + case, strict=False)
    except AssertionError as e:
        bad += 1
        print('MISMATCH case', case, 'args', S, Tn, start_t, reduced, str(e)[:300].replace('\n', ' '), flush=True)
    if case % 10 == 9:
        print('case', case + 1, 'mismatches', bad, flush=True)
print('MOTION_STEP_FUZZ_DONE cases', n, 'mismatches', bad)
sys.exit(1 if bad else 0)
