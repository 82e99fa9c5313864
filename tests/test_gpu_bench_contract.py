"""The driver's contract with bench.py, on the GPU: the ONE JSON line (metric / value / roofline / cpu_baseline keys) from a small
run started the two ways the driver starts it -- `python bench.py` and `python -m torch.distributed.run --nproc-per-node 1 ...`
(round 3: the second form hung at one rank until bench.py took the launcher's process group)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ['--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '256', '--no-secondary', '--cpu-seconds', '0.5']


def _line(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-800:]
    return json.loads(lines[0])


@pytest.mark.parametrize('launcher', ['python', 'torchrun'])
def test_bench_line_contract(launcher):
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py')] + ARGS
    if launcher == 'torchrun':
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
               '--master-port', '29541', os.path.join(ROOT, 'bench.py')] + ARGS
    d = _line(cmd)
    assert d['metric'] == base['metric'] and d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert d['dtype'] == 'f32' and d['value'] > 0 and abs(d['value'] * d['ms_per_step'] * 1e-3 - 1.0) < 1e-3
    assert 'workload' in d['config'] and 'model' not in d['config']
    rf = d['roofline']
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and 0 < rf['frac'] < 1 and abs(rf['achieved'] / rf['peak'] - rf['frac']) < 1e-3
    assert 'traffic' in rf             # a number only when the committed PMC summary describes this very run (B = 4096): None here
    cb = d['cpu_baseline']
    assert cb['kind'] in ('port', 'reference') and cb['value'] > 0 and cb['cores'] >= 1 and cb['sample']
    assert 'RCCL' in d['config']['end_of_fit_gather']
