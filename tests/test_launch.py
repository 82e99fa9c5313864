"""CPU tests of the one-process-per-GPU launcher behind `bench.py --gpus N` (saga_amd/launch.py) -- row (e)."""
import os
import subprocess
import sys
import textwrap

from saga_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rank_environment_and_argv():
    base = {'PATH': '/bin', 'KEEP': 'x'}
    envs = [launch.rank_env(r, 4, 29511, base) for r in range(4)]
    for r, e in enumerate(envs):
        assert e['RANK'] == e['LOCAL_RANK'] == str(r) and e['WORLD_SIZE'] == e['LOCAL_WORLD_SIZE'] == '4'
        assert e['MASTER_ADDR'] == '127.0.0.1' and e['MASTER_PORT'] == '29511' and e['KEEP'] == 'x'
        assert e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert 'RANK' not in base                                   # the caller's environment is not modified
    assert launch.rank_env(0, 2, 1, {'HSA_ENABLE_IPC_MODE_LEGACY': '1'})['HSA_ENABLE_IPC_MODE_LEGACY'] == '1'
    argv = launch.rank_argv(['bench.py', '--gpus', '2', '--steps', '3'], executable='/usr/bin/python3')
    assert argv == ['/usr/bin/python3', '-u', 'bench.py', '--gpus', '2', '--steps', '3']
    p1, p2 = launch.free_port(), launch.free_port()
    assert 1024 < p1 < 65536 and 1024 < p2 < 65536


def test_two_ranks_rendezvous_over_gloo(tmp_path):
    script = tmp_path / 'rank.py'
    script.write_text(textwrap.dedent('''
        import os, sys, torch, torch.distributed as dist
        rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        dist.init_process_group('gloo', rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        open(sys.argv[1] + '.%d' % rank, 'w').write('%d %d %g' % (rank, world, float(t)))
        if rank == 0:
            print('RESULT', float(t))
        dist.destroy_process_group()
    '''))
    out = str(tmp_path / 'out')
    code = subprocess.run([sys.executable, '-c', 'import sys; sys.path.insert(0, %r); from saga_amd.launch import launch_ranks; '
                           'sys.exit(launch_ranks(2, [%r, %r], check_devices=False, timeout=120))' % (ROOT, str(script), out)],
                          capture_output=True, text=True, timeout=180)
    assert code.returncode == 0, code.stderr[-2000:]
    assert 'RESULT 3.0' in code.stdout                          # rank 0's stdout is the job's stdout; rank 1's is dropped
    assert open(out + '.0').read() == '0 2 3' and open(out + '.1').read() == '1 2 3'


def test_failing_rank_fails_the_job_and_stops_the_others(tmp_path):
    script = tmp_path / 'rank.py'
    script.write_text('import os, sys, time\nif os.environ["RANK"] == "1":\n    sys.exit(7)\ntime.sleep(60)\n')
    import time
    t0 = time.time()
    rc = launch.launch_ranks(2, [str(script)], check_devices=False, timeout=100)
    assert rc == 7 and time.time() - t0 < 30


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` with fewer than N devices: non-zero exit and a clear message, never an n_gpus: 1 line."""
    import torch
    n = torch.cuda.device_count() + 1
    if n < 2:
        n = 2
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and 'visible' in r.stderr and '"n_gpus"' not in r.stdout
    # under a launcher whose world size disagrees with --gpus: refused as well
    env.update(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr
