"""One process per GPU on ONE node: the launcher behind ``bench.py --gpus N``.

The reference has no multi-process mode (serial candidate loop, ``opt_grasppose.py:211-222``); here the candidate axis is
sharded over ranks and every rank is its own process with its own HIP context.  ``launch_ranks`` starts the ranks as fresh
children (``subprocess``: nothing of the parent's state is inherited; the parent itself makes no HIP call -- it counts the
GPUs through a short-lived child, ``visible_gpus``) with the
``torch.distributed`` environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) -- the same
variables ``python -m torch.distributed.run`` sets, so a script runs unchanged under either launcher.
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    """Environment of rank ``rank`` of ``world`` on this node (rendezvous on 127.0.0.1: the hostname may not resolve)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL needs it on this driver
    return env


def rank_argv(script_argv, executable=None):
    """argv of a rank: the same script with the same arguments under the same interpreter."""
    return [executable or sys.executable, '-u'] + list(script_argv)


def visible_gpus():
    """Number of visible GPUs, counted by a short-lived child process: on ROCm builds of torch ``torch.cuda.device_count()`` ends in
    ``hipGetDeviceCount``, which brings up the HIP / HSA runtime in the process that calls it -- the launcher would then sit beside
    its N ranks with an initialised runtime for the whole job.  The child exits before the ranks start."""
    out = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'], capture_output=True, text=True,
                         timeout=600)
    if out.returncode != 0:
        raise RuntimeError('could not count the visible GPUs: %s' % out.stderr.strip()[-500:])
    return int(out.stdout.strip().splitlines()[-1])


def launch_ranks(n, script_argv, check_devices=True, timeout=None, env=None):
    """Start ``n`` ranks of ``script_argv`` (= ``sys.argv`` of the calling script), wait for all of them and return the
    largest exit code.  Rank 0 inherits stdout (it prints the result line); every rank inherits stderr.  When one rank
    fails the others are terminated, so a rendezvous that can no longer complete does not hang the job."""
    if n < 1:
        raise ValueError('launch_ranks: n = %d' % n)
    if check_devices:
        have = visible_gpus()
        if have < n:
            raise RuntimeError('--gpus %d asked for, but only %d GPU(s) are visible on this node' % (n, have))
    port = free_port()
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(rank_argv(script_argv), env=rank_env(r, n, port, env),
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    stopped = set()          # ranks this launcher terminated after another one failed: their exit codes say nothing
    try:
        pending = list(procs)
        import time
        t0 = time.time()
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and p not in stopped:
                    rc = max(rc, code if code > 0 else 128 - code)
                    for q in pending:          # exact PIDs we started, nothing by pattern
                        stopped.add(q)
                        q.terminate()
            if timeout is not None and time.time() - t0 > timeout:
                rc = max(rc, 124)
                for q in pending:
                    stopped.add(q)
                    q.terminate()
                timeout = None
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return rc
