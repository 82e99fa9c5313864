"""CPU, world_size 2, gloo: candidate sharding and the single end-of-fit all-gather + best-of-N selection
(SURVEY.md section 8e).  The data path itself has no collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from saga_amd import distributed as sd


def test_shard_range_covers_everything():
    for n, w in [(16384, 8), (10, 3), (7, 8), (4096, 1)]:
        spans = [sd.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1


def _fake(n_total):
    g = torch.Generator().manual_seed(0)
    losses = torch.rand(n_total, generator=g)
    dims = {'transl': 3, 'global_orient': 3, 'betas': 10, 'body_pose': 63, 'left_hand_pose': 24,
            'right_hand_pose': 24, 'leye_pose': 3, 'reye_pose': 3}
    params = {k: torch.randn(n_total, d, generator=g) for k, d in dims.items()}
    return losses, params


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        losses, params = _fake(n_total)
        s, e = sd.shard_range(n_total, rank, world)
        gl, gp = sd.gather_candidates(losses[s:e], {k: v[s:e] for k, v in params.items()}, n_total=n_total)
        ok = torch.equal(gl, losses) and all(torch.equal(gp[k], params[k]) for k in params)
        # without n_total the per-rank counts are exchanged: ragged shards (here 1 row against the rest) still line up
        s2, e2 = (0, 1) if rank == 0 else (1, n_total)
        gl2, gp2 = sd.gather_candidates(losses[s2:e2], {k: v[s2:e2] for k, v in params.items()})
        ok = ok and torch.equal(gl2, losses) and all(torch.equal(gp2[k], params[k]) for k in params)
        # a shard that is not this rank's block of n_total is refused on every rank (no hang: the counts were gathered first)
        try:
            sd.gather_candidates(losses[s2:e2], {k: v[s2:e2] for k, v in params.items()}, n_total=n_total + (0 if n_total > 2 else 1))
            ok = ok and (e2 - s2) == (e - s)
        except ValueError:
            pass
        idx, bl, bp = sd.best_of_n(gl, gp, group_size=n_total // 2) if n_total % 2 == 0 else (None, None, None)
        q.put((rank, bool(ok), None if idx is None else idx.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_total', [12, 7])
def test_gather_and_best_of_n_world2(n_total):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    if n_total % 2 == 0:
        losses, _ = _fake(n_total)
        expect = [int(losses[:n_total // 2].argmin()), int(losses[n_total // 2:].argmin()) + n_total // 2]
        assert all(idx == expect for _, _, idx in res)


def test_single_process_passthrough():
    losses, params = _fake(6)
    gl, gp = sd.gather_candidates(losses, params)
    assert torch.equal(gl, losses) and torch.equal(gp['betas'], params['betas'])
    idx, bl, bp = sd.best_of_n(gl, gp, 3)
    assert idx.tolist() == [int(losses[:3].argmin()), int(losses[3:].argmin()) + 3]
