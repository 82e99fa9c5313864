"""Multi-GPU layer: candidates shard across ranks, one exchange at the end.

The reference has no distributed code at all (single process, serial loop over candidates,
``opt_grasppose.py:211,220``).  Candidates are independent, so each rank (one process per GPU) fits a
contiguous block of ``B / world`` candidates with NO communication during the 1,200 iterations; a
single ``all_gather`` of per-candidate ``(loss, parameters)`` -- RCCL over xGMI when the backend is
``nccl`` -- then lets every rank pick the best-of-N.  Payload is < 10 MB at B = 16,384, i.e.
latency-bound, so one fused buffer is gathered instead of one collective per tensor.
"""
import torch
import torch.distributed as dist

PARAM_ORDER = ['transl', 'global_orient', 'betas', 'body_pose', 'left_hand_pose', 'right_hand_pose', 'leye_pose', 'reye_pose']


def shard_range(n_total, rank, world):
    """Contiguous block of candidates owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack(losses, params):
    """losses [b], params dict of [b,.] -> one [b, 1+sum(dims)] float32 buffer."""
    cols = [losses.reshape(-1, 1).float()] + [params[k].reshape(losses.shape[0], -1).float() for k in PARAM_ORDER]
    return torch.cat(cols, dim=1).contiguous()


def unpack(buf, dims):
    losses = buf[:, 0]
    out, o = {}, 1
    for k in PARAM_ORDER:
        out[k] = buf[:, o:o + dims[k]]
        o += dims[k]
    return losses, out


def gather_candidates(losses, params, n_total=None, group=None, force_collective=False):
    """All-gather every rank's candidates.  Returns (losses [N], params dict [N,.]) in global candidate order (rank order;
    within a rank, row order).  Shards may be ragged: the per-rank row counts are exchanged first (one tiny all-gather),
    rows are padded to the largest shard for the single ``all_gather_into_tensor`` and the padding is cut out again.
    ``n_total`` (optional) is checked against the gathered counts, and this rank's rows against its ``shard_range``.
    Works without an initialised process group (single process: returns the inputs); ``force_collective`` runs the
    collectives even in a group of one rank (exercises the RCCL path on a 1-GPU box)."""
    dims = {k: int(params[k].reshape(losses.shape[0], -1).shape[1]) for k in PARAM_ORDER}
    buf = pack(losses, params)
    ready = dist.is_available() and dist.is_initialized()
    if not ready or (dist.get_world_size(group) == 1 and not force_collective):
        if n_total is not None and (not ready or dist.get_world_size(group) == 1) and n_total != buf.shape[0]:
            raise ValueError('gather_candidates: %d rows on the only rank, n_total = %d' % (buf.shape[0], n_total))
        return unpack(buf, dims)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = torch.tensor([buf.shape[0]], device=buf.device, dtype=torch.int64)
    counts_t = torch.empty(world, device=buf.device, dtype=torch.int64)
    dist.all_gather_into_tensor(counts_t, mine, group=group)
    counts = [int(c) for c in counts_t.tolist()]
    if n_total is not None:
        if sum(counts) != n_total:
            raise ValueError('gather_candidates: the ranks hold %s rows, n_total = %d' % (counts, n_total))
        s, e = shard_range(n_total, rank, world)
        if e - s != buf.shape[0]:
            raise ValueError('gather_candidates: rank %d holds %d rows, its shard of %d is %d' % (rank, buf.shape[0], n_total, e - s))
    mx = max(counts)
    send = buf
    if buf.shape[0] < mx:
        send = torch.cat([buf, torch.zeros(mx - buf.shape[0], buf.shape[1], device=buf.device, dtype=buf.dtype)], 0)
    recv = torch.empty(world * mx, buf.shape[1], device=buf.device, dtype=buf.dtype)
    dist.all_gather_into_tensor(recv, send, group=group)
    if any(c != mx for c in counts):
        recv = torch.cat([recv[r * mx:r * mx + c] for r, c in enumerate(counts)], 0)
    return unpack(recv, dims)


def best_of_n(losses, params, group_size):
    """Best candidate of every consecutive group of ``group_size`` (e.g. the k random samples of one object).
    -> (best_idx [n_groups] global indices, best_loss [n_groups], best_params dict [n_groups,.])."""
    n = losses.shape[0]
    if n % group_size:
        raise ValueError('%d candidates do not split into groups of %d' % (n, group_size))
    g = losses.view(-1, group_size)
    best_loss, arg = g.min(dim=1)
    idx = arg + torch.arange(g.shape[0], device=losses.device) * group_size
    return idx, best_loss, {k: v[idx] for k, v in params.items()}
