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


def gather_candidates(losses, params, n_total=None, group=None):
    """All-gather every rank's candidates.  Returns (losses [N], params dict [N,.]) in global candidate order.
    Works without an initialised process group (single process: returns the inputs)."""
    dims = {k: int(params[k].reshape(losses.shape[0], -1).shape[1]) for k in PARAM_ORDER}
    buf = pack(losses, params)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return unpack(buf, dims)
    world = dist.get_world_size(group)
    counts = [shard_range(n_total, r, world) for r in range(world)] if n_total is not None else None
    if counts is not None and len({e - s for s, e in counts}) > 1:
        # ragged shards: pad to the largest so a single all_gather_into_tensor still applies
        mx = max(e - s for s, e in counts)
        pad = torch.zeros(mx - buf.shape[0], buf.shape[1], device=buf.device, dtype=buf.dtype)
        send = torch.cat([buf, pad], 0)
    else:
        mx = buf.shape[0]
        send = buf
    recv = torch.empty(world * mx, buf.shape[1], device=buf.device, dtype=buf.dtype)
    dist.all_gather_into_tensor(recv, send, group=group)
    if counts is not None and mx * world != (n_total or 0):
        recv = torch.cat([recv[r * mx:r * mx + (e - s)] for r, (s, e) in enumerate(counts)], 0)
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
