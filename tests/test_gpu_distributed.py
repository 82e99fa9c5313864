"""Row (e) on the GPU box: the end-of-fit exchange on DEVICE tensors through RCCL (torch.distributed backend 'nccl').
One GPU is visible there, so the group has one rank; `force_collective=True` makes gather_candidates run its two
`all_gather_into_tensor` calls anyway, so the RCCL path executes (the world-2 logic is covered on CPU over gloo)."""
import os

import pytest
import torch
import torch.distributed as dist

from saga_amd import distributed as sd
from saga_amd.launch import free_port

pytestmark = pytest.mark.gpu


def test_gather_candidates_over_rccl_single_rank():
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        g = torch.Generator().manual_seed(0)
        n = 64
        losses = torch.rand(n, generator=g).cuda()
        dims = {'transl': 3, 'global_orient': 3, 'betas': 10, 'body_pose': 63, 'left_hand_pose': 24,
                'right_hand_pose': 24, 'leye_pose': 3, 'reye_pose': 3}
        params = {k: torch.randn(n, d, generator=g).cuda() for k, d in dims.items()}
        gl, gp = sd.gather_candidates(losses, params, n_total=n, force_collective=True)
        torch.cuda.synchronize()
        assert gl.is_cuda and torch.equal(gl, losses) and all(torch.equal(gp[k], params[k]) for k in params)
        idx, bl, bp = sd.best_of_n(gl, gp, group_size=8)
        assert torch.equal(bl, losses.view(-1, 8).min(dim=1).values) and bp['body_pose'].shape == (8, 63)
        with pytest.raises(ValueError):
            sd.gather_candidates(losses, params, n_total=n + 1, force_collective=True)
    finally:
        dist.destroy_process_group()
