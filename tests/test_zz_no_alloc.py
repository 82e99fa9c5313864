"""saga_smplx_bwd_noalloc (the entry point that allocates nothing: its workspaces are a static device array inside the
library) against saga_smplx_bwd with caller-allocated workspaces, through the C-ABI: bit-identical gradients on two calls in
a row, the second with a smaller input, with and without joint gradients; sizes beyond the static scratch are refused."""
import ctypes

import numpy as np
import pytest
import torch

from tests.helpers import make_hip_model, random_body_inputs, to_cuda

pytestmark = pytest.mark.gpu


def _forward(bm, B, seed):
    """SMPL-X forward through the C-ABI; returns what the backward needs (all device tensors)."""
    from saga_amd import _lib
    dev = torch.device('cuda', 0)
    inp = to_cuda(random_body_inputs(B, seed=seed, scale=0.5, dtype=torch.float32))
    with torch.no_grad():
        out = bm(return_full_pose=True, **inp)
    full_pose = out.full_pose.contiguous()
    shape = torch.cat([inp['betas'], inp['expression']], dim=-1).contiguous()
    h = bm.handle(dev)
    V, npad, njo = bm.num_verts, bm.npad(dev), bm.num_joints_out
    f32 = dict(device=dev, dtype=torch.float32)
    t = {k: torch.empty(*s, **f32) for k, s in dict(vertices=(B, V, 3), joints=(B, njo, 3), feat=(B, _lib.FEAT_K), rot=(B, 55, 9),
                                                  jrest=(B, 55, 3), chain=(B, 55, 12), A=(B, 55, 12), v_posed=(B, npad)).items()}
    P = _lib.ptr
    _lib.call('saga_smplx_fwd', dev, h, P(shape), P(full_pose), P(inp['transl'].contiguous()), P(t['vertices']), P(t['joints']),
              P(t['feat']), P(t['rot']), P(t['jrest']), P(t['chain']), P(t['A']), P(t['v_posed']), B)
    t['full_pose'] = full_pose
    return h, t


def _backward(h, t, gv, gj, B, bm, noalloc):
    from saga_amd import _lib
    dev = torch.device('cuda', 0)
    f32 = dict(device=dev, dtype=torch.float32)
    P = _lib.ptr
    # poisoned outputs: whatever the call leaves unwritten shows up in the comparison
    g_shape, g_pose, g_transl = (torch.full((B, n), float('nan'), **f32) for n in (bm.num_shape, 165, 3))
    common = (h, P(gv), P(gj), P(t['full_pose']), P(t['rot']), P(t['jrest']), P(t['chain']), P(t['A']), P(t['v_posed']),
              P(g_shape), P(g_pose), P(g_transl))
    if noalloc:
        _lib.call('saga_smplx_bwd_noalloc', dev, *common, B)
    else:
        npad = t['v_posed'].shape[1]
        ws = [torch.full(s, float('nan'), **f32) for s in ((B, npad), (_lib.BWD_SPLITK, B, _lib.FEAT_K), (B, _lib.DA_STRIDE),
                                                         (B, bm.num_verts, 3))]
        _lib.call('saga_smplx_bwd', dev, *common, *(P(w) for w in ws), B)
    torch.cuda.synchronize()
    return g_shape, g_pose, g_transl


@pytest.mark.parametrize('with_joints', [False, True])
def test_noalloc_backward_equals_workspace_backward_two_calls_in_a_row(body_model_dir, with_joints):
    from saga_amd import _lib
    bm = make_hip_model(body_model_dir, 64)
    g = torch.Generator().manual_seed(3)
    for B, seed in ((64, 11), (5, 12)):                 # the second call is smaller: the scratch still holds the first one's rows
        h, t = _forward(bm, B, seed)
        gv = torch.randn(B, bm.num_verts, 3, generator=g).cuda()
        gj = torch.randn(B, bm.num_joints_out, 3, generator=g).cuda() if with_joints else None
        want = _backward(h, t, gv, gj, B, bm, noalloc=False)
        got = _backward(h, t, gv, gj, B, bm, noalloc=True)
        for a, b, name in zip(got, want, ('g_shape', 'g_full_pose', 'g_transl')):
            assert torch.isfinite(b).all(), name
            assert torch.equal(a, b), (B, name, float((a - b).abs().max()))
    # joint gradients alone (ws_gv is cleared inside, not by the caller)
    if with_joints:
        want = _backward(h, t, None, gj, B, bm, noalloc=False)
        got = _backward(h, t, None, gj, B, bm, noalloc=True)
        assert all(torch.equal(a, b) for a, b in zip(got, want))


def test_noalloc_rejects_sizes_beyond_its_scratch(body_model_dir):
    from saga_amd import _lib
    lib = _lib.load()
    bm = make_hip_model(body_model_dir, 2)
    h, t = _forward(bm, 2, 1)
    P = _lib.ptr
    z = torch.zeros(2, 165, device='cuda')
    args = (h, P(t['vertices']), None, P(t['full_pose']), P(t['rot']), P(t['jrest']), P(t['chain']), P(t['A']), P(t['v_posed']),
            P(z), P(z), P(z))
    with torch.cuda.device(0):
        rc = lib.saga_smplx_bwd_noalloc(*args, _lib.NOALLOC_MAX_B + 1, None)
    assert rc == -1 and b'static scratch' in lib.saga_last_error()
    with torch.cuda.device(0):
        assert lib.saga_smplx_bwd_noalloc(None, *args[1:], 2, None) == -1
        assert lib.saga_smplx_bwd_noalloc(*args, 0, None) == 0           # empty batch: nothing to do


def test_noalloc_backward_under_graph_capture(body_model_dir):
    """The property the entry point exists for: it can be CAPTURED.  A HIP-graph capture (torch.cuda.graph, global
    capture mode) fails on any hipMalloc / hipFree / synchronising call inside it; saga_smplx_bwd_noalloc is captured once, then
    replayed twice on different upstream gradients, and each replay equals saga_smplx_bwd with caller workspaces bit for bit."""
    from saga_amd import _lib
    dev = torch.device('cuda', 0)
    B = 16
    bm = make_hip_model(body_model_dir, B)
    h, t = _forward(bm, B, 21)
    g = torch.Generator().manual_seed(8)
    f32 = dict(device=dev, dtype=torch.float32)
    gv = torch.zeros(B, bm.num_verts, 3, **f32)                   # static input / output buffers of the graph
    gj = torch.zeros(B, bm.num_joints_out, 3, **f32)
    outs = [torch.zeros(B, n, **f32) for n in (bm.num_shape, 165, 3)]
    P = _lib.ptr

    def call():
        _lib.call('saga_smplx_bwd_noalloc', dev, h, P(gv), P(gj), P(t['full_pose']), P(t['rot']), P(t['jrest']), P(t['chain']), P(t['A']),
                  P(t['v_posed']), P(outs[0]), P(outs[1]), P(outs[2]), B)

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        call()                                                    # warm-up off the capture (lazy per-device set-up)
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
    for seed in (1, 2):
        gv.copy_(torch.randn(gv.shape, generator=g).cuda())
        gj.copy_(torch.randn(gj.shape, generator=g).cuda())
        for o in outs:
            o.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        want = _backward(h, t, gv, gj, B, bm, noalloc=False)
        for a, b, name in zip(outs, want, ('g_shape', 'g_full_pose', 'g_transl')):
            assert torch.isfinite(b).all() and torch.equal(a, b), (seed, name)
