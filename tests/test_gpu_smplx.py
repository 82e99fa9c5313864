"""GPU parity: saga_amd.smplx (HIP) against the float64 CPU oracle.
Bars (BASELINE.json north_star): vertices and joints within 1e-5 m absolute in FP32; gradients are
compared relative to the largest oracle gradient entry (1e-4)."""
import numpy as np
import pytest
import torch

from tests.helpers import load_oracle_model, make_hip_model, random_body_inputs, rel_err, to_cuda

pytestmark = pytest.mark.gpu

VERT_TOL = 1e-5      # metres, absolute
GRAD_TOL = 2e-4      # relative to max |grad|


def _fwd_bwd(root, B, small, seed, use_joints=True, use_verts=True):
    ref = load_oracle_model(root, B, small=small)
    hip = make_hip_model(root, B, small=small)
    inp = random_body_inputs(B, seed=seed)
    inp_ref = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    inp_hip = to_cuda(inp, requires_grad=True)
    out_ref = ref(**inp_ref)
    out_hip = hip(return_verts=True, **inp_hip)
    g = torch.Generator().manual_seed(seed + 100)
    wv = torch.randn(out_ref['vertices'].shape, generator=g, dtype=torch.float64)
    wj = torch.randn(out_ref['joints'].shape, generator=g, dtype=torch.float64)
    loss_ref = 0
    loss_hip = 0
    if use_verts:
        loss_ref = loss_ref + (out_ref['vertices'] * wv).sum()
        loss_hip = loss_hip + (out_hip.vertices * wv.float().cuda()).sum()
    if use_joints:
        loss_ref = loss_ref + (out_ref['joints'] * wj).sum()
        loss_hip = loss_hip + (out_hip.joints * wj.float().cuda()).sum()
    loss_ref.backward()
    loss_hip.backward()
    return out_ref, out_hip, inp_ref, inp_hip


@pytest.mark.parametrize('B', [1, 3, 130])
def test_forward_small_dense_weights(small_model_dir, B):
    ref = load_oracle_model(small_model_dir, B, small=True)
    hip = make_hip_model(small_model_dir, B, small=True)
    inp = random_body_inputs(B, seed=B)
    out_ref = ref(**inp)
    out_hip = hip(return_verts=True, **to_cuda(inp))
    assert out_hip.vertices.shape == (B, 203, 3) and out_hip.joints.shape == (B, 127, 3)
    assert np.abs(out_hip.vertices.detach().cpu().numpy() - out_ref['vertices'].numpy()).max() < VERT_TOL
    assert np.abs(out_hip.joints.detach().cpu().numpy() - out_ref['joints'].numpy()).max() < VERT_TOL


def test_forward_full_mesh(body_model_dir):
    B = 5
    ref = load_oracle_model(body_model_dir, B)
    hip = make_hip_model(body_model_dir, B)
    inp = random_body_inputs(B, seed=42)
    out_ref = ref(**inp)
    out_hip = hip(return_verts=True, return_full_pose=True, **to_cuda(inp))
    assert out_hip.vertices.shape == (B, 10475, 3) and out_hip.joints.shape == (B, 127, 3)
    dv = np.abs(out_hip.vertices.detach().cpu().numpy() - out_ref['vertices'].numpy()).max()
    dj = np.abs(out_hip.joints.detach().cpu().numpy() - out_ref['joints'].numpy()).max()
    assert dv < VERT_TOL and dj < VERT_TOL, (dv, dj)
    np.testing.assert_allclose(out_hip.full_pose.detach().cpu().numpy(), out_ref['full_pose'].numpy(), atol=1e-6)


def test_defaults_are_module_parameters(small_model_dir):
    """Every argument optional: defaults = zero parameters of shape [batch_size, .] (fittingop.py:149 passes
    no jaw_pose / expression)."""
    B = 2
    ref = load_oracle_model(small_model_dir, B, small=True)
    hip = make_hip_model(small_model_dir, B, small=True)
    out_ref = ref()
    out_hip = hip()
    assert np.abs(out_hip.vertices.detach().cpu().numpy() - out_ref['vertices'].numpy()).max() < VERT_TOL
    inp = random_body_inputs(B, seed=9)
    keys = ['transl', 'global_orient', 'betas', 'body_pose', 'left_hand_pose', 'right_hand_pose', 'leye_pose', 'reye_pose']
    sub = {k: inp[k] for k in keys}
    out_ref = ref(**sub)
    out_hip = hip(return_verts=True, **to_cuda(sub))
    assert np.abs(out_hip.vertices.detach().cpu().numpy() - out_ref['vertices'].numpy()).max() < VERT_TOL


@pytest.mark.parametrize('mode', ['verts', 'joints', 'both'])
def test_backward_small(small_model_dir, mode):
    out_ref, out_hip, inp_ref, inp_hip = _fwd_bwd(small_model_dir, 6, True, 5, use_joints=mode != 'verts',
                                                 use_verts=mode != 'joints')
    for k in inp_ref:
        e = rel_err(inp_hip[k].grad.cpu().numpy(), inp_ref[k].grad.numpy())
        assert e < GRAD_TOL, (k, e)


def test_backward_full_mesh(body_model_dir):
    out_ref, out_hip, inp_ref, inp_hip = _fwd_bwd(body_model_dir, 4, False, 7)
    for k in inp_ref:
        e = rel_err(inp_hip[k].grad.cpu().numpy(), inp_ref[k].grad.numpy())
        assert e < GRAD_TOL, (k, e)


def test_ragged_batch_on_the_large_tile_path_equals_small_batches_bit_for_bit(body_model_dir):
    """B = 520 is four whole 128-row tiles and one of 8 rows, on the 128 x 128 launches of the blend contraction in both
    directions (forward 5 x 246 tiles, backward 5 x 4 x 16); a batch of 5 runs the 64 x 64 launches.  Same products in the
    same order: a candidate's vertices and parameter gradients must not depend on which of the two it went through, nor
    on its position in a partly filled tile."""
    B = 520
    sel = [0, 127, 128, 511, 512, 519]
    inp = random_body_inputs(B, seed=77)
    w = torch.randn(B, 10475, 3, generator=torch.Generator().manual_seed(5))

    def run(idx):
        hip = make_hip_model(body_model_dir, len(idx))
        x = to_cuda({k: v[idx] for k, v in inp.items()}, requires_grad=True)
        out = hip(return_verts=True, **x)
        (out.vertices * w[idx].cuda()).sum().backward()
        return out.vertices.detach(), {k: v.grad for k, v in x.items()}

    v_big, g_big = run(list(range(B)))
    v_small, g_small = run(sel)
    assert torch.equal(v_big[sel], v_small)
    for k in g_small:
        assert torch.equal(g_big[k][sel], g_small[k]), k
    assert torch.isfinite(v_big).all()


def _overlapped_forward_check(body_model_dir, B, streams=True):
    from saga_amd import _lib
    inp = random_body_inputs(B, seed=4321)
    hip = make_hip_model(body_model_dir, B)
    h = hip.handle(torch.device('cuda', 0))
    assert _lib.load().saga_smplx_fwd_is_persistent(h, B) == 1 and _lib.load().saga_smplx_fwd_is_persistent(h, 128) == 0
    x = to_cuda(inp)
    with torch.no_grad():
        v_big = hip(return_verts=True, **x).vertices
        small = make_hip_model(body_model_dir, 128)
        for b0 in range(0, B, 128):
            v = small(return_verts=True, **{k: t[b0:b0 + 128] for k, t in x.items()}).vertices
            assert torch.equal(v_big[b0:b0 + 128], v), b0
        assert torch.equal(hip(return_verts=True, **x).vertices, v_big)
        torch.cuda.synchronize()
        if not streams:
            return
        hip2 = make_hip_model(body_model_dir, B)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for rep in range(3):
            with torch.cuda.stream(s1):
                a = hip(return_verts=True, **x).vertices
            with torch.cuda.stream(s2):
                b = hip2(return_verts=True, **x).vertices
            outs += [a, b]
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o, v_big)


def test_overlapped_tile_forward_equals_one_tile_launches_bit_for_bit(body_model_dir):
    """B = 2,048 runs the blend contraction's forward as the persistent kernel with overlapping tiles (16 x 246 tiles, a
    queue per XCD, stores of one tile during the k-loop of the next); 128 candidates at a time run one 64 x 64 tile per
    block.  Same products in the same order: every vertex of every candidate must agree bit for bit -- on a second launch
    into the same buffers and on two streams at once as well (the queue lives in the call's own output buffer)."""
    _overlapped_forward_check(body_model_dir, 2048)


def _overlapped_backward_check(body_model_dir, B, chunks=None):
    from saga_amd import _lib
    inp = random_body_inputs(B, seed=99)
    w = torch.randn(B, 10475, 3, generator=torch.Generator().manual_seed(3))
    hip = make_hip_model(body_model_dir, B)
    h = hip.handle(torch.device('cuda', 0))
    assert _lib.load().saga_smplx_bwd_is_persistent(h, B) == 1 and _lib.load().saga_smplx_bwd_is_persistent(h, 128) == 0

    def run(model, b0, n):
        x = to_cuda({k: v[b0:b0 + n] for k, v in inp.items()}, requires_grad=True)
        out = model(return_verts=True, **x)
        (out.vertices * w[b0:b0 + n].cuda()).sum().backward()
        return {k: v.grad for k, v in x.items()}

    g_big = run(hip, 0, B)
    g_again = run(hip, 0, B)
    for k in g_big:
        assert torch.equal(g_big[k], g_again[k]), k
        assert g_big[k].abs().max() > 0 and torch.isfinite(g_big[k]).all(), k
    small = make_hip_model(body_model_dir, 128)
    for b0 in (range(0, B, 128) if chunks is None else chunks):
        g = run(small, b0, 128)
        for k in g:
            assert torch.equal(g_big[k][b0:b0 + 128], g[k]), (k, b0)


def test_overlapped_tile_backward_equals_one_tile_launches_bit_for_bit(body_model_dir):
    """B = 3,072 runs the split-K backward of the blend contraction as the persistent kernel with overlapping tiles (24 x 4
    tiles x 16 slabs, the last slab shorter; the forward too); 128 candidates at a time run 64 x 64 tiles, one per block.
    Same products in the same order within every slab: every parameter gradient of every candidate must agree bit for bit,
    and a second backward into the same buffers as well."""
    _overlapped_backward_check(body_model_dir, 3072)


@pytest.mark.parametrize('B', [512, 1024, 2048])
def test_overlapped_tile_forward_with_one_or_two_tiles_per_block(body_model_dir, B):
    """The persistent kernel's short walks -- blocks whose first tile is their last (no stores to interleave, nothing to
    prefetch), blocks with two tiles (the first interleaved stores follow a tile that had none).  In a child process with
    SAGA_GEMM_XT=1 spelled out (the switch is read once per process; 1 is also the default since the kernels were measured
    to win from one tile per block on).  B = 512: 984 tiles for 512 blocks, B = 1,024: 1,968 -- and the backward with exactly one tile per block
    (B = 1,024: 512 tiles) and two (B = 2,048)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys; sys.path.insert(0, %r); import tests.test_gpu_smplx as t; '
            't._overlapped_forward_check(%r, %d, streams=False); %s print("CHILD_OK")'
            % (root, str(body_model_dir), B, 't._overlapped_backward_check(%r, %d);' % (str(body_model_dir), B) if B >= 1024 else ''))
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, SAGA_GEMM_XT='1'), capture_output=True, text=True,
                       timeout=280, cwd=root)
    assert r.returncode == 0 and 'CHILD_OK' in r.stdout, (r.stdout[-400:], r.stderr[-1200:])


@pytest.mark.parametrize('variant', [
    dict(n_verts=2100, n_faces=4100, dense_weights=True),   # dense table over 3 vertex ranges: the MFMA dA reduction
    dict(n_verts=3000, n_faces=5900, weights_nnz=6),        # 6 non-zeros per vertex: generic sparse path, one block per candidate
    dict(n_verts=5003, n_faces=9900, local_weights=True),   # mesh-like table, 5 ranges split 3 + 2 over two blocks per candidate
], ids=['dense', 'nnz6', 'local'])
def test_backward_weight_table_variants(tmp_path, variant):
    """Every skinning-backward kernel (saga_amd/csrc/skin.hip: sparse streaming, dense MFMA) against the oracle."""
    from saga_amd.synthetic import write_body_model
    root = str(tmp_path)
    write_body_model(root, gender='male', seed=11, **variant)
    out_ref, out_hip, inp_ref, inp_hip = _fwd_bwd(root, 5, True, 13)
    dv = np.abs(out_hip.vertices.detach().cpu().numpy() - out_ref['vertices'].detach().numpy()).max()
    assert dv < VERT_TOL, dv
    for k in inp_ref:
        e = rel_err(inp_hip[k].grad.cpu().numpy(), inp_ref[k].grad.numpy())
        assert e < GRAD_TOL, (k, e)


def test_backward_zero_pose_finite(small_model_dir):
    """Reset state of the optimiser: zero hand / eye pose (fittingop.py:134-135) must give finite grads."""
    B = 2
    ref = load_oracle_model(small_model_dir, B, small=True)
    hip = make_hip_model(small_model_dir, B, small=True)
    z = {k: torch.zeros_like(v) for k, v in random_body_inputs(B).items()}
    zr = {k: v.clone().requires_grad_(True) for k, v in z.items()}
    zh = to_cuda(z, requires_grad=True)
    w = torch.randn(B, 203, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    (ref(**zr)['vertices'] * w).sum().backward()
    (hip(**zh).vertices * w.float().cuda()).sum().backward()
    for k in zr:
        gh = zh[k].grad.cpu().numpy()
        assert np.isfinite(gh).all(), k
        assert rel_err(gh, zr[k].grad.numpy()) < GRAD_TOL, k


def test_batch_rodrigues_parity():
    from oracle.smplx_ref import batch_rodrigues as ref_rod
    from saga_amd.smplx.lbs import batch_rodrigues
    g = torch.Generator().manual_seed(3)
    r = torch.randn(257, 3, generator=g, dtype=torch.float64)
    r[0] = 0
    r[1] = torch.tensor([0.0, 0.0, 3.1])
    rr = r.clone().requires_grad_(True)
    rh = r.float().cuda().requires_grad_(True)
    w = torch.randn(257, 3, 3, generator=g, dtype=torch.float64)
    Rr = ref_rod(rr)
    Rh = batch_rodrigues(rh)
    assert np.abs(Rh.detach().cpu().numpy() - Rr.detach().numpy()).max() < 2e-6
    (Rr * w).sum().backward()
    (Rh * w.float().cuda()).sum().backward()
    assert rel_err(rh.grad.cpu().numpy(), rr.grad.numpy()) < 1e-4


def test_cpu_tensor_is_rejected(small_model_dir):
    import saga_amd.smplx as smplx
    from saga_amd.synthetic import small_vertex_ids
    bm = smplx.create(small_model_dir, model_type='smplx', gender='male', ext='npz', num_pca_comps=24, batch_size=1,
                      vertex_ids=small_vertex_ids(203))
    with pytest.raises(RuntimeError):
        bm()        # parameters still on the CPU: no fallback


@pytest.mark.parametrize('small', [True, False])
def test_vertex_grad_ids_matches_dense_backward(small_model_dir, body_model_dir, small):
    """forward(vertex_grad_ids=ids): `vertices_sel` equals `vertices[:, ids]` bit for bit, `vertices` / `joints` carry no
    gradient, and a loss on `vertices_sel` gives the input gradients of the same loss on the dense path (backward on the
    subset model: the skipped terms are exact zeros) -- also against the float64 oracle."""
    root = small_model_dir if small else body_model_dir
    B = 5
    hip = make_hip_model(root, B, small=small)
    ref = load_oracle_model(root, B, small=small)
    V = hip.num_verts
    g = torch.Generator().manual_seed(11)
    ids = torch.sort(torch.randperm(V, generator=g)[:max(7, V // 10)]).values.cuda()
    inp = random_body_inputs(B, seed=21)
    w = torch.randn(B, ids.shape[0], 3, generator=g, dtype=torch.float64)
    a = to_cuda(inp, requires_grad=True)
    b = to_cuda(inp, requires_grad=True)
    out_a = hip(return_verts=True, vertex_grad_ids=ids, **a)
    out_b = hip(return_verts=True, **b)
    assert torch.equal(out_a.vertices, out_b.vertices) and torch.equal(out_a.joints, out_b.joints)
    assert torch.equal(out_a.vertices_sel, out_b.vertices[:, ids])
    assert not out_a.vertices.requires_grad and not out_a.joints.requires_grad and out_a.vertices_sel.requires_grad
    (out_a.vertices_sel * w.float().cuda()).sum().backward()
    (out_b.vertices.index_select(1, ids) * w.float().cuda()).sum().backward()
    r = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    (ref(**r)['vertices'][:, ids.cpu()] * w).sum().backward()
    for k in ('betas', 'global_orient', 'body_pose', 'left_hand_pose', 'right_hand_pose', 'transl', 'expression', 'jaw_pose'):
        ga, gb = a[k].grad.cpu().numpy(), b[k].grad.cpu().numpy()
        assert rel_err(ga, gb) < 2e-5, (k, rel_err(ga, gb))
        assert rel_err(ga, r[k].grad.numpy()) < GRAD_TOL, (k, rel_err(ga, r[k].grad.numpy()))
    # a second call with the same id tensor reuses the cached subset model; unsorted ids are rejected
    n_handles = len(hip._handles)
    hip(return_verts=True, vertex_grad_ids=ids, **to_cuda(inp))
    assert len(hip._handles) == n_handles
    with pytest.raises(ValueError):
        hip(return_verts=True, vertex_grad_ids=torch.flip(ids, [0]), **to_cuda(inp))


def test_smplx_fuzz_against_oracle(small_model_dir, body_model_dir):
    """Random batch sizes and pose / shape magnitudes, from exactly-zero rotations (the Rodrigues series branch) through poses of
    several radians, with zeroed joints mixed in: vertices and joints within 1e-5 m of the fp64 oracle -- relative to the body's own
    extent for the large shapes (|betas| up to 5 stretch the synthetic body to tens of metres) --, gradients within 2e-4 of its
    largest entry; small dense-weight model and the full mesh."""
    g = torch.Generator().manual_seed(77)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    cases = [(small_model_dir, True)] * 14 + [(body_model_dir, False)] * 3
    for ci, (root, small) in enumerate(cases):
        B = [1, 2, 7, 64, 65, 130][ri(0, 5)] if small else [1, 5, 33][ri(0, 2)]
        pose_scale = [0.0, 1e-9, 1e-4, 0.3, 1.0, 3.0, 6.0][ri(0, 6)]
        shape_scale = [0.0, 1.0, 5.0][ri(0, 2)]
        inp = random_body_inputs(B, seed=1000 + ci, scale=1.0)
        for k in inp:
            if k in ('betas', 'expression'):
                inp[k] = inp[k] * shape_scale
            elif k != 'transl':
                inp[k] = inp[k] * pose_scale
                drop = torch.rand(inp[k].shape[0], inp[k].shape[1] // 3, 1, generator=g) < 0.2           # some joints exactly at rest
                inp[k] = (inp[k].view(B, -1, 3) * (~drop)).reshape(B, -1)
        ref, hip = load_oracle_model(root, B, small=small), make_hip_model(root, B, small=small)
        inp_ref = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
        inp_hip = to_cuda(inp, requires_grad=True)
        out_ref, out_hip = ref(**inp_ref), hip(return_verts=True, **inp_hip)
        extent = max(1.0, float(out_ref['vertices'].detach().abs().max()))
        for name, a, b in (('vertices', out_hip.vertices, out_ref['vertices']), ('joints', out_hip.joints, out_ref['joints'])):
            err = float((a.detach().cpu().double() - b.detach()).abs().max())
            assert err < VERT_TOL * extent, (ci, name, B, pose_scale, shape_scale, err, extent)
        wv = torch.randn(out_ref['vertices'].shape, generator=g, dtype=torch.float64)
        wj = torch.randn(out_ref['joints'].shape, generator=g, dtype=torch.float64)
        ((out_ref['vertices'] * wv).sum() + (out_ref['joints'] * wj).sum()).backward()
        ((out_hip.vertices * wv.float().cuda()).sum() + (out_hip.joints * wj.float().cuda()).sum()).backward()
        for k in inp_ref:
            assert torch.isfinite(inp_hip[k].grad).all(), (ci, k)
            e = rel_err(inp_hip[k].grad.cpu().numpy(), inp_ref[k].grad.numpy())
            assert e < GRAD_TOL, (ci, k, B, pose_scale, shape_scale, e)
