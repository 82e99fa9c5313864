"""CPU tests: the oracle against the golden vectors generated from the reference's own modules
(tests/golden/make_golden.py) and against the self-consistency invariants of SURVEY.md Appendix C."""
import numpy as np
import pytest
import torch

from oracle import chamfer_ref, geometry_ref, rotations_ref, smplx_ref, vposer_ref
from saga_amd.synthetic import make_body_model, small_vertex_ids


def test_rotations_match_reference_golden(golden):
    g = golden('rotations.npz')
    x = torch.from_numpy(g['x6d']).requires_grad_(True)
    R = rotations_ref.cont2rotmat(x)
    aa = rotations_ref.rotmat2aa(R)
    assert np.array_equal(R.detach().numpy(), g['R'])
    np.testing.assert_allclose(aa.detach().numpy(), g['aa'], rtol=0, atol=1e-6)
    (aa * torch.from_numpy(g['w'])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g['grad_x6d'], rtol=1e-4, atol=1e-5)
    # fittingop.py:73 initial orientation -> [0, 2.2214, 2.2214]
    np.testing.assert_allclose(g['aa'][0], [0.0, 2.2214415, 2.2214415], atol=1e-5)
    aa64 = rotations_ref.rotmat2aa(rotations_ref.cont2rotmat(torch.from_numpy(g['x6d']).double()))
    np.testing.assert_allclose(aa64.numpy(), g['aa64'], atol=1e-12)


@pytest.mark.parametrize('name', ['markers', 'rhand'])
def test_point2point_signed_matches_reference_golden(golden, name):
    g = golden('p2p_signed.npz')
    x = torch.from_numpy(g[f'{name}_x']).requires_grad_(True)
    y = torch.from_numpy(g[f'{name}_y'])
    y2x, x2y, yidx, xidx = geometry_ref.point2point_signed(x, y, torch.from_numpy(g[f'{name}_xn']), torch.from_numpy(g[f'{name}_yn']))
    assert np.array_equal(yidx.numpy(), g[f'{name}_yidx'])
    assert np.array_equal(xidx.numpy(), g[f'{name}_xidx'])
    np.testing.assert_allclose(y2x.detach().numpy(), g[f'{name}_y2x'], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(x2y.detach().numpy(), g[f'{name}_x2y'], rtol=1e-6, atol=1e-8)
    ((y2x * torch.from_numpy(g[f'{name}_wy'])).sum() + (x2y * torch.from_numpy(g[f'{name}_wx'])).sum()).backward()
    np.testing.assert_allclose(x.grad.numpy(), g[f'{name}_grad_x'], rtol=1e-5, atol=1e-6)


def test_point2point_signed_marker_quirk(golden):
    """fittingop.py:183: normal_object passed as x_normals, y_normals None -> x2y unsigned."""
    g = golden('p2p_signed.npz')
    x = torch.from_numpy(g['markers_x'])
    y = torch.from_numpy(g['markers_y'])
    y2x, x2y, _, _ = geometry_ref.point2point_signed(x, y, torch.from_numpy(g['markers_quirk_normals']))
    np.testing.assert_allclose(y2x.numpy(), g['markers_quirk_y2x'], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(x2y.numpy(), g['markers_quirk_x2y'], rtol=1e-6, atol=1e-8)
    assert (x2y >= 0).all()


def test_point2point_signed_shape_error():
    with pytest.raises(ValueError):
        geometry_ref.point2point_signed(torch.zeros(2, 5, 3), torch.zeros(3, 7, 3))


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_chamfer_c_scan_equals_literal_definition(dtype):
    """oracle/csrc/nn_ref.c (fmaf chain) against the exact-arithmetic literal scan, bit for bit."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 37, 3, generator=g).to(dtype)
    y = torch.randn(2, 53, 3, generator=g).to(dtype)
    y[0, 10] = y[0, 3]                      # exact tie: lowest index must win
    x[1, 5] = x[1, 2]
    d1, d2, i1, i2 = chamfer_ref.chamfer_distance(x, y)
    l1, l2, li1, li2 = chamfer_ref.chamfer_distance_loop(x.numpy(), y.numpy())
    assert np.array_equal(i1.numpy(), li1) and np.array_equal(i2.numpy(), li2)
    assert np.array_equal(d1.numpy(), l1) and np.array_equal(d2.numpy(), l2)
    assert not (i1[0] == 10).any()
    assert not (i2[1] == 5).any()
    assert d1.dtype == dtype and i1.dtype == torch.int32


def test_chamfer_fma_chain_differs_from_unfused_sum_only_in_last_ulp():
    """The contraction is part of the definition: it moves squared distances by at most 1 ulp."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 64, 3, generator=g)
    y = torch.randn(1, 512, 3, generator=g)
    d1, _, i1, _ = chamfer_ref.chamfer_distance(x, y)
    d = x[:, :, None, :] - y[:, None, :, :]
    un = ((d * d)[..., 0] + (d * d)[..., 1]) + (d * d)[..., 2]
    m = un.min(dim=2).values
    assert (d1 - m).abs().max() <= 2 * torch.finfo(torch.float32).eps * m.max()


def _small_model(dtype=torch.float64, **kw):
    data = make_body_model(seed=3, n_verts=200, n_faces=380, **kw)
    return smplx_ref.SMPLXRef(data, dtype=dtype, batch_size=2, vertex_ids=small_vertex_ids(200)), data


def test_smplx_zero_pose_is_shaped_template():
    """Appendix C-2: zero pose (flat hands) => vertices = v_shaped + transl."""
    data = make_body_model(seed=3, n_verts=200, n_faces=380)
    bm = smplx_ref.SMPLXRef(data, dtype=torch.float64, batch_size=2, vertex_ids=small_vertex_ids(200), flat_hand_mean=True)
    betas = torch.randn(2, 10, dtype=torch.float64)
    transl = torch.randn(2, 3, dtype=torch.float64)
    out = bm(betas=betas, transl=transl, return_aux=True)
    np.testing.assert_allclose(out['vertices'].numpy(), (out['v_shaped'] + transl[:, None]).numpy(), atol=1e-7)
    A = out['aux']['A']
    np.testing.assert_allclose(A.numpy(), np.broadcast_to(np.eye(4), A.shape), atol=1e-7)


def test_smplx_global_rotation_is_rigid():
    """Appendix C-3: only global_orient != 0 => rigid motion about the pelvis joint."""
    data = make_body_model(seed=3, n_verts=200, n_faces=380)
    bm = smplx_ref.SMPLXRef(data, dtype=torch.float64, batch_size=2, vertex_ids=small_vertex_ids(200), flat_hand_mean=True)
    go = torch.tensor([[0.3, -0.2, 0.9], [1.0, 2.0, -0.5]], dtype=torch.float64)
    out = bm(global_orient=go, return_aux=True)
    R0 = smplx_ref.batch_rodrigues(go)
    J0 = out['aux']['J'][:, 0]
    vs = out['v_shaped']
    expect = torch.einsum('bij,bvj->bvi', R0, vs - J0[:, None]) + J0[:, None]
    np.testing.assert_allclose(out['vertices'].numpy(), expect.numpy(), atol=1e-9)
    assert out['joints'].shape == (2, 127, 3)


def test_smplx_parents_topological_depth():
    from saga_amd.synthetic import SMPLX_PARENTS
    depth = np.zeros(55, int)
    for j in range(1, 55):
        assert SMPLX_PARENTS[j] < j
        depth[j] = depth[SMPLX_PARENTS[j]] + 1
    assert depth.max() == 10


def test_rodrigues_at_zero_is_identity_with_finite_grad():
    """Appendix C-7 (the reset state of hand / eye pose, fittingop.py:134-135)."""
    r = torch.zeros(3, 3, dtype=torch.float64, requires_grad=True)
    R = smplx_ref.batch_rodrigues(r)
    np.testing.assert_allclose(R.detach().numpy(), np.broadcast_to(np.eye(3), (3, 3, 3)), atol=1e-15)
    R[:, 2, 1].sum().backward()       # d R[2,1] / d r_x = 1 at the origin
    assert torch.isfinite(r.grad).all()
    np.testing.assert_allclose(r.grad.numpy()[:, 0], 1.0, atol=1e-6)


def test_vertex_normals_convention():
    # one triangle in the z=0 plane, counter-clockwise -> +z; a vertex without faces stays zero
    v = torch.tensor([[[0., 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5]]], dtype=torch.float64)
    f = torch.tensor([[0, 1, 2]])
    n = geometry_ref.vertex_normals(v, f)
    np.testing.assert_allclose(n[0, :3].numpy(), [[0, 0, 1]] * 3, atol=1e-15)
    np.testing.assert_allclose(n[0, 3].numpy(), [0, 0, 0])
    # area weighting: a big and a small face sharing vertex 0
    v = torch.tensor([[[0., 0, 0], [2, 0, 0], [0, 2, 0], [0, 0, 1], [0, 1, 0]]], dtype=torch.float64)
    f = torch.tensor([[0, 1, 2], [0, 3, 4]])
    n = geometry_ref.vertex_normals(v, f)
    raw = np.array([0, 0, 4.0]) + np.cross([0, 0, 1.0], [0, 1.0, 0])
    np.testing.assert_allclose(n[0, 0].numpy(), raw / np.linalg.norm(raw), atol=1e-15)


def test_vposer_decoder_shapes_and_unit_rotations():
    g = torch.Generator().manual_seed(1)
    w = {'w1': torch.randn(512, 32, generator=g) * 0.1, 'b1': torch.zeros(512), 'w2': torch.randn(512, 512, generator=g) * 0.05,
         'b2': torch.zeros(512), 'w3': torch.randn(126, 512, generator=g) * 0.05, 'b3': torch.zeros(126)}
    aa = vposer_ref.vposer_decode(w, torch.randn(4, 32, generator=g))
    assert aa.shape == (4, 1, 21, 3)
    assert torch.isfinite(aa).all() and (aa.norm(dim=-1) <= np.pi + 1e-4).all()


def test_fitting_results_schema_matches_reference_driver(tmp_path):
    """saga_amd.opt_grasppose.fitting_results_dict writes the keys / nesting / shapes of opt_grasppose.py:194-243
    (what visualization/vis_pose.py:31-42 reads back)."""
    from saga_amd.opt_grasppose import BODY_KEYS, fitting_results_dict
    n, O = 5, 64
    rng = np.random.default_rng(0)
    dims = {'betas': 10, 'transl': 3, 'global_orient': 3, 'body_pose': 63, 'leye_pose': 3, 'reye_pose': 3,
            'left_hand_pose': 24, 'right_hand_pose': 24}
    params = {k: torch.from_numpy(rng.normal(size=(n, d)).astype(np.float32)) for k, d in dims.items()}
    out = fitting_results_dict(rng.normal(size=(n, 143, 3)), rng.normal(size=(n, 143, 3)), params, 'male',
                               rng.random((n, O)), rng.random((n, 143)), 'camera', rng.normal(size=(n, O, 3)),
                               rng.normal(size=(n, 3)), rng.normal(size=(n, 3)))
    np.savez(tmp_path / 'fitting_results.npz', **out)
    back = np.load(tmp_path / 'fitting_results.npz', allow_pickle=True)      # our own file
    assert set(back.files) == {'markers', 'markers_fit', 'body', 'object', 'contact', 'gender', 'object_name'}
    body = back['body'][()]
    assert list(body.keys()) == BODY_KEYS and all(body[k].shape == (n, dims[k]) for k in BODY_KEYS)
    assert back['markers'].shape == (n, 143, 3) and back['markers_fit'].shape == (n, 143, 3)
    assert back['object'][()]['verts_object'].shape == (n, O, 3)
    assert back['contact'][()]['body'].shape == (n, 143) and back['contact'][()]['object'].shape == (n, O)
    assert list(back['gender']) == ['male'] * n and list(back['object_name']) == ['camera'] * n


@pytest.mark.parametrize('kw', [dict(), dict(local_weights=True), dict(weights_nnz=6), dict(dense_weights=True)])
def test_synthetic_body_model_weight_tables(kw):
    """The synthetic SMPL-X arrays and their weight-table variants: rows of the skinning weights sum to 1 with the
    advertised number of non-zeros; the mesh-like variant gives runs of consecutive vertices the same joints."""
    from saga_amd.synthetic import make_body_model
    d = make_body_model(seed=5, n_verts=550, n_faces=1090, **kw)
    w = d['weights'].astype(np.float64)
    assert w.shape == (550, 55) and (w >= 0).all()
    np.testing.assert_allclose(w.sum(1), 1.0, atol=1e-6)
    nnz = (w != 0).sum(1)
    if kw.get('dense_weights'):
        assert nnz.min() > 40
    else:
        assert (nnz == kw.get('weights_nnz', 4)).all()
    if kw.get('local_weights'):
        first = np.argmax(w != 0, axis=1)
        # 550 vertices over 55 joints: runs of 10 vertices share their joints (the wrap-around rows aside)
        assert (np.diff(first[:500]) >= 0).all() and len(np.unique(first)) >= 50
    assert d['f'].max() < 550 and np.unique(d['f']).size == 550          # every vertex is on a face
    np.testing.assert_allclose(d['J_regressor'].sum(1), 1.0, atol=1e-5)
