"""GPU parity: chamfer NN / point2point_signed / vertex normals against the oracle and the reference goldens.
Bar: nearest-neighbour indices and squared distances bit-exact (integer / index work)."""
import numpy as np
import pytest
import torch

from oracle import chamfer_ref, geometry_ref
from tests.helpers import assert_same

pytestmark = pytest.mark.gpu


def _check(x, y):
    import saga_amd.chamfer_distance as chd
    d1, d2, i1, i2 = chd.ChamferDistance(x.cuda(), y.cuda())
    r1, r2, j1, j2 = chamfer_ref.chamfer_distance(x, y)
    assert i1.dtype == torch.int32 and i2.dtype == torch.int32
    assert_same(i1.cpu().numpy(), j1.numpy())
    assert_same(i2.cpu().numpy(), j2.numpy())
    assert_same(d1.cpu().numpy(), r1.numpy())
    assert_same(d2.cpu().numpy(), r2.numpy())


@pytest.mark.parametrize('N,P1,P2', [(1, 1, 1), (2, 143, 2048), (2, 778, 2048), (3, 37, 53), (1, 1500, 1025), (2, 300, 4097)])
def test_chamfer_exact(N, P1, P2):
    g = torch.Generator().manual_seed(P1 * 7 + P2)
    _check(torch.randn(N, P1, 3, generator=g) * 0.1, torch.randn(N, P2, 3, generator=g) * 0.1)


def test_chamfer_ties_lowest_index():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 3, generator=g)
    y = torch.randn(2, 1200, 3, generator=g)
    y[:, 1100] = y[:, 7]        # duplicates across an LDS tile boundary
    y[:, 300] = y[:, 7]
    x[:, 40] = x[:, 11]
    _check(x, y)
    # on a lattice nearly everything ties
    lat = torch.stack(torch.meshgrid(torch.arange(8.), torch.arange(8.), torch.arange(4.), indexing='ij'), -1).reshape(1, -1, 3)
    _check(lat + 0.5, lat)


def _sphere(n, g):
    v = torch.randn(n, 3, generator=g)
    return v / v.norm(dim=1, keepdim=True)


@pytest.mark.parametrize('case', ['offset_far', 'straddle_zero', 'negative_box', 'equidistant', 'grasp_like', 'huge_coords',
                                  'coincident'])
def test_chamfer_two_level_scan_adversarial(case):
    """Single-tile databases take the approximate-then-exact scan (nn_fast.h); the result must stay that of the
    sequential exact scan wherever the approximation is weakest: large offsets (cancellation), boxes that straddle
    zero (no centring), every point equidistant (every chunk inside the margin -> slow path), coincident points."""
    g = torch.Generator().manual_seed(sum(map(ord, case)))
    N = 3
    if case == 'offset_far':
        c = torch.tensor([53.0, -31.0, 2.5])
        y = torch.randn(N, 2048, 3, generator=g) * 0.03 + c
        x = torch.cat([torch.randn(N, 500, 3, generator=g) * 0.05 + c, torch.randn(N, 300, 3, generator=g) * 2.0 + c], 1)
    elif case == 'straddle_zero':
        y = torch.randn(N, 2000, 3, generator=g) * 0.05
        x = torch.randn(N, 900, 3, generator=g) * 0.07
    elif case == 'negative_box':
        c = torch.tensor([-0.8, -1.3, -0.4])
        y = (torch.rand(N, 1999, 3, generator=g) - 0.5) * 0.1 + c
        x = (torch.rand(N, 921, 3, generator=g) - 0.5) * 0.3 + c
    elif case == 'equidistant':
        c = torch.tensor([0.7, 0.2, 1.1])
        y = torch.stack([_sphere(2048, g) for _ in range(N)]) * 0.05 + c
        x = torch.randn(N, 256, 3, generator=g) * 1e-4 + c          # near the centre: all 2048 points nearly tie
    elif case == 'grasp_like':
        N = 24
        c = torch.tensor([0.3, -0.4, 1.0])
        y = torch.stack([_sphere(2048, g) for _ in range(N)]) * torch.tensor([0.06, 0.04, 0.03]) + c
        near = torch.stack([_sphere(778, g) for _ in range(N)]) * torch.tensor([0.065, 0.045, 0.034]) + c
        body = torch.randn(N, 143, 3, generator=g) * torch.tensor([0.3, 0.2, 0.8]) + torch.tensor([0.0, 0.0, 0.9])
        x = torch.cat([near, body], 1)
    elif case == 'huge_coords':
        y = torch.randn(N, 1024, 3, generator=g) * 10.0 + 1000.0
        x = torch.randn(N, 300, 3, generator=g) * 10.0 + 1000.0
    else:
        y = torch.randn(N, 640, 3, generator=g) * 0.1 + 0.5
        x = y[:, ::3].clone()                                        # queries ON database points (distance 0)
        y[:, 100:140] = y[:, 5:6]                                    # and a block of duplicates
    _check(x.float().contiguous(), y.float().contiguous())


@pytest.mark.parametrize('P2', [1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 2047, 2048, 2049, 2064, 2065, 4096])
def test_chamfer_sizes_around_chunk_and_tile_edges(P2):
    """The matrix-core pass walks the database 32 points (two 16-point chunks) at a time, LDS tiles hold 2,048: sizes on
    both sides of every edge, in both roles (the reverse direction scans the 70-point cloud)."""
    g = torch.Generator().manual_seed(P2)
    _check(torch.randn(2, 70, 3, generator=g) * 0.1, torch.randn(2, P2, 3, generator=g) * 0.1 + 0.02)


@pytest.mark.parametrize('case', ['tiny_cloud', 'far_queries', 'flat_cloud', 'line_cloud', 'mixed_scales', 'one_point_many_times'])
def test_chamfer_matrix_pass_scaling(case):
    """Operand scaling / f16 splitting of the approximate pass (nn_fast.h) at the ends of its range: clouds of radius 1e-6,
    queries 1e4 radii away, clouds with no extent along one or two axes, near-coincident queries inside a large cloud,
    a cloud that is one point repeated (radius 0)."""
    g = torch.Generator().manual_seed(sum(map(ord, case)))
    N = 2
    c = torch.tensor([0.3, 0.2, 0.1])
    if case == 'tiny_cloud':
        y = torch.randn(N, 1000, 3, generator=g) * 1e-6 + c
        x = torch.cat([torch.randn(N, 200, 3, generator=g) * 1e-6 + c, torch.randn(N, 100, 3, generator=g) * 0.5 + c], 1)
    elif case == 'far_queries':
        y = torch.randn(N, 2048, 3, generator=g) * 0.05 + c
        x = torch.cat([_sphere(200, g)[None].expand(N, -1, -1) * 500.0 + c, torch.randn(N, 100, 3, generator=g) * 0.05 + c], 1)
    elif case == 'flat_cloud':
        y = torch.randn(N, 1500, 3, generator=g) * 0.1 + c
        y[..., 2] = 0.25
        x = torch.randn(N, 400, 3, generator=g) * 0.1 + c
    elif case == 'line_cloud':
        y = torch.zeros(N, 777, 3) + c
        y[..., 0] += torch.randn(N, 777, generator=g) * 0.2
        x = torch.randn(N, 300, 3, generator=g) * 0.2 + c
    elif case == 'mixed_scales':
        y = torch.randn(N, 2048, 3, generator=g) * 5.0
        x = torch.cat([y[:, ::8] + torch.randn(N, 256, 3, generator=g) * 1e-4, torch.randn(N, 100, 3, generator=g) * 5.0], 1)
    else:
        y = (torch.zeros(N, 100, 3) + c).contiguous()
        x = torch.randn(N, 65, 3, generator=g) + c
    _check(x.float().contiguous(), y.float().contiguous())


@pytest.mark.parametrize('N,P1,P2', [(2, 700, 5000), (1, 3000, 20001), (1, 64, 100000)])
def test_chamfer_dense_clouds_multi_tile(N, P1, P2):
    """Databases larger than one LDS tile (BASELINE config 5 streams 100,000 points): the exact minimum is carried
    across tiles; near-ties across tile boundaries included (points on a sphere around some of the queries)."""
    g = torch.Generator().manual_seed(P2)
    y = torch.randn(N, P2, 3, generator=g) * 0.2 + torch.tensor([0.4, -0.2, 1.0])
    x = torch.randn(N, P1, 3, generator=g) * 0.25 + torch.tensor([0.4, -0.2, 1.0])
    c = x[:, :8].mean(dim=1, keepdim=True)
    y[:, ::7] = c + _sphere(y[:, ::7].shape[1], g)[None] * 0.3      # every 7th point equidistant from a query-side centre
    x[:, 0] = c[:, 0]
    y[:, P2 - 1] = y[:, 3]                                            # duplicate in the last tile: lowest index must win
    _check(x.contiguous(), y.contiguous())


def _pruned_cases():
    g = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g)
    lat = torch.stack(torch.meshgrid(*[torch.arange(17.)] * 3, indexing='ij'), -1).reshape(1, -1, 3)
    y6 = r(1, 6000, 3)
    return {
        'gauss': (r(3, 5000, 3), r(3, 9000, 3) * 0.5 + 0.3),
        'body_vs_scene': (r(2, 10475, 3) * torch.tensor([0.3, 0.15, 0.85]) + torch.tensor([0., 0., 0.9]), r(2, 30000, 3) * 0.25 + torch.tensor([0.3, 0.1, 1.0])),
        'far_apart': (r(2, 4096, 3) + 100.0, r(2, 4097, 3) * 0.01),
        'duplicates': (r(1, 5000, 3), torch.cat([y6, y6], 1)),                    # every database point twice, 6,000 indices apart
        'all_identical': (torch.zeros(2, 4100, 3) + 0.5, torch.zeros(2, 4100, 3) + 0.5),
        'lattice_ties': (lat + 0.5, lat),                                         # every query has eight equidistant neighbours
        'planar': (torch.cat([r(2, 7000, 2), torch.zeros(2, 7000, 1)], -1), torch.cat([r(2, 20000, 2), torch.zeros(2, 20000, 1) + 0.1], -1)),
        'few_queries': (r(4, 143, 3), r(4, 50000, 3)),
        'one_query': (r(1, 1, 3), r(1, 4096, 3)),
        'ragged': (r(3, 4999, 3) * 3, r(3, 6001, 3)),
        'two_clusters': (torch.cat([r(2, 4500, 3) * 0.01, r(2, 4500, 3) * 0.01 + 5], 1), torch.cat([r(2, 4500, 3) * 0.01 + 5, r(2, 4500, 3) * 0.01], 1)),
        'sphere_shell': (r(2, 300, 3) * 0.01, torch.nn.functional.normalize(r(2, 8192, 3), dim=-1)),    # all points ~equidistant from every query
    }


@pytest.mark.parametrize('case', sorted(_pruned_cases()))
def test_chamfer_pruned_search_equals_exact_scan(case):
    """Clouds of >= 4,096 points take the pruned search (nn_prune.h: Morton-cell counting sort, box bounds per 32 / 512 sorted
    points, per-query refinement; ChamferDistance allocates its workspace).  It must return what the sequential exact scan
    returns -- same squared distances, same ORIGINAL indices, lowest index on ties -- although it visits the points in another
    order: against the plain exact-scan kernel (SAGA_NN_EXACT_SCAN=1) on every case and against the C oracle where that takes
    seconds.  Cases: duplicates far apart in index, lattices (8-way ties), coincident clouds, planes, far / tiny clusters, shells."""
    import os
    import saga_amd.chamfer_distance as chd
    from saga_amd import _lib
    x, y = (t.float().contiguous() for t in _pruned_cases()[case])
    assert _lib.load().saga_chamfer_workspace_bytes(x.shape[0], x.shape[1], y.shape[1]) > 0, 'the case does not reach the pruned search'
    got = chd.ChamferDistance(x.cuda(), y.cuda())
    os.environ['SAGA_NN_EXACT_SCAN'] = '1'
    try:
        want = chd.ChamferDistance(x.cuda(), y.cuda())
    finally:
        del os.environ['SAGA_NN_EXACT_SCAN']
    for a, b, name in zip(got, want, ('dist1', 'dist2', 'idx1', 'idx2')):
        assert_same(a.cpu().numpy(), b.cpu().numpy(), name)
    if x.shape[0] * x.shape[1] * y.shape[1] <= 150e6:
        r1, r2, j1, j2 = chamfer_ref.chamfer_distance(x, y)
        assert_same(got[2].cpu().numpy(), j1.numpy(), 'idx1 vs oracle')
        assert_same(got[3].cpu().numpy(), j2.numpy(), 'idx2 vs oracle')
        assert_same(got[0].cpu().numpy(), r1.numpy(), 'dist1 vs oracle')


def test_chamfer_workspace_entry_point_through_the_c_abi():
    """saga_chamfer_fwd_ws: a workspace that is too small is refused with a message; without a workspace (or for small clouds) it
    is saga_chamfer_fwd; any output may be NULL; the scratch contents do not matter (poisoned workspace, two calls)."""
    from saga_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    N, P1, P2 = 2, 600, 9000
    x, y = torch.randn(N, P1, 3, generator=g).cuda(), (torch.randn(N, P2, 3, generator=g) * 0.7).cuda()
    nb = lib.saga_chamfer_workspace_bytes(N, P1, P2)
    assert nb > 0 and lib.saga_chamfer_workspace_bytes(N, 600, 2048) == 0 and lib.saga_chamfer_workspace_bytes(0, P1, P2) == 0
    P = _lib.ptr
    f32, i32 = dict(device='cuda', dtype=torch.float32), dict(device='cuda', dtype=torch.int32)
    ref = [torch.empty(N, P1, **f32), torch.empty(N, P2, **f32), torch.empty(N, P1, **i32), torch.empty(N, P2, **i32)]
    _lib.call('saga_chamfer_fwd', x.device, P(x), P(y), *(P(t) for t in ref), N, P1, P2)
    for fill in (0x00, 0xff):
        ws = torch.full((nb,), fill, device='cuda', dtype=torch.uint8)
        out = [torch.empty_like(t) for t in ref]
        _lib.call('saga_chamfer_fwd_ws', x.device, P(x), P(y), *(P(t) for t in out), N, P1, P2, P(ws), nb)
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
    only_idx1 = torch.empty(N, P1, **i32)
    _lib.call('saga_chamfer_fwd_ws', x.device, P(x), P(y), None, None, P(only_idx1), None, N, P1, P2, P(ws), nb)
    assert torch.equal(only_idx1, ref[2])
    with torch.cuda.device(0):
        rc = lib.saga_chamfer_fwd_ws(P(x), P(y), None, None, P(only_idx1), None, N, P1, P2, P(ws), nb - 1, None)
    assert rc == -1 and b'workspace' in lib.saga_last_error()
    no_ws = torch.empty(N, P1, **i32)
    _lib.call('saga_chamfer_fwd_ws', x.device, P(x), P(y), None, None, P(no_ws), None, N, P1, P2, None, 0)      # brute force
    assert torch.equal(no_ws, ref[2])


@pytest.mark.parametrize('directions', ['both', 'x2y', 'y2x'])
def test_p2p_signed_large_clouds_take_the_pruned_search(directions):
    """point2point_signed on clouds large enough for the pruned search (saga_p2p_signed_fwd_ws; reference call shape
    WholeGraspPose/trainer.py:280-281: a body mesh against a large object cloud): every output equals the scan path's
    (SAGA_NN_PRUNE=0) bit for bit -- signed distances, indices, vectors -- for both directions and for one direction alone, and
    the backward (which reads the indices) gives the same gradients."""
    import os
    from saga_amd.geometry import point2point_signed
    g = torch.Generator().manual_seed(21)
    N, P1, P2 = 2, 5000, 6000
    x = (torch.randn(N, P1, 3, generator=g) * torch.tensor([0.3, 0.15, 0.8])).cuda()
    y = (torch.randn(N, P2, 3, generator=g) * 0.3).cuda()
    y[:, 100:200] = y[:, 3000:3100]                                    # duplicates: exact ties, the lowest index must win
    xn = torch.nn.functional.normalize(torch.randn(N, P1, 3, generator=g), dim=-1).cuda()
    yn = torch.nn.functional.normalize(torch.randn(N, P2, 3, generator=g), dim=-1).cuda()
    wy, wx = torch.randn(N, P2, generator=g).cuda(), torch.randn(N, P1, generator=g).cuda()
    res = {}
    for mode in ('pruned', 'scan'):
        if mode == 'scan':
            os.environ['SAGA_NN_PRUNE'] = '0'
        try:
            xx, yy = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
            out = point2point_signed(xx, yy, xn, yn, return_vector=True, directions=directions)
            loss = sum((o * w).sum() for o, w in zip(out[:2], (wy, wx)) if o is not None)
            loss = loss + sum((o ** 2).sum() for o in out[4:] if o is not None)
            loss.backward()
            res[mode] = [o.detach() if o is not None else None for o in out] + [xx.grad, yy.grad]
        finally:
            os.environ.pop('SAGA_NN_PRUNE', None)
    for a, b in zip(res['pruned'], res['scan']):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    y2x, x2y, yidx, xidx = res['pruned'][:4]
    assert (y2x is None) == (directions == 'x2y') and (x2y is None) == (directions == 'y2x')
    if xidx is not None:
        assert not ((xidx >= 3000) & (xidx < 3100)).any()                # a duplicate at the higher index never wins its tie


@pytest.mark.parametrize('bad', [float('nan'), float('inf'), float('-inf'), 3e38, 1e10, 1e7], ids=['nan', 'inf', '-inf', '3e38', '1e10', '1e7'])
def test_bad_points_do_not_disturb_their_neighbours(bad):
    """A cloud with a non-finite, absurdly large or merely far point (a sentinel at 1e7 among points of size 1): the plain scan of the
    reference's extension just never picks it (NaN / Inf distances lose every '<').  The two-level scan and the pruned search must do
    the same for every query that is finite itself -- round 3 found both losing neighbours: a NaN / Inf coordinate turned its whole
    16-point chunk's minimum into NaN (the tile was then skipped for the query), and ONE far query in a wave set operand scales under
    which its wave-mates' coordinates underflowed below the pass's error bound (nn_fast.h: pad test on every coordinate,
    nnf_mfma_error_floor).  Both directions, both search forms, against SAGA_NN_EXACT_SCAN=1; bit-equal distances."""
    import os
    import saga_amd.chamfer_distance as chd
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(2, 5000, 3, generator=g) * 0.3).cuda()
    y = (torch.randn(2, 6000, 3, generator=g) * 0.3).cuda()
    y[0, 17, 1] = bad           # one coordinate of a point
    y[1, 4000] = bad            # a whole point
    x[0, 100, 0] = bad          # ... and in the other cloud (a bad QUERY of the x -> y direction, a bad database point of y -> x)

    def run(env):
        os.environ.update(env)
        try:
            out = chd.ChamferDistance(x, y)
            torch.cuda.synchronize()
        finally:
            for k in env:
                del os.environ[k]
        return out

    exact = run({'SAGA_NN_EXACT_SCAN': '1'})
    ok = (torch.isfinite(x).all(-1) & (x.abs().amax(-1) < 1e6), torch.isfinite(y).all(-1) & (y.abs().amax(-1) < 1e6))
    for name, env in (('pruned', {}), ('two-level scan', {'SAGA_NN_PRUNE': '0'})):
        out = run(env)
        for d in range(2):
            assert torch.equal(out[2 + d][ok[d]], exact[2 + d][ok[d]]), '%s: neighbour indices, direction %d' % (name, d)
            assert torch.equal(out[d][ok[d]], exact[d][ok[d]]), '%s: distances, direction %d' % (name, d)
    # the bad points are nobody's neighbour (queries that are themselves bad aside)
    assert not (exact[2][0][ok[0][0]] == 17).any() and not (exact[2][1][ok[0][1]] == 4000).any() and not (exact[3][0][ok[1][0]] == 100).any()


def test_nn_search_fuzz_against_exact_scan():
    """200 random clouds (gaussian / uniform / clusters / lines / lattices with exact ties / duplicates / planes with outliers /
    shells; scales 1e-3 .. 1e3, offsets up to 30 scales; sizes 1 .. 20,000 on either side, both sides of the 4,096-point threshold
    of the pruned search) through scripts/exp/nn_fuzz.py: the pruned search and the two-level scan equal SAGA_NN_EXACT_SCAN=1
    bit for bit.  (6,300 further cases were run by hand in round 3: no mismatch.)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts', 'exp'))
    import nn_fuzz
    assert nn_fuzz.fuzz(200, seed=2026) == 0


def test_p2p_signed_fuzz_against_oracle():
    """40 random cloud pairs (scripts/exp/p2p_fuzz.py: shapes / scales / sizes of the NN fuzz, with and without normals) through
    point2point_signed: indices equal the exact scan's, forward within 2e-5 of the fp64 oracle evaluated with those indices, both
    backward forms (fixed-point accumulation, fixed-order gather) within 2e-4 of its autograd.  (1,260 further cases by hand.)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts', 'exp'))
    import p2p_fuzz
    assert p2p_fuzz.fuzz(40, seed=2026) == 0


def test_chamfer_both_call_forms_and_errors():
    import saga_amd.chamfer_distance as chd
    x = torch.randn(1, 5, 3).cuda()
    y = torch.randn(1, 6, 3).cuda()
    a = chd.ChamferDistance(x, y)
    b = chd.ChamferDistance()(x, y)
    for s, t in zip(a, b):
        assert torch.equal(s, t)
    with pytest.raises(ValueError):
        chd.ChamferDistance(x, torch.randn(2, 6, 3).cuda())
    with pytest.raises(RuntimeError):
        chd.ChamferDistance(x, torch.zeros(1, 0, 3).cuda())     # empty cloud has no neighbour
    with pytest.raises(RuntimeError):
        chd.ChamferDistance(x.cpu(), y.cpu())                   # no CPU fallback


def test_chamfer_backward_matches_autograd_of_gathered_distance():
    import saga_amd.chamfer_distance as chd
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 50, 3, generator=g, dtype=torch.float64)
    y = torch.randn(2, 70, 3, generator=g, dtype=torch.float64)
    w1 = torch.randn(2, 50, generator=g, dtype=torch.float64)
    w2 = torch.randn(2, 70, generator=g, dtype=torch.float64)
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    _, _, i1, i2 = chamfer_ref.chamfer_distance(x.float(), y.float())
    d1 = ((xr - yr.gather(1, i1.long()[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    d2 = ((yr - xr.gather(1, i2.long()[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    ((d1 * w1).sum() + (d2 * w2).sum()).backward()
    xh, yh = x.float().cuda().requires_grad_(True), y.float().cuda().requires_grad_(True)
    h1, h2, _, _ = chd.ChamferDistance(xh, yh)
    ((h1 * w1.float().cuda()).sum() + (h2 * w2.float().cuda()).sum()).backward()
    np.testing.assert_allclose(xh.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(yh.grad.cpu().numpy(), yr.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_chamfer_backward_is_deterministic_with_heavy_collisions():
    """saga_chamfer_bwd is a fixed-order gather (no float atomics): many points share one neighbour (5,003 points against 37,
    across two LDS index tiles), the result is bit-equal run to run, independent of the batch around an element, only one
    direction's gradient may be given, and it matches float64 autograd of the gathered distances."""
    import saga_amd.chamfer_distance as chd
    g = torch.Generator().manual_seed(9)
    N, P1, P2 = 3, 37, 5003
    x = torch.randn(N, P1, 3, generator=g)
    y = torch.randn(N, P2, 3, generator=g)
    w1, w2 = torch.randn(N, P1, generator=g), torch.randn(N, P2, generator=g)

    def run(xc, yc, a, b):
        xh, yh = xc.cuda().requires_grad_(True), yc.cuda().requires_grad_(True)
        h1, h2, i1, i2 = chd.ChamferDistance(xh, yh)
        loss = 0
        if a is not None:
            loss = loss + (h1 * a.cuda()).sum()
        if b is not None:
            loss = loss + (h2 * b.cuda()).sum()
        loss.backward()
        return xh.grad.cpu(), yh.grad.cpu(), i1.cpu(), i2.cpu()

    gx, gy, i1, i2 = run(x, y, w1, w2)
    gx2, gy2, _, _ = run(x, y, w1, w2)
    assert torch.equal(gx, gx2) and torch.equal(gy, gy2)
    gx_b, gy_b, _, _ = run(x[1:2], y[1:2], w1[1:2], w2[1:2])           # element 1 alone
    assert torch.equal(gx_b[0], gx[1]) and torch.equal(gy_b[0], gy[1])
    xr, yr = x.double().requires_grad_(True), y.double().requires_grad_(True)
    d1 = ((xr - yr.gather(1, i1.long()[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    d2 = ((yr - xr.gather(1, i2.long()[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    ((d1 * w1.double()).sum() + (d2 * w2.double()).sum()).backward()
    np.testing.assert_allclose(gx.numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-4)    # ~135 terms per x point
    np.testing.assert_allclose(gy.numpy(), yr.grad.numpy(), rtol=1e-4, atol=1e-5)
    # one direction only
    gx1, gy1, _, _ = run(x, y, None, w2)
    xr.grad = yr.grad = None
    d2 = ((yr - xr.gather(1, i2.long()[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    (d2 * w2.double()).sum().backward()
    np.testing.assert_allclose(gx1.numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gy1.numpy(), yr.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('name', ['markers', 'rhand'])
def test_point2point_signed_against_reference_golden(golden, name):
    """The product function on the HIP NN search against outputs of the reference's own function."""
    from saga_amd.geometry import point2point_signed
    g = golden('p2p_signed.npz')
    c = lambda k: torch.from_numpy(g[f'{name}_{k}']).cuda()
    x = c('x').requires_grad_(True)
    y2x, x2y, yidx, xidx = point2point_signed(x, c('y'), c('xn'), c('yn'))
    assert_same(yidx.cpu().numpy(), g[f'{name}_yidx'])
    assert_same(xidx.cpu().numpy(), g[f'{name}_xidx'])
    np.testing.assert_allclose(y2x.detach().cpu().numpy(), g[f'{name}_y2x'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(x2y.detach().cpu().numpy(), g[f'{name}_x2y'], rtol=1e-5, atol=1e-7)
    ((y2x * c('wy')).sum() + (x2y * c('wx')).sum()).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[f'{name}_grad_x'], rtol=1e-4, atol=1e-5)


def test_p2p_signed_operator_against_oracle_with_vectors_and_both_gradients():
    """saga_p2p_signed_fwd/bwd (one HIP operator) against the pinned oracle (oracle/geometry_ref.point2point_signed, itself checked
    against the reference function's outputs): signed distances, difference vectors, indices, and the gradients w.r.t. BOTH clouds
    of a loss that uses all four float outputs (the MotionFill consumer reads the vectors, fittingop.py:154).  Many object points
    share one hand vertex (3,001 points against 97), the index tiles are ragged; the backward is bit-reproducible run to run."""
    from saga_amd.geometry import point2point_signed
    g = torch.Generator().manual_seed(21)
    N, P1, P2 = 3, 97, 3001
    x = torch.randn(N, P1, 3, generator=g) * 0.1
    y = torch.randn(N, P2, 3, generator=g) * 0.1
    xn = torch.nn.functional.normalize(torch.randn(N, P1, 3, generator=g), dim=-1)
    yn = torch.nn.functional.normalize(torch.randn(N, P2, 3, generator=g), dim=-1)
    w = [torch.randn(s, generator=g) for s in ((N, P2), (N, P1), (N, P2, 3), (N, P1, 3))]

    def run(device, dtype, fn, normals=True):
        xr = x.to(device, dtype).requires_grad_(True)
        yr = y.to(device, dtype).requires_grad_(True)
        out = fn(xr, yr, xn.to(device, dtype) if normals else None, yn.to(device, dtype) if normals else None, return_vector=True)
        loss = sum((o * wi.to(device, dtype)).sum() for o, wi in zip((out[0], out[1], out[4], out[5]), w))
        loss.backward()
        return [o.detach().cpu() for o in out], xr.grad.cpu(), yr.grad.cpu()

    for normals in (True, False):
        out_h, gx_h, gy_h = run('cuda', torch.float32, point2point_signed, normals)
        out_r, gx_r, gy_r = run('cpu', torch.float64, geometry_ref.point2point_signed, normals)
        # the oracle's float64 search can pick another index at a float32 near-tie: compare where the indices agree (all, here)
        assert_same(out_h[2].numpy(), out_r[2].numpy(), 'yidx')
        assert_same(out_h[3].numpy(), out_r[3].numpy(), 'xidx')
        for k in (0, 1, 4, 5):
            np.testing.assert_allclose(out_h[k].numpy(), out_r[k].numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(gx_h.numpy(), gx_r.numpy(), rtol=1e-4, atol=1e-4)       # ~31 scattered terms per x point
        np.testing.assert_allclose(gy_h.numpy(), gy_r.numpy(), rtol=1e-4, atol=1e-5)
        if not normals:
            assert (out_h[0] >= 0).all() and (out_h[1] >= 0).all()
        out_2, gx_2, gy_2 = run('cuda', torch.float32, point2point_signed, normals)
        assert torch.equal(gx_2, gx_h) and torch.equal(gy_2, gy_h)
        # the other backward kernel (fixed-order gather instead of fixed-point LDS accumulation; the default for clouds beyond
        # 2,048 targets): same gradients to float32 rounding, and bit-reproducible too
        import os
        os.environ['SAGA_P2P_GATHER'] = '1'
        try:
            _, gx_g, gy_g = run('cuda', torch.float32, point2point_signed, normals)
            _, gx_g2, gy_g2 = run('cuda', torch.float32, point2point_signed, normals)
        finally:
            del os.environ['SAGA_P2P_GATHER']
        assert torch.equal(gx_g, gx_g2) and torch.equal(gy_g, gy_g2)
        np.testing.assert_allclose(gx_g.numpy(), gx_h.numpy(), rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(gy_g.numpy(), gy_h.numpy(), rtol=1e-5, atol=1e-6)
    # only x needs a gradient (the SAGA call sites): y's gradient is not computed
    xr = x.cuda().requires_grad_(True)
    o = point2point_signed(xr, y.cuda(), xn.cuda(), yn.cuda())
    (o[0].sum() + o[1].sum()).backward()
    assert xr.grad is not None and len(o) == 4 and o[2].dtype == torch.int32
    # directions='x2y': the y -> x search is skipped, its outputs are None, the x -> y outputs and their gradients are those of
    # the full call with the y -> x outputs unused
    res = {}
    for d in ('both', 'x2y'):
        xr, yr = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
        o = point2point_signed(xr, yr, None, yn.cuda(), return_vector=True, directions=d)
        ((o[1] * w[1].cuda()).sum() + (o[5] * w[3].cuda()).sum()).backward()
        res[d] = (o, xr.grad.clone(), yr.grad.clone())
    assert res['x2y'][0][0] is None and res['x2y'][0][2] is None and res['x2y'][0][4] is None
    for k in (1, 3, 5):
        assert torch.equal(res['x2y'][0][k], res['both'][0][k])
    assert torch.equal(res['x2y'][1], res['both'][1]) and torch.equal(res['x2y'][2], res['both'][2])
    res = {}
    for d in ('both', 'y2x'):
        xr, yr = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
        o = point2point_signed(xr, yr, xn.cuda(), None, return_vector=True, directions=d)
        ((o[0] * w[0].cuda()).sum() + (o[4] * w[2].cuda()).sum()).backward()
        res[d] = (o, xr.grad.clone(), yr.grad.clone())
    assert res['y2x'][0][1] is None and res['y2x'][0][3] is None and res['y2x'][0][5] is None
    for k in (0, 2, 4):
        assert torch.equal(res['y2x'][0][k], res['both'][0][k])
    assert torch.equal(res['y2x'][1], res['both'][1]) and torch.equal(res['y2x'][2], res['both'][2])
    with pytest.raises(ValueError):
        point2point_signed(x.cuda(), y.cuda(), directions='neither')


def test_p2p_backward_non_finite_upstream_terms_poison_their_target_only():
    """non-finite upstream gradients.  Three (or more) NaN / Inf terms that land on ONE target used to wrap the 64-bit
    fixed-point accumulator into a finite garbage gradient; now the target's row is NaN (what torch's scatter-add gives) and every
    other row is what it is without the poisoned terms."""
    from saga_amd import _lib
    g = torch.Generator().manual_seed(2)
    N, P1, P2 = 2, 50, 400
    x = (torch.randn(N, P1, 3, generator=g) * 0.1).cuda()
    y = (torch.randn(N, P2, 3, generator=g) * 0.1).cuda()
    f32, i32 = dict(device='cuda', dtype=torch.float32), dict(device='cuda', dtype=torch.int32)
    y2x, x2y = torch.empty(N, P2, **f32), torch.empty(N, P1, **f32)
    yidx, xidx = torch.empty(N, P2, **i32), torch.empty(N, P1, **i32)
    P = _lib.ptr
    _lib.call('saga_p2p_signed_fwd', x.device, P(x), P(y), None, None, P(y2x), P(x2y), P(yidx), P(xidx), None, None, N, P1, P2)
    g_y2x = torch.randn(N, P2, generator=g).cuda()
    target = int(torch.bincount(yidx[0].long()).argmax())           # the x point most y points of batch element 0 send to
    senders = torch.nonzero(yidx[0] == target).flatten()
    assert senders.numel() >= 4
    clean = g_y2x.clone()
    clean[0, senders[:4]] = 0.0
    bad = g_y2x.clone()
    bad[0, senders[0]], bad[0, senders[1]], bad[0, senders[2]], bad[0, senders[3]] = float('nan'), float('inf'), float('inf'), float('-inf')

    def bwd(gy):
        gx = torch.empty(N, P1, 3, **f32)
        _lib.call('saga_p2p_signed_bwd', x.device, P(x), P(y), P(yidx), P(xidx), P(y2x), P(x2y), P(gy), None, None, None, P(gx), None,
                  N, P1, P2)
        return gx

    gx_clean, gx_bad = bwd(clean), bwd(bad)
    assert torch.isnan(gx_bad[0, target]).all()
    keep = torch.ones(N, P1, dtype=torch.bool, device='cuda')
    keep[0, target] = False
    assert torch.isfinite(gx_bad[keep]).all()
    np.testing.assert_allclose(gx_bad[keep].cpu().numpy(), gx_clean[keep].cpu().numpy(), rtol=1e-6, atol=1e-9)


def test_point2point_signed_value_error():
    from saga_amd.geometry import point2point_signed
    with pytest.raises(ValueError):
        point2point_signed(torch.zeros(2, 5, 3).cuda(), torch.zeros(3, 7, 3).cuda())


def test_vertex_normals_full_and_subset(body_model_dir):
    from saga_amd.geometry import vertex_normals, vertex_normals_subset
    from tests.helpers import make_hip_model
    bm = make_hip_model(body_model_dir, 3)
    g = torch.Generator().manual_seed(2)
    verts = torch.randn(3, 10475, 3, generator=g)
    ref = geometry_ref.vertex_normals(verts.double(), torch.from_numpy(bm.faces)).numpy()
    full = vertex_normals(bm, verts.cuda()).cpu().numpy()
    # float32 on the device vs Open3D's double: compare directions
    assert np.abs(full - ref).max() < 1e-4
    sel = torch.arange(7331, 7331 + 778, dtype=torch.int32).cuda()
    sub = vertex_normals_subset(bm, verts.cuda(), sel).cpu().numpy()
    assert_same(sub, full[:, 7331:7331 + 778])


def test_vertex_normals_lds_staged_kernel_equals_gather_kernel(body_model_dir):
    """Batches of >= 512 meshes take the kernel that stages each mesh in LDS first (saga_vertex_normals[_subset]); it must give
    the gather kernel's bits (SAGA_NORMALS_GATHER=1 forces that one), for the subset and the full output, whatever the 16-B
    alignment of a mesh's row (V * 3 = 31,425 floats: every alignment occurs)."""
    import os
    from saga_amd.geometry import vertex_normals, vertex_normals_subset
    from tests.helpers import make_hip_model
    bm = make_hip_model(body_model_dir, 1)
    g = torch.Generator().manual_seed(5)
    base = torch.randn(1, 10475, 3, generator=g).cuda()
    verts = (base + 0.05 * torch.randn(515, 1, 3, generator=g).cuda()).contiguous()
    verts[7] = 0.0                                  # degenerate faces: zero normals stay zero
    sel = torch.cat([torch.arange(7331, 7331 + 778), torch.tensor([0, 10474, 5, 5])]).to(torch.int32).cuda()
    got_sub, got_full = vertex_normals_subset(bm, verts, sel), vertex_normals(bm, verts)
    os.environ['SAGA_NORMALS_GATHER'] = '1'
    try:
        ref_sub, ref_full = vertex_normals_subset(bm, verts, sel), vertex_normals(bm, verts)
    finally:
        del os.environ['SAGA_NORMALS_GATHER']
    assert torch.equal(got_sub, ref_sub) and torch.equal(got_full, ref_full)
    assert torch.equal(got_sub, got_full[:, sel.long()]) and got_full[7].abs().max() == 0
    small = vertex_normals_subset(bm, verts[:4].contiguous(), sel)           # below the threshold: the gather kernel
    assert torch.equal(small, got_sub[:4])
    # saga_vertex_normals_subset_window: the frames t0.. of [S,T,V,3] without copying the slice -- both kernels
    for S, T, t0 in [(103, 5, 2), (2, 5, 4), (3, 4, 0)]:
        seq = verts[:S * T].reshape(S, T, 10475, 3)
        win = vertex_normals_subset(bm, seq, sel, first_frame=t0)
        ref = vertex_normals_subset(bm, seq[:, t0:].reshape(S * (T - t0), 10475, 3).contiguous(), sel)
        assert win.shape == ref.shape and torch.equal(win, ref), (S, T, t0)
    with pytest.raises(ValueError):
        vertex_normals_subset(bm, verts[:6].reshape(2, 3, 10475, 3), sel, first_frame=3)
