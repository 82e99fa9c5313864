"""Shared helpers for the GPU parity tests (HIP path vs the CPU oracle on the same seeded inputs)."""
import os

import numpy as np
import torch

from oracle import smplx_ref
from saga_amd.synthetic import small_vertex_ids


def model_path(root, gender='male'):
    return os.path.join(root, 'smplx', 'SMPLX_%s.npz' % gender.upper())


def load_oracle_model(root, batch_size, dtype=torch.float64, small=False, **kw):
    data = dict(np.load(model_path(root), allow_pickle=False))
    vid = small_vertex_ids(data['v_template'].shape[0]) if small else None
    return smplx_ref.SMPLXRef(data, num_pca_comps=24, batch_size=batch_size, dtype=dtype, vertex_ids=vid, **kw)


def make_hip_model(root, batch_size, small=False, **kw):
    import saga_amd.smplx as smplx
    extra = {}
    if small:
        data = np.load(model_path(root), allow_pickle=False)
        extra['vertex_ids'] = small_vertex_ids(data['v_template'].shape[0])
    extra.update(kw)
    bm = smplx.create(root, model_type='smplx', gender='male', ext='npz', num_pca_comps=24,
                      create_global_orient=True, create_body_pose=True, create_betas=True,
                      create_left_hand_pose=True, create_right_hand_pose=True, create_expression=True,
                      create_jaw_pose=True, create_leye_pose=True, create_reye_pose=True, create_transl=True,
                      batch_size=batch_size, **extra)
    return bm.to('cuda').eval()


def random_body_inputs(B, seed=0, scale=1.0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, dtype=torch.float64) * sc * scale).to(dtype)
    return {
        'betas': r(B, 10), 'global_orient': r(B, 3), 'body_pose': r(B, 63, sc=0.4),
        'left_hand_pose': r(B, 24, sc=0.5), 'right_hand_pose': r(B, 24, sc=0.5),
        'transl': r(B, 3), 'expression': r(B, 10, sc=0.5), 'jaw_pose': r(B, 3, sc=0.2),
        'leye_pose': r(B, 3, sc=0.1), 'reye_pose': r(B, 3, sc=0.1),
    }


def to_cuda(d, requires_grad=False):
    out = {}
    for k, v in d.items():
        t = v.detach().to(torch.float32).cuda()
        out[k] = t.requires_grad_(True) if requires_grad else t
    return out


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def assert_same(a, b, name=''):
    """Bit-exact array equality with a SHORT failure message (no array dumps)."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, '%s: shape %s vs %s' % (name, a.shape, b.shape)
    bad = a != b
    if bad.any():
        i = np.argwhere(bad)[0]
        raise AssertionError('%s: %d of %d entries differ; first at %s: %r vs %r'
                             % (name, int(bad.sum()), a.size, tuple(i), a[tuple(i)], b[tuple(i)]))
