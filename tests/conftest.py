import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a ROCm GPU (MI355X); run with -m gpu')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    return lambda name: np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def body_model_dir(tmp_path_factory):
    """Full-size synthetic SMPL-X (.npz) written where smplx.create() expects it."""
    from saga_amd.synthetic import write_body_model
    root = tmp_path_factory.mktemp('body_models')
    write_body_model(str(root), gender='male', seed=7)
    return str(root)


@pytest.fixture(scope='session')
def small_model_dir(tmp_path_factory):
    """Reduced mesh (V=203, ragged vs every tile size) with DENSE skinning weights."""
    from saga_amd.synthetic import write_body_model
    root = tmp_path_factory.mktemp('body_models_small')
    write_body_model(str(root), gender='male', seed=3, n_verts=203, n_faces=390, dense_weights=True)
    return str(root)


@pytest.fixture(scope='session')
def mesh_model_dir(tmp_path_factory):
    """Full-size synthetic SMPL-X whose template and faces are a mesh-like tube (face corners a few ids apart) with local weights."""
    from saga_amd.synthetic import write_body_model
    root = tmp_path_factory.mktemp('body_models_mesh')
    write_body_model(str(root), gender='male', seed=7, mesh_faces=True, local_weights=True)
    return str(root)
