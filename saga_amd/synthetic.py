"""Synthetic, shape-compatible inputs for the GraspPose-Opt hot path.

The licensed SMPL-X ``.npz``, the VPoser snapshot, the GRAB object meshes and the
pretrained WholeGrasp-VAE are absent (reference ``README.md:62-130``, ``.gitignore:1-7``),
so tests, ``bench.py`` and ``smoke()`` run on random arrays that have the shapes, dtypes
and structural properties of the real ones:

* ``make_body_model``  -> dict with the fields of ``SMPLX_<GENDER>.npz``
* ``write_body_model`` -> stores it as ``<dir>/smplx/SMPLX_MALE.npz`` so that the
  ``smplx.create(model_path, model_type='smplx', gender=..., ext='npz')`` path logic
  (reference call site ``WholeGraspPose/models/fittingop.py:30-44``) is exercised
* ``make_object_cloud`` -> 2,048 points on an ellipsoid with analytic normals, placed like
  ``opt_grasppose.py:31-47`` places an object (height U(0.6,1.6), yaw U(-pi/2,pi/2))
* ``make_candidate_params`` -> seeded optimiser variables of ``FittingOP``
  (``WholeGraspPose/models/fittingop.py:71-76``)
* ``make_grasp_targets`` -> marker targets / contact maps with the shapes the
  WholeGrasp-VAE emits (``opt_grasppose.py:115-117``)

Nothing here is on the timed path; this is input generation only.
"""
import os

import numpy as np

# SMPL-X kinematic tree (55 joints). Public model constant; corroborated in-tree by
# the joint names / limb tables of utils/como/como_utils.py:411-530.
SMPLX_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
     20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int64)

NUM_JOINTS = 55
NUM_BODY_JOINTS = 21
NUM_HAND_JOINTS = 15
NUM_VERTS_SMPLX = 10475
NUM_FACES_SMPLX = 20908
NUM_LANDMARKS = 51


def _random_faces(rng, n_verts, n_faces):
    """Random valid triangles (3 distinct ids each) that cover every vertex at least once."""
    n_cover = (n_verts + 2) // 3
    n_faces = max(n_faces, n_cover)
    faces = np.empty((n_faces, 3), dtype=np.int64)
    perm = rng.permutation(n_verts)
    pad = (-n_verts) % 3
    if pad:
        perm = np.concatenate([perm, perm[:pad]])
    cover = perm.reshape(-1, 3)
    # a wrapped-around row can repeat an id when n_verts < 6; patch it
    for row in cover:
        while len(set(row.tolist())) < 3:
            row[:] = rng.choice(n_verts, 3, replace=False)
    faces[:n_cover] = cover
    rest = n_faces - n_cover
    if rest:
        a = rng.integers(0, n_verts, rest)
        b = (a + 1 + rng.integers(0, n_verts - 1, rest)) % n_verts
        c = rng.integers(0, n_verts, rest)
        bad = (c == a) | (c == b)
        while bad.any():
            c[bad] = rng.integers(0, n_verts, int(bad.sum()))
            bad = (c == a) | (c == b)
        faces[n_cover:] = np.stack([a, b, c], 1)
    return faces


def _tube_mesh(rng, n_verts, n_faces, cols=25):
    """A mesh-like stand-in for the body surface: vertices on a closed tube (rings of ``cols`` vertices stacked along z, radius
    varying smoothly), numbered ring by ring, faces joining each ring to the next -- so that, as on a real scan-derived mesh,
    the corners of a face are a few ids apart, a vertex has valence ~6, and a run of ids (the hand: 7331..8128) is a compact
    patch with a thin ring of neighbours.  -> (v_template [V,3], faces [>= n_faces, 3])."""
    V = int(n_verts)
    rows = (V + cols - 1) // cols
    r, c = np.arange(V) // cols, np.arange(V) % cols
    theta = 2 * np.pi * c / cols
    z = -0.85 + 1.7 * r / max(rows - 1, 1)
    rad = 0.5 + 0.5 * np.sin(np.pi * r / max(rows - 1, 1)) + 0.05 * np.sin(11 * z)
    v = np.stack([0.15 * rad * np.cos(theta), 0.1 * rad * np.sin(theta), z], 1) + rng.normal(0, 2e-3, (V, 3))
    faces = []
    vid = lambda rr, cc: rr * cols + (cc % cols)
    for rr in range(rows - 1):
        for cc in range(cols):
            a, b, d, e = vid(rr, cc), vid(rr, cc + 1), vid(rr + 1, cc), vid(rr + 1, cc + 1)
            if max(a, b, d, e) >= V:
                continue
            faces.append((a, b, d))
            faces.append((b, e, d))
    faces = np.asarray(faces, dtype=np.int64)
    extra = max(n_faces - faces.shape[0], 0)              # caps / fillers: fans over the first ring
    if extra:
        k = np.arange(extra) % max(cols - 2, 1)
        faces = np.concatenate([faces, np.stack([np.zeros(extra, np.int64), k + 1, k + 2], 1)])
    return v, faces


def make_body_model(seed=7, n_verts=NUM_VERTS_SMPLX, n_faces=NUM_FACES_SMPLX,
                    weights_nnz=4, dense_weights=False, local_weights=False, mesh_faces=False, n_shape=20, dtype=np.float32):
    """Random arrays with the layout of a SMPL-X ``.npz`` (10 shape + 10 expression dirs).  ``mesh_faces``: template and faces
    of a mesh-like tube (:func:`_tube_mesh`) instead of SURVEY 8d's random points joined by random triangles."""
    rng = np.random.default_rng(seed)
    V, J = int(n_verts), NUM_JOINTS
    v_template = rng.uniform([-0.3, -0.15, -0.85], [0.3, 0.15, 0.85], size=(V, 3))
    shapedirs = rng.normal(0.0, 0.01, size=(V, 3, n_shape))
    # real files store posedirs as [V, 3, 486]; smplx reshapes to [486, V*3]
    posedirs = rng.normal(0.0, 1e-3, size=(V, 3, (J - 1) * 9))

    nnz_j = min(32, V)
    J_regressor = np.zeros((J, V))
    for j in range(J):
        ids = rng.choice(V, nnz_j, replace=False)
        w = rng.uniform(0.1, 1.0, nnz_j)
        J_regressor[j, ids] = w / w.sum()

    if dense_weights:
        weights = rng.uniform(0.0, 1.0, size=(V, J)) ** 4
    elif local_weights:
        # mesh-like structure: runs of consecutive vertices share their joints (as the vertices of a body part do), instead
        # of every vertex drawing its own random joints (the default: the worst case for anything that caches transforms)
        weights = np.zeros((V, J))
        k = min(weights_nnz, J)
        first = (np.arange(V) * J) // V
        for d in range(k):
            weights[np.arange(V), (first + d) % J] = rng.uniform(0.05, 1.0, V)
    else:
        weights = np.zeros((V, J))
        k = min(weights_nnz, J)
        for v in range(V):
            ids = rng.choice(J, k, replace=False)
            weights[v, ids] = rng.uniform(0.05, 1.0, k)
    weights /= weights.sum(1, keepdims=True)

    def _orth(n):
        q, _ = np.linalg.qr(rng.normal(size=(n, n)))
        return q

    faces = _random_faces(rng, V, n_faces)
    if mesh_faces:
        v_template, faces = _tube_mesh(rng, V, n_faces)
    F = faces.shape[0]
    lmk_faces_idx = rng.integers(0, F, NUM_LANDMARKS)
    lmk_bary = rng.dirichlet(np.ones(3), NUM_LANDMARKS)
    kintree = np.stack([np.where(SMPLX_PARENTS < 0, 2 ** 32 - 1, SMPLX_PARENTS),
                        np.arange(J)]).astype(np.int64)
    return {
        'v_template': v_template.astype(dtype),
        'f': faces.astype(np.uint32),
        'shapedirs': shapedirs.astype(dtype),
        'posedirs': posedirs.astype(dtype),
        'J_regressor': J_regressor.astype(dtype),
        'kintree_table': kintree,
        'weights': weights.astype(dtype),
        'hands_componentsl': _orth(45).astype(dtype),
        'hands_componentsr': _orth(45).astype(dtype),
        'hands_meanl': rng.normal(0.0, 0.1, 45).astype(dtype),
        'hands_meanr': rng.normal(0.0, 0.1, 45).astype(dtype),
        'lmk_faces_idx': lmk_faces_idx.astype(np.int64),
        'lmk_bary_coords': lmk_bary.astype(dtype),
        'dynamic_lmk_faces_idx': rng.integers(0, F, (79, 17)).astype(np.int64),
        'dynamic_lmk_bary_coords': rng.dirichlet(np.ones(3), (79, 17)).astype(dtype),
    }


def write_body_model(root, gender='male', **kw):
    """Write ``<root>/smplx/SMPLX_<GENDER>.npz`` and return ``root`` (the ``model_path``)."""
    d = os.path.join(root, 'smplx')
    os.makedirs(d, exist_ok=True)
    np.savez(os.path.join(d, 'SMPLX_%s.npz' % gender.upper()), **make_body_model(**kw))
    return root


def small_vertex_ids(n_verts):
    """Extra-joint vertex ids for a reduced test mesh (the real ids exceed a small V)."""
    names = ['nose', 'reye', 'leye', 'rear', 'lear', 'LBigToe', 'LSmallToe', 'LHeel', 'RBigToe',
             'RSmallToe', 'RHeel', 'lthumb', 'lindex', 'lmiddle', 'lring', 'lpinky',
             'rthumb', 'rindex', 'rmiddle', 'rring', 'rpinky']
    return {n: int((37 * i + 11) % n_verts) for i, n in enumerate(names)}


def make_object_cloud(n_samples, n_points=2048, seed=0, semi_axes=(0.06, 0.04, 0.03), dtype=np.float32):
    """Ellipsoid 'camera' stand-in: verts [n,P,3], outward unit normals [n,P,3]."""
    rng = np.random.default_rng(seed)
    a = np.asarray(semi_axes)
    u = rng.normal(size=(n_points, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    pts = u * a
    nrm = pts / (a * a)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    height = rng.uniform(0.6, 1.6, n_samples)          # opt_grasppose.py:31
    yaw = rng.uniform(-np.pi / 2, np.pi / 2, n_samples)  # opt_grasppose.py:32
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.zeros((n_samples, 3, 3))
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = c, -s, s, c, 1.0
    verts = np.einsum('nij,pj->npi', R, pts)
    normals = np.einsum('nij,pj->npi', R, nrm)
    verts[:, :, 2] += height[:, None]
    return verts.astype(dtype), normals.astype(dtype), height.astype(dtype), yaw.astype(dtype)


def make_candidate_params(batch, seed=1234, hand_ncomps=24, dtype=np.float32, scale=1.0):
    """Seeded optimiser variables, as numpy arrays."""
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(batch, 3))
    b = rng.normal(size=(batch, 3))
    rot6d = np.stack([a, b], axis=-1).reshape(batch, 6)    # columns of a 3x2 block
    out = {
        'betas': scale * rng.normal(0, 1.0, (batch, 10)),
        'transl_rec': rng.normal(0, 0.1, (batch, 3)) + np.array([0.0, 0.0, 1.0]),
        'glo_rot_rec': rot6d,
        'vpose_rec': scale * rng.normal(0, 1.0, (batch, 32)),
        'hand_pose': scale * rng.normal(0, 0.5, (batch, 2 * hand_ncomps)),
        'eye_pose': scale * rng.normal(0, 0.05, (batch, 6)),
    }
    return {k: v.astype(dtype) for k, v in out.items()}


def make_grasp_targets(batch, obj_verts, seed=4815, n_markers=143, dtype=np.float32):
    """Marker targets near the object + contact maps in [0,1] (shapes of opt_grasppose.py:115-117)."""
    rng = np.random.default_rng(seed)
    centre = obj_verts.mean(axis=1, keepdims=True)            # [B or 1,1,3]
    markers = centre + rng.normal(0, 0.25, (batch, n_markers, 3))
    markers_contact = rng.uniform(0, 1, (batch, n_markers)) ** 3
    object_contact = rng.uniform(0, 1, (batch, obj_verts.shape[1])) ** 3
    return markers.astype(dtype), markers_contact.astype(dtype), object_contact.astype(dtype)
