"""Drop-in for the part of the ``smplx`` package the reference uses:
``smplx.create(model_path, model_type='smplx', gender=..., ext='npz', num_pca_comps=24,
create_*=True, batch_size=B)`` -> module ``__call__(return_verts=True, **body_param)``
(``WholeGraspPose/models/fittingop.py:30-59,149``; ``utils/utils_body.py:108-134``).

Hand-PCA expansion, concatenation into ``full_pose`` and the ``pose_mean`` offset are a few
tiny torch ops (autograd handles them); everything from ``full_pose`` on -- Rodrigues, blend
shapes, joint regression, kinematic chain, skinning, extra joints, landmarks -- is one call into
``libsaga_hip.so`` (:func:`saga_amd.smplx.lbs.lbs`).
"""
import ctypes
import os
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from . import lbs as _lbs

NUM_BODY_JOINTS = 21
NUM_HAND_JOINTS = 15
NUM_JOINTS = 55
SHAPE_SPACE_DIM = 300

# vertex ids smplx' VertexJointSelector appends as joints (public constants of the smplx package);
# order: face (5), feet (6), left finger tips (5), right finger tips (5)
VERTEX_IDS_SMPLX = {
    'nose': 9120, 'reye': 9929, 'leye': 9448, 'rear': 616, 'lear': 6,
    'LBigToe': 5770, 'LSmallToe': 5780, 'LHeel': 8846, 'RBigToe': 8463, 'RSmallToe': 8474, 'RHeel': 8635,
    'lthumb': 5361, 'lindex': 4933, 'lmiddle': 5058, 'lring': 5169, 'lpinky': 5286,
    'rthumb': 8079, 'rindex': 7669, 'rmiddle': 7794, 'rring': 7905, 'rpinky': 8022,
}
_EXTRA_ORDER = ['nose', 'reye', 'leye', 'rear', 'lear', 'LBigToe', 'LSmallToe', 'LHeel', 'RBigToe', 'RSmallToe',
                'RHeel', 'lthumb', 'lindex', 'lmiddle', 'lring', 'lpinky', 'rthumb', 'rindex', 'rmiddle', 'rring', 'rpinky']

_OUT_FIELDS = ['vertices', 'joints', 'full_pose', 'global_orient', 'transl', 'v_shaped', 'betas', 'body_pose',
               'expression', 'left_hand_pose', 'right_hand_pose', 'jaw_pose', 'vertices_sel']
SMPLXOutput = namedtuple('SMPLXOutput', _OUT_FIELDS, defaults=[None] * len(_OUT_FIELDS))


class SMPLX(nn.Module):
    """SMPL-X body model whose forward/backward run on hand-written gfx950 kernels."""

    NUM_BODY_JOINTS = NUM_BODY_JOINTS
    NUM_HAND_JOINTS = NUM_HAND_JOINTS
    NUM_JOINTS = NUM_JOINTS

    def __init__(self, model_path, gender='neutral', ext='npz', num_betas=10, num_expression_coeffs=10,
                 num_pca_comps=6, use_pca=True, flat_hand_mean=False, batch_size=1,
                 create_global_orient=True, create_body_pose=True, create_betas=True,
                 create_left_hand_pose=True, create_right_hand_pose=True, create_expression=True,
                 create_jaw_pose=True, create_leye_pose=True, create_reye_pose=True, create_transl=True,
                 v_template=None, vertex_ids=None, use_face_contour=False, dtype=torch.float32, vertex_subset=None, **kwargs):
        super().__init__()
        if dtype != torch.float32:
            raise TypeError('saga_amd.smplx computes in float32 only')
        if use_face_contour:
            raise NotImplementedError('use_face_contour=True (dynamic landmarks) is not on the GraspPose-Opt path')
        if os.path.isdir(model_path):
            model_path = os.path.join(model_path, 'SMPLX_{}.{}'.format(gender.upper(), ext))
        if not os.path.exists(model_path):
            raise FileNotFoundError('Path {} does not exist!'.format(model_path))
        if ext != 'npz':
            raise ValueError('only ext="npz" model files are supported (pickled .pkl models are not loaded)')
        data = np.load(model_path, allow_pickle=False)   # members are read lazily; only plain arrays are touched

        self.gender = gender
        self.batch_size = batch_size
        self.use_pca = use_pca
        self.num_pca_comps = num_pca_comps
        self.flat_hand_mean = flat_hand_mean

        f64 = lambda k: np.asarray(data[k], dtype=np.float64)
        vt = f64('v_template') if v_template is None else np.asarray(
            v_template.detach().cpu().numpy() if torch.is_tensor(v_template) else v_template, dtype=np.float64)
        V = vt.shape[0]
        sd = np.asarray(data['shapedirs'])
        if sd.ndim < 3:
            sd = sd[:, :, None]
        if sd.shape[-1] < SHAPE_SPACE_DIM:
            num_betas = min(num_betas, 10)
            e0 = 10
            num_expression_coeffs = min(num_expression_coeffs, 10)
        else:
            num_betas = min(num_betas, SHAPE_SPACE_DIM)
            e0 = SHAPE_SPACE_DIM
        self.num_betas = num_betas
        self.num_expression_coeffs = num_expression_coeffs
        shapedirs = np.concatenate([sd[:, :, :num_betas], sd[:, :, e0:e0 + num_expression_coeffs]], axis=-1)
        pd = np.asarray(data['posedirs'])
        posedirs = pd.reshape(-1, pd.shape[-1]).T                      # [486, V*3]
        parents = np.asarray(data['kintree_table'])[0].astype(np.int64)
        parents[0] = -1
        faces = np.asarray(data['f']).astype(np.int64)
        ids = dict(VERTEX_IDS_SMPLX)
        if vertex_ids is not None:
            ids.update(vertex_ids.get('smplx', vertex_ids) if isinstance(vertex_ids, dict) else {})
        extra = np.asarray([ids[k] for k in _EXTRA_ORDER], dtype=np.int32)
        if extra.max() >= V:
            raise ValueError('extra-joint vertex ids exceed the mesh (V=%d); pass vertex_ids=' % V)

        # host copies used to build the device model (float32, C-contiguous)
        c32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        self._host = {
            'v_template': c32(vt), 'shapedirs': c32(shapedirs), 'posedirs': c32(posedirs),
            'J_regressor': c32(np.asarray(data['J_regressor'], dtype=np.float64)), 'parents': i32(parents),
            'weights': c32(data['weights']), 'faces': i32(faces),
            'lmk_faces_idx': i32(data['lmk_faces_idx']), 'lmk_bary': c32(data['lmk_bary_coords']),
            'extra': extra,
        }
        # vertex_subset (not in smplx): evaluate the listed vertices only (ascending ids).  ``vertices`` is then [B, n, 3] in
        # list order, bit-identical to the same rows of the full mesh; ``faces`` are the faces inside the subset, renumbered;
        # ``joints`` are the 55 skeleton joints only (picked-vertex joints and landmarks need the whole mesh).
        self.vertex_subset = None
        if vertex_subset is not None:
            sub = np.unique(np.asarray(vertex_subset, dtype=np.int64))
            if sub.size == 0 or sub[0] < 0 or sub[-1] >= V:
                raise ValueError('vertex_subset ids must lie in [0, %d)' % V)
            self.vertex_subset = i32(sub)
            pos = np.full(V, -1, dtype=np.int64)
            pos[sub] = np.arange(sub.size)
            keep = (pos[faces] >= 0).all(axis=1)
            faces = pos[faces[keep]]
            self._host['lmk_faces_idx'] = i32(np.zeros(0))
            self._host['lmk_bary'] = c32(np.zeros((0, 3)))
            self._host['extra'] = extra = np.zeros(0, dtype=np.int32)
            self.num_verts_full, V, vt = V, int(sub.size), vt[sub]
        self.num_verts = V
        self.num_shape = num_betas + num_expression_coeffs
        self.num_joints_out = NUM_JOINTS + len(extra) + self._host['lmk_faces_idx'].shape[0]
        self.joints_touch_verts = (len(extra) + self._host['lmk_faces_idx'].shape[0]) > 0
        self.faces = faces                                              # ndarray [F,3] (fittingop.py:348)
        self.register_buffer('faces_tensor', torch.from_numpy(faces))
        self.register_buffer('parents', torch.from_numpy(parents))
        self.register_buffer('v_template', torch.from_numpy(c32(vt)))

        lhc = c32(np.asarray(data['hands_componentsl'])[:num_pca_comps])
        rhc = c32(np.asarray(data['hands_componentsr'])[:num_pca_comps])
        if use_pca:
            self.register_buffer('left_hand_components', torch.from_numpy(lhc))
            self.register_buffer('right_hand_components', torch.from_numpy(rhc))
        lm = np.zeros(45) if flat_hand_mean else f64('hands_meanl')
        rm = np.zeros(45) if flat_hand_mean else f64('hands_meanr')
        self.register_buffer('left_hand_mean', torch.from_numpy(c32(lm)))
        self.register_buffer('right_hand_mean', torch.from_numpy(c32(rm)))
        pose_mean = np.concatenate([np.zeros(3 + NUM_BODY_JOINTS * 3 + 9), lm, rm])
        self.register_buffer('pose_mean', torch.from_numpy(c32(pose_mean)))

        hand_n = num_pca_comps if use_pca else 3 * NUM_HAND_JOINTS
        def param(flag, name, n):
            if flag:
                self.register_parameter(name, nn.Parameter(torch.zeros(batch_size, n, dtype=torch.float32), requires_grad=True))
        param(create_global_orient, 'global_orient', 3)
        param(create_body_pose, 'body_pose', NUM_BODY_JOINTS * 3)
        param(create_betas, 'betas', num_betas)
        param(create_left_hand_pose, 'left_hand_pose', hand_n)
        param(create_right_hand_pose, 'right_hand_pose', hand_n)
        param(create_expression, 'expression', num_expression_coeffs)
        param(create_jaw_pose, 'jaw_pose', 3)
        param(create_leye_pose, 'leye_pose', 3)
        param(create_reye_pose, 'reye_pose', 3)
        param(create_transl, 'transl', 3)
        self._handles = {}

    # ---- device model handle ------------------------------------------------------------------
    def handle(self, device):
        """Opaque ``saga_model_t*`` for ``device`` (created on first use, freed with the module)."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        h = self._handles.get(idx)
        if h is None:
            lib = _lib.load()
            hp = self._host
            p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
            desc = _lib.ModelDesc(
                n_verts=hp['v_template'].shape[0], n_shape=self.num_shape, n_faces=hp['faces'].shape[0],
                n_landmarks=hp['lmk_faces_idx'].shape[0], n_extra_joints=hp['extra'].shape[0],
                v_template_host=p(hp['v_template']), shapedirs_host=p(hp['shapedirs']), posedirs_host=p(hp['posedirs']),
                J_regressor_host=p(hp['J_regressor']), parents_host=p(hp['parents']), lbs_weights_host=p(hp['weights']),
                faces_host=p(hp['faces']), lmk_faces_idx_host=p(hp['lmk_faces_idx']), lmk_bary_host=p(hp['lmk_bary']),
                extra_joint_verts_host=p(hp['extra']))
            out = ctypes.c_void_p()
            with torch.cuda.device(idx):
                if self.vertex_subset is None:
                    _lib.check(lib.saga_model_create(ctypes.byref(out), ctypes.byref(desc)), 'saga_model_create')
                else:
                    _lib.check(lib.saga_model_create_subset(ctypes.byref(out), ctypes.byref(desc), p(self.vertex_subset),
                                                            int(self.vertex_subset.shape[0])), 'saga_model_create_subset')
            h = out
            self._handles[idx] = h
        return h

    def subset_handle(self, device, vertex_ids):
        """``saga_model_t*`` of this model restricted to ``vertex_ids`` (rows of ``vertices``: ascending, unique) on
        ``device``: the model ``saga_fit_step_sparse`` runs its backward on.  Cached per device and id list, freed with the
        module."""
        ids = np.ascontiguousarray(vertex_ids, dtype=np.int32)
        if ids.ndim != 1 or ids.size == 0 or np.any(np.diff(ids) <= 0) or ids[0] < 0 or ids[-1] >= self.num_verts:
            raise ValueError('subset_handle: vertex ids must be ascending, unique and inside the mesh')
        if self.vertex_subset is not None:          # ids are rows of THIS module's vertices: a subset of the subset
            ids = np.ascontiguousarray(self.vertex_subset[ids])
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = ('subset', idx, ids.tobytes())
        h = self._handles.get(key)
        if h is None:
            hp = self._host
            p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
            desc = _lib.ModelDesc(
                n_verts=hp['v_template'].shape[0], n_shape=self.num_shape, n_faces=hp['faces'].shape[0], n_landmarks=0,
                n_extra_joints=0, v_template_host=p(hp['v_template']), shapedirs_host=p(hp['shapedirs']),
                posedirs_host=p(hp['posedirs']), J_regressor_host=p(hp['J_regressor']), parents_host=p(hp['parents']),
                lbs_weights_host=p(hp['weights']), faces_host=p(hp['faces']), lmk_faces_idx_host=None, lmk_bary_host=None,
                extra_joint_verts_host=None)
            h = ctypes.c_void_p()
            with torch.cuda.device(idx):
                _lib.check(_lib.load().saga_model_create_subset(ctypes.byref(h), ctypes.byref(desc), p(ids), int(ids.size)),
                           'saga_model_create_subset')
            self._handles[key] = h
        return h

    def _grad_subset(self, ids, device):
        """(ids as a long tensor on ``device``, subset model handle) for ``forward(vertex_grad_ids=ids)``; cached by the
        identity of ``ids`` (the entry keeps it alive), so the host copy is made once."""
        cache = self.__dict__.setdefault('_grad_subsets', {})
        key = (id(ids), str(device))
        hit = cache.get(key)
        if hit is None:
            host = np.asarray(ids.detach().cpu().numpy() if torch.is_tensor(ids) else ids, dtype=np.int64).reshape(-1)
            sel = torch.as_tensor(host, dtype=torch.long, device=device)
            if len(cache) >= 8:
                cache.clear()
            hit = cache[key] = (ids, sel, self.subset_handle(device, host))
        return hit[1], hit[2]

    def npad(self, device):
        return _lib.load().saga_model_npad(self.handle(device))

    def __del__(self):
        try:
            lib = _lib.load()
            for h in getattr(self, '_handles', {}).values():
                lib.saga_model_destroy(h)
        except Exception:
            pass

    def name(self):
        return 'SMPL-X'

    def extra_repr(self):
        return 'gender={}, V={}, betas={}, expr={}, pca={}'.format(self.gender, self.num_verts, self.num_betas,
                                                                   self.num_expression_coeffs, self.num_pca_comps)

    # ---- forward ------------------------------------------------------------------------------------
    def forward(self, betas=None, global_orient=None, body_pose=None, left_hand_pose=None, right_hand_pose=None,
                transl=None, expression=None, jaw_pose=None, leye_pose=None, reye_pose=None, return_verts=True,
                return_full_pose=False, pose2rot=True, return_shaped=True, vertex_grad_ids=None, **kwargs):
        """``vertex_grad_ids`` (not in smplx): ascending unique vertex ids (long tensor; pass the SAME tensor object every
        call -- its subset model is cached by identity).  The output then carries ``vertices_sel`` = ``vertices[:, ids]``
        and ONLY that tensor is differentiable: ``vertices`` and ``joints`` are returned without gradient.  For losses that
        read a few vertices (GraspMotion-Opt reads ~1,100 of 10,475): the backward never forms the dense [B,V,3] gradient."""
        if not pose2rot:
            raise NotImplementedError('pose2rot=False (rotation-matrix input) is not on the GraspPose-Opt path')

        def default(x, name):
            if x is not None:
                return x
            if not hasattr(self, name):
                raise ValueError('%s was neither passed nor created with the module' % name)
            return getattr(self, name)

        global_orient = default(global_orient, 'global_orient')
        body_pose = default(body_pose, 'body_pose')
        betas = default(betas, 'betas')
        left_hand_pose = default(left_hand_pose, 'left_hand_pose')
        right_hand_pose = default(right_hand_pose, 'right_hand_pose')
        jaw_pose = default(jaw_pose, 'jaw_pose')
        leye_pose = default(leye_pose, 'leye_pose')
        reye_pose = default(reye_pose, 'reye_pose')
        expression = default(expression, 'expression')
        if transl is None and hasattr(self, 'transl'):
            transl = self.transl

        if self.use_pca:
            left_hand_pose = torch.einsum('bi,ij->bj', left_hand_pose, self.left_hand_components)
            right_hand_pose = torch.einsum('bi,ij->bj', right_hand_pose, self.right_hand_components)
        B = max(betas.shape[0], global_orient.shape[0], body_pose.shape[0])
        full_pose = torch.cat([global_orient.reshape(-1, 3), body_pose.reshape(-1, NUM_BODY_JOINTS * 3),
                               jaw_pose.reshape(-1, 3), leye_pose.reshape(-1, 3), reye_pose.reshape(-1, 3),
                               left_hand_pose.reshape(-1, 45), right_hand_pose.reshape(-1, 45)], dim=1)
        full_pose = full_pose + self.pose_mean
        if betas.shape[0] != B:
            betas = betas.expand(B, -1)
        if expression.shape[0] != B:
            expression = expression.expand(B, -1)
        shape = torch.cat([betas, expression], dim=-1)

        vertices_sel = None
        if vertex_grad_ids is not None:
            sel, hs = self._grad_subset(vertex_grad_ids, full_pose.device)
            vertices, joints, vertices_sel = _lbs.lbs_selected(self, sel, hs, shape, full_pose, transl)
        else:
            vertices, joints = _lbs.lbs(self, shape, full_pose, transl)
        return SMPLXOutput(vertices=vertices if return_verts else None, joints=joints, betas=betas, vertices_sel=vertices_sel,
                           expression=expression, global_orient=global_orient, body_pose=body_pose,
                           left_hand_pose=left_hand_pose, right_hand_pose=right_hand_pose, jaw_pose=jaw_pose,
                           transl=transl, full_pose=full_pose if return_full_pose else None)


def create(model_path, model_type='smplx', **kwargs):
    """``smplx.create``: ``<model_path>/<model_type>/SMPLX_<GENDER>.npz`` -> :class:`SMPLX`."""
    if os.path.isdir(model_path):
        model_path = os.path.join(model_path, model_type)
    else:
        model_type = os.path.basename(model_path).split('_')[0].lower()
    if model_type.lower() != 'smplx':
        raise ValueError('Unknown model type {}: only "smplx" is on the GraspPose-Opt path'.format(model_type))
    return SMPLX(model_path, **kwargs)
