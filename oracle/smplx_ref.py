"""CPU restatement (torch, any float dtype) of the public ``smplx`` SMPL-X forward.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED: the ``smplx`` package is a
third-party dependency of the reference that is not in the container (``README.md:58``, no
version pinned).  The algorithm below follows SURVEY.md §3.2 step by step; the reference
call sites are ``WholeGraspPose/models/fittingop.py:30-59,149`` and ``utils/utils_body.py:93-134``.
Everything is written with differentiable torch ops so ``autograd`` supplies the gradients the
HIP backward kernels are compared against.
"""
import os

import numpy as np
import torch

NUM_BODY_JOINTS = 21
NUM_HAND_JOINTS = 15
SHAPE_SPACE_DIM = 300

# vertex ids picked as extra joints by smplx' VertexJointSelector (public constants,
# smplx/vertex_ids.py); order: 5 face, 6 feet, 5 left + 5 right finger tips.
SMPLX_EXTRA_JOINT_VERTS = {
    'nose': 9120, 'reye': 9929, 'leye': 9448, 'rear': 616, 'lear': 6,
    'LBigToe': 5770, 'LSmallToe': 5780, 'LHeel': 8846,
    'RBigToe': 8463, 'RSmallToe': 8474, 'RHeel': 8635,
    'lthumb': 5361, 'lindex': 4933, 'lmiddle': 5058, 'lring': 5169, 'lpinky': 5286,
    'rthumb': 8079, 'rindex': 7669, 'rmiddle': 7794, 'rring': 7905, 'rpinky': 8022,
}
EXTRA_JOINT_ORDER = ['nose', 'reye', 'leye', 'rear', 'lear',
                     'LBigToe', 'LSmallToe', 'LHeel', 'RBigToe', 'RSmallToe', 'RHeel',
                     'lthumb', 'lindex', 'lmiddle', 'lring', 'lpinky',
                     'rthumb', 'rindex', 'rmiddle', 'rring', 'rpinky']


def batch_rodrigues(rot_vecs):
    """[N,3] axis-angle -> [N,3,3].  angle = ||r + 1e-8|| (used directly by
    WholeGraspPose/models/objectmodel.py:38 and opt_grasppose.py:38)."""
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    axis = rot_vecs / angle
    c = torch.cos(angle)[:, None]
    s = torch.sin(angle)[:, None]
    rx, ry, rz = axis[:, 0:1], axis[:, 1:2], axis[:, 2:3]
    z = torch.zeros_like(rx)
    K = torch.cat([z, -rz, ry, rz, z, -rx, -ry, rx, z], dim=1).view(n, 3, 3)
    eye = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None]
    return eye + s * K + (1 - c) * torch.bmm(K, K)


def blend_shapes(coeffs, dirs):
    """coeffs [B,L], dirs [V,3,L] -> [B,V,3]."""
    return torch.einsum('bl,mkl->bmk', coeffs, dirs)


def vertices2joints(J_regressor, vertices):
    """[J,V] x [B,V,3] -> [B,J,3]."""
    return torch.einsum('bik,ji->bjk', vertices, J_regressor)


def batch_rigid_transform(rot_mats, joints, parents):
    """Kinematic chain.  rot_mats [B,J,3,3], joints [B,J,3] -> posed joints [B,J,3], A [B,J,4,4]."""
    B, J = joints.shape[:2]
    rel = joints.clone()
    rel[:, 1:] = joints[:, 1:] - joints[:, parents[1:]]
    M = torch.zeros(B, J, 4, 4, dtype=joints.dtype, device=joints.device)
    M[:, :, :3, :3] = rot_mats
    M[:, :, :3, 3] = rel
    M[:, :, 3, 3] = 1
    chain = [M[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[int(parents[i])], M[:, i]))
    G = torch.stack(chain, dim=1)
    posed = G[:, :, :3, 3]
    jh = torch.cat([joints, torch.zeros(B, J, 1, dtype=joints.dtype, device=joints.device)], dim=2)
    corr = torch.matmul(G, jh[..., None])            # [B,J,4,1]
    A = G - torch.nn.functional.pad(corr, [3, 0])    # subtract G[:3,:3]·J from the last column
    return posed, A


def lbs(shape_coeffs, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
    """Linear blend skinning. pose [B,J*3] axis-angle; posedirs [P,V*3]; returns verts, joints,
    and the intermediates the HIP kernels are checked against."""
    B = max(shape_coeffs.shape[0], pose.shape[0])
    V = v_template.shape[0]
    v_shaped = v_template[None] + blend_shapes(shape_coeffs, shapedirs)
    J = vertices2joints(J_regressor, v_shaped)
    R = batch_rodrigues(pose.reshape(-1, 3)).view(B, -1, 3, 3)
    eye = torch.eye(3, dtype=pose.dtype, device=pose.device)
    pose_feat = (R[:, 1:] - eye).reshape(B, -1)
    v_posed = v_shaped + torch.matmul(pose_feat, posedirs).view(B, V, 3)
    J_tf, A = batch_rigid_transform(R, J, parents)
    nj = J_regressor.shape[0]
    T = torch.matmul(lbs_weights[None].expand(B, -1, -1), A.view(B, nj, 16)).view(B, V, 4, 4)
    vh = torch.cat([v_posed, torch.ones(B, V, 1, dtype=pose.dtype, device=pose.device)], dim=2)
    verts = torch.matmul(T, vh[..., None])[:, :, :3, 0]
    aux = {'v_shaped': v_shaped, 'J': J, 'R': R, 'pose_feat': pose_feat, 'v_posed': v_posed, 'A': A}
    return verts, J_tf, aux


def vertices2landmarks(vertices, faces, lmk_faces_idx, lmk_bary_coords):
    """Barycentric landmarks: vertices [B,V,3], faces [F,3], idx [L], bary [L,3] -> [B,L,3]."""
    tri = faces[lmk_faces_idx]                       # [L,3]
    corners = vertices[:, tri]                       # [B,L,3,3]
    return torch.einsum('blfi,lf->bli', corners, lmk_bary_coords)


class SMPLXRef:
    """Restatement of ``smplx.create(..., model_type='smplx')`` + ``SMPLX.forward``.

    ``data`` is the dict of ``.npz`` arrays (or a path to the file).  Defaults follow the
    reference call (``fittingop.py:30-44``): ``use_pca=True``, ``flat_hand_mean=False``,
    ``num_betas=10``, ``num_expression_coeffs=10``, ``use_face_contour=False``.
    """

    def __init__(self, data, num_pca_comps=24, num_betas=10, num_expression_coeffs=10,
                 batch_size=1, dtype=torch.float64, vertex_ids=None, use_pca=True,
                 flat_hand_mean=False):
        if isinstance(data, (str, os.PathLike)):
            data = dict(np.load(data, allow_pickle=False))
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64)).to(dtype)
        self.dtype = dtype
        self.batch_size = batch_size
        self.use_pca = use_pca
        self.num_pca_comps = num_pca_comps
        self.v_template = t(data['v_template'])
        V = self.v_template.shape[0]
        sd = np.asarray(data['shapedirs'])
        if sd.shape[-1] < SHAPE_SPACE_DIM:
            num_betas = min(num_betas, 10)
            e0, e1 = 10, 10 + min(num_expression_coeffs, 10)
        else:
            num_betas = min(num_betas, SHAPE_SPACE_DIM)
            e0, e1 = SHAPE_SPACE_DIM, SHAPE_SPACE_DIM + num_expression_coeffs
        self.num_betas, self.num_expression_coeffs = num_betas, e1 - e0
        self.shapedirs = t(np.concatenate([sd[:, :, :num_betas], sd[:, :, e0:e1]], axis=-1))
        pd = np.asarray(data['posedirs'])
        self.posedirs = t(pd.reshape(-1, pd.shape[-1]).T)       # [P, V*3]
        self.J_regressor = t(data['J_regressor'])
        parents = np.asarray(data['kintree_table'])[0].astype(np.int64)
        parents[0] = -1
        self.parents = parents
        self.lbs_weights = t(data['weights'])
        self.faces = np.asarray(data['f']).astype(np.int64)
        self.faces_tensor = torch.as_tensor(self.faces)
        self.lmk_faces_idx = torch.as_tensor(np.asarray(data['lmk_faces_idx']).astype(np.int64))
        self.lmk_bary_coords = t(data['lmk_bary_coords'])
        self.left_hand_components = t(np.asarray(data['hands_componentsl'])[:num_pca_comps])
        self.right_hand_components = t(np.asarray(data['hands_componentsr'])[:num_pca_comps])
        lm = np.zeros(45) if flat_hand_mean else np.asarray(data['hands_meanl'], dtype=np.float64)
        rm = np.zeros(45) if flat_hand_mean else np.asarray(data['hands_meanr'], dtype=np.float64)
        self.pose_mean = t(np.concatenate([np.zeros(3), np.zeros(NUM_BODY_JOINTS * 3), np.zeros(9), lm, rm]))
        ids = dict(SMPLX_EXTRA_JOINT_VERTS)
        if vertex_ids is not None:
            ids.update(vertex_ids)
        self.extra_joint_verts = torch.as_tensor([ids[k] for k in EXTRA_JOINT_ORDER], dtype=torch.long)
        assert int(self.extra_joint_verts.max()) < V, 'extra-joint vertex ids exceed the mesh'

    def _default(self, n):
        return torch.zeros(self.batch_size, n, dtype=self.dtype)

    def full_pose(self, global_orient, body_pose, jaw_pose, leye_pose, reye_pose, left_hand_pose, right_hand_pose):
        if self.use_pca:
            left_hand_pose = torch.einsum('bi,ij->bj', left_hand_pose, self.left_hand_components)
            right_hand_pose = torch.einsum('bi,ij->bj', right_hand_pose, self.right_hand_components)
        B = body_pose.shape[0]
        fp = torch.cat([global_orient.reshape(B, 3), body_pose.reshape(B, NUM_BODY_JOINTS * 3),
                        jaw_pose.reshape(B, 3), leye_pose.reshape(B, 3), reye_pose.reshape(B, 3),
                        left_hand_pose.reshape(B, 45), right_hand_pose.reshape(B, 45)], dim=1)
        return fp + self.pose_mean, left_hand_pose, right_hand_pose

    def forward(self, betas=None, global_orient=None, body_pose=None, left_hand_pose=None,
                right_hand_pose=None, transl=None, expression=None, jaw_pose=None,
                leye_pose=None, reye_pose=None, return_aux=False):
        hand_n = self.num_pca_comps if self.use_pca else 45
        global_orient = self._default(3) if global_orient is None else global_orient
        body_pose = self._default(NUM_BODY_JOINTS * 3) if body_pose is None else body_pose
        betas = self._default(self.num_betas) if betas is None else betas
        expression = self._default(self.num_expression_coeffs) if expression is None else expression
        left_hand_pose = self._default(hand_n) if left_hand_pose is None else left_hand_pose
        right_hand_pose = self._default(hand_n) if right_hand_pose is None else right_hand_pose
        jaw_pose = self._default(3) if jaw_pose is None else jaw_pose
        leye_pose = self._default(3) if leye_pose is None else leye_pose
        reye_pose = self._default(3) if reye_pose is None else reye_pose
        transl = self._default(3) if transl is None else transl

        full_pose, lh, rh = self.full_pose(global_orient, body_pose, jaw_pose, leye_pose, reye_pose,
                                           left_hand_pose, right_hand_pose)
        B = max(betas.shape[0], global_orient.shape[0], body_pose.shape[0])
        if betas.shape[0] != B:
            betas = betas.expand(B, -1)
        shape = torch.cat([betas, expression.expand(B, -1) if expression.shape[0] != B else expression], dim=-1)
        verts, joints, aux = lbs(shape, full_pose, self.v_template, self.shapedirs, self.posedirs,
                                 self.J_regressor, self.parents, self.lbs_weights)
        landmarks = vertices2landmarks(verts, self.faces_tensor, self.lmk_faces_idx, self.lmk_bary_coords)
        joints = torch.cat([joints, verts[:, self.extra_joint_verts], landmarks], dim=1)
        joints = joints + transl[:, None]
        verts = verts + transl[:, None]
        out = {'vertices': verts, 'joints': joints, 'full_pose': full_pose, 'betas': betas,
               'expression': expression, 'global_orient': global_orient, 'body_pose': body_pose,
               'left_hand_pose': lh, 'right_hand_pose': rh, 'jaw_pose': jaw_pose,
               'v_shaped': aux['v_shaped']}
        if return_aux:
            out['aux'] = aux
        return out

    __call__ = forward
