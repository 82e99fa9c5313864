"""Batched GraspPose-Opt: the counterpart of ``WholeGraspPose/models/fittingop.py`` (``FittingOP``)
and of the per-sample loop in ``opt_grasppose.py:164-243``.

The reference fits ONE candidate at a time (``batch_size=1``, serial Python loops
``opt_grasppose.py:211,220``).  Here B candidates are fitted at once as B independent problems:
every mean in the loss is per candidate, the objective handed to ``backward`` is the sum over
candidates, and Adam is element-wise -- so each candidate follows the update sequence it would
follow alone.  The per-iteration arithmetic (SMPL-X forward/backward, vertex normals, the two
bidirectional nearest-neighbour searches) runs in ``libsaga_hip.so``; nothing is copied to the host
inside the loop (the reference crosses device->host->device for normals, ``fittingop.py:158-164``, and
pulls 14 scalars per iteration, ``:225-242``).
"""
import ctypes
import os

import numpy as np
import torch
import torch.optim as optim

from . import _lib
from . import smplx as _smplx
from .geometry import point2point_signed, vertex_normals_subset
from .grasp_loss import GraspTables, grasp_loss_terms
from .rotations import cont6d_to_aa
from .vposer import VPoserDecoder

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'grasp_indices.npz')

LOSS_KEYS = ['total', 'rec', 'body_rec', 'marker contact', 'object contact', 'prior contact',
             'hand collision', 'object collision', 'foot', 'reg']
COUNT_KEYS = ['hand colli', 'obj colli', 'contact', 'hand markers colli']
PARAM_KEYS = ['transl', 'global_orient', 'betas', 'body_pose', 'left_hand_pose', 'right_hand_pose', 'leye_pose', 'reye_pose']


def load_grasp_indices(path=None):
    """markers_143 / rhand / foot vertex ids and rhand_weight (``fittingop.py:86-124``), plus the 79 'f0_p5' markers
    GraspMotion-Opt fits (``opt_graspmotion.py:408``).

    The packaged ``data/grasp_indices.npz`` was derived from the reference's ``body_utils`` files with
    the reference's own expressions (``tests/golden/make_golden.py``), including the CPython-``set``
    ordering that selects the 172 foot vertices."""
    d = np.load(path or _DATA, allow_pickle=False)
    return {k: d[k] for k in ('markers_143', 'markers_f0_p5', 'rhand', 'lhand', 'foot', 'rhand_weight', 'foot_sizes')}


def create_decoder(vposer, bm, device):
    """``saga_decoder_t*`` (VPoser decoder weights + the hand PCA basis / pose mean of ``bm``) on ``device``; the caller frees it
    with ``saga_decoder_destroy``."""
    for attr in ('bodyprior_dec_fc1', 'bodyprior_dec_fc2', 'bodyprior_dec_out'):
        if not hasattr(vposer, attr):
            raise RuntimeError('native fitting needs a VPoser decoder with bodyprior_dec_fc1/fc2/out layers')
    h = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
    arrs = [h(vposer.bodyprior_dec_fc1.weight), h(vposer.bodyprior_dec_fc1.bias), h(vposer.bodyprior_dec_fc2.weight),
            h(vposer.bodyprior_dec_fc2.bias), h(vposer.bodyprior_dec_out.weight), h(vposer.bodyprior_dec_out.bias),
            h(bm.left_hand_components), h(bm.right_hand_components), h(bm.pose_mean)]
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    desc = _lib.DecoderDesc(arrs[0].shape[1], arrs[0].shape[0], arrs[4].shape[0] // 6, arrs[6].shape[0], *[p(a) for a in arrs])
    out = ctypes.c_void_p()
    with torch.cuda.device(device):
        _lib.check(_lib.load().saga_decoder_create(ctypes.byref(out), ctypes.byref(desc)), 'saga_decoder_create')
    return out


class _ParamMaps(torch.autograd.Function):
    """Packed variable rows [B,112] -> (full_pose [B,165], shape [B,n_shape], transl [B,3]) on ``saga_params_fwd/bwd``."""

    @staticmethod
    def forward(ctx, packed, decoder, n_shape):
        _lib.require_device(packed)
        pk = packed.detach().contiguous()
        B, dev = pk.shape[0], pk.device
        f32 = dict(device=dev, dtype=torch.float32)
        full_pose, shape, transl = torch.empty(B, 165, **f32), torch.empty(B, n_shape, **f32), torch.empty(B, 3, **f32)
        ws = torch.empty(_lib.load().saga_params_workspace_floats(decoder, B), **f32)
        _lib.call('saga_params_fwd', dev, decoder, _lib.ptr(pk), _lib.ptr(full_pose), _lib.ptr(shape), _lib.ptr(transl), n_shape,
                  _lib.ptr(ws), B)
        ctx.decoder, ctx.n_shape = decoder, n_shape
        ctx.save_for_backward(pk, ws)
        ctx.set_materialize_grads(False)
        return full_pose, shape, transl

    @staticmethod
    def backward(ctx, g_pose, g_shape, g_transl):
        pk, ws = ctx.saved_tensors
        if g_pose is None and g_shape is None and g_transl is None:
            return None, None, None
        c = lambda g: None if g is None else g.contiguous()
        g_pose = torch.zeros(pk.shape[0], 165, device=pk.device) if g_pose is None else g_pose.contiguous()
        grad = torch.empty_like(pk)
        _lib.call('saga_params_bwd', pk.device, ctx.decoder, _lib.ptr(pk), _lib.ptr(g_pose), _lib.ptr(c(g_shape)), _lib.ptr(c(g_transl)),
                  _lib.ptr(grad), ctx.n_shape, _lib.ptr(ws), pk.shape[0])
        return grad, None, None


def param_maps(packed, decoder, n_shape=20):
    """The optimiser-variable maps of the fitting ops as one operator: packed rows [B, ``_lib.VAR_STRIDE``] (``_lib.VAR_SLICES``:
    VPoser latent 32 | hand PCA 48 | eyes 6 | betas 10 | transl 3 | 6D global orientation 6) -> (full_pose [B,165] with the pose
    mean, shape [B,n_shape], transl [B,3]); ``decoder`` from :func:`create_decoder`.  Differentiable w.r.t. ``packed``."""
    return _ParamMaps.apply(packed, decoder, int(n_shape))


def loss_vertex_subset(faces, idx):
    """Ascending ids of the vertices one fitting iteration reads: the 143 markers, the right hand, the feet, and every corner
    of a face that touches a right-hand vertex (the hand's vertex normals sum over its incident faces)."""
    faces = np.asarray(faces)
    hand = np.zeros(int(faces.max()) + 1 if faces.size else 0, dtype=bool)
    rh = np.asarray(idx['rhand'])
    hand[rh[rh < hand.shape[0]]] = True
    ring = faces[hand[faces].any(axis=1)].reshape(-1) if faces.size else np.zeros(0, dtype=np.int64)
    return np.unique(np.concatenate([np.asarray(idx['markers_143']), rh, np.asarray(idx['foot']), ring]).astype(np.int64))


class FittingOP:
    """Same constructor contract as the reference (``FittingOP(fittingconfig)``, keys as at
    ``opt_grasppose.py:179-191``) plus optional keys: ``body_model_path`` (default
    ``./body_utils/body_models``), ``vposer`` (a module with ``decode(z, output_type='aa')``; default: a
    random-initialised :class:`VPoserDecoder`), ``indices`` (dict from :func:`load_grasp_indices`),
    ``smplx_kwargs`` (extra ``smplx.create`` kwargs, e.g. ``vertex_ids`` for reduced test meshes),
    ``genders`` (which models to create; default both like the reference), ``native`` (default True: ``fitting``
    drives ``saga_fit_step`` -- the whole iteration incl. Adam as one kernel sequence; False: PyTorch autograd over
    the drop-in operators + ``torch.optim.Adam``, the form the reference has), ``record_history``, ``sparse_backward``
    (native path only, default False: ``saga_fit_step_sparse`` -- the backward runs over the ~1,050 vertices the loss reads
    instead of multiplying the structural zeros of the dense [B,V,3] vertex gradient through the whole mesh; the forward
    is unchanged and the gradients agree to rounding), ``log_gated_terms`` (native path only, default True = the reference:
    stages 1 and 2 evaluate and log the contact / collision terms although their weights are zero there; False: those
    stages skip the nearest-neighbour search and the hand normals and evaluate the body model on the 315 marker / foot
    vertices only -- nothing else is read there -- and ``save_loss`` holds 0 for the gated keys)."""

    def __init__(self, fittingconfig):
        cfg = dict(init_lr_h=0.008, num_iter=[300, 400, 500], batch_size=1, num_markers=143, device='cuda',
                   verbose=False, hand_ncomps=24, only_rec=False, contact_loss='contact', logger=print,
                   data_type='markers_143', body_model_path='./body_utils/body_models', vposer=None, indices=None,
                   smplx_kwargs={}, genders=('male', 'female'), fused=True, native=True,
                   record_history=True, loss_vertices_only=False, sparse_backward=False, log_gated_terms=True)
        cfg.update(fittingconfig)
        for key, val in cfg.items():
            setattr(self, key, val)
        self.device = torch.device(self.device)
        if self.device.type != 'cuda':
            raise RuntimeError('saga_amd.FittingOP runs on a ROCm GPU only (device=%s)' % self.device)
        B = self.batch_size
        idx = dict(self.indices if self.indices is not None else load_grasp_indices())
        subset = [None]

        def make(gender):
            kw = dict(model_type='smplx', gender=gender, ext='npz', num_pca_comps=self.hand_ncomps, create_global_orient=True,
                      create_body_pose=True, create_betas=True, create_left_hand_pose=True, create_right_hand_pose=True,
                      create_expression=True, create_jaw_pose=True, create_leye_pose=True, create_reye_pose=True,
                      create_transl=True, batch_size=B, **self.smplx_kwargs)
            if self.loss_vertices_only:
                # the full module is only read for its faces (no device model is created for it)
                sub = loss_vertex_subset(_smplx.create(self.body_model_path, **kw).faces, idx)
                if subset[0] is not None and not np.array_equal(subset[0], sub):
                    raise ValueError('the body models of the two genders have different faces around the loss vertices')
                subset[0] = sub
                kw['vertex_subset'] = sub
            return _smplx.create(self.body_model_path, **kw).to(self.device).eval()

        self.bm_male = make('male') if 'male' in self.genders else None
        self.bm_female = make('female') if 'female' in self.genders else None
        self.bm = None
        if self.vposer is None:
            self.vposer = VPoserDecoder(seed=0)
        self.vposer = self.vposer.to(self.device).eval()
        for p in self.vposer.parameters():
            p.requires_grad_(False)
        self.fittingconfig = fittingconfig

        # optimisation variables (fittingop.py:71-76): one packed row per candidate (the layout saga_fit_step works on);
        # the named tensors are leaf views of it, so the autograd path and torch.optim see the same storage
        self.vars = torch.zeros(B, _lib.VAR_STRIDE, device=self.device)
        if 2 * self.hand_ncomps != 48:
            self.native = False          # the packed row holds 2 x 24 hand PCA coefficients
            self.hand_pose = torch.zeros(B, 2 * self.hand_ncomps, device=self.device, requires_grad=True)
        for name, (a, b) in _lib.VAR_SLICES.items():
            if name == 'hand_pose' and not self.native:
                continue
            setattr(self, name, self.vars[:, a:b].detach().requires_grad_(True))
        with torch.no_grad():
            self.glo_rot_rec.copy_(torch.tensor([[-1., 0, 0, 0, 0, 1]], device=self.device).expand(B, 6))
        self._make_optimizers()
        self._decoder = None
        self._ws = None

        if subset[0] is not None:          # vertex ids -> positions in the subset the model evaluates
            for k in ('markers_143', 'rhand', 'foot'):
                idx[k] = np.searchsorted(subset[0], np.asarray(idx[k]))
        self.vertex_subset = subset[0]
        dev = self.device
        self.v_weights = torch.as_tensor(np.asarray(idx['rhand_weight']), dtype=torch.float32, device=dev)
        self.v_weights2 = torch.pow(self.v_weights, 1.0 / 2.5)
        self.markers_143 = torch.as_tensor(np.asarray(idx['markers_143']), dtype=torch.long, device=dev)
        self.rhand_verts = torch.as_tensor(np.asarray(idx['rhand']), dtype=torch.long, device=dev)
        self.foot_markers_all = torch.as_tensor(np.asarray(idx['foot']), dtype=torch.long, device=dev)
        self.rhand_verts_i32 = self.rhand_verts.to(torch.int32)
        self.tables = GraspTables(idx['markers_143'], idx['rhand'], idx['foot'], idx['rhand_weight'], dev)
        self._idx_np = {k: np.asarray(idx[k]) for k in ('markers_143', 'foot')}
        self._lite = None

    def _make_optimizers(self):
        lr = self.init_lr_h
        self.optimizer_s1 = optim.Adam([self.transl_rec, self.glo_rot_rec], lr=lr * 2.0)
        self.optimizer_s2 = optim.Adam([self.betas, self.transl_rec, self.glo_rot_rec, self.vpose_rec], lr=lr * 1.5)
        self.optimizer_s3 = optim.Adam([self.vpose_rec, self.hand_pose, self.eye_pose], lr=lr)
        self.optimizers = [self.optimizer_s1, self.optimizer_s2, self.optimizer_s3]
        # state of the same three optimisers for the native step: exp_avg / exp_avg_sq rows + step counts
        G = _lib.VAR_GROUPS
        self._native_opt = [
            dict(lr=lr * 2.0, mask=G['transl_rec'] | G['glo_rot_rec']),
            dict(lr=lr * 1.5, mask=G['betas'] | G['transl_rec'] | G['glo_rot_rec'] | G['vpose_rec']),
            dict(lr=lr, mask=G['vpose_rec'] | G['hand_pose'] | G['eye_pose']),
        ]
        for o in self._native_opt:
            o['m'] = torch.zeros_like(self.vars)
            o['v'] = torch.zeros_like(self.vars)
            o['step'] = 0

    def variables(self):
        return {'betas': self.betas, 'transl_rec': self.transl_rec, 'glo_rot_rec': self.glo_rot_rec,
                'vpose_rec': self.vpose_rec, 'hand_pose': self.hand_pose, 'eye_pose': self.eye_pose}

    def set_variables(self, values):
        with torch.no_grad():
            for k, v in values.items():
                getattr(self, k).copy_(torch.as_tensor(v, dtype=torch.float32, device=self.device))

    def init_betas(self, betas):
        with torch.no_grad():
            self.betas.copy_(torch.as_tensor(betas, dtype=torch.float32, device=self.device).reshape(1, -1).expand_as(self.betas))

    def reset(self):
        """fittingop.py:129-135.  DELIBERATE difference: the reference builds its three Adam optimisers once
        (:77-84) and ``reset()`` only re-initialises the variables, so the exp_avg / exp_avg_sq moments and step counts
        of sample k leak into sample k+1 (an artefact of looping over samples at batch size 1).  Here the
        candidates of a batch are independent problems and every batch starts from fresh optimisers -- what the
        reference's FIRST sample sees, and what makes a candidate's result independent of the batch it is in."""
        with torch.no_grad():
            for t in (self.betas, self.transl_rec, self.vpose_rec, self.hand_pose, self.eye_pose):
                t.zero_()
            self.glo_rot_rec.copy_(torch.tensor([[-1., 0, 0, 0, 0, 1]], device=self.device).expand_as(self.glo_rot_rec))
        self._make_optimizers()

    # ------------------------------------------------------------------------------------------------
    def body_params(self):
        B, n = self.batch_size, self.hand_ncomps
        return {
            'transl': self.transl_rec,
            'global_orient': cont6d_to_aa(self.glo_rot_rec),
            'betas': self.betas,
            'body_pose': self.vposer.decode(self.vpose_rec, output_type='aa').view(B, -1),
            'left_hand_pose': self.hand_pose[:, :n],
            'right_hand_pose': self.hand_pose[:, n:],
            'leye_pose': self.eye_pose[:, :3],
            'reye_pose': self.eye_pose[:, 3:],
        }

    def calc_loss_contact_map(self, body_markers, verts_object, normal_object, contacts_object, contacts_markers,
                              gender, betas, stage, alpha):
        """fittingop.py:137-244 for B candidates.  ``loss`` and every ``loss_dict`` / ``vertices_info``
        entry is a [B] device tensor (per-candidate values); call ``loss.sum().backward()``.
        Default: the fused HIP loss (``saga_grasp_loss_fwd/bwd``); ``fused=False`` in the config composes the
        same loss from the drop-in operators the way the reference does (used to cross-check the fused path)."""
        if not self.fused:
            return self.calc_loss_contact_map_composed(body_markers, verts_object, normal_object, contacts_object,
                                                       contacts_markers, gender, betas, stage, alpha)
        B = self.batch_size
        body_param = self.body_params()
        output = self.bm(return_verts=True, **body_param)
        r = grasp_loss_terms(output.vertices, self.bm, self.tables, verts_object.float(), normal_object.float(),
                             body_markers, contacts_markers, contacts_object)
        t = r['terms']
        loss_rec, loss_body_rec, loss_foot = t[:, 0], t[:, 1], t[:, 2]
        loss_marker_contact, loss_object_contact, loss_prior_contact = t[:, 3], t[:, 4], t[:, 5]
        h_collision, loss_dist_o = t[:, 6], t[:, 7]
        loss_vpose_reg = 0.0001 * (self.vpose_rec ** 2).mean(dim=1)
        loss_hand_pose_reg = 0.0005 * (self.hand_pose ** 2).mean(dim=1)
        loss_eye_pose_reg = 0.0001 * (self.eye_pose ** 2).mean(dim=1)
        on = 0.0 if self.only_rec else 1.0
        loss = (1 * loss_rec
                + on * (self.contact_loss == 'contact') * (stage > 1) * 15 * (loss_marker_contact + loss_object_contact)
                + on * (self.contact_loss == 'prior') * (stage > 1) * 15 * loss_prior_contact
                + on * (stage > 1) * 10 * (h_collision + loss_dist_o)
                + (stage > 1) * loss_hand_pose_reg
                + (stage > 1) * loss_eye_pose_reg
                + (stage > 0) * loss_vpose_reg
                + 0.1 * loss_foot)
        d = lambda x: x.detach()
        loss_dict = {
            'total': d(loss), 'rec': d(loss_rec), 'body_rec': d(loss_body_rec),
            'marker contact': d(loss_marker_contact), 'object contact': d(loss_object_contact),
            'prior contact': d(loss_prior_contact), 'hand collision': d(h_collision),
            'object collision': d(loss_dist_o), 'foot': d(loss_foot),
            'reg': d(loss_vpose_reg + loss_hand_pose_reg + loss_eye_pose_reg),
        }
        td = d(t)
        vertices_info = {'hand colli': td[:, 8].long(), 'obj colli': td[:, 9].long(), 'contact': td[:, 10].long(),
                         'hand markers colli': td[:, 11].long()}
        return (loss, loss_dict, r['markers_rec'], body_param, vertices_info, r['hand_rec'], r['hand_normals'],
                r['h2o_signed'], r['o2h_signed'])

    def calc_loss_contact_map_composed(self, body_markers, verts_object, normal_object, contacts_object, contacts_markers,
                                       gender, betas, stage, alpha):
        """The same loss written the way the reference writes it, on the drop-in operators
        (smplx forward, vertex normals, point2point_signed) + torch element-wise ops."""
        B = self.batch_size
        body_param = self.body_params()
        output = self.bm(return_verts=True, **body_param)
        verts_full = output.vertices
        # index_select (backward = index_add) instead of advanced indexing (backward = sorted index_put)
        body_markers_rec = verts_full.index_select(1, self.markers_143)
        foot_markers_rec = verts_full.index_select(1, self.foot_markers_all)
        rhand_verts_rec = verts_full.index_select(1, self.rhand_verts)
        # vertex normals of the right hand, on the device, for every candidate (:158-164)
        rh_normals = vertex_normals_subset(self.bm, verts_full, self.rhand_verts_i32)

        tgt = body_markers.detach()
        loss_rec = (body_markers_rec - tgt).abs().mean(dim=(1, 2))
        loss_body_rec = (body_markers_rec[:, :49] - tgt[:, :49]).abs().mean(dim=(1, 2))
        loss_foot = 0.1 * foot_markers_rec[:, :, -1].abs().mean(dim=1)
        loss_vpose_reg = 0.0001 * (self.vpose_rec ** 2).mean(dim=1)
        loss_hand_pose_reg = 0.0005 * (self.hand_pose ** 2).mean(dim=1)
        loss_eye_pose_reg = 0.0001 * (self.eye_pose ** 2).mean(dim=1)

        vo, no = verts_object.float(), normal_object.float()
        # :183 passes normal_object positionally as x_normals (kept: the signed output is unused)
        o2h_marker, h2o_signed_marker, o2h_idx_marker, _ = point2point_signed(body_markers_rec, vo, no)
        o2h_signed, h2o_signed, o2h_idx, _ = point2point_signed(rhand_verts_rec, vo, rh_normals, no)

        loss_marker_contact = (h2o_signed_marker.abs() * contacts_markers.view(B, -1)).mean(dim=1)
        loss_object_contact = (o2h_signed.abs() * contacts_object.view(B, -1)).mean(dim=1)
        zero_h = torch.zeros_like(h2o_signed)
        v_dist = (h2o_signed < 0.02) & (h2o_signed > 0) & (self.v_weights2[None] > 0.7)
        v_dist_neg = h2o_signed < 0
        v_dist_marker_neg = h2o_signed_marker < 0
        v_contact = torch.where(v_dist, self.v_weights[None].expand_as(h2o_signed), zero_h)
        v_collision = torch.where(v_dist_neg, torch.full_like(h2o_signed, 10.0), zero_h)
        w_dist_neg = o2h_signed < 0
        w = torch.where(w_dist_neg, torch.full_like(o2h_signed, 20.0), torch.zeros_like(o2h_signed))
        loss_prior_contact = (h2o_signed.abs() * v_contact).mean(dim=1)
        h_collision = (h2o_signed.abs() * v_collision).mean(dim=1)
        loss_dist_o = (o2h_signed.abs() * w).mean(dim=1)

        on = 0.0 if self.only_rec else 1.0
        loss = (1 * loss_rec
                + on * (self.contact_loss == 'contact') * (stage > 1) * 15 * (loss_marker_contact + loss_object_contact)
                + on * (self.contact_loss == 'prior') * (stage > 1) * 15 * loss_prior_contact
                + on * (stage > 1) * 10 * (h_collision + loss_dist_o)
                + (stage > 1) * loss_hand_pose_reg
                + (stage > 1) * loss_eye_pose_reg
                + (stage > 0) * loss_vpose_reg
                + 0.1 * loss_foot)

        d = lambda t: t.detach()
        loss_dict = {
            'total': d(loss), 'rec': d(loss_rec), 'body_rec': d(loss_body_rec),
            'marker contact': d(loss_marker_contact), 'object contact': d(loss_object_contact),
            'prior contact': d(loss_prior_contact), 'hand collision': d(h_collision),
            'object collision': d(loss_dist_o), 'foot': d(loss_foot),
            'reg': d(loss_vpose_reg + loss_hand_pose_reg + loss_eye_pose_reg),
        }
        vertices_info = {
            'hand colli': v_dist_neg.sum(dim=1), 'obj colli': w_dist_neg.sum(dim=1),
            'contact': ((h2o_signed < 0.001) & (h2o_signed > -0.001)).sum(dim=1),
            'hand markers colli': v_dist_marker_neg.sum(dim=1),
        }
        return loss, loss_dict, body_markers_rec, body_param, vertices_info, rhand_verts_rec, rh_normals, h2o_signed, o2h_signed

    # ---- native iteration (saga_fit_step) --------------------------------------------------------------
    def _native_handles(self):
        """Decoder constants (VPoser weights + hand PCA of the current body model) and the workspace, created once."""
        key = id(self.bm)
        if self._decoder is None or self._decoder[0] != key:
            out = create_decoder(self.vposer, self.bm, self.device)
            if self._decoder is not None:
                _lib.load().saga_decoder_destroy(self._decoder[1])
            self._decoder = (key, out)
        return self._decoder[1]

    def close(self):
        """Release the decoder constants held by the library (also done when the object is collected)."""
        dec, self._decoder = getattr(self, '_decoder', None), None
        if dec is not None:
            try:
                _lib.load().saga_decoder_destroy(dec[1])
            except Exception:      # interpreter shutdown: the library may already be gone
                pass

    def __del__(self):
        self.close()

    def _tstruct(self):
        """The loss tables with the mesh-dependent part (the hand's face ring) of the current body model: lets ``saga_fit_step``
        write the vertices the loss reads as compact rows in its skinning kernel and take the hand normals from LDS."""
        return self.tables.struct_for_mesh(self.bm.faces, self.bm.num_verts)

    def _workspace(self, n_obj):
        n = _lib.load().saga_fit_workspace_floats(self.bm.handle(self.device), ctypes.byref(self._tstruct()), self.batch_size, n_obj)
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.zeros(n, device=self.device, dtype=torch.float32)      # contract: zero-initialised
        return self._ws

    def make_problem(self, body_markers, verts_object, normal_object, contacts_object, contacts_markers):
        """Device-resident fitting targets of a batch (kept alive by the returned object)."""
        B = self.batch_size
        c = lambda t, shape: t.detach().to(self.device, torch.float32).reshape(shape).contiguous()
        O = verts_object.shape[1]
        t = dict(obj=c(verts_object, (B, O, 3)), obj_normals=c(normal_object, (B, O, 3)),
                 markers=c(body_markers, (B, self.tables.n_markers, 3)), c_markers=c(contacts_markers, (B, self.tables.n_markers)),
                 c_object=c(contacts_object, (B, O)))
        P = lambda x: ctypes.c_void_p(x.data_ptr())
        t['struct'] = _lib.FitProblem(P(t['obj']), P(t['obj_normals']), P(t['markers']), P(t['c_markers']), P(t['c_object']), O)
        t['loss'] = torch.empty(B, device=self.device)
        t['markers_rec'] = torch.empty(B, self.tables.n_markers, 3, device=self.device)
        return t

    def _lite_handles(self, n_obj):
        """(model, tables, workspace) of the stages whose contact / collision weights are all zero when ``log_gated_terms`` is
        off: nothing reads a vertex other than the 143 markers and the 172 foot vertices there, so those stages run the whole
        iteration on the subset model over them (vertices bit-identical to the full mesh's rows; ``saga_model_create_subset``)."""
        if self._lite is None or self._lite[0] is not self.bm:
            mk, ft = self._idx_np['markers_143'], self._idx_np['foot']
            ids = np.unique(np.concatenate([mk, ft]).astype(np.int64))
            pos = lambda a: np.searchsorted(ids, a)
            # the hand tables are not read in these stages; one valid entry keeps the table checks happy
            tables = GraspTables(pos(mk), pos(mk[:1]), pos(ft), np.ones(1, dtype=np.float32), self.device)
            self._lite = [self.bm, self.bm.subset_handle(self.device, ids), tables, None]
        _, model, tables, ws = self._lite
        n = _lib.load().saga_fit_workspace_floats(model, ctypes.byref(tables.struct), self.batch_size, n_obj)
        if ws is None or ws.numel() < n:
            ws = self._lite[3] = torch.zeros(n, device=self.device, dtype=torch.float32)
        return model, tables, ws

    def native_step(self, ss, problem, loss_terms=None):
        """One iteration of stage ``ss`` (loss, backward, Adam update of that stage's variables) in ``saga_fit_step``.
        -> (loss [B], markers_rec [B,143,3]) of BEFORE the update, like the reference's loop body."""
        if self.bm is None:
            raise RuntimeError('select a body model first (self.bm = self.bm_male / bm_female)')
        o = self._native_opt[ss]
        o['step'] += 1
        cfg = _lib.FitStepCfg(ss, int(bool(self.only_rec)), int(self.contact_loss == 'prior'), o['mask'], o['step'],
                              o['lr'], 0.9, 0.999, 1e-8, int(not self.log_gated_terms))
        P = _lib.ptr
        if not self.log_gated_terms and (ss < 2 or self.only_rec) and self.vertex_subset is None:
            model, tables, ws = self._lite_handles(problem['struct'].n_obj)
            _lib.call('saga_fit_step', self.device,
                      model, self._native_handles(), ctypes.byref(tables.struct), ctypes.byref(problem['struct']), ctypes.byref(cfg),
                      P(self.vars), P(o['m']), P(o['v']), P(ws), P(problem['loss']), P(loss_terms), P(problem['markers_rec']),
                      self.batch_size)
            return problem['loss'], problem['markers_rec']
        tstruct = self._tstruct()
        rest = (self._native_handles(), ctypes.byref(tstruct), ctypes.byref(problem['struct']),
                ctypes.byref(cfg), P(self.vars), P(o['m']), P(o['v']), P(self._workspace(problem['struct'].n_obj)), P(problem['loss']),
                P(loss_terms), P(problem['markers_rec']), self.batch_size)
        if self.sparse_backward:
            cached = getattr(self, '_m_loss', None)
            if cached is None or cached[0] is not self.bm:       # subset model of the current body model, looked up once
                cached = self._m_loss = (self.bm, self.bm.subset_handle(self.device, self.tables.unique_ids_host))
            m_loss = cached[1]
            _lib.call('saga_fit_step_sparse', self.device, self.bm.handle(self.device), m_loss, *rest)
        else:
            _lib.call('saga_fit_step', self.device, self.bm.handle(self.device), *rest)
        return problem['loss'], problem['markers_rec']

    def workspace_view(self, name, n_obj, shape, dtype=torch.float32):
        """Intermediate of the last ``native_step`` by name (``saga_fit_workspace_offset``), e.g. ``'vertices'``."""
        off = _lib.load().saga_fit_workspace_offset(self.bm.handle(self.device), ctypes.byref(self._tstruct()),
                                                    self.batch_size, n_obj, name.encode())
        if off < 0:
            raise KeyError(name)
        n = int(np.prod(shape))
        t = self._ws[off:off + n]
        return (t.view(dtype) if dtype != torch.float32 else t).view(*shape)

    def decode_vars(self, packed=None):
        """Packed variable rows -> the SMPL-X parameter dict the reference saves (fittingop.py:466-475)."""
        packed = self.vars if packed is None else packed
        B = packed.shape[0]
        out = torch.empty(B, _lib.BODY_PARAM_COUNT, device=self.device)
        n = _lib.load().saga_fit_workspace_floats(self.bm.handle(self.device), ctypes.byref(self._tstruct()), B, 1)
        ws = self._ws if (self._ws is not None and self._ws.numel() >= n) else torch.empty(n, device=self.device)
        _lib.call('saga_fit_decode', self.device, self._native_handles(), _lib.ptr(packed), _lib.ptr(out), _lib.ptr(ws), B)
        return {k: out[:, a:b].clone() for k, (a, b) in _lib.BODY_PARAM_SLICES.items()}

    def _fitting_native(self, body_markers, object_contact, markers_contact, verts_object, normal_object):
        B, M = self.batch_size, self.tables.n_markers
        prob = self.make_problem(body_markers, verts_object, normal_object, object_contact, markers_contact)
        n_total = sum(self.num_iter)
        hist = torch.zeros(n_total, B, 16, device=self.device) if self.record_history else None
        best_loss = torch.full((B,), 10000.0, device=self.device)
        best_vars = torch.zeros_like(self.vars)
        best_markers = torch.zeros(B, M, 3, device=self.device)
        any_best = False
        smplxparams_list, markers_fit_list = [], []
        lib = _lib.load()
        P = _lib.ptr
        it = 0
        for ss in range(3):
            for ii in range(self.num_iter[ss]):
                loss, mk = self.native_step(ss, prob, None if hist is None else hist[it])
                it += 1
                if self.verbose and ii % 100 == 99:
                    self.logger('[INFO][fitting][stage%d] iter=%d, mean loss %.5f' % (ss, ii, float(loss.mean())))
                if not self.only_rec and ss == 2 and ii > 200:
                    _lib.call('saga_fit_best_update', self.device, P(loss), P(self.vars), P(mk), P(best_loss), P(best_vars),
                              P(best_markers), B, M)
                    any_best = True
            has_iter = self.num_iter[ss] > 0
            if ss == 2 and any_best:
                never = best_loss >= 10000.0           # such a candidate keeps its final parameters (tmp_info is None, :464)
                out = self.decode_vars(torch.where(never[:, None], self.vars, best_vars))
                mfit = torch.where(never[:, None, None], prob['markers_rec'], best_markers)
                self.last_fit_loss = torch.where(never, prob['loss'], best_loss)
            else:
                out, mfit = self.decode_vars(), (prob['markers_rec'].clone() if has_iter else None)
                if has_iter:
                    self.last_fit_loss = prob['loss'].clone()
            smplxparams_list.append(out)
            markers_fit_list.append(None if mfit is None else mfit.cpu().numpy())
        self.reset()
        if hist is None:
            save_loss = {k: np.zeros((0, B)) for k in LOSS_KEYS + COUNT_KEYS}
        else:
            h = hist.cpu().numpy()
            save_loss = {k: h[:, :, i] for i, k in enumerate(LOSS_KEYS)}
            save_loss.update({k: h[:, :, 10 + i].astype(np.int64) for i, k in enumerate(COUNT_KEYS)})
        return markers_fit_list, smplxparams_list, save_loss

    def _snapshot(self):
        with torch.no_grad():
            return {k: v.detach().clone() for k, v in self.body_params().items()}

    def fitting(self, body_markers, object_contact, markers_contact, verts_object, normal_object, gender, betas=None,
                log_every=0):
        """fittingop.py:343-499 for a batch.  Returns ``(markers_fit_list, smplxparams_list, save_loss)``:
        three entries (one per stage) of [B,143,3] arrays / parameter dicts of [B,.] tensors; ``save_loss``
        maps every key of the reference's ``save_loss`` to an array [n_iterations, B].  ``self.last_fit_loss`` [B] is the
        loss of the state returned last (the best stage-3 iteration's where the rule of :429-447 picked one, else the last
        iteration's): what a best-of-N selection over candidates compares."""
        self.bm = self.bm_male if gender == 'male' else self.bm_female
        if self.bm is None:
            raise ValueError('no body model was created for gender %r' % gender)
        self.faces = self.bm.faces
        if self.native:
            return self._fitting_native(body_markers, object_contact, markers_contact, verts_object, normal_object)
        B = self.batch_size
        smplxparams_list, markers_fit_list = [], []
        best_eval = torch.full((B,), 10000.0, device=self.device)
        best_params, best_markers = None, None
        hist = {k: [] for k in LOSS_KEYS + COUNT_KEYS}

        for ss, optimizer in enumerate(self.optimizers):
            markers_fit = None
            for ii in range(self.num_iter[ss]):
                alpha = min(ii / self.num_iter[ss] * 2, 1)
                optimizer.zero_grad()
                (loss, loss_dict, markers_fit, body_param, vertices_info, _rh, _rn, _h2o, _o2h) = self.calc_loss_contact_map(
                    body_markers, verts_object, normal_object, object_contact, markers_contact, gender, betas, ss, alpha)
                loss.sum().backward()
                optimizer.step()
                for k in LOSS_KEYS:
                    hist[k].append(loss_dict[k])
                for k in COUNT_KEYS:
                    hist[k].append(vertices_info[k])
                if self.verbose and log_every and not (ii + 1) % log_every:
                    self.logger('[INFO][fitting][stage%d] iter=%d, mean loss %.5f' % (ss, ii, float(loss.mean())))
                if not self.only_rec and ss == 2 and ii > 200:
                    # best-of-iteration snapshot (:430-447): parameters AFTER the step, markers from before it
                    ev = loss.detach()
                    better = ev < best_eval
                    snap = self._snapshot()
                    mk = markers_fit.detach()
                    if best_params is None:
                        best_params = {k: v.clone() for k, v in snap.items()}
                        best_markers = mk.clone()
                    else:
                        for k in snap:
                            best_params[k] = torch.where(better[:, None], snap[k], best_params[k])
                        best_markers = torch.where(better[:, None, None], mk, best_markers)
                    best_eval = torch.where(better, ev, best_eval)
            final = self._snapshot()
            if ss == 2 and not self.only_rec and best_params is not None:
                # a candidate whose loss never dropped below 10000 keeps its final parameters (tmp_info is None, :464)
                never = best_eval >= 10000.0
                out = {k: torch.where(never[:, None], final[k], best_params[k]) for k in final}
                mfit = torch.where(never[:, None, None], markers_fit.detach(), best_markers)
                self.last_fit_loss = torch.where(never, loss.detach(), best_eval)
            else:
                out, mfit = final, (markers_fit.detach() if markers_fit is not None else None)
                if markers_fit is not None:
                    self.last_fit_loss = loss.detach().clone()
            smplxparams_list.append(out)
            markers_fit_list.append(None if mfit is None else mfit.cpu().numpy())
        self.reset()
        save_loss = {k: (torch.stack(v).cpu().numpy() if v else np.zeros((0, B))) for k, v in hist.items()}
        return markers_fit_list, smplxparams_list, save_loss
