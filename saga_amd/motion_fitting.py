"""GraspMotion-Opt on the HIP operators: the counterpart of ``MotionFill/models/fittingop.py:25-428`` (SURVEY.md
section 8f, f2), batched over S independent sequences.

The reference optimises ONE sequence per ``FittingOP`` (``opt_graspmotion.py:469-513``): stage 1 fits the 79 markers of
each of the T = 61 frames in turn (one Adam, whose moments carry over from frame to frame, ``:283-340``); stage 2
optimises all frames of the sequence at once (``:342-414``) with marker, end-pose, skating, contact, collision,
hand-angle, hand-smoothness and regulariser terms.  Here S sequences share every launch: frames x sequences are the
batch of the SMPL-X forward/backward, of the vertex-normal kernel and of the nearest-neighbour search (BASELINE config
4: S = 256 -> 15,616 bodies per stage-2 iteration).  The objective handed to ``backward`` is the SUM over sequences of
the reference's per-sequence loss and Adam is element-wise, so each sequence follows exactly the update sequence of
its own B = 1 run (tests/test_gpu_motion.py checks S = 2 against two oracle runs).

Not part of this path: the learned smoothness prior (``body_smooth_loss`` ``:70-130``, a convolutional auto-encoder
with a pretrained snapshot).  Pass ``body_smooth_loss=callable(body_verts [S,T,V,3], body_params [S,T,96]) -> [S]`` in
the config to add it; without one the term is left out, whatever ``stage2_weight_loss_smooth`` says.
"""
import ctypes

import numpy as np
import torch
import torch.optim as optim

from . import _lib
from . import smplx as _smplx
from .smplx import lbs as _lbs
from . import motion_losses as ml
from .fitting import create_decoder, load_grasp_indices, param_maps
from .geometry import point2point_signed, vertex_normals_subset
from .rotations import cont6d_to_aa
from .vposer import VPoserDecoder


def angle_axis_to_rotation_matrix(aa, eps=1e-6):
    """``torchgeometry.angle_axis_to_rotation_matrix`` (upper-left 3x3) as ``ContinousRotReprDecoder.aa2matrot`` uses it
    (``utils/como/como_utils.py:219-226``): Rodrigues with the axis normalised by (theta + 1e-6), first-order Taylor
    form where theta^2 <= 1e-6.  torchgeometry is not in the reference tree: restated from its published source."""
    aa = aa.reshape(-1, 3)
    theta2 = (aa * aa).sum(-1, keepdim=True)
    theta = torch.sqrt(theta2)
    w = aa / (theta + eps)
    wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
    c, s = torch.cos(theta), torch.sin(theta)
    one = 1.0
    normal = torch.cat([c + wx * wx * (one - c), wx * wy * (one - c) - wz * s, wy * s + wx * wz * (one - c),
                        wz * s + wx * wy * (one - c), c + wy * wy * (one - c), -wx * s + wy * wz * (one - c),
                        -wy * s + wx * wz * (one - c), wx * s + wy * wz * (one - c), c + wz * wz * (one - c)], dim=1)
    rx, ry, rz = aa[:, 0:1], aa[:, 1:2], aa[:, 2:3]
    k1 = torch.ones_like(rx)
    taylor = torch.cat([k1, -rz, ry, rz, k1, -rx, -ry, rx, k1], dim=1)
    return torch.where(theta2 > eps, normal, taylor).reshape(-1, 3, 3)


def convert_to_6D_all(aa):
    """``como_utils.py:311-314``: axis-angle -> first two columns of R, row-major [N,6]."""
    return angle_axis_to_rotation_matrix(aa)[:, :, :-1].reshape(-1, 6)


def convert_to_3D_rot(x):
    """``como_utils.py:295-307``: [transl 3 | rot 6D | rest] -> [transl 3 | axis-angle 3 | rest] (fused 6D->aa kernel)."""
    return torch.cat([x[:, :3], cont6d_to_aa(x[:, 3:9]), x[:, 9:]], dim=-1)


def gen_body_mesh_v1(body_params, smplx_model, vposer_model, vertex_grad_ids=None):
    """``utils/utils_body.py:93-106``: [N, 3+3+10+32+24+24] -> (vertices [N,V,3], SMPL-X keyword dict).  With
    ``vertex_grad_ids`` -> ((vertices without gradient, vertices[:, ids] with gradient), dict)."""
    bs = body_params.shape[0]
    d = {'transl': body_params[:, 0:3], 'global_orient': body_params[:, 3:6], 'betas': body_params[:, 6:16],
         'body_pose': vposer_model.decode(body_params[:, 16:48], output_type='aa').view(bs, -1),
         'left_hand_pose': body_params[:, 48:72], 'right_hand_pose': body_params[:, 72:]}
    if vertex_grad_ids is not None:
        out = smplx_model(return_verts=True, vertex_grad_ids=vertex_grad_ids, **d)
        return (out.vertices, out.vertices_sel), d
    return smplx_model(return_verts=True, **d).vertices, d


class FittingOP:
    """``FittingOP(fittingconfig)`` with the reference's keys (``opt_graspmotion.py:470-495``): T, gender, smplx_beta,
    init_lr_stage1/2, num_iter, device, markers_ids, markers_ids_143, stage1_weight_loss_*, stage2_weight_loss_*.
    Additional optional keys: ``n_seq`` (S; ``smplx_beta`` is then [S,10]), ``start_t`` (30), ``body_model_path``,
    ``vposer``, ``smplx_kwargs``, ``indices`` (foot groups), ``body_smooth_loss`` (see the module docstring),
    ``sparse_backward`` (default True: stage 2 differentiates through the vertices its losses read only -- same gradients up
    to summation order, no dense [S*T,V,3] gradient; switched off automatically with a ``body_smooth_loss`` hook, which may
    read any vertex), ``reduced_forward`` (default True, with ``sparse_backward``: stage 2 evaluates only the vertices its losses
    and the hand normals read -- the same numbers, see ``_make_picker``), ``fused_terms`` (default True: the seven geometric
    terms of stage 2 as one HIP operator, ``saga_motion_terms_fwd/bwd``; False: their autograd composition in
    ``motion_losses``, the form the tests check it against), ``graph_stage1`` (default True: stage 1 replays one captured
    iteration, see ``fitting_stage1``), ``native_stage1`` (default True: every stage-1 iteration is one ``saga_fit_step`` call,
    see ``_fitting_stage1_native``; False: the autograd composition, replayed as a graph or eager), ``graph_stage2`` ('auto':
    stage 2 replays one captured iteration when an iteration is launch-bound, i.e. up to 2,048 bodies; True / False force it),
    ``native_params`` (default True: stage 2 maps its variables to SMPL-X arguments with ``fitting.param_maps``, one operator,
    instead of the torch VPoser MLP + 6D->aa + hand PCA; off with a ``body_smooth_loss`` hook, which takes the torch parameters),
    ``native_stage2`` (default True: every stage-2 iteration is ONE library call, ``saga_motion_step`` -- the same operators in
    the same order as the autograd composition of ``stage2_losses``, plus the regularisers, the weighted sum and Adam, with no
    PyTorch kernel in the loop; needs ``sparse_backward``, ``fused_terms``, ``native_params`` and no ``body_smooth_loss`` hook,
    else the autograd composition runs)."""

    def __init__(self, fittingconfig):
        cfg = dict(T=61, gender='male', device='cuda', n_seq=1, start_t=30, init_lr_stage1=0.1, init_lr_stage2=0.01,
                   num_iter=[100, 300], body_model_path='./body_utils/body_models', vposer=None, smplx_kwargs={},
                   indices=None, body_smooth_loss=None, sparse_backward=True, reduced_forward=True, fused_terms=True,
                   native_stage1=True, graph_stage1=True, graph_stage2='auto', native_params=True, native_stage2=True,
                   stage1_weight_loss_rec_markers=1.0, stage1_weight_loss_vposer=0.02, stage1_weight_loss_shape=0.01,
                   stage1_weight_loss_hand=0.01,
                   stage2_weight_loss_rec_markers=0.1, stage2_weight_loss_end_markers_fit=10, stage2_weight_loss_vposer=0.02,
                   stage2_weight_loss_hand=0.02, stage2_weight_loss_skating=0.05, stage2_weight_loss_smooth=2e7,
                   stage2_weight_loss_collision=10, stage2_weight_loss_contact=60, stage2_weight_loss_hand_smooth=1,
                   stage2_weight_loss_hand_angle=1)
        cfg.update(fittingconfig)
        for key, val in cfg.items():
            setattr(self, key, val)
        self.device = torch.device(self.device)
        if self.device.type != 'cuda':
            raise RuntimeError('saga_amd.motion_fitting.FittingOP runs on a ROCm GPU only (device=%s)' % self.device)
        S, T = self.n_seq, self.T

        self.smplx_model = self._make_body_model(S)                    # one frame of every sequence (stage 1)
        self.smplx_model_batch = self._make_body_model(S * T)          # every frame of every sequence (stage 2)
        self._loss_model = None                       # stage 2, the vertices its losses read only (see _make_picker)
        if self.vposer is None:
            self.vposer = VPoserDecoder(seed=0)
        self.vposer_model = self.vposer.to(self.device).eval()
        for p in self.vposer_model.parameters():
            p.requires_grad_(False)
        idx = self.indices if self.indices is not None else load_grasp_indices()
        self.foot_groups = ml.foot_groups(idx)
        as_ids = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.long, device=self.device)
        self.markers_ids = as_ids(self.markers_ids)
        self.markers_ids_143 = as_ids(self.markers_ids_143)
        self._picker = None
        self.init_param_t()

    def _make_body_model(self, batch, **extra):
        kw = dict(self.smplx_kwargs)
        kw.update(extra)
        bm = _smplx.create(self.body_model_path, model_type='smplx', gender=self.gender, ext='npz', num_pca_comps=24,
                           create_global_orient=True, create_body_pose=True, create_betas=True,
                           create_left_hand_pose=True, create_right_hand_pose=True, create_expression=True,
                           create_jaw_pose=True, create_leye_pose=True, create_reye_pose=True, create_transl=True,
                           batch_size=batch, **kw)
        return bm.to(self.device).eval()

    # ---- variables -----------------------------------------------------------------------------------------------------
    def init_param_t(self):
        """``:40-55``: one frame's variables, here one row per sequence."""
        S, dev = self.n_seq, self.device
        transl = torch.zeros(S, 3, device=dev)
        transl[:, 1] = 0.4
        transl[:, 2] = 1.0
        rot = torch.zeros(S, 3, device=dev)
        rot[:, 1] = 1.6
        rot[:, 2] = 3.14
        self.transl_opt_t = transl.requires_grad_(True)
        self.rot_6d_opt_t = convert_to_6D_all(rot).requires_grad_(True)
        self.other_params_opt_t = torch.zeros(S, 56 + 24, device=dev, requires_grad=True)    # VPoser 32 + hands 24 + 24
        self.shape_t = torch.as_tensor(np.asarray(self.smplx_beta), dtype=torch.float32, device=dev).reshape(S, -1)

    def init_param_T(self, transl_opt_T, rot_6d_opt_T, other_params_opt_T, shape_T):
        """``:57-61``: whole-sequence variables [S,T,.]."""
        dev = self.device
        self.transl_opt_T = transl_opt_T.to(dev).clone().requires_grad_(True)
        self.rot_6d_opt_T = rot_6d_opt_T.to(dev).clone().requires_grad_(True)
        self.other_params_opt_T = other_params_opt_T.to(dev).clone().requires_grad_(True)
        self.shape_T = torch.as_tensor(shape_T, dtype=torch.float32, device=dev)

    @staticmethod
    def _set_lr(opt, lr):
        for group in opt.param_groups:
            group['lr'] = lr

    # ---- stage 1 -------------------------------------------------------------------------------------------------------
    def _stage1_model(self):
        """Stage 1 reads the 79 markers only: the body model over those vertices (rows bit-identical to the full mesh's)."""
        if self.__dict__.get('_s1_model') is None:
            ids = self.markers_ids.cpu().numpy()
            keep = np.unique(ids)
            self._s1_model = (self._make_body_model(self.n_seq, vertex_subset=keep),
                              torch.as_tensor(np.searchsorted(keep, ids), dtype=torch.long, device=self.device))
        return self._s1_model

    def _stage1_iteration(self, markers_rec_t):
        """One Adam iteration of the per-frame marker fit (``:300-331``)."""
        w = self
        self.optimizer_s1.zero_grad(set_to_none=True)
        p = convert_to_3D_rot(torch.cat([self.transl_opt_t, self.rot_6d_opt_t, self.shape_t, self.other_params_opt_t], dim=-1))
        if self.reduced_forward:
            bm, pos = self._stage1_model()
            markers = gen_body_mesh_v1(p, bm, self.vposer_model)[0].index_select(1, pos)
        else:
            markers = gen_body_mesh_v1(p, self.smplx_model, self.vposer_model)[0].index_select(1, self.markers_ids)
        loss_marker = (markers - markers_rec_t).abs().mean(dim=(1, 2))
        loss = (w.stage1_weight_loss_rec_markers * loss_marker
                + w.stage1_weight_loss_vposer * (p[:, 16:48] ** 2).mean(dim=1)
                + w.stage1_weight_loss_shape * (p[:, 6:16] ** 2).mean(dim=1)
                + w.stage1_weight_loss_hand * (p[:, 48:] ** 2).mean(dim=1))
        loss.sum().backward()
        self.optimizer_s1.step()

    def fitting_stage1(self, body_markers_rec):
        """``:283-340``.  body_markers_rec [S,T,79,3] (or [T,79,3] when S = 1) -> transl [S,T,3], rot6d [S,T,6],
        shape [S,T,10], other [S,T,80], markers [S,T,79,3].

        The T x num_iter[0] iterations are small (S bodies, 79 vertices) and, composed from the operators under autograd,
        launch-bound -- ~150 launches of a few microseconds each.  Default (``native_stage1``): every iteration is one
        ``saga_fit_step`` call (``_fitting_stage1_native``).  Otherwise, with ``graph_stage1``, ONE autograd iteration is captured
        in a HIP graph (``torch.cuda.CUDAGraph``; the
        library's operators enqueue on the capturing stream and allocate nothing) and replayed: the frame's targets and the
        learning rate are device tensors the graph reads."""
        S, T = self.n_seq, self.T
        rec = torch.as_tensor(body_markers_rec).to(self.device).float().reshape(S, T, -1, 3)
        if self.native_stage1:
            return self._fitting_stage1_native(rec)
        params = [self.transl_opt_t, self.rot_6d_opt_t, self.other_params_opt_t]
        graph = bool(self.graph_stage1) and self.num_iter[0] > 0 and T > 0
        if graph:
            lr = torch.tensor(float(self.init_lr_stage1), device=self.device)
            self.optimizer_s1 = optim.Adam(params, lr=lr, capturable=True)
            target = rec[:, 0].clone()
            # warm-up off the capture (lazy handles, autotuning, allocator), then put variables and Adam state back
            keep = [q.detach().clone() for q in params]
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._stage1_iteration(target)
            torch.cuda.current_stream(self.device).wait_stream(side)
            with torch.no_grad():
                for q, k in zip(params, keep):
                    q.copy_(k)
                for st in self.optimizer_s1.state.values():
                    st['step'].zero_(); st['exp_avg'].zero_(); st['exp_avg_sq'].zero_()
            self.optimizer_s1.zero_grad(set_to_none=True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._stage1_iteration(target)
            set_lr = lambda v: lr.fill_(v)
            run = g.replay
        else:
            self.optimizer_s1 = optim.Adam(params, lr=self.init_lr_stage1)
            target = None
            set_lr = lambda v: self._set_lr(self.optimizer_s1, v)
        out = {k: [] for k in ('transl', 'rot', 'shape', 'other')}
        for t in range(T):
            if graph:
                target.copy_(rec[:, t])
            else:
                markers_rec_t = rec[:, t]
                run = lambda: self._stage1_iteration(markers_rec_t)
            if t > 0:
                set_lr(0.005)
            for step in range(self.num_iter[0]):
                if step == 61:
                    set_lr(0.01)
                if step == 81:
                    set_lr(0.003)
                run()
            out['transl'].append(self.transl_opt_t.detach().clone())
            out['rot'].append(self.rot_6d_opt_t.detach().clone())
            out['shape'].append(self.shape_t.detach().clone())
            out['other'].append(self.other_params_opt_t.detach().clone())
        st = lambda k: torch.stack(out[k], dim=1)
        return st('transl'), st('rot'), st('shape'), st('other'), rec.clone()

    def _stage1_native_handles(self):
        """(model over the markers, loss tables, decoder, workspace) for ``saga_fit_step`` with the stage-1 objective."""
        h = self.__dict__.get('_s1_native')
        if h is None:
            from .fitting import create_decoder
            from .grasp_loss import GraspTables
            bm, pos = self._stage1_model()
            pos = pos.cpu().numpy()
            # the marker-fit objective reads the markers only; the hand / foot tables need one valid entry each
            tables = GraspTables(pos, pos[:1], pos[:1], np.ones(1, dtype=np.float32), self.device)
            vp = self.vposer_model
            if not hasattr(vp, 'bodyprior_dec_fc1'):
                raise RuntimeError('native_stage1 needs a VPoser decoder with bodyprior_dec_fc1/fc2/out layers (native_stage1=False '
                                   'composes the iteration from the operators instead)')
            dec = create_decoder(vp, bm, self.device)
            n = _lib.load().saga_fit_workspace_floats(bm.handle(self.device), ctypes.byref(tables.struct), self.n_seq, 1)
            h = self._s1_native = dict(bm=bm, tables=tables, decoder=dec, ws=torch.zeros(n, device=self.device))
        return h

    def _decoder_handle(self):
        """``saga_decoder_t*`` of the VPoser decoder and the body model's hand PCA (stage 2's ``param_maps``)."""
        if self.__dict__.get('_decoder') is None:
            self._decoder = create_decoder(self.vposer_model, self.smplx_model_batch, self.device)
        return self._decoder

    def __del__(self):
        h = self.__dict__.get('_s1_native')
        for dec in ([h['decoder']] if h is not None else []) + ([self.__dict__['_decoder']] if self.__dict__.get('_decoder') else []):
            try:
                _lib.load().saga_decoder_destroy(dec)
            except Exception:
                pass

    def _fitting_stage1_native(self, rec):
        """Stage 1 with every iteration as ONE library call: ``saga_fit_step`` with ``custom_weights`` -- variable maps (VPoser
        MLP, 6D->aa, hand PCA), SMPL-X forward / backward over the 79 markers, marker L1 + regularisers (``:309-325``) and the
        Adam update of transl / rot6d / latent / hands (torch.optim.Adam arithmetic; the moments carry over from frame to frame as
        in ``:283-340``) as one sequence of ~20 kernels, no PyTorch kernel in the loop."""
        S, T, dev = self.n_seq, self.T, self.device
        h = self._stage1_native_handles()
        sl = _lib.VAR_SLICES
        packed = torch.zeros(S, _lib.VAR_STRIDE, device=dev)
        with torch.no_grad():
            packed[:, sl['vpose_rec'][0]:sl['vpose_rec'][1]] = self.other_params_opt_t[:, :32]
            packed[:, sl['hand_pose'][0]:sl['hand_pose'][1]] = self.other_params_opt_t[:, 32:]
            packed[:, sl['betas'][0]:sl['betas'][1]] = self.shape_t
            packed[:, sl['transl_rec'][0]:sl['transl_rec'][1]] = self.transl_opt_t
            packed[:, sl['glo_rot_rec'][0]:sl['glo_rot_rec'][1]] = self.rot_6d_opt_t
        m, v = torch.zeros_like(packed), torch.zeros_like(packed)
        dummy3, dummy1 = torch.zeros(S, 1, 3, device=dev), torch.zeros(S, max(h['tables'].n_markers, 1), device=dev)
        loss, mk_out = torch.empty(S, device=dev), torch.empty(S, h['tables'].n_markers, 3, device=dev)
        G = _lib.VAR_GROUPS
        mask = G['vpose_rec'] | G['hand_pose'] | G['transl_rec'] | G['glo_rot_rec']
        P = _lib.ptr
        out = {k: [] for k in ('transl', 'rot', 'shape', 'other')}
        step_count, lr = 0, float(self.init_lr_stage1)
        for t in range(T):
            target = rec[:, t].contiguous()
            prob = _lib.FitProblem(P(dummy3), P(dummy3), P(target), P(dummy1), P(dummy1), 1)
            if t > 0:
                lr = 0.005
            for step in range(self.num_iter[0]):
                if step > 60:
                    lr = 0.01
                if step > 80:
                    lr = 0.003
                step_count += 1
                cfg = _lib.FitStepCfg(0, 0, 0, mask, step_count, lr, 0.9, 0.999, 1e-8, 1, 1,
                                      float(self.stage1_weight_loss_rec_markers), 0.0, float(self.stage1_weight_loss_vposer),
                                      float(self.stage1_weight_loss_hand), 0.0, float(self.stage1_weight_loss_shape))
                _lib.call('saga_fit_step', dev, h['bm'].handle(dev), h['decoder'], ctypes.byref(h['tables'].struct), ctypes.byref(prob),
                          ctypes.byref(cfg), P(packed), P(m), P(v), P(h['ws']), P(loss), None, P(mk_out), S)
            out['transl'].append(packed[:, sl['transl_rec'][0]:sl['transl_rec'][1]].clone())
            out['rot'].append(packed[:, sl['glo_rot_rec'][0]:sl['glo_rot_rec'][1]].clone())
            out['shape'].append(self.shape_t.detach().clone())
            out['other'].append(torch.cat([packed[:, sl['vpose_rec'][0]:sl['vpose_rec'][1]],
                                           packed[:, sl['hand_pose'][0]:sl['hand_pose'][1]]], dim=1))
        with torch.no_grad():            # the module's variables end where the last frame's fit ended, as in the reference
            if T > 0:
                self.transl_opt_t.copy_(out['transl'][-1])
                self.rot_6d_opt_t.copy_(out['rot'][-1])
                self.other_params_opt_t.copy_(out['other'][-1])
        self.last_stage1_loss = loss
        st = lambda k: torch.stack(out[k], dim=1)
        return st('transl'), st('rot'), st('shape'), st('other'), rec.clone()

    # ---- stage 2 -------------------------------------------------------------------------------------------------------
    def _make_picker(self, rhand_verts):
        key = bytes(np.asarray(rhand_verts.detach().cpu() if torch.is_tensor(rhand_verts) else rhand_verts, dtype=np.int64).data)
        if self._picker is not None and self.__dict__.get('_picker_key') == (key, self.start_t, bool(self.sparse_backward),
                                                                              bool(self.reduced_forward), self.body_smooth_loss is None):
            return                                  # same hand, same switches: the tables and the loss-vertex model stand
        self._picker_key = (key, self.start_t, bool(self.sparse_backward), bool(self.reduced_forward), self.body_smooth_loss is None)
        sets = {'markers': self.markers_ids, 'markers_143': self.markers_ids_143, 'rhand': rhand_verts}
        sets.update(self.foot_groups)
        # (the objective reads the 143 markers and the hand on the frames start_t.. only: picked as windows; the sets stay in
        # the union, which is the gradient subset of the body model)
        self._picker = pk = ml.VertexPicker(sets, self.device, windows={'markers_143_tail': ('markers_143', self.start_t),
                                                                        'rhand_tail': ('rhand', self.start_t)})
        self._loss_model = None
        if not (self.sparse_backward and self.reduced_forward and self.body_smooth_loss is None):
            return
        # Stage 2 reads ~1,100 vertices, and the normals of the right hand and its palm markers read the corners of the faces
        # around those: evaluate that set only (a vertex-subset body model: its rows are bit-identical to the same rows of the
        # full mesh and it keeps every face inside the set, in order, so the normals are the full mesh's too).
        faces = np.asarray(self.smplx_model_batch.faces)
        union = pk.union.cpu().numpy()
        normal_ids = np.unique(np.concatenate([rhand_verts.cpu().numpy(), self.markers_ids_143[-22:].cpu().numpy()]))
        touched = np.zeros(self.smplx_model_batch.num_verts, dtype=bool)
        touched[normal_ids] = True
        ring = np.unique(faces[touched[faces].any(axis=1)])
        keep = np.union1d(union, ring)
        if keep.size * 10 > self.smplx_model_batch.num_verts * 9:
            return                                              # (nearly) the whole mesh anyway
        key = keep.tobytes()
        cached = self.__dict__.get('_loss_model_cache')
        if cached is None or cached[0] != key:
            cached = self.__dict__['_loss_model_cache'] = (key, self._make_body_model(self.n_seq * self.T, vertex_subset=keep))
        as_pos = lambda ids: torch.as_tensor(np.searchsorted(keep, ids.cpu().numpy() if torch.is_tensor(ids) else ids),
                                             dtype=torch.long, device=self.device)
        self._loss_model = dict(bm=cached[1], grad_rows=as_pos(union), rhand=as_pos(rhand_verts), markers_143=as_pos(self.markers_ids_143))

    def stage2_losses(self, markers_rec_T, markers_end_new, rhand_verts, contact_lbl_rec, contact_object, contact_markers,
                      normal_object, verts_object):
        """One evaluation of the stage-2 objective (``:356-409``) -> (total [S], dict of the weighted terms' raw values)."""
        S, T, start_t, w = self.n_seq, self.T, self.start_t, self
        if self._picker is None:
            self._make_picker(rhand_verts)
        pk = self._picker
        pp = None
        if self.native_params and self.body_smooth_loss is None and hasattr(self.vposer_model, 'bodyprior_dec_fc1'):
            # variables -> SMPL-X arguments as ONE operator (saga_params_fwd/bwd: VPoser MLP on the library's GEMM, 22 x (6D -> aa
            # + Jacobian), hand PCA, pose mean) instead of ~60 PyTorch launches forward and as many backward
            other = self.other_params_opt_T
            packed = torch.cat([other.reshape(S * T, -1), other.new_zeros(S * T, 6), self.shape_T.reshape(S * T, -1),
                                self.transl_opt_T.reshape(S * T, 3), self.rot_6d_opt_T.reshape(S * T, 6),
                                other.new_zeros(S * T, _lib.VAR_STRIDE - 105)], dim=1)
            smplx_args = param_maps(packed, self._decoder_handle(), self.smplx_model_batch.num_shape)
            latent, hands = other[..., :32], other[..., 32:]

            def bodies(bm, grad_ids=None):
                shape, full_pose, transl = smplx_args[1], smplx_args[0], smplx_args[2]
                if grad_ids is None:
                    return _lbs.lbs(bm, shape, full_pose, transl)[0], None
                sel, hs = bm._grad_subset(grad_ids, full_pose.device)
                v, _, v_sel = _lbs.lbs_selected(bm, sel, hs, shape, full_pose, transl)
                return v, v_sel
        else:
            p = convert_to_3D_rot(torch.cat([self.transl_opt_T, self.rot_6d_opt_T, self.shape_T, self.other_params_opt_T],
                                            dim=-1).reshape(S * T, -1))
            pp = p.reshape(S, T, -1)
            latent, hands = pp[:, :, 16:48], pp[:, :, 48:]

            def bodies(bm, grad_ids=None):
                if grad_ids is None:
                    return gen_body_mesh_v1(p, bm, self.vposer_model)[0], None
                return gen_body_mesh_v1(p, bm, self.vposer_model, vertex_grad_ids=grad_ids)[0]
        bm, normal_ids = self.smplx_model_batch, (rhand_verts, self.markers_ids_143)
        if self._loss_model is not None:
            lm = self._loss_model
            bm, normal_ids = lm['bm'], (lm['rhand'], lm['markers_143'])
            verts, sub = bodies(bm, lm['grad_rows'])
            verts, sub = verts.reshape(S, T, -1, 3), sub.reshape(S, T, -1, 3)
        elif self.sparse_backward and self.body_smooth_loss is None:
            # every vertex any term reads, as the differentiable output: the backward runs over those ~1,100 vertices only
            # (the full mesh is still evaluated -- the normals read it -- but carries no gradient)
            verts, sub = bodies(self.smplx_model_batch, pk.union)
            verts, sub = verts.reshape(S, T, -1, 3), sub.reshape(S, T, -1, 3)
        else:
            verts = bodies(self.smplx_model_batch)[0].reshape(S, T, -1, 3)
            sub = pk.gather(verts)                               # every vertex any term reads, gathered once
        picked = pk.pick(sub)                                    # every set the terms read: one kernel, one backward gather
        if self.fused_terms:
            # the seven geometric terms as one operator over the picks and the two nearest-neighbour calls (:154-157)
            n_tail, O = T - start_t, verts_object.shape[-2]
            cache = self.__dict__.setdefault('_fused_const', {})
            if cache.get('key') != (id(pk), str(self.device)):
                mask = torch.ones(picked['markers'].shape[2], device=self.device)      # (:360-364: hand, forearm, fingers out)
                mask[49:49 + 6] = 0
                mask[33:33 + 6] = 0
                mask[69:] = 0
                cache.update(key=(id(pk), str(self.device)), mask=mask,
                             sel=torch.cat([normal_ids[0], normal_ids[1][-22:]]).to(torch.int32), n_rh=int(normal_ids[0].shape[0]))
            normals = vertex_normals_subset(bm, verts, cache['sel'], first_frame=start_t)
            vo, no = verts_object.float().reshape(S * n_tail, O, 3), normal_object.float().reshape(S * n_tail, O, 3)
            _, h2o_mk, _, _, _, h2o_vec = point2point_signed(picked['markers_143_tail'].reshape(S * n_tail, -1, 3), vo, None, no,
                                                             return_vector=True, directions='x2y')
            o2h, _, _, _ = point2point_signed(picked['rhand_tail'].reshape(S * n_tail, -1, 3), vo, normals[:, :cache['n_rh']], no,
                                              directions='y2x')
            t7 = ml.motion_terms(picked['markers'], picked['feet'], pk.foot_sizes, picked['rhand_tail'], o2h, h2o_mk, h2o_vec,
                                 markers_rec_T, cache['mask'], markers_end_new, contact_lbl_rec, contact_object, contact_markers,
                                 normals[:, cache['n_rh']:])
            terms = dict(zip(ml.TERM_NAMES, t7.unbind(1)))
            terms['vposer'] = (latent ** 2).mean(dim=(1, 2))
            terms['hand'] = (hands ** 2).mean(dim=(1, 2))
            return self._total(terms, verts, pp)
        markers = picked['markers']
        mask = torch.ones(1, 1, markers.shape[2], 1, device=self.device)
        mask[:, :, 49:49 + 6] = 0        # hand
        mask[:, :, 33:33 + 6] = 0        # forearm
        mask[:, :, 69:] = 0              # finger
        terms = {}
        terms['rec_markers'] = (markers * mask - markers_rec_T * mask).abs().mean(dim=(1, 2, 3))
        terms['end_markers_fit'] = (markers[:, -1] - markers_end_new).abs().mean(dim=(1, 2))
        terms['skating'] = ml.skating_loss_seq(sub, contact_lbl_rec, {k: pk.pos[k] for k in self.foot_groups}, picked=picked)
        terms['contact'], state = ml.contact_loss_seq(bm, start_t, sub, pk.pos['markers_143'], pk.pos['rhand'],
                                                       contact_object, contact_markers, normal_object, verts_object, n_frames=T,
                                                       full_verts=verts, full_ids=normal_ids, picked=picked)
        terms['hand_smooth'] = ml.hand_smooth_loss_seq(sub, start_t, pk.pos['rhand'], picked=picked)
        terms['collision'] = ml.collision_loss_seq(state)
        terms['hand_angle'] = ml.angle_loss_seq(state, start_t, T)
        terms['vposer'] = (latent ** 2).mean(dim=(1, 2))
        terms['hand'] = (hands ** 2).mean(dim=(1, 2))
        return self._total(terms, verts, pp)

    def _total(self, terms, verts, pp):
        """Weighted sum of the stage-2 terms (``:391-409``) -> (total [S], terms)."""
        w = self
        total = (w.stage2_weight_loss_rec_markers * terms['rec_markers']
                 + w.stage2_weight_loss_end_markers_fit * terms['end_markers_fit']
                 + w.stage2_weight_loss_vposer * terms['vposer']
                 + w.stage2_weight_loss_hand * terms['hand']
                 + w.stage2_weight_loss_skating * terms['skating']
                 + w.stage2_weight_loss_hand_smooth * terms['hand_smooth']
                 + w.stage2_weight_loss_collision * terms['collision']
                 + w.stage2_weight_loss_contact * terms['contact']
                 + w.stage2_weight_loss_hand_angle * terms['hand_angle'])
        if self.body_smooth_loss is not None:
            terms['smooth'] = self.body_smooth_loss(verts, pp)
            total = total + w.stage2_weight_loss_smooth * terms['smooth']
        return total, terms

    def fitting_stage2(self, markers_rec_T, markers_end_new, rhand_verts, contact_lbl_rec, contact_object, contact_markers,
                       normal_object, verts_object, record=None):
        """``:342-414``.  markers_rec_T [S,T,79,3]; markers_end_new [S,79,3]; rhand_verts: 778 ids; contact_lbl_rec
        [S,T,4]; contact_object [S,O]; contact_markers [S,143]; normal_object / verts_object [S,T-start_t,O,3]."""
        S, T, dev = self.n_seq, self.T, self.device
        f = lambda a, *shape: torch.as_tensor(a).to(dev).float().reshape(*shape)
        markers_rec_T = f(markers_rec_T, S, T, -1, 3)
        markers_end_new = f(markers_end_new, S, -1, 3)
        contact_lbl_rec = f(contact_lbl_rec, S, T, 4)
        contact_object, contact_markers = f(contact_object, S, -1), f(contact_markers, S, -1)
        O = contact_object.shape[1]
        normal_object, verts_object = f(normal_object, S, T - self.start_t, O, 3), f(verts_object, S, T - self.start_t, O, 3)
        rhand_verts = torch.as_tensor(np.asarray(rhand_verts), dtype=torch.long, device=dev)
        self._make_picker(rhand_verts)
        params = [self.transl_opt_T, self.rot_6d_opt_T, self.other_params_opt_T]
        args = (markers_rec_T, markers_end_new, rhand_verts, contact_lbl_rec, contact_object, contact_markers, normal_object, verts_object)

        if self._stage2_native_ok():
            return self._fitting_stage2_native(args, record)

        def iteration():
            self.optimizer_s2.zero_grad(set_to_none=True)
            total, terms = self.stage2_losses(*args)
            total.sum().backward()
            self.optimizer_s2.step()
            return total.detach(), {k: v.detach() for k, v in terms.items()}

        # Few bodies per iteration (one sequence = 61: what the reference runs) leave the ~250 launches of an iteration
        # launch-bound (1.9 ms at S = 1 against 0.2 ms of kernel time): one iteration is captured in a HIP graph and replayed,
        # the learning rate a device tensor.  Large batches are GPU-bound and run eagerly.  (A body_smooth_loss hook is user code:
        # eager.)
        graph = self.graph_stage2
        if graph == 'auto':
            graph = S * T <= 2048
        graph = bool(graph) and self.num_iter[1] > 0 and self.body_smooth_loss is None
        if not graph:
            self.optimizer_s2 = optim.Adam(params, lr=self.init_lr_stage2)
            for step in range(self.num_iter[1]):
                if step > 150:
                    self._set_lr(self.optimizer_s2, 0.005)
                total, terms = iteration()
                if record is not None:
                    record(step, total, terms)
            return self.transl_opt_T, self.rot_6d_opt_T, self.shape_T, self.other_params_opt_T
        lr = torch.tensor(float(self.init_lr_stage2), device=dev)
        self.optimizer_s2 = optim.Adam(params, lr=lr, capturable=True)
        keep = [q.detach().clone() for q in params]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):          # warm-up off the capture, then variables and Adam state go back
            for _ in range(2):
                iteration()
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.no_grad():
            for q, k in zip(params, keep):
                q.copy_(k)
            for st in self.optimizer_s2.state.values():
                st['step'].zero_(); st['exp_avg'].zero_(); st['exp_avg_sq'].zero_()
        self.optimizer_s2.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            total, terms = iteration()
        for step in range(self.num_iter[1]):
            if step == 151:
                lr.fill_(0.005)
            g.replay()
            if record is not None:             # (the graph's output tensors: valid until the next replay)
                record(step, total.clone(), {k: v.clone() for k, v in terms.items()})
        return self.transl_opt_T, self.rot_6d_opt_T, self.shape_T, self.other_params_opt_T

    # ---- stage 2 as one library call per iteration ----------------------------------------------------------------------------
    def _stage2_native_ok(self):
        return bool(self.native_stage2 and self.native_params and self.fused_terms and self.sparse_backward
                    and self.body_smooth_loss is None and hasattr(self.vposer_model, 'bodyprior_dec_fc1'))

    def pack_stage2_vars(self):
        """The whole-sequence variables as ``saga_motion_step``'s packed rows [S*T, 112] (a copy)."""
        S, T = self.n_seq, self.T
        other = self.other_params_opt_T.detach()
        return torch.cat([other.reshape(S * T, -1), other.new_zeros(S * T, 6), self.shape_T.reshape(S * T, -1),
                          self.transl_opt_T.detach().reshape(S * T, 3), self.rot_6d_opt_T.detach().reshape(S * T, 6),
                          other.new_zeros(S * T, _lib.VAR_STRIDE - 105)], dim=1).contiguous()

    def unpack_stage2_vars(self, packed):
        S, T, sl = self.n_seq, self.T, _lib.VAR_SLICES
        with torch.no_grad():
            self.other_params_opt_T.copy_(packed[:, :80].reshape(S, T, 80))
            self.transl_opt_T.copy_(packed[:, sl['transl_rec'][0]:sl['transl_rec'][1]].reshape(S, T, 3))
            self.rot_6d_opt_T.copy_(packed[:, sl['glo_rot_rec'][0]:sl['glo_rot_rec'][1]].reshape(S, T, 6))

    def stage2_native_setup(self, markers_rec_T, markers_end_new, rhand_verts, contact_lbl_rec, contact_object, contact_markers,
                            normal_object, verts_object):
        """Handles, device tables, targets and workspace of ``saga_motion_step`` for these (device, float32, shaped as
        ``fitting_stage2`` shapes them) targets -> a context for :meth:`stage2_native_step`."""
        S, T, dev = self.n_seq, self.T, self.device
        if self._picker is None:
            self._make_picker(rhand_verts)
        pk, lm = self._picker, self._loss_model
        if pk.rows.names != ['markers', 'feet', 'markers_143_tail', 'rhand_tail']:
            raise RuntimeError('native_stage2: unexpected pick sets %s' % (pk.rows.names,))
        if lm is not None:
            bm, grad_ids, rh_rows, mk_rows = lm['bm'], lm['grad_rows'], lm['rhand'], lm['markers_143']
        else:
            bm, grad_ids, rh_rows, mk_rows = self.smplx_model_batch, pk.union, rhand_verts, self.markers_ids_143
        _, hs = bm._grad_subset(grad_ids, dev)
        O, n_tail = verts_object.shape[-2], T - self.start_t
        c = lambda t: t.detach().float().contiguous()
        keep = dict(markers_rec=c(markers_rec_T), markers_end=c(markers_end_new), contact_lbl=c(contact_lbl_rec),
                    contact_object=c(contact_object), contact_markers=c(contact_markers), verts_object=c(verts_object),
                    normal_object=c(normal_object))
        mask = torch.ones(len(self.markers_ids), device=dev)              # (:360-364: hand, forearm, fingers out)
        mask[49:49 + 6] = 0
        mask[33:33 + 6] = 0
        mask[69:] = 0
        keep.update(mask=mask, normal_sel=torch.cat([rh_rows, mk_rows[-22:]]).to(torch.int32).contiguous())
        P = _lib.ptr
        tabs = pk.rows.tables(dev)
        tables = _lib.MotionTables(normal_sel=P(keep['normal_sel']), marker_mask=P(mask))
        for k, (tb, n, t0) in enumerate(zip(tabs, pk.rows.sizes, pk.rows.t0)):
            tables.sets[k] = _lib.PickSet(pos=P(tb[0]), inv_ptr=P(tb[1]), inv_idx=P(tb[2]), n=n, t0=t0)
        prob = _lib.MotionProblem(**{k: P(keep[k]) for k in ('markers_rec', 'markers_end', 'contact_lbl', 'contact_object',
                                                             'contact_markers', 'verts_object', 'normal_object')})
        cfg = _lib.MotionStepCfg(beta1=0.9, beta2=0.999, eps=1e-8, lr=float(self.init_lr_stage2), adam_step=0,
                                 **{'w_' + k: float(getattr(self, 'stage2_weight_loss_' + k)) for k in _lib.MOTION_WEIGHTS})
        cfg.terms = _lib.MotionTermsCfg(n_seq=S, n_frames=T, n_tail=n_tail, n_markers=len(self.markers_ids),
                                        foot_group_size=(ctypes.c_int32 * 4)(*pk.foot_sizes), n_hand=int(rh_rows.shape[0]), n_obj=O,
                                        n_contact_markers=len(self.markers_ids_143), n_palm=22, n_last_frames=5, vel_thres=0.1)
        lib = _lib.load()
        h, dec = bm.handle(dev), self._decoder_handle()
        n = lib.saga_motion_step_workspace_floats(h, hs, dec, ctypes.byref(cfg))
        if n == 0:
            raise RuntimeError('saga_motion_step_workspace_floats: %s' % lib.saga_last_error().decode())
        return dict(bm=bm, h=h, hs=hs, dec=dec, tables=tables, prob=prob, cfg=cfg, keep=keep, ws=torch.empty(n, device=dev),
                    loss=torch.empty(S, device=dev), terms=torch.empty(S, 16, device=dev))

    def stage2_native_step(self, ctx, packed, adam_m=None, adam_v=None, adam_step=0, lr=None):
        """One ``saga_motion_step`` on the packed rows (``adam_step`` 0: objective and gradient only) -> (loss [S], terms [S,16]:
        views of the context's buffers, valid until the next call)."""
        cfg = ctx['cfg']
        cfg.adam_step = int(adam_step)
        if lr is not None:
            cfg.lr = float(lr)
        P = _lib.ptr
        _lib.call('saga_motion_step', self.device, ctx['h'], ctx['hs'], ctx['dec'], ctypes.byref(ctx['tables']), ctypes.byref(ctx['prob']),
                  ctypes.byref(cfg), P(packed), P(adam_m), P(adam_v), P(ctx['ws']), P(ctx['loss']), P(ctx['terms']))
        return ctx['loss'], ctx['terms']

    def stage2_native_buffer(self, ctx, name, shape, dtype=torch.float32):
        """A named intermediate of the last ``stage2_native_step`` inside the context's workspace (a view)."""
        off = _lib.load().saga_motion_step_workspace_offset(ctx['h'], ctx['hs'], ctx['dec'], ctypes.byref(ctx['cfg']), name.encode())
        if off < 0:
            raise KeyError(name)
        n = int(np.prod(shape))
        return ctx['ws'][off:off + n].view(dtype).reshape(shape)

    def _fitting_stage2_native(self, args, record):
        ctx = self.stage2_native_setup(*args)
        packed = self.pack_stage2_vars()
        m, v = torch.zeros_like(packed), torch.zeros_like(packed)
        lr = float(self.init_lr_stage2)
        for step in range(self.num_iter[1]):
            if step > 150:
                lr = 0.005
            total, terms = self.stage2_native_step(ctx, packed, m, v, step + 1, lr)
            if record is not None:
                record(step, total.clone(), {k: terms[:, i].clone() for i, k in enumerate(_lib.MOTION_STEP_TERMS) if i > 0})
        self.unpack_stage2_vars(packed)
        self.last_stage2 = ctx
        return self.transl_opt_T, self.rot_6d_opt_T, self.shape_T, self.other_params_opt_T

    def fitting(self, body_markers_rec, markers_end_new, rhand_verts, contact_lbl_rec, contact_object, contact_markers,
                normal_object, verts_object):
        """``:417-428``: stage 1, then stage 2 from its result; returns both parameter sets."""
        transl, rot, shape, other, markers_rec_T = self.fitting_stage1(body_markers_rec)
        self.init_param_T(transl.detach(), rot.detach(), other.detach(), shape.detach())
        final = self.fitting_stage2(markers_rec_T, markers_end_new, rhand_verts, contact_lbl_rec, contact_object,
                                    contact_markers, normal_object, verts_object)
        return (transl, rot, shape, other) + tuple(final)
