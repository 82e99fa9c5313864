"""``smplx.lbs`` surface used by the reference: ``batch_rodrigues`` (imported directly at
``WholeGraspPose/models/objectmodel.py:6,38`` and ``opt_grasppose.py:10,38``) and the fused
``lbs`` entry used by :class:`saga_amd.smplx.SMPLX`.  All arithmetic runs in ``libsaga_hip.so``.
"""
import torch

from .. import _lib


class _BatchRodrigues(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rot_vecs):
        _lib.require_device(rot_vecs)
        rv = rot_vecs.contiguous()
        out = torch.empty(rv.shape[0], 3, 3, device=rv.device, dtype=torch.float32)
        lib = _lib.load()
        _lib.call('saga_rodrigues_fwd', rv.device, _lib.ptr(rv), _lib.ptr(out), rv.shape[0])
        ctx.save_for_backward(rv)
        return out

    @staticmethod
    def backward(ctx, g):
        (rv,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(rv)
        lib = _lib.load()
        _lib.call('saga_rodrigues_bwd', rv.device, _lib.ptr(rv), _lib.ptr(g), _lib.ptr(out), rv.shape[0])
        return out


def batch_rodrigues(rot_vecs, epsilon=1e-8, dtype=torch.float32):
    """[N,3] axis-angle -> [N,3,3] rotation matrices (angle = ||r + 1e-8||)."""
    if rot_vecs.dim() != 2 or rot_vecs.shape[1] != 3:
        raise ValueError('rot_vecs must be [N,3], got %s' % (tuple(rot_vecs.shape),))
    return _BatchRodrigues.apply(rot_vecs)


class _LBS(torch.autograd.Function):
    """(shape[B,S], full_pose[B,165], transl[B,3]) -> vertices[B,V,3], joints[B,55+E+L,3]."""

    @staticmethod
    def forward(ctx, model, shape, full_pose, transl):
        _lib.require_device(shape, full_pose, transl)
        lib = _lib.load()
        B = full_pose.shape[0]
        dev = full_pose.device
        shape = shape.contiguous()
        full_pose = full_pose.contiguous()
        transl = None if transl is None else transl.contiguous()
        h = model.handle(dev)
        V, npad, njo = model.num_verts, model.npad(dev), model.num_joints_out
        f32 = dict(device=dev, dtype=torch.float32)
        vertices = torch.empty(B, V, 3, **f32)
        joints = torch.empty(B, njo, 3, **f32)
        feat = torch.empty(B, _lib.FEAT_K, **f32)
        rot = torch.empty(B, 55, 9, **f32)
        jrest = torch.empty(B, 55, 3, **f32)
        chain = torch.empty(B, 55, 12, **f32)
        A = torch.empty(B, 55, 12, **f32)
        v_posed = torch.empty(B, npad, **f32)
        _lib.call('saga_smplx_fwd', dev, h, _lib.ptr(shape), _lib.ptr(full_pose), _lib.ptr(transl), _lib.ptr(vertices),
                                      _lib.ptr(joints), _lib.ptr(feat), _lib.ptr(rot), _lib.ptr(jrest), _lib.ptr(chain),
                                      _lib.ptr(A), _lib.ptr(v_posed), B)
        ctx.model = model
        ctx.has_transl = transl is not None
        ctx.save_for_backward(full_pose, rot, jrest, chain, A, v_posed)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable()
        return vertices, joints

    @staticmethod
    def backward(ctx, g_vertices, g_joints):
        full_pose, rot, jrest, chain, A, v_posed = ctx.saved_tensors
        model = ctx.model
        if g_vertices is None and g_joints is None:
            return None, None, None, None
        lib = _lib.load()
        B = full_pose.shape[0]
        dev = full_pose.device
        f32 = dict(device=dev, dtype=torch.float32)
        g_vertices = None if g_vertices is None else g_vertices.contiguous()
        g_joints = None if g_joints is None else g_joints.contiguous()
        g_shape = torch.empty(B, model.num_shape, **f32)
        g_pose = torch.empty(B, 165, **f32)
        g_transl = torch.empty(B, 3, **f32)
        has_v = g_vertices is not None or (g_joints is not None and model.joints_touch_verts)
        ws_gvp = torch.empty(B, v_posed.shape[1], **f32) if has_v else None
        ws_slab = torch.empty(_lib.BWD_SPLITK, B, _lib.FEAT_K, **f32) if has_v else None
        ws_dA = torch.empty(B, _lib.DA_STRIDE, **f32)
        ws_gv = torch.empty(B, model.num_verts, 3, **f32) if (g_joints is not None and model.joints_touch_verts) else None
        _lib.call('saga_smplx_bwd', dev, model.handle(dev), _lib.ptr(g_vertices), _lib.ptr(g_joints), _lib.ptr(full_pose),
                                      _lib.ptr(rot), _lib.ptr(jrest), _lib.ptr(chain), _lib.ptr(A), _lib.ptr(v_posed),
                                      _lib.ptr(g_shape), _lib.ptr(g_pose), _lib.ptr(g_transl), _lib.ptr(ws_gvp),
                                      _lib.ptr(ws_slab), _lib.ptr(ws_dA), _lib.ptr(ws_gv), B)
        return None, g_shape, g_pose, (g_transl if ctx.has_transl else None)


class _LBSSelected(torch.autograd.Function):
    """(shape, full_pose, transl) -> vertices [B,V,3] (no gradient), joints (no gradient), vertices_sel [B,U,3] = the rows
    ``sel`` of ``vertices`` (differentiable).  For callers whose loss reads a few vertices only: the backward takes the
    compact gradient [B,U,3] and runs on the subset model (``SMPLX.subset_handle``) -- no dense [B,V,3] gradient, no
    zero-fill / scatter of it, and the products with its structural zeros are never formed.  Only the selected rows of
    ``v_posed`` are kept for the backward."""

    @staticmethod
    def forward(ctx, model, sel, hs, shape, full_pose, transl):
        _lib.require_device(shape, full_pose, transl)         # sel: long index tensor, used by torch gathers only
        lib = _lib.load()
        B = full_pose.shape[0]
        dev = full_pose.device
        shape = shape.contiguous()
        full_pose = full_pose.contiguous()
        transl = None if transl is None else transl.contiguous()
        h = model.handle(dev)
        V, npad, njo = model.num_verts, model.npad(dev), model.num_joints_out
        f32 = dict(device=dev, dtype=torch.float32)
        vertices = torch.empty(B, V, 3, **f32)
        joints = torch.empty(B, njo, 3, **f32)
        feat = torch.empty(B, _lib.FEAT_K, **f32)
        rot = torch.empty(B, 55, 9, **f32)
        jrest = torch.empty(B, 55, 3, **f32)
        chain = torch.empty(B, 55, 12, **f32)
        A = torch.empty(B, 55, 12, **f32)
        v_posed = torch.empty(B, npad, **f32)
        U = sel.shape[0]
        v_posed_s = torch.empty(B, lib.saga_model_npad(hs), **f32)     # the subset model's layout: [B, Npad_s], pad written 0
        vertices_sel = torch.empty(B, U, 3, **f32)
        # the skinning kernel writes the selected rows compact while it has them in registers (no gather passes)
        _lib.call('saga_smplx_fwd_selected', dev, h, hs, _lib.ptr(shape), _lib.ptr(full_pose), _lib.ptr(transl), _lib.ptr(vertices),
                  _lib.ptr(joints), _lib.ptr(feat), _lib.ptr(rot), _lib.ptr(jrest), _lib.ptr(chain), _lib.ptr(A), _lib.ptr(v_posed),
                  _lib.ptr(v_posed_s), _lib.ptr(vertices_sel), B)
        ctx.model, ctx.subset = model, hs
        ctx.has_transl = transl is not None
        ctx.save_for_backward(full_pose, rot, jrest, chain, A, v_posed_s)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(vertices, joints)
        return vertices, joints, vertices_sel

    @staticmethod
    def backward(ctx, _gv, _gj, g_sel):
        full_pose, rot, jrest, chain, A, v_posed_s = ctx.saved_tensors
        model = ctx.model
        if g_sel is None:
            return None, None, None, None, None, None
        lib = _lib.load()
        B = full_pose.shape[0]
        dev = full_pose.device
        f32 = dict(device=dev, dtype=torch.float32)
        g_sel = g_sel.contiguous()
        g_shape = torch.empty(B, model.num_shape, **f32)
        g_pose = torch.empty(B, 165, **f32)
        g_transl = torch.empty(B, 3, **f32)
        ws_gvp = torch.empty(B, v_posed_s.shape[1], **f32)
        ws_slab = torch.empty(_lib.BWD_SPLITK, B, _lib.FEAT_K, **f32)
        ws_dA = torch.empty(B, _lib.DA_STRIDE, **f32)
        _lib.call('saga_smplx_bwd', dev, ctx.subset, _lib.ptr(g_sel), None, _lib.ptr(full_pose), _lib.ptr(rot), _lib.ptr(jrest),
                                      _lib.ptr(chain), _lib.ptr(A), _lib.ptr(v_posed_s), _lib.ptr(g_shape), _lib.ptr(g_pose),
                                      _lib.ptr(g_transl), _lib.ptr(ws_gvp), _lib.ptr(ws_slab), _lib.ptr(ws_dA), None, B)
        return None, None, None, g_shape, g_pose, (g_transl if ctx.has_transl else None)


def lbs_selected(model, sel, hs, shape, full_pose, transl=None):
    """``sel``: ascending unique vertex ids (long tensor on the device), ``hs``: ``model.subset_handle`` over them
    -> (vertices, joints, vertices_sel)."""
    return _LBSSelected.apply(model, sel, hs, shape, full_pose, transl)


def lbs(model, shape, full_pose, transl=None):
    """Fused SMPL-X skinning on the HIP path; ``model`` is a :class:`saga_amd.smplx.SMPLX`."""
    return _LBS.apply(model, shape, full_pose, transl)
