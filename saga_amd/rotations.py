"""Rotation maps evaluated every iteration at ``WholeGraspPose/models/fittingop.py:140``:
``RotConverter.rotmat2aa(RotConverter.cont2rotmat(glo_rot_rec))`` (``utils/utils.py:373-417``).
``cont6d_to_aa`` runs the whole chain (Gram-Schmidt, 4-branch quaternion, atan2) and its exact Jacobian in ONE
HIP kernel (``saga_cont6d_to_aa_fwd/bwd``) -- the eager chain is ~90 tiny kernels forward and ~200 backward per
call, and the iteration makes two calls (global orientation, VPoser's 21 joints).  There is no torch restatement of
the two-step ``RotConverter`` in the product (the CPU restatement lives in ``oracle/rotations_ref.py``, tests only): a
reference call site ``RotConverter.rotmat2aa(RotConverter.cont2rotmat(x))`` becomes ``cont6d_to_aa(x)``."""
import torch

from . import _lib


class _Cont6dToAA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x6):
        _lib.require_device(x6)
        x = x6.detach().reshape(-1, 6).contiguous()
        n = x.shape[0]
        aa = torch.empty(n, 3, device=x.device, dtype=torch.float32)
        jac = torch.empty(n, 18, device=x.device, dtype=torch.float32)
        _lib.call('saga_cont6d_to_aa_fwd', x.device, _lib.ptr(x), _lib.ptr(aa), _lib.ptr(jac), n)
        ctx.save_for_backward(jac)
        ctx.in_shape = x6.shape
        return aa

    @staticmethod
    def backward(ctx, g_aa):
        jac, = ctx.saved_tensors
        g = g_aa.contiguous()
        n = jac.shape[0]
        gx = torch.empty(n, 6, device=g.device, dtype=torch.float32)
        _lib.call('saga_cont6d_to_aa_bwd', jac.device, _lib.ptr(jac), _lib.ptr(g), _lib.ptr(gx), n)
        return gx.reshape(ctx.in_shape)


def cont6d_to_aa(x6):
    """[..., 6] (or any shape with 6k elements per row, viewed as [-1, 6]) -> [N, 3] axis-angle;
    ``rotmat2aa(cont2rotmat(x6))`` of ``utils/utils.py:385-417`` fused, differentiable."""
    return _Cont6dToAA.apply(x6)
