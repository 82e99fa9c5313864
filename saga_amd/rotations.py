"""Rotation maps evaluated every iteration at ``WholeGraspPose/models/fittingop.py:140``:
``RotConverter.rotmat2aa(RotConverter.cont2rotmat(glo_rot_rec))`` (``utils/utils.py:373-417``).
``cont6d_to_aa`` runs the whole chain (Gram-Schmidt, 4-branch quaternion, atan2) and its exact Jacobian in ONE
HIP kernel (``saga_cont6d_to_aa_fwd/bwd``) -- the eager chain is ~90 tiny kernels forward and ~200 backward per
call, and the iteration makes two calls (global orientation, VPoser's 21 joints).  ``RotConverter`` keeps the
reference's two-step interface as plain torch ops on device tensors."""
import torch
import torch.nn.functional as F

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


class RotConverter(torch.nn.Module):
    @staticmethod
    def cont2rotmat(module_input):
        """6D continuous representation -> rotation matrix, columns (b1, b2, b3). utils.py:385-392."""
        m = module_input.reshape(-1, 3, 2)
        b1 = F.normalize(m[:, :, 0], dim=1)
        dot = torch.sum(b1 * m[:, :, 1], dim=1, keepdim=True)
        b2 = F.normalize(m[:, :, 1] - dot * b1, dim=-1)
        b3 = torch.cross(b1, b2, dim=1)
        return torch.stack([b1, b2, b3], dim=-1)

    @staticmethod
    def rotmat2aa(pose_matrot):
        """Rotation matrix -> axis-angle through the 4-branch quaternion route. utils.py:409-417, 244-370."""
        R = pose_matrot.reshape(-1, 3, 3)
        return quaternion_to_angle_axis(rotation_matrix_to_quaternion(R)).reshape(-1, 3).contiguous()


def rotation_matrix_to_quaternion(R, eps=1e-6):
    """[N,3,3] -> (w,x,y,z); same branch conditions as utils.py:279-320 (which index the transpose)."""
    t = R.transpose(1, 2)
    m00, m11, m22 = t[:, 0, 0], t[:, 1, 1], t[:, 2, 2]
    neg_z = m22 < eps
    x_dom = m00 > m11
    xy_neg = m00 < -m11
    t0 = 1 + m00 - m11 - m22
    t1 = 1 - m00 + m11 - m22
    t2 = 1 - m00 - m11 + m22
    t3 = 1 + m00 + m11 + m22
    q0 = torch.stack([t[:, 1, 2] - t[:, 2, 1], t0, t[:, 0, 1] + t[:, 1, 0], t[:, 2, 0] + t[:, 0, 2]], -1)
    q1 = torch.stack([t[:, 2, 0] - t[:, 0, 2], t[:, 0, 1] + t[:, 1, 0], t1, t[:, 1, 2] + t[:, 2, 1]], -1)
    q2 = torch.stack([t[:, 0, 1] - t[:, 1, 0], t[:, 2, 0] + t[:, 0, 2], t[:, 1, 2] + t[:, 2, 1], t2], -1)
    q3 = torch.stack([t3, t[:, 1, 2] - t[:, 2, 1], t[:, 2, 0] - t[:, 0, 2], t[:, 0, 1] - t[:, 1, 0]], -1)
    masks = [neg_z & x_dom, neg_z & ~x_dom, ~neg_z & xy_neg, ~neg_z & ~xy_neg]
    masks = [m[:, None].to(R.dtype) for m in masks]
    q = q0 * masks[0] + q1 * masks[1] + q2 * masks[2] + q3 * masks[3]
    den = t0[:, None] * masks[0] + t1[:, None] * masks[1] + t2[:, None] * masks[2] + t3[:, None] * masks[3]
    return q / torch.sqrt(den) * 0.5


def quaternion_to_angle_axis(q):
    """(w,x,y,z) -> axis-angle. utils.py:323-370."""
    q1, q2, q3 = q[..., 1], q[..., 2], q[..., 3]
    sin_sq = q1 * q1 + q2 * q2 + q3 * q3
    sin_t = torch.sqrt(sin_sq)
    cos_t = q[..., 0]
    two_theta = 2.0 * torch.where(cos_t < 0.0, torch.atan2(-sin_t, -cos_t), torch.atan2(sin_t, cos_t))
    k = torch.where(sin_sq > 0.0, two_theta / sin_t, 2.0 * torch.ones_like(sin_t))
    return torch.stack([q1 * k, q2 * k, q3 * k], dim=-1)
