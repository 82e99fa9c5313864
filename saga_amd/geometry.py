"""Geometry / loss helpers of the hot path on the HIP kernels.

* :func:`point2point_signed`  mirrors ``utils/train_helper.py:17-86`` (same arguments, return order
  and ``ValueError``) as ONE HIP operator with its own deterministic backward (``saga_p2p_signed_fwd/bwd``).
* :func:`vertex_normals` / :func:`vertex_normals_subset` replace Open3D's
  ``compute_vertex_normals`` (``WholeGraspPose/models/fittingop.py:158-164``) on the device.
"""
import torch

from . import _lib


class _P2PSigned(torch.autograd.Function):
    """One HIP operator for ``point2point_signed`` (``saga_p2p_signed_fwd/bwd``): NN search + gathered difference vectors +
    norms + normal-dot signs forward; a deterministic (fixed-order gather, no atomics) backward to both clouds."""

    @staticmethod
    def forward(ctx, x, y, x_normals, y_normals, want_vec, do_y2x=True, do_x2y=True):
        _lib.require_device(x, y, x_normals, y_normals)
        xc, yc = x.detach().contiguous(), y.detach().contiguous()
        xn = None if x_normals is None else x_normals.detach().contiguous()
        yn = None if y_normals is None else y_normals.detach().contiguous()
        N, P1, _ = xc.shape
        P2 = yc.shape[1]
        f32, i32 = dict(device=xc.device, dtype=torch.float32), dict(device=xc.device, dtype=torch.int32)
        y2x_s, x2y_s = (torch.empty(N, P2, **f32) if do_y2x else None), (torch.empty(N, P1, **f32) if do_x2y else None)
        yidx, xidx = (torch.empty(N, P2, **i32) if do_y2x else None), (torch.empty(N, P1, **i32) if do_x2y else None)
        y2x_v = torch.empty(N, P2, 3, **f32) if (want_vec and do_y2x) else None
        x2y_v = torch.empty(N, P1, 3, **f32) if (want_vec and do_x2y) else None
        P = _lib.ptr
        # large clouds: the pruned search, with a workspace allocated here (same indices as the scan, bit for bit)
        nbytes = _lib.load().saga_chamfer_workspace_bytes(N, P1, P2) if N > 0 and P1 > 0 and P2 > 0 else 0
        ws = torch.empty(nbytes, device=xc.device, dtype=torch.uint8) if nbytes else None
        _lib.call('saga_p2p_signed_fwd_ws', xc.device, P(xc), P(yc), P(xn), P(yn), P(y2x_s), P(x2y_s), P(yidx), P(xidx), P(y2x_v),
                  P(x2y_v), N, P1, P2, P(ws), nbytes)
        ctx.save_for_backward(xc, yc, yidx, xidx, y2x_s, x2y_s)
        ctx.mark_non_differentiable(*[t for t in (yidx, xidx) if t is not None])
        ctx.set_materialize_grads(False)
        if want_vec:
            return y2x_s, x2y_s, yidx, xidx, y2x_v, x2y_v
        return y2x_s, x2y_s, yidx, xidx

    @staticmethod
    def backward(ctx, g_y2x_s, g_x2y_s, _gi1, _gi2, g_y2x_v=None, g_x2y_v=None):
        xc, yc, yidx, xidx, y2x_s, x2y_s = ctx.saved_tensors
        need_x, need_y = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_x or need_y) or all(g is None for g in (g_y2x_s, g_x2y_s, g_y2x_v, g_x2y_v)):
            return None, None, None, None, None, None, None
        c = lambda g: None if g is None else g.contiguous()
        gx = torch.empty_like(xc) if need_x else None
        gy = torch.empty_like(yc) if need_y else None
        N, P1, _ = xc.shape
        P = _lib.ptr
        _lib.call('saga_p2p_signed_bwd', xc.device, P(xc), P(yc), P(yidx), P(xidx), P(y2x_s), P(x2y_s), P(c(g_y2x_s)), P(c(g_x2y_s)),
                  P(c(g_y2x_v)), P(c(g_x2y_v)), P(gx), P(gy), N, P1, yc.shape[1])
        return gx, gy, None, None, None, None, None


def point2point_signed(x, y, x_normals=None, y_normals=None, return_vector=False, directions='both'):
    """Signed distance between two point clouds (``utils/train_helper.py:17-86``: same arguments, return order and
    ``ValueError``).  x [N,P1,3], y [N,P2,3] -> y2x_signed [N,P2], x2y_signed [N,P1], yidx_near [N,P2] int32,
    xidx_near [N,P1] int32 (, y2x [N,P2,3], x2y [N,P1,3]).  Differentiable w.r.t. ``x`` and ``y`` (not the normals: only
    the sign of their dot product is used).  ``directions='x2y'`` / ``'y2x'`` (not in the reference): the other direction's
    search is not run and its outputs (signed distance, index, vector) are None -- for call sites that discard them."""
    if directions not in ('both', 'x2y', 'y2x'):
        raise ValueError("directions must be 'both', 'x2y' or 'y2x'")
    N, P1, D = x.shape
    if y.shape[0] != N or y.shape[2] != D:
        raise ValueError("y does not have the correct shape.")
    if D != 3:
        raise NotImplementedError('point2point_signed on the HIP path takes 3-D points (got D=%d)' % D)
    return _P2PSigned.apply(x, y, x_normals, y_normals, bool(return_vector), directions != 'x2y', directions != 'y2x')


class _PickRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, picker):
        _lib.require_device(src)
        if src.dim() < 3 or src.shape[-1] != 3 or src.shape[-2] != picker.n_rows:
            raise ValueError('pick_rows: src must be [..., T, %d, 3], got %s' % (picker.n_rows, tuple(src.shape)))
        sc = src.detach().contiguous()
        T = sc.shape[-3] if sc.dim() >= 3 else 1
        S = sc.numel() // (T * picker.n_rows * 3)
        if max(picker.t0) >= T:
            raise ValueError('pick_rows: a set starts at frame %d, src has %d frames' % (max(picker.t0), T))
        lead = sc.shape[:-3]
        outs = [torch.empty(lead + (T - t0, n, 3), device=sc.device, dtype=torch.float32) for n, t0 in zip(picker.sizes, picker.t0)]
        tabs = picker.tables(sc.device)
        sets = (_lib.PickSet * len(outs))(*[_lib.PickSet(pos=_lib.ptr(tb[0]), out=_lib.ptr(o), n=n, t0=t0)
                                            for tb, o, n, t0 in zip(tabs, outs, picker.sizes, picker.t0)])
        _lib.call('saga_pick_sets_fwd', sc.device, _lib.ptr(sc), sets, len(outs), S, T, picker.n_rows)
        ctx.picker, ctx.src_shape, ctx.dims = picker, src.shape, (S, T)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        picker = ctx.picker
        if all(g is None for g in grads):
            return None, None
        dev = next(g for g in grads if g is not None).device
        grads = [None if g is None else g.contiguous() for g in grads]
        gs = torch.empty(ctx.src_shape, device=dev, dtype=torch.float32)
        tabs = picker.tables(dev)
        sets = (_lib.PickSet * len(grads))(*[_lib.PickSet(inv_ptr=_lib.ptr(tb[1]), inv_idx=_lib.ptr(tb[2]), g_out=_lib.ptr(g), n=n, t0=t0)
                                             for tb, g, n, t0 in zip(tabs, grads, picker.sizes, picker.t0)])
        S, T = ctx.dims
        _lib.call('saga_pick_sets_bwd', dev, sets, len(grads), _lib.ptr(gs), S, T, picker.n_rows)
        return gs, None


class RowPicker:
    """Several named row sets of one [..., T, U, 3] cloud picked by ONE kernel (``saga_pick_sets_fwd``) and differentiated by
    one deterministic gather (``saga_pick_sets_bwd``).  ``sets``: {name: positions in [0, U)} or {name: (positions, t0)} --
    repeats and overlaps allowed; with ``t0`` the set is taken from frames t0.. of the T axis only.  ``picker(src)`` ->
    {name: [..., T - t0, n, 3]} (contiguous tensors)."""

    def __init__(self, sets, n_rows):
        import numpy as np
        if not 0 < len(sets) <= _lib.PICK_MAX_SETS:
            raise ValueError('RowPicker: 1..%d sets' % _lib.PICK_MAX_SETS)
        self.names = list(sets)
        self.n_rows = int(n_rows)
        self.sizes, self.t0, self._host = [], [], []
        for k in self.names:
            v, t0 = sets[k] if isinstance(sets[k], tuple) else (sets[k], 0)
            pos = np.asarray(v.cpu() if torch.is_tensor(v) else v, dtype=np.int64).reshape(-1)
            if pos.size == 0 or pos.min() < 0 or pos.max() >= n_rows or t0 < 0:
                raise ValueError('RowPicker: positions of %r must lie in [0, %d), t0 >= 0' % (k, n_rows))
            order = np.argsort(pos, kind='stable')                      # picks of a row in ascending pick order
            ptr = np.zeros(n_rows + 1, dtype=np.int64)
            np.add.at(ptr, pos + 1, 1)
            self.sizes.append(int(pos.size))
            self.t0.append(int(t0))
            self._host.append((pos.astype(np.int32), np.cumsum(ptr).astype(np.int32), order.astype(np.int32)))
        self._dev = {}

    def tables(self, device):
        t = self._dev.get(str(device))
        if t is None:
            t = self._dev[str(device)] = [tuple(torch.from_numpy(a).to(device) for a in h) for h in self._host]
        return t

    def __call__(self, src):
        return dict(zip(self.names, _PickRows.apply(src, self)))


def vertex_normals(model, verts):
    """All vertex normals of the model's mesh: verts [B,V,3] -> [B,V,3] (detached)."""
    _lib.require_device(verts)
    verts = verts.detach().contiguous()
    B, V, _ = verts.shape
    if V != model.num_verts:
        raise ValueError('verts has %d vertices, the model %d' % (V, model.num_verts))
    out = torch.empty_like(verts)
    lib = _lib.load()
    _lib.call('saga_vertex_normals', verts.device, model.handle(verts.device), _lib.ptr(verts), _lib.ptr(out), B)
    return out


def vertex_normals_subset(model, verts, sel, first_frame=None):
    """Normals of the listed vertices only: verts [B,V,3], sel int32 [n] (device) -> [B,n,3] (detached).  With
    ``first_frame`` = t0: verts [S,T,V,3] -> [S*(T-t0),n,3], the frames t0.. of every sequence (no copy of the slice)."""
    _lib.require_device(verts, sel)
    verts = verts.detach().contiguous()
    if first_frame is not None:
        S, T, V, _ = verts.shape
        if V != model.num_verts or not 0 <= first_frame < T:
            raise ValueError('verts has %d vertices (model: %d), first_frame %d of %d' % (V, model.num_verts, first_frame, T))
        out = torch.empty(S * (T - first_frame), sel.shape[0], 3, device=verts.device, dtype=torch.float32)
        _lib.call('saga_vertex_normals_subset_window', verts.device, model.handle(verts.device), _lib.ptr(verts), _lib.ptr(sel),
                  sel.shape[0], _lib.ptr(out), S, T, int(first_frame))
        return out
    B, V, _ = verts.shape
    if V != model.num_verts:
        raise ValueError('verts has %d vertices, the model %d' % (V, model.num_verts))
    out = torch.empty(B, sel.shape[0], 3, device=verts.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.call('saga_vertex_normals_subset', verts.device, model.handle(verts.device), _lib.ptr(verts), _lib.ptr(sel), sel.shape[0],
                                              _lib.ptr(out), B)
    return out
