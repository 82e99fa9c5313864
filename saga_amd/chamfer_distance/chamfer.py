import torch

from .. import _lib


def _nn_search(x, y, dist1, dist2, idx1, idx2):
    """saga_chamfer_fwd, or -- when a cloud is large enough for the pruned search (saga_chamfer_workspace_bytes > 0: >= 4,096
    database points) -- saga_chamfer_fwd_ws with a workspace allocated here.  Same outputs either way, bit for bit."""
    N, P1, P2 = x.shape[0], x.shape[1], y.shape[1]
    lib = _lib.load()
    nbytes = lib.saga_chamfer_workspace_bytes(N, P1, P2) if N > 0 and P1 > 0 and P2 > 0 else 0
    P = _lib.ptr
    if nbytes:
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        _lib.call('saga_chamfer_fwd_ws', x.device, P(x), P(y), P(dist1), P(dist2), P(idx1), P(idx2), N, P1, P2, P(ws), nbytes)
    else:
        _lib.call('saga_chamfer_fwd', x.device, P(x), P(y), P(dist1), P(dist2), P(idx1), P(idx2), N, P1, P2)


class _ChamferFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        _lib.require_device(x, y)
        if x.dim() != 3 or y.dim() != 3 or x.shape[2] != 3 or y.shape[2] != 3 or x.shape[0] != y.shape[0]:
            raise ValueError('ChamferDistance expects x [N,P1,3] and y [N,P2,3], got %s and %s'
                             % (tuple(x.shape), tuple(y.shape)))
        x = x.contiguous()
        y = y.contiguous()
        N, P1, _ = x.shape
        P2 = y.shape[1]
        dev = x.device
        dist1 = torch.empty(N, P1, device=dev, dtype=torch.float32)
        dist2 = torch.empty(N, P2, device=dev, dtype=torch.float32)
        idx1 = torch.empty(N, P1, device=dev, dtype=torch.int32)
        idx2 = torch.empty(N, P2, device=dev, dtype=torch.int32)
        _nn_search(x, y, dist1, dist2, idx1, idx2)
        ctx.save_for_backward(x, y, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        ctx.set_materialize_grads(False)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, g1, g2, _gi1, _gi2):
        x, y, idx1, idx2 = ctx.saved_tensors
        if g1 is None and g2 is None:
            return None, None
        N, P1, _ = x.shape
        P2 = y.shape[1]
        gx = torch.zeros_like(x)
        gy = torch.zeros_like(y)
        g1 = None if g1 is None else g1.contiguous()
        g2 = None if g2 is None else g2.contiguous()
        lib = _lib.load()
        _lib.call('saga_chamfer_bwd', x.device, _lib.ptr(x), _lib.ptr(y), _lib.ptr(idx1), _lib.ptr(idx2), _lib.ptr(g1),
                                        _lib.ptr(g2), _lib.ptr(gx), _lib.ptr(gy), N, P1, P2)
        return gx, gy


def chamfer_nn(x, y):
    """Indices only (what ``point2point_signed`` consumes): -> idx1 [N,P1], idx2 [N,P2] int32."""
    _lib.require_device(x, y)
    if x.dim() != 3 or y.dim() != 3 or x.shape[2] != 3 or y.shape[2] != 3 or x.shape[0] != y.shape[0]:
        raise ValueError('chamfer_nn expects x [N,P1,3] and y [N,P2,3]')
    x = x.detach().contiguous()
    y = y.detach().contiguous()
    N, P1, _ = x.shape
    P2 = y.shape[1]
    idx1 = torch.empty(N, P1, device=x.device, dtype=torch.int32)
    idx2 = torch.empty(N, P2, device=x.device, dtype=torch.int32)
    _nn_search(x, y, None, None, idx1, idx2)
    return idx1, idx2


class _ChamferMeta(type(torch.nn.Module)):
    # `chd.ChamferDistance(x, y)` (train_helper.py:57) and `chd.ChamferDistance()(x, y)` (:55) both work
    def __call__(cls, *args, **kwargs):
        if args or kwargs:
            return _ChamferFn.apply(*args, **kwargs)
        return super().__call__()


class ChamferDistance(torch.nn.Module, metaclass=_ChamferMeta):
    """(x [N,P1,3], y [N,P2,3]) -> (dist1 [N,P1], dist2 [N,P2], idx1 int32, idx2 int32); squared distances."""

    def forward(self, x, y):
        return _ChamferFn.apply(x, y)
