"""HIP-backed drop-in for ``chamfer_distance`` (github otaheri/chamfer_distance) as the reference
calls it at ``utils/train_helper.py:57``: ``chd.ChamferDistance(x, y)`` as a plain callable
(the instantiate-then-call form, commented out at ``:55``, also works)."""
from .chamfer import ChamferDistance, chamfer_nn

__all__ = ['ChamferDistance', 'chamfer_nn']
