"""HIP-backed drop-in for the ``smplx`` package surface the reference uses
(``create``, the SMPL-X module, ``SMPLXOutput``, ``lbs.batch_rodrigues``)."""
from . import lbs
from .body_models import SMPLX, SMPLXOutput, VERTEX_IDS_SMPLX, create
from .lbs import batch_rodrigues

__all__ = ['create', 'SMPLX', 'SMPLXOutput', 'lbs', 'batch_rodrigues', 'VERTEX_IDS_SMPLX']
