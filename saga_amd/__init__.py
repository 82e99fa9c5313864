"""saga_amd -- MI355X-native GraspPose-Opt hot path (SMPL-X fwd/bwd + chamfer NN + contact losses).

Sub-packages mirror the operator packages the reference imports:
  saga_amd.smplx             ~ smplx             (create / SMPLX / lbs.batch_rodrigues)
  saga_amd.chamfer_distance  ~ chamfer_distance  (ChamferDistance)
  saga_amd.geometry          ~ utils/train_helper.point2point_signed + Open3D vertex normals
  saga_amd.rotations         ~ utils/utils.RotConverter: rotmat2aa(cont2rotmat(x)) as one HIP op, cont6d_to_aa(x)
  saga_amd.vposer            ~ human_body_prior VPoser v1 decoder
  saga_amd.fitting           ~ WholeGraspPose/models/fittingop.FittingOP (batched)
All arithmetic of the hot path runs in saga_amd/lib/libsaga_hip.so (HIP, gfx950); there is no CPU fallback.
"""
import os
import sys

__version__ = '0.1.0'


def install_dropins():
    """Make ``import smplx`` / ``import chamfer_distance`` resolve to the HIP-backed packages."""
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dropin')
    if d not in sys.path:
        sys.path.insert(0, d)
    return d
