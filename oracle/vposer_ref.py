"""CPU restatement of the VPoser v1.0 decoder used at ``WholeGraspPose/models/fittingop.py:65,142``
(``human_body_prior.tools.model_loader.load_vposer`` -> ``vposer.decode(z, output_type='aa')``).

TEST INFRASTRUCTURE.  PARITY UNPINNED: ``human_body_prior`` and the ``TR00_E096.pt`` snapshot
are absent (``README.md:57,75-78``).  Architecture per SURVEY.md §2-D3: 32 -> 512 -> 512 -> 126,
leaky-relu(0.2), dropout inactive in eval mode, 21 x 6D -> rotation matrix (Gram-Schmidt)
-> axis-angle through the same quaternion route as ``utils/utils.py:223-370``.
"""
import torch
import torch.nn.functional as F

from .rotations_ref import cont2rotmat, rotmat2aa


def vposer_decode(weights, z):
    """weights: dict w1[512,32] b1 w2[512,512] b2 w3[126,512] b3; z [B,32] -> [B,1,21,3]."""
    h = F.leaky_relu(F.linear(z, weights['w1'], weights['b1']), 0.2)
    h = F.leaky_relu(F.linear(h, weights['w2'], weights['b2']), 0.2)
    x = F.linear(h, weights['w3'], weights['b3'])
    R = cont2rotmat(x.reshape(-1, 6))
    return rotmat2aa(R).reshape(z.shape[0], 1, 21, 3)
