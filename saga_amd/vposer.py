"""VPoser v1.0 decoder as used at ``WholeGraspPose/models/fittingop.py:65,142``
(``human_body_prior``: ``vposer.decode(z, output_type='aa')``): 32 -> 512 -> 512 -> 126 MLP with
leaky-relu(0.2), 21 x 6D -> rotation matrix -> axis-angle.  This module is the autograd-path form (three
``nn.Linear`` + the fused ``saga_cont6d_to_aa`` kernel) and the holder of the weights; ``saga_fit_step`` runs the same
decoder inside the library.  The pretrained snapshot is not available, so weights are random-initialised unless a
state dict with the ``bodyprior_dec_*`` keys is loaded."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .rotations import cont6d_to_aa


class VPoserDecoder(nn.Module):
    def __init__(self, latentD=32, num_neurons=512, num_joints=21, seed=None):
        super().__init__()
        self.latentD, self.num_joints = latentD, num_joints
        gen = None if seed is None else torch.Generator().manual_seed(seed)
        self.bodyprior_dec_fc1 = nn.Linear(latentD, num_neurons)
        self.bodyprior_dec_fc2 = nn.Linear(num_neurons, num_neurons)
        self.bodyprior_dec_out = nn.Linear(num_neurons, num_joints * 6)
        if gen is not None:
            for lin in (self.bodyprior_dec_fc1, self.bodyprior_dec_fc2, self.bodyprior_dec_out):
                bound = 1.0 / lin.in_features ** 0.5
                with torch.no_grad():
                    lin.weight.copy_((torch.rand(lin.weight.shape, generator=gen) * 2 - 1) * bound)
                    lin.bias.copy_((torch.rand(lin.bias.shape, generator=gen) * 2 - 1) * bound)

    def decode(self, Zin, output_type='matrot'):
        assert output_type in ['matrot', 'aa']
        x = F.leaky_relu(self.bodyprior_dec_fc1(Zin), negative_slope=0.2)
        x = F.leaky_relu(self.bodyprior_dec_fc2(x), negative_slope=0.2)   # dropout is inactive in eval mode
        x = self.bodyprior_dec_out(x)
        # fused 6D -> axis-angle kernel (one launch instead of ~90); GPU only, like every op of the package
        aa = cont6d_to_aa(x.reshape(-1, 6))
        if output_type == 'aa':
            return aa.reshape(Zin.shape[0], 1, self.num_joints, 3)
        # 'matrot' is not on the GraspPose-Opt path (fittingop.py:142 asks for 'aa'): R(aa) on the Rodrigues kernel, equal to
        # the Gram-Schmidt matrix to rounding
        from .smplx.lbs import batch_rodrigues
        return batch_rodrigues(aa).reshape(Zin.shape[0], 1, self.num_joints, 9)

    def weights_dict(self, dtype=torch.float64):
        """Plain tensors for the CPU oracle (tests only)."""
        g = lambda t: t.detach().cpu().to(dtype)
        return {'w1': g(self.bodyprior_dec_fc1.weight), 'b1': g(self.bodyprior_dec_fc1.bias),
                'w2': g(self.bodyprior_dec_fc2.weight), 'b2': g(self.bodyprior_dec_fc2.bias),
                'w3': g(self.bodyprior_dec_out.weight), 'b3': g(self.bodyprior_dec_out.bias)}
