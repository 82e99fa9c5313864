"""Batched counterpart of the optimisation part of the reference's GraspMotion-Opt driver (``opt_graspmotion.py:456-561``).

The reference loops over the samples of an object, builds one ``FittingOP`` per sample (``:469-513``), decodes the
stage-1 and the final parameters to SMPL-X keyword dicts with ``gen_body_mesh_v1`` (``:515-523``) and saves
``fitting_results.npy`` (``:525-561``), which ``visualization/vis_motion.py`` consumes.  Here all samples are the S
sequences of ONE :class:`saga_amd.motion_fitting.FittingOP`; the dict returned (and saved) has the same keys, nesting and
array shapes ``[S, T, ...]``.  What precedes the optimisation in the reference (trajectory / local-motion CVAEs, GRAB
meshes) is outside this path: the caller hands over its outputs.
"""
import os

import numpy as np
import torch

from .motion_fitting import convert_to_3D_rot, gen_body_mesh_v1

BODY_KEYS = ['transl', 'global_orient', 'betas', 'body_pose', 'left_hand_pose', 'right_hand_pose']


def decode_params(fittingop, transl, rot_6d, shape, other):
    """[S,T,.] optimiser variables -> SMPL-X keyword dict of [S,T,.] numpy arrays (``:515-523``)."""
    S, T = transl.shape[:2]
    with torch.no_grad():
        p = convert_to_3D_rot(torch.cat([transl, rot_6d, shape, other], dim=-1).reshape(S * T, -1).detach())
        _, d = gen_body_mesh_v1(p, fittingop.smplx_model_batch, fittingop.vposer_model)
    return {k: d[k].detach().reshape(S, T, -1).cpu().numpy() for k in BODY_KEYS}


def motion_opt(fittingop, body_markers_rec, markers_end_new, rhand_verts, contact_lbl_rec, contact_object, contact_markers,
               normal_object, verts_object, object_name, object_transl, object_global_orient, save_dir=None):
    """Fit S sequences at once and return the ``saved_results`` dict of ``opt_graspmotion.py:549-556``:
    ``body_orig`` (stage 1) / ``body_opt`` (final): dicts of [S,T,.]; ``object_name``; ``object``: transl, global_orient."""
    res = fittingop.fitting(body_markers_rec, markers_end_new, rhand_verts, contact_lbl_rec, contact_object, contact_markers,
                            normal_object, verts_object)
    a = lambda x: np.asarray(x.detach().cpu().numpy() if torch.is_tensor(x) else x)
    out = {'body_orig': decode_params(fittingop, *res[:4]), 'body_opt': decode_params(fittingop, *res[4:]),
           'object_name': str(object_name),
           'object': {'transl': a(object_transl), 'global_orient': a(object_global_orient)}}
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
        np.save(os.path.join(save_dir, 'fitting_results'), out)            # same call as opt_graspmotion.py:561
    return out
