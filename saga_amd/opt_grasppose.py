"""Batched counterpart of the reference's GraspPose-Opt driver (``opt_grasppose.py:164-243``).

The reference fits ``n_samples x n_random_samples`` candidates one after the other at batch size 1
(``:211-230``) and saves ``fitting_results.npz`` (``:194-208,232-243``), which
``visualization/vis_pose.py:31-42`` and ``opt_graspmotion.py:30-53`` consume.  Here all candidates of an
object (or a shard of them on this rank) are ONE batch of :class:`saga_amd.fitting.FittingOP`; the file
written has the same keys, nesting, shapes and row order (sample-major, then random sample).
"""
import os

import numpy as np
import torch

BODY_KEYS = ['betas', 'transl', 'global_orient', 'body_pose', 'leye_pose', 'reye_pose', 'left_hand_pose', 'right_hand_pose']


def fitting_results_dict(markers, markers_fit, smplxparams, gender, object_contact, body_contact, object_name,
                         verts_object, global_orient_object, transf_transl_object):
    """The ``save_data_gen`` dict of ``opt_grasppose.py:194-243`` for N fitted candidates (plain numpy in, numpy out).

    markers, markers_fit [N,143,3]; smplxparams: dict of [N,.]; object_contact [N,O]; body_contact [N,143];
    verts_object [N,O,3]; global_orient_object, transf_transl_object [N,.]."""
    n = markers.shape[0]
    a = lambda x: np.asarray(x.detach().cpu().numpy() if torch.is_tensor(x) else x)
    return {
        'markers': a(markers).reshape(n, -1, 3),
        'markers_fit': a(markers_fit).reshape(n, -1, 3),
        'body': {k: a(smplxparams[k]).reshape(n, -1) for k in BODY_KEYS},
        'object': {'transl': a(transf_transl_object).reshape(n, -1), 'global_orient': a(global_orient_object).reshape(n, -1),
                   'verts_object': a(verts_object).reshape(n, -1, 3)},
        'contact': {'body': a(body_contact).reshape(n, -1), 'object': a(object_contact).reshape(n, -1)},
        'gender': [gender] * n,
        'object_name': [object_name] * n,
    }


def pose_opt(fittingop, samples_results, n_random_samples, obj, gender, save_dir=None, rank=0, world=1, gather=False):
    """``pose_opt`` of the reference for a batch: ``samples_results`` holds ``verts_object``/``normal_object``
    [n,O,3], ``global_orient_object``, ``transf_transl`` [n,.] and ``markers_gen`` [n*k,143,3],
    ``object_contact_gen`` [n*k,O], ``markers_contact_gen`` [n*k,143] (``:166-176``).  Candidate ``i*k + j`` is
    random sample j of object pose i.  With ``world > 1`` this rank fits its contiguous block of candidates
    (``saga_amd.distributed.shard_range``) and the returned dict holds that block only.
    ``fittingop.batch_size`` must equal the number of candidates fitted here.  ``gather=True`` (needs an initialised
    ``torch.distributed`` group when ``world > 1``): the per-candidate loss and parameters of every rank are all-gathered
    (``saga_amd.distributed.gather_candidates``, one fused buffer over RCCL), the returned dict holds ALL N candidates plus
    ``'loss'`` [N] and ``'best_of_n'`` (index of the best random sample of every object pose); rank 0 writes the file."""
    from .distributed import shard_range
    dev = fittingop.device
    t = lambda k: torch.as_tensor(np.asarray(samples_results[k]), dtype=torch.float32)
    n = t('verts_object').shape[0]
    k = n_random_samples
    rep = lambda x: x.repeat_interleave(k, dim=0)                       # object pose i serves candidates i*k .. i*k+k-1
    verts_o, normals_o = rep(t('verts_object')), rep(t('normal_object'))
    go, tt = rep(t('global_orient_object')), rep(t('transf_transl'))
    markers = t('markers_gen')
    N = markers.shape[0]
    if N != n * k:
        raise ValueError('markers_gen has %d rows, expected n_samples * n_random_samples = %d' % (N, n * k))
    oc = t('object_contact_gen').reshape(N, -1)
    mc = t('markers_contact_gen').reshape(N, -1)
    lo, hi = shard_range(N, rank, world)
    if fittingop.batch_size != hi - lo:
        raise ValueError('FittingOP was built for batch_size=%d, this rank fits %d candidates' % (fittingop.batch_size, hi - lo))
    sl = slice(lo, hi)
    d = lambda x: x[sl].to(dev)
    markers_fit, smplxparams, _loss = fittingop.fitting(d(markers), d(oc), d(mc), d(verts_o), d(normals_o), gender)
    if gather:
        from .distributed import best_of_n, gather_candidates
        loss = fittingop.last_fit_loss
        p = dict(smplxparams[-1])
        # the fitted markers ride along in the fused buffer as one more 'parameter' block of body_pose-like rows
        mf = torch.as_tensor(markers_fit[-1], dtype=torch.float32, device=dev).reshape(hi - lo, -1)
        p['body_pose'] = torch.cat([p['body_pose'].reshape(hi - lo, -1), mf], dim=1)
        gl, gp = gather_candidates(loss, p, n_total=N)
        mf_all = gp['body_pose'][:, 63:].reshape(N, -1, 3)
        gp['body_pose'] = gp['body_pose'][:, :63]
        out = fitting_results_dict(markers, mf_all, gp, gender, oc, mc, obj, verts_o, go, tt)
        if save_dir is not None and rank == 0:
            os.makedirs(save_dir, exist_ok=True)
            np.savez(os.path.join(save_dir, 'fitting_results.npz'), **out)
        out['loss'] = gl.detach().cpu().numpy()
        out['best_of_n'] = best_of_n(gl, gp, k)[0].cpu().numpy()
        return out
    out = fitting_results_dict(markers[sl], markers_fit[-1], smplxparams[-1], gender, oc[sl], mc[sl], obj, verts_o[sl], go[sl], tt[sl])
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
        name = 'fitting_results.npz' if world == 1 else 'fitting_results_rank%d.npz' % rank
        np.savez(os.path.join(save_dir, name), **out)                   # same call as opt_grasppose.py:243
    return out
