#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference's own Python
modules from /root/reference (read-only) in the build container.  Run once, here:

    python tests/golden/make_golden.py

Nothing from the reference travels to the GPU box: only the .npz data written below does.
What is imported (these imports work in this container):

* ``utils.utils.RotConverter``            -> rotations.npz   (cont2rotmat, rotmat2aa, grads)
* ``utils.train_helper.point2point_signed`` on top of a brute-force ``chamfer_distance`` stub
  module (the CUDA extension is absent)  -> p2p_signed.npz  (outputs + grads w.r.t. x)
* ``WholeGraspPose.models.models.FullBodyGraspNet`` (random-init, torch seed 4815 =
  ``WholeGraspPose.yaml:17``)             -> vae_samples.npz (markers / contact maps used as
  optimiser inputs for config 1; NOT a parity target, just realistic inputs)

Index arrays the hot path reads (``WholeGraspPose/models/fittingop.py:86-124``) are derived with the
same Python expressions the reference uses and written to ``grasp_indices.npz`` (also copied to
``saga_amd/data/`` because the product's FittingOP needs them when the reference tree is absent):
markers_143, rhand (778), lhand (778), foot (172; CPython ``set`` iteration order, quirk a13q-iii),
rhand_weight.  ``MANO_SMPLX_vertex_ids.pkl`` is NOT unpickled: its two int64[778] payloads are cut
out of the opcode stream with ``pickletools.genops`` (a disassembler; executes nothing).
"""
import json
import os
import pickletools
import shutil
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def brute_force_chamfer(x, y):
    """Stand-in for the absent CUDA extension: squared distances, lowest-index arg-min."""
    d = ((x[:, :, None, :] - y[:, None, :, :]) ** 2)
    d = (d[..., 0] + d[..., 1]) + d[..., 2]
    n2 = d.shape[2]
    n1 = d.shape[1]
    m1 = d.min(2).values
    i1 = torch.where(d == m1[:, :, None], torch.arange(n2).view(1, 1, -1), n2).min(2).values
    m2 = d.min(1).values
    i2 = torch.where(d == m2[:, None, :], torch.arange(n1).view(1, -1, 1), n1).min(1).values
    return m1, m2, i1.to(torch.int32), i2.to(torch.int32)


def import_reference():
    sys.path.insert(0, REF)
    stub = types.ModuleType('chamfer_distance')
    stub.ChamferDistance = brute_force_chamfer
    sys.modules['chamfer_distance'] = stub
    from utils.utils import RotConverter
    from utils.train_helper import point2point_signed
    return RotConverter, point2point_signed


def gen_rotations(RotConverter):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(63, 6, generator=g, dtype=torch.float32)
    x = torch.cat([torch.tensor([[-1., 0, 0, 0, 0, 1]]), x], 0)          # fittingop.py:73 init
    x.requires_grad_(True)
    R = RotConverter.cont2rotmat(x)
    aa = RotConverter.rotmat2aa(R)
    w = torch.randn(aa.shape, generator=g)
    (aa * w).sum().backward()
    x64 = x.detach().double()
    aa64 = RotConverter.rotmat2aa(RotConverter.cont2rotmat(x64))
    np.savez(os.path.join(HERE, 'rotations.npz'), x6d=x.detach().numpy(), R=R.detach().numpy(),
             aa=aa.detach().numpy(), w=w.numpy(), grad_x6d=x.grad.numpy(), aa64=aa64.numpy())


def gen_p2p(point2point_signed):
    g = torch.Generator().manual_seed(5)
    out = {}
    for name, P1 in (('markers', 143), ('rhand', 778)):
        x = (torch.randn(2, P1, 3, generator=g) * 0.05).requires_grad_(True)
        y = torch.randn(2, 2048, 3, generator=g) * 0.05
        xn = torch.nn.functional.normalize(torch.randn(2, P1, 3, generator=g), dim=-1)
        yn = torch.nn.functional.normalize(torch.randn(2, 2048, 3, generator=g), dim=-1)
        y2x, x2y, yidx, xidx = point2point_signed(x, y, xn, yn)
        wy = torch.randn(y2x.shape, generator=g)
        wx = torch.randn(x2y.shape, generator=g)
        ((y2x * wy).sum() + (x2y * wx).sum()).backward()
        out.update({f'{name}_x': x.detach().numpy(), f'{name}_y': y.numpy(), f'{name}_xn': xn.numpy(),
                    f'{name}_yn': yn.numpy(), f'{name}_y2x': y2x.detach().numpy(),
                    f'{name}_x2y': x2y.detach().numpy(), f'{name}_yidx': yidx.numpy(),
                    f'{name}_xidx': xidx.numpy(), f'{name}_wy': wy.numpy(), f'{name}_wx': wx.numpy(),
                    f'{name}_grad_x': x.grad.numpy()})
        # the marker call of fittingop.py:183: normal_object positionally as x_normals, no y_normals
        if name == 'markers':
            x2 = x.detach().clone().requires_grad_(True)
            nq = torch.nn.functional.normalize(torch.randn(2, 2048, 3, generator=g), dim=-1)
            y2x_q, x2y_q, _, _ = point2point_signed(x2, y, nq)
            out.update({'markers_quirk_normals': nq.numpy(), 'markers_quirk_y2x': y2x_q.detach().numpy(),
                        'markers_quirk_x2y': x2y_q.detach().numpy()})
    np.savez(os.path.join(HERE, 'p2p_signed.npz'), **out)


def gen_indices():
    with open(os.path.join(REF, 'body_utils/smplx_markerset.json')) as f:
        markerset = json.load(f)['markersets']
    markers_143 = []
    for marker in markerset:                                     # fittingop.py:92-95
        if marker['type'] not in ['palm_5']:
            markers_143 += list(marker['indices'].values())
    markers_f0_p5 = []                                           # utils_body.py:311-329 with 'f0_p5' (opt_graspmotion.py:408)
    for marker in markerset:
        if marker['type'] == 'finger':
            continue
        if 'palm' in marker['type']:
            if marker['type'] == 'palm_5':
                markers_f0_p5 += list(marker['indices'].values())
            continue
        markers_f0_p5 += list(marker['indices'].values())
    raw = open(os.path.join(REF, 'body_utils/smplx_mano_flame_correspondences/MANO_SMPLX_vertex_ids.pkl'), 'rb').read()
    hands, key = {}, None
    for op, arg, _ in pickletools.genops(raw):
        if op.name == 'BINUNICODE' and arg in ('left_hand', 'right_hand'):
            key = arg
        elif op.name == 'BINUNICODE' and key is not None and len(arg) == 778 * 8:
            hands[key] = np.frombuffer(arg.encode('latin1'), dtype='<i8').copy()
            key = None
    assert hands['right_hand'].shape == (778,) and hands['left_hand'].shape == (778,)
    foot = {}
    for side, name in (('L', 'left'), ('R', 'right')):
        with open(os.path.join(REF, 'body_utils/body_segments', f'{side}_Leg.json')) as f:
            data = json.load(f)
        ids = np.asarray(list(set(data['verts_ind'])))           # fittingop.py:106,116 (set order!)
        foot[f'{name}_heel'] = ids[np.load(os.path.join(REF, f'body_utils/{name}_heel_verts_id.npy'))]
        foot[f'{name}_toe'] = ids[np.load(os.path.join(REF, f'body_utils/{name}_toe_verts_id.npy'))]
    foot_all = np.concatenate([foot['right_heel'], foot['right_toe'], foot['left_heel'], foot['left_toe']])  # :124
    w = np.load(os.path.join(REF, 'WholeGraspPose/configs/rhand_weight.npy'))
    path = os.path.join(HERE, 'grasp_indices.npz')
    np.savez(path, markers_143=np.asarray(markers_143, dtype=np.int64),
             markers_f0_p5=np.asarray(markers_f0_p5, dtype=np.int64), rhand=hands['right_hand'],
             lhand=hands['left_hand'], foot=foot_all.astype(np.int64), rhand_weight=w,
             foot_sizes=np.asarray([len(foot['right_heel']), len(foot['right_toe']),
                                    len(foot['left_heel']), len(foot['left_toe'])]))
    os.makedirs(os.path.join(REPO, 'saga_amd', 'data'), exist_ok=True)
    shutil.copy(path, os.path.join(REPO, 'saga_amd', 'data', 'grasp_indices.npz'))
    print('indices:', len(markers_143), hands['right_hand'][:3], foot_all.shape, w.shape)


def gen_vae_samples():
    sys.path.insert(0, REF)
    from utils.cfg_parser import Config
    from WholeGraspPose.models.models import FullBodyGraspNet
    sys.path.insert(0, REPO)
    from saga_amd.synthetic import make_object_cloud
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        cfg = Config(default_cfg_path='WholeGraspPose/configs/WholeGraspPose.yaml')
    finally:
        os.chdir(cwd)
    torch.manual_seed(4815)
    np.random.seed(0)
    net = FullBodyGraspNet(cfg).eval()
    n = 10
    verts, normals, height, yaw = make_object_cloud(n, seed=0)
    # object-centred cloud + [normal(3), rot6d(6)] features, as opt_grasppose.py:44-52
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.zeros((n, 3, 3), np.float32)
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = c, -s, s, c, 1
    centred = verts.copy()
    centred[:, :, 2] -= height[:, None]
    rot6 = R.reshape(n, 1, 9)[:, :, :6].repeat(2048, axis=1)
    feat = np.concatenate([normals, rot6], axis=-1)
    transl = np.zeros((n, 3), np.float32)
    transl[:, 2] = height
    with torch.no_grad():
        mk, mc, oc = net.sample(torch.from_numpy(centred).permute(0, 2, 1),
                                torch.from_numpy(feat).permute(0, 2, 1), torch.from_numpy(transl))
    markers = mk.numpy() + transl[:, None]                       # opt_grasppose.py:111
    np.savez(os.path.join(HERE, 'vae_samples.npz'), markers_gen=markers.astype(np.float32),
             markers_contact_gen=mc.numpy().reshape(n, -1).astype(np.float32),
             object_contact_gen=oc.numpy().reshape(n, -1).astype(np.float32),
             verts_object=verts, normal_object=normals, transf_transl=transl, yaw=yaw)
    print('vae:', markers.shape, mc.shape, oc.shape)


if __name__ == '__main__':
    RotConverter, p2p = import_reference()
    gen_rotations(RotConverter)
    gen_p2p(p2p)
    gen_indices()
    gen_vae_samples()
    print('golden fixtures written to', HERE)
