"""ctypes binding of ``libsaga_hip.so`` (declared in ``include/saga_hip.h``).

The library is the only compute path of this package: if it is missing, or the tensors are
not on a ROCm device, the ops raise -- there is no CPU or eager-PyTorch fallback.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SAGA_HIP_LIB') or os.path.join(_HERE, 'lib', 'libsaga_hip.so')   # override: A/B builds

c_float_p = ctypes.c_void_p
c_int_p = ctypes.c_void_p
c_stream = ctypes.c_void_p

NUM_JOINTS = 55
FEAT_K = 512
N_ALIGN = 128
BWD_SPLITK = 16
DA_STRIDE = 840
NOALLOC_MAX_B = 64
NOALLOC_MAX_NPAD = 31488


class ModelDesc(ctypes.Structure):
    _fields_ = [
        ('n_verts', ctypes.c_int32), ('n_shape', ctypes.c_int32), ('n_faces', ctypes.c_int32),
        ('n_landmarks', ctypes.c_int32), ('n_extra_joints', ctypes.c_int32),
        ('v_template_host', ctypes.c_void_p), ('shapedirs_host', ctypes.c_void_p),
        ('posedirs_host', ctypes.c_void_p), ('J_regressor_host', ctypes.c_void_p),
        ('parents_host', ctypes.c_void_p), ('lbs_weights_host', ctypes.c_void_p),
        ('faces_host', ctypes.c_void_p), ('lmk_faces_idx_host', ctypes.c_void_p),
        ('lmk_bary_host', ctypes.c_void_p), ('extra_joint_verts_host', ctypes.c_void_p),
    ]


class GraspTables(ctypes.Structure):
    _fields_ = [('n_markers', ctypes.c_int32), ('n_hand', ctypes.c_int32), ('n_foot', ctypes.c_int32),
                ('n_body_markers', ctypes.c_int32), ('n_unique', ctypes.c_int32),
                ('marker_ids', ctypes.c_void_p), ('hand_ids', ctypes.c_void_p), ('foot_ids', ctypes.c_void_p),
                ('hand_weights', ctypes.c_void_p), ('unique_ids', ctypes.c_void_p), ('marker_slot', ctypes.c_void_p),
                ('hand_slot', ctypes.c_void_p), ('foot_slot', ctypes.c_void_p),
                ('n_ring', ctypes.c_int32), ('ring_ids', ctypes.c_void_p),
                ('hand_tri_ptr', ctypes.c_void_p), ('hand_tri', ctypes.c_void_p), ('hand_prior_w', ctypes.c_void_p)]


class PickSet(ctypes.Structure):
    """saga_pick_set_t"""
    _fields_ = [('pos', ctypes.c_void_p), ('inv_ptr', ctypes.c_void_p), ('inv_idx', ctypes.c_void_p), ('out', ctypes.c_void_p),
                ('g_out', ctypes.c_void_p), ('n', ctypes.c_int32), ('t0', ctypes.c_int32)]


PICK_MAX_SETS = 8


class MotionTermsCfg(ctypes.Structure):
    """saga_motion_terms_cfg_t"""
    _fields_ = [('n_seq', ctypes.c_int32), ('n_frames', ctypes.c_int32), ('n_tail', ctypes.c_int32), ('n_markers', ctypes.c_int32),
                ('foot_group_size', ctypes.c_int32 * 4), ('n_hand', ctypes.c_int32), ('n_obj', ctypes.c_int32),
                ('n_contact_markers', ctypes.c_int32), ('n_palm', ctypes.c_int32), ('n_last_frames', ctypes.c_int32),
                ('vel_thres', ctypes.c_float)]


MOTION_NTERMS = 12


class MotionTables(ctypes.Structure):
    """saga_motion_tables_t"""
    _fields_ = [('sets', PickSet * 4), ('normal_sel', ctypes.c_void_p),
                ('marker_mask', ctypes.c_void_p)]


class MotionProblem(ctypes.Structure):
    """saga_motion_problem_t"""
    _fields_ = [(k, ctypes.c_void_p) for k in ('markers_rec', 'markers_end', 'contact_lbl', 'contact_object', 'contact_markers',
                                               'verts_object', 'normal_object')]


MOTION_WEIGHTS = ('rec_markers', 'end_markers_fit', 'vposer', 'hand', 'skating', 'hand_smooth', 'collision', 'contact', 'hand_angle')
MOTION_STEP_TERMS = ('total', 'rec_markers', 'end_markers_fit', 'skating', 'hand_smooth', 'contact', 'collision', 'hand_angle',
                     'vposer', 'hand')                       # columns 0..9 of saga_motion_step's terms_out [S,16]


class MotionStepCfg(ctypes.Structure):
    """saga_motion_step_cfg_t"""
    _fields_ = ([('terms', MotionTermsCfg)] + [('w_' + k, ctypes.c_float) for k in MOTION_WEIGHTS] +
                [('adam_step', ctypes.c_int32), ('lr', ctypes.c_float), ('beta1', ctypes.c_float), ('beta2', ctypes.c_float),
                 ('eps', ctypes.c_float)])

GRASP_NTERMS = 16

# packed optimiser-variable row (include/saga_hip.h SAGA_VAR_*)
VAR_STRIDE = 112
VAR_SLICES = {'vpose_rec': (0, 32), 'hand_pose': (32, 80), 'eye_pose': (80, 86), 'betas': (86, 96),
              'transl_rec': (96, 99), 'glo_rot_rec': (99, 105)}
VAR_GROUPS = {'vpose_rec': 1, 'hand_pose': 2, 'eye_pose': 4, 'betas': 8, 'transl_rec': 16, 'glo_rot_rec': 32}
BODY_PARAM_COUNT = 133
BODY_PARAM_SLICES = {'transl': (0, 3), 'global_orient': (3, 6), 'betas': (6, 16), 'body_pose': (16, 79),
                     'left_hand_pose': (79, 103), 'right_hand_pose': (103, 127), 'leye_pose': (127, 130),
                     'reye_pose': (130, 133)}


class DecoderDesc(ctypes.Structure):
    _fields_ = [('latent', ctypes.c_int32), ('hidden', ctypes.c_int32), ('n_body_joints', ctypes.c_int32),
                ('n_pca', ctypes.c_int32),
                ('w1_host', ctypes.c_void_p), ('b1_host', ctypes.c_void_p), ('w2_host', ctypes.c_void_p),
                ('b2_host', ctypes.c_void_p), ('w3_host', ctypes.c_void_p), ('b3_host', ctypes.c_void_p),
                ('lhand_components_host', ctypes.c_void_p), ('rhand_components_host', ctypes.c_void_p),
                ('pose_mean_host', ctypes.c_void_p)]


class FitProblem(ctypes.Structure):
    _fields_ = [('obj', ctypes.c_void_p), ('obj_normals', ctypes.c_void_p), ('markers_target', ctypes.c_void_p),
                ('c_markers', ctypes.c_void_p), ('c_object', ctypes.c_void_p), ('n_obj', ctypes.c_int32)]


class FitStepCfg(ctypes.Structure):
    _fields_ = [('stage', ctypes.c_int32), ('only_rec', ctypes.c_int32), ('prior_contact', ctypes.c_int32),
                ('update_mask', ctypes.c_int32), ('adam_step', ctypes.c_int32),
                ('lr', ctypes.c_float), ('beta1', ctypes.c_float), ('beta2', ctypes.c_float), ('eps', ctypes.c_float),
                ('skip_gated_terms', ctypes.c_int32), ('custom_weights', ctypes.c_int32),
                ('w_rec', ctypes.c_float), ('w_foot', ctypes.c_float), ('reg_vpose', ctypes.c_float), ('reg_hand', ctypes.c_float),
                ('reg_eye', ctypes.c_float), ('reg_shape', ctypes.c_float)]

# name -> (restype, argtypes); every symbol include/saga_hip.h declares
PROTOTYPES = {
    'saga_last_error': (ctypes.c_char_p, []),
    'saga_version': (ctypes.c_int, []),
    'saga_model_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ModelDesc)]),
    'saga_model_create_subset': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_int]),
    'saga_model_num_faces': (ctypes.c_int, [ctypes.c_void_p]),
    'saga_model_destroy': (None, [ctypes.c_void_p]),
    'saga_model_npad': (ctypes.c_int, [ctypes.c_void_p]),
    'saga_model_num_joints_out': (ctypes.c_int, [ctypes.c_void_p]),
    'saga_model_weight_nnz': (ctypes.c_int, [ctypes.c_void_p]),
    'saga_smplx_fwd': (ctypes.c_int, [ctypes.c_void_p] + [c_float_p] * 11 + [ctypes.c_int, c_stream]),
    'saga_smplx_fwd_selected': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p] + [c_float_p] * 13 + [ctypes.c_int, c_stream]),
    'saga_smplx_bwd': (ctypes.c_int, [ctypes.c_void_p] + [c_float_p] * 15 + [ctypes.c_int, c_stream]),
    'saga_smplx_bwd_noalloc': (ctypes.c_int, [ctypes.c_void_p] + [c_float_p] * 11 + [ctypes.c_int, c_stream]),
    'saga_rodrigues_fwd': (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, c_stream]),
    'saga_rodrigues_bwd': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, ctypes.c_int, c_stream]),
    'saga_cont6d_to_aa_fwd': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, ctypes.c_int, c_stream]),
    'saga_cont6d_to_aa_bwd': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, ctypes.c_int, c_stream]),
    'saga_chamfer_fwd': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_int_p, c_int_p,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    'saga_chamfer_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'saga_chamfer_fwd_ws': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_int_p, c_int_p,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, c_stream]),
    'saga_chamfer_bwd': (ctypes.c_int, [c_float_p, c_float_p, c_int_p, c_int_p, c_float_p, c_float_p,
                                        c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    'saga_p2p_signed_fwd': (ctypes.c_int, [c_float_p] * 6 + [c_int_p, c_int_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, c_stream]),
    'saga_p2p_signed_fwd_ws': (ctypes.c_int, [c_float_p] * 6 + [c_int_p, c_int_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, c_stream]),
    'saga_p2p_signed_bwd': (ctypes.c_int, [c_float_p, c_float_p, c_int_p, c_int_p] + [c_float_p] * 8 + [ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, c_stream]),
    'saga_pick_sets_fwd': (ctypes.c_int, [c_float_p, ctypes.POINTER(PickSet), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    'saga_pick_sets_bwd': (ctypes.c_int, [ctypes.POINTER(PickSet), ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    'saga_motion_step_workspace_floats': (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(MotionStepCfg)]),
    'saga_motion_step_workspace_offset': (ctypes.c_longlong, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                              ctypes.POINTER(MotionStepCfg), ctypes.c_char_p]),
    'saga_motion_step': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(MotionTables),
                                        ctypes.POINTER(MotionProblem), ctypes.POINTER(MotionStepCfg)] + [c_float_p] * 6 + [c_stream]),
    'saga_params_workspace_floats': (ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_int]),
    'saga_params_fwd': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, c_stream]),
    'saga_params_bwd': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, c_float_p,
                                       ctypes.c_int, c_stream]),
    'saga_motion_terms_fwd': (ctypes.c_int, [ctypes.POINTER(MotionTermsCfg)] + [c_float_p] * 14 + [c_stream]),
    'saga_motion_terms_bwd': (ctypes.c_int, [ctypes.POINTER(MotionTermsCfg)] + [c_float_p] * 21 + [c_stream]),
    'saga_vertex_normals': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, ctypes.c_int, c_stream]),
    'saga_vertex_normals_subset': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_int_p, ctypes.c_int, c_float_p,
                                                  ctypes.c_int, c_stream]),
    'saga_vertex_normals_subset_window': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_int_p, ctypes.c_int, c_float_p,
                                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, c_stream]),
    'saga_grasp_loss_fwd': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(GraspTables)] + [ctypes.c_void_p] * 16 +
                            [ctypes.c_int, ctypes.c_int, c_stream]),
    'saga_grasp_loss_bwd': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(GraspTables)] + [ctypes.c_void_p] * 14 +
                            [ctypes.c_int, ctypes.c_int, c_stream]),
    'saga_decoder_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(DecoderDesc)]),
    'saga_decoder_destroy': (None, [ctypes.c_void_p]),
    'saga_fit_workspace_floats': (ctypes.c_size_t, [ctypes.c_void_p, ctypes.POINTER(GraspTables), ctypes.c_int, ctypes.c_int]),
    'saga_fit_workspace_offset': (ctypes.c_longlong, [ctypes.c_void_p, ctypes.POINTER(GraspTables), ctypes.c_int, ctypes.c_int,
                                                      ctypes.c_char_p]),
    'saga_fit_step': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(GraspTables), ctypes.POINTER(FitProblem),
                                     ctypes.POINTER(FitStepCfg)] + [ctypes.c_void_p] * 7 + [ctypes.c_int, c_stream]),
    'saga_fit_step_sparse': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(GraspTables),
                                            ctypes.POINTER(FitProblem), ctypes.POINTER(FitStepCfg)] + [ctypes.c_void_p] * 7 +
                             [ctypes.c_int, c_stream]),
    'saga_fit_decode': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, c_stream]),
    'saga_fit_best_update': (ctypes.c_int, [c_float_p] * 6 + [ctypes.c_int, ctypes.c_int, c_stream]),
    'saga_profile_enable': (ctypes.c_int, [ctypes.c_int]),
    'saga_profile_read': (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]),
    'saga_smplx_fwd_is_persistent': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'saga_smplx_bwd_is_persistent': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
}

PROF_KERNELS = ['pose_fwd', 'blend_gemm_fwd', 'skin_fwd', 'joints', 'skin_bwd', 'blend_gemm_bwd', 'pose_bwd',
                'chamfer_nn', 'vertex_normals', 'grasp_loss_fwd', 'grasp_loss_bwd', 'params_fwd', 'loss_sum', 'params_bwd',
                'adam']

_lib = None


def load():
    """Load the shared library once; raise a clear error when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'saga_amd: %s not found. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(or `make -C saga_amd/csrc`). There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.saga_version() <= 0 and not os.environ.get('SAGA_ALLOW_EXPERIMENT_LIB'):
        # -DSAGA_EXPERIMENT_BUILD: the timing experiments of scripts/exp, several of which give wrong results by design
        raise RuntimeError('saga_amd: %s is an EXPERIMENT build (saga_version() = %d): not a library to compute with. '
                           'scripts/exp sets SAGA_ALLOW_EXPERIMENT_LIB=1 to load one.' % (LIB_PATH, lib.saga_version()))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().saga_last_error()
        raise RuntimeError('%s failed (code %d): %s' % (what, rc, msg.decode() if msg else '?'))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream(device=None):
    """hipStream_t of torch's current stream ON ``device`` (default: the current device)."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def call(name, device, *args):
    """Run entry point ``name`` with ``device`` current and torch's current stream of THAT device appended as the last
    argument -- the kernels launch where the tensors live, whatever the process's current device is (the library checks
    its model / decoder handles against the current device and returns SAGA_EINVAL on a mismatch)."""
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('saga_amd ops run on the GPU only (device %s); there is no CPU fallback' % device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    fn = getattr(load(), name)
    with torch.cuda.device(device):
        check(fn(*args, stream(device)), name)


def require_device(*tensors):
    """Every tensor on ONE ROCm device, float32 / int32; -> that device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('saga_amd ops run on the GPU only (got a %s tensor); there is no CPU fallback' % t.device)
        if t.dtype not in (torch.float32, torch.int32):
            raise TypeError('saga_amd ops take float32 / int32 tensors, got %s' % t.dtype)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError('saga_amd ops take tensors of one device, got %s and %s' % (dev, t.device))
    return dev


def profile_enable(on=True):
    """True / 1: bracket every kernel; 2: only the blend GEMM launches; False / 0: off."""
    check(load().saga_profile_enable(int(on)), 'saga_profile_enable')


def profile_read():
    """-> {kernel name: (total_ms, launches)} since the last read (synchronises the recorded events)."""
    n = len(PROF_KERNELS)
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_int32 * n)()
    check(load().saga_profile_read(ms, cnt), 'saga_profile_read')
    return {PROF_KERNELS[i]: (ms[i], cnt[i]) for i in range(n)}
