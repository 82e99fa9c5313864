/*
 * saga_hip.h -- C-ABI of libsaga_hip.so: the MI355X (gfx950) kernels behind the GraspPose-Opt hot path.
 *
 * The reference (JiahaoPlus/SAGA) is pure Python; its FFI for this path is the pair of
 * third-party operator packages it imports, `smplx` and `chamfer_distance` (plus Open3D's
 * vertex normals).  Each entry point below names the reference interface it replaces.
 * Host side above this ABI: saga_amd/ (Python, ctypes) mirroring those packages.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; tensors are dense,
 *     row-major, FP32 (indices int32) -- the caller (PyTorch) owns and allocates all buffers;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it, nothing synchronises;
 *   - return 0 on success, a negative SAGA_E* code for argument errors, or a positive hipError_t;
 *     saga_last_error() returns a static, thread-local message for the last failure;
 *   - no entry point allocates device memory except saga_model_create (model constants).
 */
#ifndef SAGA_HIP_H
#define SAGA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAGA_EINVAL   (-1)   /* bad argument / shape */
#define SAGA_ENOMEM   (-2)
#define SAGA_EDEVICE  (-3)   /* wrong architecture (not gfx950) */

#define SAGA_NUM_JOINTS      55
#define SAGA_FEAT_K          512     /* padded K of the blend contraction: n_shape + 486 -> 512 */
#define SAGA_N_ALIGN         128     /* v_posed rows are padded to a multiple of this many floats */
#define SAGA_BWD_SPLITK      16      /* split-K slabs written by saga_smplx_bwd's blend GEMM */
#define SAGA_DA_STRIDE       840     /* floats per candidate of ws_dA: 56*12 dA + 55*3 g_posed_joints + 3 */

const char* saga_last_error(void);
int saga_version(void);

/* ------------------------------------------------------------------------------------------
 * Body model constants.  Replaces the buffers `smplx.create(...)` registers on the module
 * (reference call site WholeGraspPose/models/fittingop.py:30-59; utils/utils_body.py:115-130).
 * All pointers are HOST pointers; the arrays are copied (and re-laid-out) to the current device.
 * ------------------------------------------------------------------------------------------ */
typedef struct saga_model saga_model_t;

typedef struct {
    int32_t n_verts;          /* V (10475 for SMPL-X) */
    int32_t n_shape;          /* S = betas + expression coefficients (20) */
    int32_t n_faces;          /* F */
    int32_t n_landmarks;      /* L (51) */
    int32_t n_extra_joints;   /* E (21 vertex-picked joints) */
    const float*   v_template_host;     /* [V,3] */
    const float*   shapedirs_host;      /* [V,3,S] */
    const float*   posedirs_host;       /* [486, V*3] */
    const float*   J_regressor_host;    /* [55, V] */
    const int32_t* parents_host;        /* [55], parents[0] = -1, parents[i] < i */
    const float*   lbs_weights_host;    /* [V,55] */
    const int32_t* faces_host;          /* [F,3] */
    const int32_t* lmk_faces_idx_host;  /* [L] */
    const float*   lmk_bary_host;       /* [L,3] */
    const int32_t* extra_joint_verts_host; /* [E] */
} saga_model_desc_t;

int  saga_model_create(saga_model_t** out, const saga_model_desc_t* desc);
void saga_model_destroy(saga_model_t* m);
/* padded row length (floats) of v_posed / grad_v_posed buffers: roundup(3V, SAGA_N_ALIGN) */
int  saga_model_npad(const saga_model_t* m);
/* number of output joints: 55 + E + L (127 for SMPL-X) */
int  saga_model_num_joints_out(const saga_model_t* m);
/* entries per vertex of the compressed LBS-weight table (max non-zeros per row) */
int  saga_model_weight_nnz(const saga_model_t* m);

/* ------------------------------------------------------------------------------------------
 * SMPL-X forward.  Replaces `smplx` SMPLX.forward -> lbs() (reference call site
 * WholeGraspPose/models/fittingop.py:149; algorithm SURVEY.md section 3.2) from `full_pose`
 * onwards: Rodrigues, shape+pose blend shapes (one FP32-MFMA contraction), joint regression,
 * 55-joint rigid chain, linear-blend skinning, extra joints + landmarks, translation.
 *   shape      [B,S]     betas ++ expression
 *   full_pose  [B,165]   axis-angle, pose_mean already added
 *   transl     [B,3] or NULL
 *   vertices   [B,V,3]   out
 *   joints     [B,55+E+L,3] out
 * saved for backward (all out, caller-allocated):
 *   feat    [B,512]      [shape | vec(R_j - I), j=1..54 | 0]
 *   rot     [B,55,9]     R_j
 *   jrest   [B,55,3]     regressed rest joints J_j
 *   chain   [B,55,12]    world transforms G_j = [Rg | t] (3x4 row-major)
 *   A       [B,55,12]    skinning transforms
 *   v_posed [B,Npad]     v_template + blend offsets (pad columns are written as 0)
 * ------------------------------------------------------------------------------------------ */
int saga_smplx_fwd(const saga_model_t* m, const float* shape, const float* full_pose, const float* transl,
                   float* vertices, float* joints,
                   float* feat, float* rot, float* jrest, float* chain, float* A, float* v_posed,
                   int B, void* stream);

/* SMPL-X backward (what loss.backward() drives through smplx at fittingop.py:383).
 *   g_vertices [B,V,3] or NULL; g_joints [B,55+E+L,3] or NULL (at least one non-NULL)
 *   g_shape [B,S], g_full_pose [B,165], g_transl [B,3] : out
 * workspace (caller-allocated, contents undefined on entry):
 *   ws_gvp   [B,Npad]                 grad wrt v_posed
 *   ws_slab  [SAGA_BWD_SPLITK,B,512]  split-K partials of grad wrt feat
 *   ws_dA    [B,SAGA_DA_STRIDE]       grad wrt A ([56,12], row 55 = sum_v g) + compact joint grads
 *   ws_gv    [B,V,3] or NULL          required when g_joints is given (private copy of g_vertices
 *                                     that the extra-joint / landmark gradients are added into)
 * `full_pose` is the forward's input; rot/jrest/chain/A/v_posed are the forward's saved outputs.
 */
int saga_smplx_bwd(const saga_model_t* m, const float* g_vertices, const float* g_joints,
                   const float* full_pose, const float* rot, const float* jrest, const float* chain,
                   const float* A, const float* v_posed,
                   float* g_shape, float* g_full_pose, float* g_transl,
                   float* ws_gvp, float* ws_slab, float* ws_dA, float* ws_gv,
                   int B, void* stream);

/* smplx.lbs.batch_rodrigues(rot_vecs[N,3]) -> [N,3,3]
 * (imported directly at WholeGraspPose/models/objectmodel.py:6,38 and opt_grasppose.py:10,38). */
int saga_rodrigues_fwd(const float* rot_vecs, float* rot_mats, int N, void* stream);
int saga_rodrigues_bwd(const float* rot_vecs, const float* g_rot_mats, float* g_rot_vecs, int N, void* stream);

/* 6D continuous rotation -> axis-angle in one pass: RotConverter.rotmat2aa(RotConverter.cont2rotmat(x))
 * (utils/utils.py:385-417 -> :223-370), evaluated every iteration for the global orientation
 * (WholeGraspPose/models/fittingop.py:140) and for VPoser's 21 decoded joints (:142).
 *   x6 [N,6] -> aa [N,3] and jac [N,3,6] = d aa / d x6 (forward-mode, same branch as the value);
 *   backward: g_x6 [N,6] = jac^T g_aa. */
int saga_cont6d_to_aa_fwd(const float* x6, float* aa, float* jac, int N, void* stream);
int saga_cont6d_to_aa_bwd(const float* jac, const float* g_aa, float* g_x6, int N, void* stream);

/* ------------------------------------------------------------------------------------------
 * chamfer_distance.ChamferDistance(x, y) -> (dist1, dist2, idx1, idx2)
 * (reference call site utils/train_helper.py:57).  Brute-force bidirectional nearest
 * neighbour: squared distances, int32 indices, strict '<' (lowest index wins ties).
 *   x [N,P1,3], y [N,P2,3]; dist1/idx1 [N,P1]; dist2/idx2 [N,P2]
 * Any of the four outputs may be NULL (that direction / array is skipped).
 * ------------------------------------------------------------------------------------------ */
int saga_chamfer_fwd(const float* x, const float* y, float* dist1, float* dist2,
                     int32_t* idx1, int32_t* idx2, int N, int P1, int P2, void* stream);
/* the extension's own backward (NmDistanceGrad): g_x, g_y must be zero-initialised by the caller;
 * never driven by SAGA itself but part of the package's API. */
int saga_chamfer_bwd(const float* x, const float* y, const int32_t* idx1, const int32_t* idx2,
                     const float* g_dist1, const float* g_dist2, float* g_x, float* g_y,
                     int N, int P1, int P2, void* stream);

/* ------------------------------------------------------------------------------------------
 * Vertex normals.  Replaces Open3D TriangleMesh.compute_vertex_normals()
 * (WholeGraspPose/models/fittingop.py:158-164): area-weighted sum of face normals, normalised.
 *   saga_vertex_normals        : all vertices, verts [B,V,3] -> normals [B,V,3]
 *   saga_vertex_normals_subset : only the listed vertices (model's faces), -> [B,n_sel,3]
 * ------------------------------------------------------------------------------------------ */
int saga_vertex_normals(const saga_model_t* m, const float* verts, float* normals, int B, void* stream);
int saga_vertex_normals_subset(const saga_model_t* m, const float* verts, const int32_t* sel, int n_sel,
                               float* normals, int B, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused contact / penetration loss.  Replaces, for B independent candidates, everything
 * FittingOP.calc_loss_contact_map does between the SMPL-X call and the weighted sum
 * (WholeGraspPose/models/fittingop.py:152-212): the marker / foot / right-hand gathers, the Open3D
 * hand normals (:158-164), both point2point_signed calls (:183-184 -> utils/train_helper.py:17-86)
 * and the per-candidate means.  Index tables are DEVICE arrays built once by the host from
 * body_utils (fittingop.py:86-124); `unique_ids` is the sorted union of marker / hand / foot ids
 * and the *_slot arrays map every table entry to its position in it.
 *
 * terms [B,SAGA_GRASP_NTERMS]: 0 rec (:168)  1 body_rec (:169)  2 foot = 0.1*mean|z| (:173)
 *   3 marker contact (:190)  4 object contact (:191)  5 prior contact (:210)  6 hand collision (:211)
 *   7 object collision (:212)  8-11 counts: hand colli, obj colli, contact, hand markers colli (:238-242)
 * Also returned (what calc_loss_contact_map hands back / what the backward needs):
 *   markers_rec [B,M,3], hand_rec [B,H,3], hand_normals [B,H,3], h2o_marker [B,M] (unsigned, :183 quirk),
 *   h2o_signed [B,H], o2h_signed [B,O], idx_m2o [B,M], idx_h2o [B,H], idx_o2h [B,O] (int32)
 * saga_grasp_loss_bwd: g_terms [B,SAGA_GRASP_NTERMS] (entries 0..7 used) -> g_verts [B,V,3] (fully
 * written: zero-filled, then the touched vertices).
 * ------------------------------------------------------------------------------------------ */
#define SAGA_GRASP_NTERMS 16
typedef struct {
    int32_t n_markers, n_hand, n_foot, n_body_markers, n_unique;
    const int32_t* marker_ids;   /* [n_markers] vertex ids (markers_143) */
    const int32_t* hand_ids;     /* [n_hand]    right-hand vertex ids    */
    const int32_t* foot_ids;     /* [n_foot]                              */
    const float*   hand_weights; /* [n_hand]    rhand_weight.npy          */
    const int32_t* unique_ids;   /* [n_unique]                            */
    const int32_t* marker_slot;  /* [n_markers] index into unique_ids     */
    const int32_t* hand_slot;    /* [n_hand]                              */
    const int32_t* foot_slot;    /* [n_foot]                              */
} saga_grasp_tables_t;

int saga_grasp_loss_fwd(const saga_model_t* m, const saga_grasp_tables_t* tables, const float* verts,
                        const float* obj, const float* obj_normals, const float* markers_target,
                        const float* c_markers, const float* c_object,
                        float* terms, float* markers_rec, float* hand_rec, float* hand_normals,
                        float* h2o_marker, float* h2o_signed, float* o2h_signed,
                        int32_t* idx_m2o, int32_t* idx_h2o, int32_t* idx_o2h, int B, int O, void* stream);
int saga_grasp_loss_bwd(const saga_model_t* m, const saga_grasp_tables_t* tables, const float* g_terms,
                        const float* verts, const float* obj, const float* markers_target,
                        const float* c_markers, const float* c_object, const float* markers_rec,
                        const float* hand_rec, const float* h2o_signed, const float* o2h_signed,
                        const int32_t* idx_m2o, const int32_t* idx_h2o, const int32_t* idx_o2h,
                        float* g_verts, int B, int O, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-kernel timing for bench.py (no counterpart in the reference, whose only timing hook is the
 * disabled `time per sample` log at WholeGraspPose/models/fittingop.py:376,491-492).
 * When enabled, every kernel launch of the entry points above is bracketed by a hipEvent pair on
 * the launch stream; saga_profile_read() synchronises them and returns, per kernel id, the summed
 * elapsed milliseconds and the number of launches since the last read.  Not thread-safe.
 * ------------------------------------------------------------------------------------------ */
#define SAGA_PROF_POSE_FWD   0
#define SAGA_PROF_BLEND_FWD  1   /* FP32-MFMA blend contraction, forward  */
#define SAGA_PROF_SKIN_FWD   2
#define SAGA_PROF_JOINTS     3
#define SAGA_PROF_SKIN_BWD   4   /* g_v_posed + dA reduction */
#define SAGA_PROF_BLEND_BWD  5   /* FP32-MFMA blend contraction, backward (split-K) */
#define SAGA_PROF_POSE_BWD   6
#define SAGA_PROF_CHAMFER    7   /* both scan directions of one saga_chamfer_fwd call */
#define SAGA_PROF_NORMALS    8
#define SAGA_PROF_GRASP_FWD  9   /* fused normals + NN + signed distances + loss terms */
#define SAGA_PROF_GRASP_BWD  10
#define SAGA_PROF_NKERNELS   11
int saga_profile_enable(int on);
int saga_profile_read(double* total_ms_host, int32_t* count_host);

#ifdef __cplusplus
}
#endif
#endif /* SAGA_HIP_H */
