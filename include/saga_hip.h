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
 *   - no entry point allocates device memory except saga_model_create / saga_decoder_create (constants).
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
/* The same model restricted to the listed vertices (ascending ids into the descriptor's arrays): every per-vertex table
 * holds those vertices only (V = n_ids), the joint regression still runs over the full arrays, the faces kept are those
 * with all three corners listed (renumbered, original order); n_landmarks and n_extra_joints must be 0.  A listed
 * vertex gets bit-identical coordinates to the full model's.  For callers that read a few vertices only -- the fitting
 * loss reads ~1,050 markers / hand / foot vertices of 10,475 (the "V_sel" variant); no
 * counterpart in smplx, which always evaluates the whole mesh (fittingop.py:149). */
int  saga_model_create_subset(saga_model_t** out, const saga_model_desc_t* desc, const int32_t* vertex_ids_host, int n_ids);
int  saga_model_num_faces(const saga_model_t* m);
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
 *   vertices   [B,V,3]   out (its first 32 bytes also serve as the tile counters of the persistent blend kernel while the call
 *                        runs, before the skinning kernel writes the rows: output memory only, nothing for the caller to do)
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
/* The same forward with the rows of a vertex subset written compact as well -- for callers whose loss reads those vertices only
 * and whose backward runs on the subset model `sub` = saga_model_create_subset(...) (smplx forward(vertex_grad_ids=...) of the
 * drop-in; no counterpart in the reference): v_posed_sel [B, saga_model_npad(sub)] in sub's layout (what saga_smplx_bwd(sub, ...)
 * takes as v_posed) and vertices_sel [B, V(sub), 3] = the rows of `vertices`.  `m` may be a subset model of the same mesh too;
 * every vertex of `sub` must then be a vertex of `m`. */
int saga_smplx_fwd_selected(const saga_model_t* m, const saga_model_t* sub, const float* shape, const float* full_pose,
                            const float* transl, float* vertices, float* joints, float* feat, float* rot, float* jrest,
                            float* chain, float* A, float* v_posed, float* v_posed_sel, float* vertices_sel, int B, void* stream);

/* SMPL-X backward (what loss.backward() drives through smplx at fittingop.py:383).
 *   g_vertices [B,V,3] or NULL; g_joints [B,55+E+L,3] or NULL (at least one non-NULL)
 *   g_shape [B,S], g_full_pose [B,165], g_transl [B,3] : out (the first 32 bytes of g_full_pose serve as the tile counters of the
 *                        persistent split-K kernel in between, as `vertices` does in the forward)
 * workspace (caller-allocated, contents undefined on entry):
 *   ws_gvp   [B,Npad]                 grad wrt v_posed
 *   ws_slab  [SAGA_BWD_SPLITK,B,512]  split-K partials of grad wrt feat (before that: partial sums of ws_dA)
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

/* saga_smplx_bwd for callers that cannot allocate.  NO counterpart in the reference: it exists for callers under HIP graph
 * capture, tight serving loops and harnesses that forbid hipMalloc -- they still pass inputs and outputs, but the four
 * workspaces (ws_gvp, ws_slab, ws_dA, ws_gv) live in a static device array inside the library (18.4 MB per GPU the library runs
 * on, held until the process ends).  Same arguments otherwise, same results bit for bit.  Limits: B <= SAGA_NOALLOC_MAX_B
 * and saga_model_npad(m) <= SAGA_NOALLOC_MAX_NPAD (SMPL-X's 10,475 vertices fit), else SAGA_EINVAL.  It calls no hipMalloc /
 * hipFree / allocating routine (per-kernel profiling is suspended inside it).  Every call on a GPU shares that GPU's array:
 * calls that overlap on one GPU (two streams, two threads) race -- serialise them. */
#define SAGA_NOALLOC_MAX_B     64
#define SAGA_NOALLOC_MAX_NPAD  31488
int saga_smplx_bwd_noalloc(const saga_model_t* m, const float* g_vertices, const float* g_joints,
                           const float* full_pose, const float* rot, const float* jrest, const float* chain,
                           const float* A, const float* v_posed,
                           float* g_shape, float* g_full_pose, float* g_transl, int B, void* stream);

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
/* The same search for LARGE clouds with a caller-provided workspace (BASELINE config 5: 10,475 body vertices against 100,000 scene
 * points; reference call shape utils/train_helper.py:57, WholeGraspPose/trainer.py:280-281).  A direction whose database cloud has
 * >= 4,096 points is searched with spatial pruning -- both clouds counting-sorted into a 32^3 Morton grid, boxes per 32 / 2,048
 * sorted points, database tiles visited nearest first, (query tile, point group) pairs skipped by box-to-box bounds -- and returns
 * EXACTLY what saga_chamfer_fwd returns (same squared distances, same original indices, lowest index on ties).
 *   saga_chamfer_workspace_bytes: bytes of device workspace (16-byte aligned, contents undefined) the call needs for these sizes;
 *   0 = no direction is large enough, saga_chamfer_fwd_ws then IS saga_chamfer_fwd (ws may be NULL). */
size_t saga_chamfer_workspace_bytes(int N, int P1, int P2);
int saga_chamfer_fwd_ws(const float* x, const float* y, float* dist1, float* dist2, int32_t* idx1, int32_t* idx2,
                        int N, int P1, int P2, void* ws, size_t ws_bytes, void* stream);
/* the extension's own backward (NmDistanceGrad): the gradients are ADDED to g_x, g_y (zero-initialise them for the plain
 * gradient); never driven by SAGA itself but part of the package's API.  Deterministic: a gather in
 * a fixed order per point instead of the extension's float atomics -- bit-reproducible, batch-independent. */
int saga_chamfer_bwd(const float* x, const float* y, const int32_t* idx1, const int32_t* idx2,
                     const float* g_dist1, const float* g_dist2, float* g_x, float* g_y,
                     int N, int P1, int P2, void* stream);

/* point2point_signed(x, y, x_normals, y_normals) of utils/train_helper.py:17-86 as one operator (call sites
 * WholeGraspPose/models/fittingop.py:183-184, MotionFill/models/fittingop.py:154-157, WholeGraspPose/trainer.py:280-281):
 * the nearest-neighbour search above, then x2y = x - y[xidx], y2x = y - x[yidx], their norms, and the sign of the dot product
 * with the OTHER cloud's normal at the neighbour (a NULL normal array leaves that direction unsigned; sign(0) = 0).
 *   y2x_signed [N,P2], x2y_signed [N,P1], yidx [N,P2], xidx [N,P1] int32; y2x_vec [N,P2,3] / x2y_vec [N,P1,3] optional (NULL).
 *   y2x_signed = yidx = y2x_vec = NULL (or x2y_signed = xidx = x2y_vec = NULL): that direction is not searched -- for callers
 *   that discard it (MotionFill/models/fittingop.py:154-158: the marker call's object -> marker outputs, the hand call's hand ->
 *   object outputs).
 * Backward: gradients of the signed distances and / or of the difference vectors (each may be NULL) -> g_x [N,P1,3], g_y
 * [N,P2,3] (written; either may be NULL).  The reference's gather backward is a scatter-add with float atomics; here it is a
 * fixed-order gather: bit-reproducible.  A point that coincides with its neighbour gets no distance gradient (torch: NaN). */
int saga_p2p_signed_fwd(const float* x, const float* y, const float* x_normals, const float* y_normals,
                        float* y2x_signed, float* x2y_signed, int32_t* yidx, int32_t* xidx, float* y2x_vec, float* x2y_vec,
                        int N, int P1, int P2, void* stream);
/* The same forward for LARGE clouds (WholeGraspPose/trainer.py:280-281 measures body meshes against object / scene clouds): with
 * ws = saga_chamfer_workspace_bytes(N, P1, P2) bytes of device workspace a direction whose database cloud has >= 4,096 points is
 * searched with the pruned search of saga_chamfer_fwd_ws -- same indices, hence the same outputs bit for bit.  ws NULL: the call
 * above. */
int saga_p2p_signed_fwd_ws(const float* x, const float* y, const float* x_normals, const float* y_normals,
                           float* y2x_signed, float* x2y_signed, int32_t* yidx, int32_t* xidx, float* y2x_vec, float* x2y_vec,
                           int N, int P1, int P2, void* ws, size_t ws_bytes, void* stream);
int saga_p2p_signed_bwd(const float* x, const float* y, const int32_t* yidx, const int32_t* xidx,
                        const float* y2x_signed, const float* x2y_signed,
                        const float* g_y2x_signed, const float* g_x2y_signed, const float* g_y2x_vec, const float* g_x2y_vec,
                        float* g_x, float* g_y, int N, int P1, int P2, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row picks.  Replaces the per-set vertex gathers and frame slices of the GraspMotion-Opt objective and their autograd
 * backward (MotionFill/models/fittingop.py:356-409: body_verts[:, markers], body_verts[start_t:, rhand], heel / toe ids ...:
 * one index_select (+ slice) per set, each with a zero-filled full-size gradient and an index_add in the backward).
 *   src [S,T,U,3]; set k: pos [n] (values in [0,U), repeats allowed), t0 (first frame kept)
 *   saga_pick_sets_fwd : out_k [S,T-t0,n,3] = src[:, t0:, pos]                      (every set in one launch)
 *   saga_pick_sets_bwd : g_src [S,T,U,3] (overwritten) = sum over the sets (in order; a NULL g_out is skipped) and over
 *                        the picks of each row in ascending pick order (inv_ptr [U+1], inv_idx [n]: CSR of pick positions
 *                        per source row) -- deterministic, no atomics
 * `sets` is a HOST array of n_sets <= SAGA_PICK_MAX_SETS descriptors whose pointers are device pointers.
 * ------------------------------------------------------------------------------------------ */
#define SAGA_PICK_MAX_SETS 8
typedef struct {
    const int32_t* pos;       /* fwd */
    const int32_t* inv_ptr;   /* bwd */
    const int32_t* inv_idx;   /* bwd */
    float* out;               /* fwd */
    const float* g_out;       /* bwd (NULL: no gradient for this set) */
    int32_t n;
    int32_t t0;
} saga_pick_set_t;
int saga_pick_sets_fwd(const float* src, const saga_pick_set_t* sets, int n_sets, int S, int T, int U, void* stream);
int saga_pick_sets_bwd(const saga_pick_set_t* sets, int n_sets, float* g_src, int S, int T, int U, void* stream);

/* ------------------------------------------------------------------------------------------
 * The geometric terms of the GraspMotion-Opt stage-2 objective for S sequences of T frames as one operator.  Replaces the
 * tensor expressions of MotionFill/models/fittingop.py:159-176 (contact), :178-191 (collision), :193-263 (skating), :265-280
 * (hand angle), :360-368 (marker / end-pose fit), :386-387 (hand smoothness) and their autograd backward.
 *   markers [S,T,M,3], markers_rec [S,T,M,3], marker_mask [M], markers_end [S,M,3]; feet [S,T,NF,3] = the four heel / toe
 *   groups side by side in the column order of contact_lbl [S,T,4] (left heel, right heel, left toe, right toe);
 *   hand_tail [S,n_tail,H,3]; o2h_signed [S,n_tail,O], h2o_marker [S,n_tail,CM], h2o_marker_vec [S,n_tail,CM,3] = outputs
 *   of saga_p2p_signed_fwd; contact_object [S,O], contact_markers [S,CM]; palm_normals [S,n_tail,NP,3] (normals of the last
 *   NP of the CM markers; no gradient).
 *   terms [S,12]: rec_markers, end_markers_fit, skating, hand_smooth, contact, collision, hand_angle, 0, and the four
 *   skating counts (read by the backward).   g_terms [S,7] -> gradients of the six differentiable inputs (each written whole).
 * Fixed summation order: bit-reproducible, a sequence's terms do not depend on the batch.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n_seq, n_frames, n_tail;       /* S, T, T - start_t */
    int32_t n_markers;                     /* M (79) */
    int32_t foot_group_size[4];            /* rows of `feet` per group, in contact_lbl column order */
    int32_t n_hand, n_obj;                 /* H (778), O (2048) */
    int32_t n_contact_markers, n_palm;     /* CM (143), NP (22) */
    int32_t n_last_frames;                 /* frames at the end of the tail on which the contact term attracts (5, :163-165) */
    float vel_thres;                       /* skating speed threshold (0.1) */
} saga_motion_terms_cfg_t;
#define SAGA_MOTION_NTERMS 12
int saga_motion_terms_fwd(const saga_motion_terms_cfg_t* cfg, const float* markers, const float* markers_rec,
                          const float* marker_mask, const float* markers_end, const float* feet, const float* contact_lbl,
                          const float* hand_tail, const float* o2h_signed, const float* contact_object, const float* h2o_marker,
                          const float* contact_markers, const float* h2o_marker_vec, const float* palm_normals, float* terms,
                          void* stream);
int saga_motion_terms_bwd(const saga_motion_terms_cfg_t* cfg, const float* markers, const float* markers_rec,
                          const float* marker_mask, const float* markers_end, const float* feet, const float* contact_lbl,
                          const float* hand_tail, const float* o2h_signed, const float* contact_object, const float* h2o_marker,
                          const float* contact_markers, const float* h2o_marker_vec, const float* palm_normals, const float* terms,
                          const float* g_terms, float* g_markers, float* g_feet, float* g_hand_tail, float* g_o2h_signed,
                          float* g_h2o_marker, float* g_h2o_marker_vec, void* stream);

/* ------------------------------------------------------------------------------------------
 * Vertex normals.  Replaces Open3D TriangleMesh.compute_vertex_normals()
 * (WholeGraspPose/models/fittingop.py:158-164): area-weighted sum of face normals, normalised.
 *   saga_vertex_normals        : all vertices, verts [B,V,3] -> normals [B,V,3]
 *   saga_vertex_normals_subset : only the listed vertices (model's faces), -> [B,n_sel,3]
 *   saga_vertex_normals_subset_window : verts [S,T,V,3], the listed vertices on the frames t0.. only -> [S,T-t0,n_sel,3]
 *       (MotionFill/models/fittingop.py:137-146 computes the normals of the frames start_t.. only)
 * ------------------------------------------------------------------------------------------ */
int saga_vertex_normals(const saga_model_t* m, const float* verts, float* normals, int B, void* stream);
int saga_vertex_normals_subset(const saga_model_t* m, const float* verts, const int32_t* sel, int n_sel,
                               float* normals, int B, void* stream);
int saga_vertex_normals_subset_window(const saga_model_t* m, const float* verts, const int32_t* sel, int n_sel, float* normals,
                                      int S, int T, int t0, void* stream);

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
    /* Optional (all NULL / 0: not used), built by the host from the mesh's faces for saga_fit_step: the "ring" = every corner
     * of a face that touches a hand vertex (the vertices the hand's vertex normals read), so that one small kernel can stage
     * exactly those in LDS and produce the normals there.
     *   ring_ids [n_ring]          ascending vertex ids of the ring (n_ring < 65,536)
     *   hand_tri_ptr [n_hand+1], hand_tri [n_inc,2]: CSR over the hand vertices (table order) of their incident faces in
     *                ascending face order; the three corners of a face as RING positions packed into two words:
     *                (corner0 | corner1 << 16, corner2) */
    int32_t n_ring;
    const int32_t* ring_ids;
    const int32_t* hand_tri_ptr;
    const int32_t* hand_tri;
    /* Optional (NULL: the kernels evaluate the power themselves): hand_prior_w [n_hand] = hand_weights where
     * hand_weights^(1/2.5) > 0.7 (the vertex mask of the prior-contact term, fittingop.py:195), else 0 -- a constant of the table */
    const float* hand_prior_w;
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
 * Whole fitting iteration on the device.  Replaces, for B independent candidates, one pass of the
 * loop body `optimizer.zero_grad(); calc_loss_contact_map(...); loss.backward(); optimizer.step()`
 * (WholeGraspPose/models/fittingop.py:377-384) with no host work between kernels:
 *   optimiser variables -> SMPL-X arguments (:138-147: 6D -> axis-angle for the global orientation,
 *   VPoser decode 32 -> 512 -> 512 -> 126 -> 21 x 6D -> axis-angle, hand PCA + pose mean),
 *   saga_smplx_fwd, saga_grasp_loss_fwd, the weighted sum with its stage gates (:215-223),
 *   saga_grasp_loss_bwd, saga_smplx_bwd, the reverse parameter maps, and torch.optim.Adam's update
 *   (:77-84; default betas/eps, no weight decay) for the variable groups the stage's optimiser owns.
 *
 * The optimiser variables of a candidate (:71-76) are one packed row of SAGA_VAR_STRIDE floats:
 * ------------------------------------------------------------------------------------------ */
#define SAGA_VAR_STRIDE   112
#define SAGA_VAR_VPOSE    0     /* vpose_rec [32]                      */
#define SAGA_VAR_HAND     32    /* hand_pose [48] = left 24 | right 24 */
#define SAGA_VAR_EYE      80    /* eye_pose  [6]  = leye 3 | reye 3    */
#define SAGA_VAR_BETAS    86    /* betas     [10]                      */
#define SAGA_VAR_TRANSL   96    /* transl_rec [3]                      */
#define SAGA_VAR_ROT6D    99    /* glo_rot_rec [6]                     */
#define SAGA_VAR_COUNT    105
#define SAGA_GROUP_VPOSE  1     /* update_mask bits */
#define SAGA_GROUP_HAND   2
#define SAGA_GROUP_EYE    4
#define SAGA_GROUP_BETAS  8
#define SAGA_GROUP_TRANSL 16
#define SAGA_GROUP_ROT6D  32
#define SAGA_BODY_PARAM_COUNT 133  /* transl 3 | global_orient 3 | betas 10 | body_pose 63 | lh 24 | rh 24 | leye 3 | reye 3 */

/* Parameter decoder constants: VPoser v1.0 decoder weights (human_body_prior `bodyprior_dec_fc1/fc2/out`,
 * fittingop.py:65,142) and the SMPL-X hand PCA basis / means (inside `smplx`).  HOST pointers, copied. */
typedef struct saga_decoder saga_decoder_t;
typedef struct {
    int32_t latent;           /* 32  (multiple of 16) */
    int32_t hidden;           /* 512 (multiple of 128) */
    int32_t n_body_joints;    /* 21  (decoder output = n_body_joints * 6 <= 128) */
    int32_t n_pca;            /* 24  hand PCA components per hand */
    const float* w1_host;     /* [hidden, latent]  torch Linear layout [out, in] */
    const float* b1_host;     /* [hidden] */
    const float* w2_host;     /* [hidden, hidden] */
    const float* b2_host;     /* [hidden] */
    const float* w3_host;     /* [n_body_joints*6, hidden] */
    const float* b3_host;     /* [n_body_joints*6] */
    const float* lhand_components_host;  /* [n_pca, 45] */
    const float* rhand_components_host;  /* [n_pca, 45] */
    const float* pose_mean_host;         /* [165] (zeros | hands_meanl | hands_meanr) */
} saga_decoder_desc_t;
int  saga_decoder_create(saga_decoder_t** out, const saga_decoder_desc_t* desc);
void saga_decoder_destroy(saga_decoder_t* d);

/* The parameter maps alone, for callers that keep their own objective under autograd (GraspMotion-Opt stage 2,
 * MotionFill/models/fittingop.py:356-358 -> utils/utils_body.py:93-106: convert_to_3D_rot + VPoser decode + the hand PCA inside
 * smplx): packed variable rows [B,SAGA_VAR_STRIDE] -> full_pose [B,165] (pose mean added), shape [B,n_shape] (betas, then
 * zeros), transl [B,3]; and the reverse map of the gradients -> grad_vars [B,SAGA_VAR_STRIDE] (g_shape / g_transl may be NULL).
 *   ws: saga_params_workspace_floats(d, B) floats, the SAME buffer for the forward and the backward of one evaluation (it
 *       carries the decoder activations and the 6D->aa Jacobians, which the backward reads and leaves intact: it may run any
 *       number of times per forward, e.g. backward(retain_graph=True) as the reference calls it every iteration). */
size_t saga_params_workspace_floats(const saga_decoder_t* d, int B);
int saga_params_fwd(const saga_decoder_t* d, const float* vars, float* full_pose, float* shape, float* transl, int n_shape,
                    float* ws, int B, void* stream);
int saga_params_bwd(const saga_decoder_t* d, const float* vars, const float* g_full_pose, const float* g_shape,
                    const float* g_transl, float* grad_vars, int n_shape, float* ws, int B, void* stream);

typedef struct {                 /* per-candidate fitting targets, DEVICE pointers (opt_grasppose.py:211-226) */
    const float* obj;            /* [B,O,3] object points   */
    const float* obj_normals;    /* [B,O,3]                 */
    const float* markers_target; /* [B,M,3] sampled markers */
    const float* c_markers;      /* [B,M]   marker contact map */
    const float* c_object;       /* [B,O]   object contact map */
    int32_t n_obj;               /* O */
} saga_fit_problem_t;

typedef struct {
    int32_t stage;          /* 0,1,2: the loss gates of fittingop.py:215-223 */
    int32_t only_rec;       /* fittingconfig['only_rec'] */
    int32_t prior_contact;  /* fittingconfig['contact_loss'] == 'prior' */
    int32_t update_mask;    /* SAGA_GROUP_* bits: the variables the optimiser of this stage owns (:77-79) */
    int32_t adam_step;      /* 1-based step count of that optimiser */
    float lr, beta1, beta2, eps;
    int32_t skip_gated_terms; /* 1: in a stage whose contact / collision weights are all zero (stage < 2, or only_rec) those terms
                                 and the four counts are not evaluated and come back as 0 (the reference evaluates and logs them
                                 although they do not reach the loss, fittingop.py:215-242); 0: the reference's behaviour */
    int32_t custom_weights; /* 0: the GraspPose-Opt objective of fittingop.py:215-223.  1: a marker-fit objective with the
                               weights below -- total = w_rec * rec + reg_vpose * mean(vpose^2) + reg_shape * mean(betas^2) +
                               reg_hand * mean(hand^2) + w_foot * foot + reg_eye * mean(eye^2) -- which is stage 1 of
                               GraspMotion-Opt (MotionFill/models/fittingop.py:309-325: 1, 0.02, 0.01, 0.01, 0, 0) when the
                               tables list its 79 markers; needs stage = 0 and skip_gated_terms = 1 */
    float w_rec, w_foot, reg_vpose, reg_hand, reg_eye, reg_shape;
} saga_fit_step_cfg_t;

/* floats of workspace saga_fit_step / saga_fit_decode need for B candidates and O object points */
size_t saga_fit_workspace_floats(const saga_model_t* m, const saga_grasp_tables_t* tb, int B, int O);

/* offset (in floats) of a named intermediate inside the workspace after saga_fit_step, or -1: "vertices" [B,V,3],
 * "joints", "full_pose" [B,165], "shape", "terms" [B,16], "hand_rec", "hand_normals", "h2o_marker", "h2o_signed",
 * "o2h_signed", "idx_m2o" / "idx_h2o" / "idx_o2h" (int32), "g_vertices", "g_full_pose", "g_shape", "g_transl",
 * "grad" [B,SAGA_VAR_STRIDE] (gradient rows; the vpose columns are in "g_z" [B,128], regularisers excluded), "x6" */
long long saga_fit_workspace_offset(const saga_model_t* m, const saga_grasp_tables_t* tb, int B, int O, const char* name);

/*   vars   [B,SAGA_VAR_STRIDE]  in/out   optimiser variables
 *   adam_m, adam_v [B,SAGA_VAR_STRIDE]  in/out  exp_avg / exp_avg_sq of the stage's optimiser
 *   ws     workspace: ZERO-INITIALISED by the caller before its first use and not written by the caller afterwards
 *          (it holds every intermediate of the iteration, and the dense vertex-gradient rows are zero-filled only once:
 *          later iterations rewrite the same vertices -- a workspace that was not zeroed is still handled, its rows are
 *          filled on first use unless a stale word happens to equal the 32-bit tag)
 *   loss   [B]       out  per-candidate total loss (before the update)
 *   loss_terms [B,16] out or NULL: total, rec, body_rec, marker contact, object contact, prior contact, hand collision,
 *              object collision, foot, reg (loss_dict :225-236), then the four counts of :238-242, 2 pads
 *   markers_rec [B,M,3] out  fitted markers (before the update) */
int saga_fit_step(const saga_model_t* m, const saga_decoder_t* dec, const saga_grasp_tables_t* tb,
                  const saga_fit_problem_t* prob, const saga_fit_step_cfg_t* cfg,
                  float* vars, float* adam_m, float* adam_v, float* ws,
                  float* loss, float* loss_terms, float* markers_rec, int B, void* stream);

/* The same iteration with a structurally sparse backward (no counterpart in the reference, whose autograd forms the dense
 * [B,V,3] vertex gradient at fittingop.py:383).  The loss reads only the tables' `unique_ids` vertices (markers, right
 * hand, feet: ~1,050 of 10,475), so the vertex gradient is zero everywhere else and the dense backward multiplies those
 * zeros through the skinning transpose and the blend contraction.  Here the loss backward writes the compact gradient
 * [B,n_unique,3] and the SMPL-X backward runs on `m_loss` = saga_model_create_subset(desc, unique_ids, n_unique) -- the
 * forward still evaluates the whole mesh with `m`.  Same gradients as saga_fit_step up to the rounding of the shorter
 * sums (the skipped terms are exact zeros); same workspace, same arguments otherwise.  "g_vertices" of
 * saga_fit_workspace_offset is not written by this variant. */
int saga_fit_step_sparse(const saga_model_t* m, const saga_model_t* m_loss, const saga_decoder_t* dec,
                         const saga_grasp_tables_t* tb, const saga_fit_problem_t* prob, const saga_fit_step_cfg_t* cfg,
                         float* vars, float* adam_m, float* adam_v, float* ws,
                         float* loss, float* loss_terms, float* markers_rec, int B, void* stream);

/* optimiser variables -> the SMPL-X parameter dict the reference saves (fittingop.py:466-475):
 * body_params [B,SAGA_BODY_PARAM_COUNT] */
int saga_fit_decode(const saga_decoder_t* dec, const float* vars, float* body_params, float* ws, int B, void* stream);

/* best-of-iteration snapshot (fittingop.py:429-447) per candidate: where loss < best_loss, copy vars (taken after
 * the update) and markers_rec (from before it) and lower best_loss. */
int saga_fit_best_update(const float* loss, const float* vars, const float* markers_rec,
                         float* best_loss, float* best_vars, float* best_markers, int B, int M, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole GraspMotion-Opt stage-2 iteration on the device.  Replaces, for S independent sequences of T frames, one pass of the
 * loop body of FittingOP.fitting_stage2 (MotionFill/models/fittingop.py:342-414: zero_grad, the objective :356-409 with its
 * helpers :132-280, loss.backward, optimizer.step) with no host work and no framework kernel between the launches:
 *   saga_params_fwd, saga_smplx_fwd_selected, saga_pick_sets_fwd, saga_vertex_normals_subset_window (hand + palm markers),
 *   saga_p2p_signed_fwd x2 (markers -> object :154, object -> hand :157), saga_motion_terms_fwd, the two regularisers and the
 *   weighted sum (:391-409), the reverse of each, and torch.optim.Adam's update (:344-350; no weight decay) of transl, rot6d,
 *   the VPoser latent and the hand coefficients.  The learned smoothness prior (:70-130) is not part of it.
 * The variables of the S*T bodies are packed rows [S*T,SAGA_VAR_STRIDE] (sequence-major; the layout of saga_fit_step: betas
 * are read, not updated; the eye columns are ignored and must be 0).
 *   m       the body model the objective is evaluated on: the full mesh, or a vertex-subset model that holds every vertex the
 *           terms read plus the face ring around the hand / palm-marker vertices (their normals)
 *   m_grad  saga_model_create_subset over the vertices the terms read (rows of m's mesh; U = its vertex count): the backward
 *           runs on it
 *   tables  DEVICE index arrays: the four pick sets out of m_grad's rows -- 0: markers [M] (every frame), 1: the four heel / toe
 *           groups side by side in contact_lbl's column order (every frame), 2: the contact markers [CM] (frames T-n_tail..),
 *           3: the right hand [H] (frames T-n_tail..) -- each with pos, inv_ptr, inv_idx, n, t0 (out / g_out are ignored);
 *           normal_sel [H+NP]: rows of m's vertices of the right hand (order of set 3), then of the last NP contact markers
 *           (the vertices whose normals the terms read); marker_mask [M] (:360-364)
 *   problem DEVICE targets: markers_rec [S,T,M,3], markers_end [S,M,3], contact_lbl [S,T,4], contact_object [S,O],
 *           contact_markers [S,CM], verts_object / normal_object [S,n_tail,O,3]
 *   cfg     sizes (terms), the nine stage2_weight_loss_* weights (opt_graspmotion.py:470-495), Adam state: adam_step is the
 *           1-based step count (0: evaluate the objective and its gradient only, no update), lr (0.01, 0.005 after step 150)
 *   ws      saga_motion_step_workspace_floats floats, contents undefined; after the call it holds every intermediate
 *           (saga_motion_step_workspace_offset: "vertices", "vertices_sel", "markers", "feet", "contact_markers_tail", "hand_tail",
 *           "hand_normals", "palm_normals", "h2o_marker", "h2o_marker_vec", "o2h_signed", "idx_m2o" / "idx_o2h" (int32), "terms"
 *           [S,SAGA_MOTION_NTERMS], "g_terms" [S,7], "g_vertices_sel", "g_full_pose", "g_shape", "g_transl", "full_pose", "shape",
 *           "transl", "grad" [S*T,SAGA_VAR_STRIDE] = gradient rows, regularisers included; else -1)
 *   loss [S] out: per-sequence objective (before the update).  terms_out [S,16] out or NULL: total, rec_markers,
 *           end_markers_fit, skating, hand_smooth, contact, collision, hand_angle, vposer, hand, the four skating counts, 2 pads.
 * S*T <= 65535.  Bit-reproducible; a sequence's result does not depend on the batch it is in.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    saga_pick_set_t sets[4];
    const int32_t* normal_sel;
    const float* marker_mask;
} saga_motion_tables_t;
typedef struct {
    const float* markers_rec;
    const float* markers_end;
    const float* contact_lbl;
    const float* contact_object;
    const float* contact_markers;
    const float* verts_object;
    const float* normal_object;
} saga_motion_problem_t;
typedef struct {
    saga_motion_terms_cfg_t terms;
    float w_rec_markers, w_end_markers_fit, w_vposer, w_hand, w_skating, w_hand_smooth, w_collision, w_contact, w_hand_angle;
    int32_t adam_step;
    float lr, beta1, beta2, eps;
} saga_motion_step_cfg_t;
size_t saga_motion_step_workspace_floats(const saga_model_t* m, const saga_model_t* m_grad, const saga_decoder_t* dec,
                                         const saga_motion_step_cfg_t* cfg);
long long saga_motion_step_workspace_offset(const saga_model_t* m, const saga_model_t* m_grad, const saga_decoder_t* dec,
                                            const saga_motion_step_cfg_t* cfg, const char* name);
int saga_motion_step(const saga_model_t* m, const saga_model_t* m_grad, const saga_decoder_t* dec,
                     const saga_motion_tables_t* tables, const saga_motion_problem_t* prob, const saga_motion_step_cfg_t* cfg,
                     float* vars, float* adam_m, float* adam_v, float* ws, float* loss, float* terms_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-kernel timing for bench.py (no counterpart in the reference, whose only timing hook is the
 * disabled `time per sample` log at WholeGraspPose/models/fittingop.py:376,491-492).
 * saga_profile_enable(1): every kernel launch of the entry points above is bracketed by a hipEvent pair on
 * the launch stream; saga_profile_read() synchronises them and returns, per kernel id, the summed
 * elapsed milliseconds and the number of launches since the last read.  saga_profile_enable(2) brackets only the
 * blend-contraction GEMM launches (the roofline kernel: two event pairs per iteration instead of seventeen, which
 * cost ~0.08 ms of a 5 ms iteration); 0 switches it off.  Not thread-safe.
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
#define SAGA_PROF_PARAMS_FWD 11  /* saga_fit_step: decoder MLP + 6D->aa + hand PCA */
#define SAGA_PROF_LOSS_SUM   12  /* weighted sum + stage gates */
#define SAGA_PROF_PARAMS_BWD 13  /* reverse parameter maps + decoder MLP backward */
#define SAGA_PROF_ADAM       14
#define SAGA_PROF_NKERNELS   15
int saga_profile_enable(int on);
int saga_profile_read(double* total_ms_host, int32_t* count_host);
/* Diagnostic (no counterpart in the reference): 1 if saga_smplx_fwd on a batch of B candidates of model m runs the blend
 * contraction as the persistent kernel with overlapping tiles (whole 128-candidate tiles and enough of them; the environment
 * switch SAGA_GEMM_XT), 0 if as one tile per block, -1 on a null model.  Same bits either way; tests use it to know which
 * launch form they have exercised. */
int saga_smplx_fwd_is_persistent(const saga_model_t* m, int B);
/* The same for the split-K backward contraction of saga_smplx_bwd (B >= 1,024 on the whole mesh with 256 CUs; only
 * behind the streaming skinning backward of sparse weight tables; switches SAGA_GEMM_XT, SAGA_GEMM_XT_BWD). */
int saga_smplx_bwd_is_persistent(const saga_model_t* m, int B);

#ifdef __cplusplus
}
#endif
#endif /* SAGA_HIP_H */
