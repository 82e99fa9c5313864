// One whole GraspMotion-Opt stage-2 iteration on the device: the loop body of FittingOP.fitting_stage2
// (MotionFill/models/fittingop.py:342-414 -- zero_grad, the objective :356-409, backward, Adam step) for S independent
// sequences of T frames, as one sequence of kernel launches with no host work and no PyTorch kernel in between.
//
//   packed variable rows [S*T,112] -> SMPL-X arguments            saga_params_fwd (VPoser MLP, 6D -> aa, hand PCA)
//   bodies over the vertices the objective reads                   saga_smplx_fwd_selected (m = loss model, m_grad = its union)
//   marker / foot / contact-marker / hand rows                     saga_pick_sets_fwd (one launch)
//   hand + palm-marker normals on the frames start_t..             saga_vertex_normals_subset_window
//   markers -> object, object -> hand signed distances             saga_p2p_signed_fwd (x2, one direction each: :154-158)
//   the seven geometric terms                                      saga_motion_terms_fwd
//   regularisers + weighted sum (:391-409)                         motion_total_kernel (below)
//   and the reverse of each, then torch.optim.Adam's update of transl / rot6d / latent / hands (:344-350).
//
// Every piece is the operator the autograd composition in saga_amd/motion_fitting.py calls, in the same order on the same
// buffers' contents: the terms are bit-identical to that composition.
#include <math.h>

#include "saga_common.h"

int saga_params_bwd_reg(const saga_decoder_t* d, const float* vars, const float* g_full_pose, const float* g_shape,
                        const float* g_transl, float* grad_vars, int n_shape, float* ws, int B, float c_vpose, float c_hand,
                        void* stream);
int saga_vertex_normals_window_split(const saga_model_t* m, const float* verts, const int32_t* sel, int n_a, int n_b, float* normals_a,
                                     float* normals_b, int S, int T, int t0, void* stream);
int saga_adam_rows(float* vars, float* adam_m, float* adam_v, const float* grad, int mask, int step, float lr, float beta1,
                   float beta2, float eps, int B, void* stream);

#define VS SAGA_VAR_STRIDE
#define MS_THREADS 256

struct MsWs {
    float *params, *full_pose, *shape, *transl;                                              // parameter maps
    float *feat, *rot, *jrest, *chain, *A, *v_posed, *verts, *joints, *v_posed_sel, *verts_sel;   // SMPL-X forward
    float *markers, *feet, *mk_tail, *hand_tail, *nrm_hand, *nrm_palm;                       // picks, normals
    float *h2o_marker, *h2o_vec, *o2h_signed;                                                // signed distances
    int32_t *idx_m2o, *idx_o2h;
    float *terms, *g_terms;
    float *g_markers, *g_feet, *g_hand_tail, *g_o2h, *g_h2o_marker, *g_h2o_vec, *g_mk_tail, *g_hand_p2p;   // reverse
    float *g_sub, *g_shape, *g_pose, *g_transl, *gvp, *slab, *dA, *grad;
    size_t total;
};

static MsWs ms_carve(float* base, const saga_model* m, const saga_model* mg, size_t params_floats, const saga_motion_terms_cfg_t& c) {
    MsWs w;
    size_t off = 0;
    auto take = [&](size_t n) {
        float* p = base ? base + off : nullptr;
        off += (n + 63) / 64 * 64;       // 256-B aligned sub-buffers
        return p;
    };
    const size_t S = (size_t)c.n_seq, T = (size_t)c.n_frames, b = S * T, nt = S * (size_t)c.n_tail;
    const size_t M = c.n_markers, NF = (size_t)c.foot_group_size[0] + c.foot_group_size[1] + c.foot_group_size[2] + c.foot_group_size[3];
    const size_t H = c.n_hand, O = c.n_obj, CM = c.n_contact_markers, NP = c.n_palm, U = mg->V;
    const size_t njo = NJ + m->E + m->L;
    w.params = take(params_floats);
    w.full_pose = take(b * 165); w.shape = take(b * m->S); w.transl = take(b * 3);
    w.feat = take(b * FEAT_K); w.rot = take(b * NJ * 9); w.jrest = take(b * NJ * 3); w.chain = take(b * NJ * 12);
    w.A = take(b * NJ * 12); w.v_posed = take(b * m->Npad); w.verts = take(b * m->V * 3); w.joints = take(b * njo * 3);
    w.v_posed_sel = take(b * mg->Npad); w.verts_sel = take(b * U * 3);
    w.markers = take(b * M * 3); w.feet = take(b * NF * 3); w.mk_tail = take(nt * CM * 3); w.hand_tail = take(nt * H * 3);
    w.nrm_hand = take(nt * H * 3); w.nrm_palm = take(nt * NP * 3);
    w.h2o_marker = take(nt * CM); w.h2o_vec = take(nt * CM * 3); w.o2h_signed = take(nt * O);
    w.idx_m2o = (int32_t*)take(nt * CM); w.idx_o2h = (int32_t*)take(nt * O);
    w.terms = take(S * SAGA_MOTION_NTERMS); w.g_terms = take(S * 8);
    w.g_markers = take(b * M * 3); w.g_feet = take(b * NF * 3); w.g_hand_tail = take(nt * H * 3); w.g_o2h = take(nt * O);
    w.g_h2o_marker = take(nt * CM); w.g_h2o_vec = take(nt * CM * 3); w.g_mk_tail = take(nt * CM * 3); w.g_hand_p2p = take(nt * H * 3);
    w.g_sub = take(b * U * 3); w.g_shape = take(b * m->S); w.g_pose = take(b * 165); w.g_transl = take(b * 3);
    w.gvp = take(b * mg->Npad); w.slab = take((size_t)SAGA_BWD_SPLITK * b * FEAT_K); w.dA = take(b * SAGA_DA_STRIDE);
    w.grad = take(b * VS);
    w.total = off;
    return w;
}

static int ms_check(const saga_model_t* m, const saga_model_t* mg, const saga_decoder_t* dec, const saga_motion_step_cfg_t* cfg,
                    const char* what) {
    SAGA_CHECK_ARG(m && mg && dec && cfg, "%s: null argument", what);
    const saga_motion_terms_cfg_t& c = cfg->terms;
    SAGA_CHECK_ARG(c.n_seq >= 0 && c.n_frames >= 2 && c.n_tail >= 2 && c.n_tail <= c.n_frames, "%s: S=%d T=%d n_tail=%d", what,
                   c.n_seq, c.n_frames, c.n_tail);
    SAGA_CHECK_ARG((long long)c.n_seq * c.n_frames <= 65535, "%s: %d x %d bodies exceed 65535", what, c.n_seq, c.n_frames);
    SAGA_CHECK_ARG(c.n_markers > 0 && c.n_hand > 0 && c.n_obj > 0 && c.n_contact_markers > 0 && c.n_palm > 0 &&
                       c.n_palm <= c.n_contact_markers && c.foot_group_size[0] > 0 && c.foot_group_size[1] > 0 &&
                       c.foot_group_size[2] > 0 && c.foot_group_size[3] > 0,
                   "%s: empty set in the term sizes", what);
    SAGA_CHECK_ARG(m->S >= 10, "%s: model has %d shape coefficients (< 10 betas)", what, m->S);
    SAGA_CHECK_ARG(mg->sel_ids && mg->VF == m->VF && mg->S == m->S,
                   "%s: m_grad must be a vertex-subset model of m's mesh (subset over a mesh of %d vertices / %d shape coefficients; m: %d / %d)",
                   what, mg->VF, mg->S, m->VF, m->S);
    return 0;
}

extern "C" size_t saga_motion_step_workspace_floats(const saga_model_t* m, const saga_model_t* m_grad, const saga_decoder_t* dec,
                                                    const saga_motion_step_cfg_t* cfg) {
    if (ms_check(m, m_grad, dec, cfg, "saga_motion_step_workspace_floats") || cfg->terms.n_seq == 0) return 0;
    const int B = cfg->terms.n_seq * cfg->terms.n_frames;
    return ms_carve(nullptr, m, m_grad, saga_params_workspace_floats(dec, B), cfg->terms).total;
}

extern "C" long long saga_motion_step_workspace_offset(const saga_model_t* m, const saga_model_t* m_grad, const saga_decoder_t* dec,
                                                       const saga_motion_step_cfg_t* cfg, const char* name) {
    if (!name || ms_check(m, m_grad, dec, cfg, "saga_motion_step_workspace_offset") || cfg->terms.n_seq == 0) return -1;
    const int B = cfg->terms.n_seq * cfg->terms.n_frames;
    float* base = (float*)(uintptr_t)256;         // carve against a fake (never dereferenced) base to read the offsets
    MsWs w = ms_carve(base, m, m_grad, saga_params_workspace_floats(dec, B), cfg->terms);
    struct { const char* n; const void* p; } tab[] = {
        {"full_pose", w.full_pose}, {"shape", w.shape}, {"transl", w.transl}, {"vertices", w.verts}, {"vertices_sel", w.verts_sel},
        {"markers", w.markers}, {"feet", w.feet}, {"contact_markers_tail", w.mk_tail}, {"hand_tail", w.hand_tail},
        {"hand_normals", w.nrm_hand}, {"palm_normals", w.nrm_palm}, {"h2o_marker", w.h2o_marker}, {"h2o_marker_vec", w.h2o_vec},
        {"o2h_signed", w.o2h_signed}, {"idx_m2o", w.idx_m2o}, {"idx_o2h", w.idx_o2h}, {"terms", w.terms}, {"g_terms", w.g_terms},
        {"g_vertices_sel", w.g_sub}, {"g_full_pose", w.g_pose}, {"g_shape", w.g_shape}, {"g_transl", w.g_transl}, {"grad", w.grad}};
    for (auto& e : tab)
        if (!strcmp(e.n, name)) return (long long)((const float*)e.p - base);
    return -1;
}

struct MsWeights { float rec, end, vposer, hand, skating, hand_smooth, collision, contact, hand_angle; };

// One block per sequence: the two regularisers of :391-392 (mean of latent^2 / hands^2 over the sequence's T frames, fixed
// summation order), the weighted sum of :393-409 in its order of additions, and the gradient of it w.r.t. the seven fused terms.
__global__ __launch_bounds__(MS_THREADS) void motion_total_kernel(const float* __restrict__ terms, const float* __restrict__ vars,
                                                                  MsWeights w, int T, float* __restrict__ g_terms,
                                                                  float* __restrict__ loss, float* __restrict__ out) {
#pragma clang fp contract(off)
    __shared__ float sv[MS_THREADS], sh[MS_THREADS];
    const int s = blockIdx.x, tid = threadIdx.x;
    const float* v = vars + (size_t)s * T * VS;
    float av = 0.f, ah = 0.f;
    for (int i = tid; i < T * 32; i += MS_THREADS) {
        const float x = v[(size_t)(i >> 5) * VS + SAGA_VAR_VPOSE + (i & 31)];
        av += x * x;
    }
    for (int i = tid; i < T * 48; i += MS_THREADS) {
        const float x = v[(size_t)(i / 48) * VS + SAGA_VAR_HAND + (i % 48)];
        ah += x * x;
    }
    sv[tid] = av; sh[tid] = ah;
    __syncthreads();
    for (int k = MS_THREADS / 2; k > 0; k >>= 1) {
        if (tid < k) { sv[tid] += sv[tid + k]; sh[tid] += sh[tid + k]; }
        __syncthreads();
    }
    if (tid != 0) return;
    const float* t = terms + (size_t)s * SAGA_MOTION_NTERMS;      // rec_markers, end_markers_fit, skating, hand_smooth, contact, collision, hand_angle
    const float vposer = sv[0] / (float)(T * 32), hand = sh[0] / (float)(T * 48);
    float total = w.rec * t[0];
    total += w.end * t[1];
    total += w.vposer * vposer;
    total += w.hand * hand;
    total += w.skating * t[2];
    total += w.hand_smooth * t[3];
    total += w.collision * t[5];
    total += w.contact * t[4];
    total += w.hand_angle * t[6];
    loss[s] = total;
    float* g = g_terms + (size_t)s * 7;
    g[0] = w.rec; g[1] = w.end; g[2] = w.skating; g[3] = w.hand_smooth; g[4] = w.contact; g[5] = w.collision; g[6] = w.hand_angle;
    if (out) {
        float* o = out + (size_t)s * 16;
        o[0] = total;
#pragma unroll
        for (int k = 0; k < 7; ++k) o[1 + k] = t[k];
        o[8] = vposer; o[9] = hand;
        o[10] = t[8]; o[11] = t[9]; o[12] = t[10]; o[13] = t[11]; o[14] = 0.f; o[15] = 0.f;
    }
}

__global__ void add_into_kernel(float* __restrict__ a, const float* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] + b[i];
}

extern "C" int saga_motion_step(const saga_model_t* m, const saga_model_t* m_grad, const saga_decoder_t* dec,
                                const saga_motion_tables_t* tb, const saga_motion_problem_t* prob, const saga_motion_step_cfg_t* cfg,
                                float* vars, float* adam_m, float* adam_v, float* ws, float* loss, float* terms_out, void* stream) {
    if (int rc = ms_check(m, m_grad, dec, cfg, "saga_motion_step")) return rc;
    SAGA_CHECK_DEVICE(m, "saga_motion_step (model)");
    SAGA_CHECK_DEVICE(m_grad, "saga_motion_step (gradient model)");
    const saga_motion_terms_cfg_t& c = cfg->terms;
    const int S = c.n_seq, T = c.n_frames, B = S * T, nt = c.n_tail, t0 = T - nt, N = S * nt, U = m_grad->V;
    if (S == 0) return 0;
    SAGA_CHECK_ARG(tb && prob && vars && ws && loss, "saga_motion_step: null argument");
    SAGA_CHECK_ARG(cfg->adam_step >= 0 && (cfg->adam_step == 0 || (adam_m && adam_v)), "saga_motion_step: adam step %d / null moments",
                   cfg->adam_step);
    SAGA_CHECK_ARG(tb->normal_sel && tb->marker_mask, "saga_motion_step: incomplete tables");
    SAGA_CHECK_ARG(prob->markers_rec && prob->markers_end && prob->contact_lbl && prob->contact_object && prob->contact_markers &&
                       prob->verts_object && prob->normal_object,
                   "saga_motion_step: incomplete problem");
    const int NF = c.foot_group_size[0] + c.foot_group_size[1] + c.foot_group_size[2] + c.foot_group_size[3];
    const int want_n[4] = {c.n_markers, NF, c.n_contact_markers, c.n_hand}, want_t0[4] = {0, 0, t0, t0};
    for (int k = 0; k < 4; ++k)
        SAGA_CHECK_ARG(tb->sets[k].n == want_n[k] && tb->sets[k].t0 == want_t0[k] && tb->sets[k].pos && tb->sets[k].inv_ptr && tb->sets[k].inv_idx,
                       "saga_motion_step: pick set %d (markers, feet, contact markers, hand) has n=%d t0=%d, expected n=%d t0=%d, with "
                       "pos / inv_ptr / inv_idx", k, tb->sets[k].n, tb->sets[k].t0, want_n[k], want_t0[k]);
    hipStream_t st = (hipStream_t)stream;
    MsWs w = ms_carve(ws, m, m_grad, saga_params_workspace_floats(dec, B), c);
    int rc;
    // ---- forward ----------------------------------------------------------------------------------------------------------
    if ((rc = saga_params_fwd(dec, vars, w.full_pose, w.shape, w.transl, m->S, w.params, B, stream))) return rc;
    if ((rc = saga_smplx_fwd_selected(m, m_grad, w.shape, w.full_pose, w.transl, w.verts, w.joints, w.feat, w.rot, w.jrest, w.chain, w.A,
                                      w.v_posed, w.v_posed_sel, w.verts_sel, B, stream)))
        return rc;
    saga_pick_set_t sets[4];
    float* outs[4] = {w.markers, w.feet, w.mk_tail, w.hand_tail};
    for (int k = 0; k < 4; ++k) { sets[k] = tb->sets[k]; sets[k].out = outs[k]; sets[k].g_out = nullptr; }
    if ((rc = saga_pick_sets_fwd(w.verts_sel, sets, 4, S, T, U, stream))) return rc;
    if ((rc = saga_vertex_normals_window_split(m, w.verts, tb->normal_sel, c.n_hand, c.n_palm, w.nrm_hand, w.nrm_palm, S, T, t0, stream)))
        return rc;
    // (:154) markers -> object: signed by the object normals, with the difference vectors; the object -> marker direction is discarded
    if ((rc = saga_p2p_signed_fwd(w.mk_tail, prob->verts_object, nullptr, prob->normal_object, nullptr, w.h2o_marker, nullptr, w.idx_m2o,
                                  nullptr, w.h2o_vec, N, c.n_contact_markers, c.n_obj, stream)))
        return rc;
    // (:157) object -> hand: signed by the hand normals; the hand -> object direction is never read
    if ((rc = saga_p2p_signed_fwd(w.hand_tail, prob->verts_object, w.nrm_hand, prob->normal_object, w.o2h_signed, nullptr, w.idx_o2h,
                                  nullptr, nullptr, nullptr, N, c.n_hand, c.n_obj, stream)))
        return rc;
    if ((rc = saga_motion_terms_fwd(&c, w.markers, prob->markers_rec, tb->marker_mask, prob->markers_end, w.feet, prob->contact_lbl,
                                    w.hand_tail, w.o2h_signed, prob->contact_object, w.h2o_marker, prob->contact_markers, w.h2o_vec,
                                    w.nrm_palm, w.terms, stream)))
        return rc;
    MsWeights wt{cfg->w_rec_markers, cfg->w_end_markers_fit, cfg->w_vposer, cfg->w_hand, cfg->w_skating, cfg->w_hand_smooth,
                 cfg->w_collision, cfg->w_contact, cfg->w_hand_angle};
    motion_total_kernel<<<S, MS_THREADS, 0, st>>>(w.terms, vars, wt, T, w.g_terms, loss, terms_out);
    SAGA_LAUNCH_CHECK();
    // ---- backward ---------------------------------------------------------------------------------------------------------
    if ((rc = saga_motion_terms_bwd(&c, w.markers, prob->markers_rec, tb->marker_mask, prob->markers_end, w.feet, prob->contact_lbl,
                                    w.hand_tail, w.o2h_signed, prob->contact_object, w.h2o_marker, prob->contact_markers, w.h2o_vec,
                                    w.nrm_palm, w.terms, w.g_terms, w.g_markers, w.g_feet, w.g_hand_tail, w.g_o2h, w.g_h2o_marker,
                                    w.g_h2o_vec, stream)))
        return rc;
    if ((rc = saga_p2p_signed_bwd(w.hand_tail, prob->verts_object, w.idx_o2h, nullptr, w.o2h_signed, nullptr, w.g_o2h, nullptr, nullptr,
                                  nullptr, w.g_hand_p2p, nullptr, N, c.n_hand, c.n_obj, stream)))
        return rc;
    if ((rc = saga_p2p_signed_bwd(w.mk_tail, prob->verts_object, nullptr, w.idx_m2o, nullptr, w.h2o_marker, nullptr, w.g_h2o_marker,
                                  nullptr, w.g_h2o_vec, w.g_mk_tail, nullptr, N, c.n_contact_markers, c.n_obj, stream)))
        return rc;
    {   // the hand rows feed the smoothness term and the object -> hand distances: one gradient
        const size_t n = (size_t)N * c.n_hand * 3;
        add_into_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(w.g_hand_tail, w.g_hand_p2p, n);
        SAGA_LAUNCH_CHECK();
    }
    const float* gouts[4] = {w.g_markers, w.g_feet, w.g_mk_tail, w.g_hand_tail};
    for (int k = 0; k < 4; ++k) { sets[k].out = nullptr; sets[k].g_out = gouts[k]; }
    if ((rc = saga_pick_sets_bwd(sets, 4, w.g_sub, S, T, U, stream))) return rc;
    if ((rc = saga_smplx_bwd(m_grad, w.g_sub, nullptr, w.full_pose, w.rot, w.jrest, w.chain, w.A, w.v_posed_sel, w.g_shape, w.g_pose,
                             w.g_transl, w.gvp, w.slab, w.dA, nullptr, B, stream)))
        return rc;
    // d/dx of w * mean over (T, n) of x^2 = (2 w / (T n)) x
    const float c_vpose = 2.f * cfg->w_vposer / (float)(T * 32), c_hand = 2.f * cfg->w_hand / (float)(T * 48);
    if ((rc = saga_params_bwd_reg(dec, vars, w.g_pose, w.g_shape, w.g_transl, w.grad, m->S, w.params, B, c_vpose, c_hand, stream))) return rc;
    if (cfg->adam_step > 0) {
        const int mask = SAGA_GROUP_VPOSE | SAGA_GROUP_HAND | SAGA_GROUP_TRANSL | SAGA_GROUP_ROT6D;      // (:344-350)
        if ((rc = saga_adam_rows(vars, adam_m, adam_v, w.grad, mask, cfg->adam_step, cfg->lr, cfg->beta1, cfg->beta2, cfg->eps, B, stream)))
            return rc;
    }
    return 0;
}
