// One whole GraspPose-Opt iteration on the device: the loop body of FittingOP.fitting
// (WholeGraspPose/models/fittingop.py:377-384 -- zero_grad, calc_loss_contact_map :137-244, backward, Adam step)
// for B independent candidates, as one sequence of kernel launches with no host work in between.
//
//   parameter maps (:138-147)   VPoser decoder 32->512->512->126 on the FP32-MFMA GEMM of blend_gemm.hip (bias +
//                               leaky-relu fused in the epilogue), then ONE kernel per candidate block: 22 x (6D ->
//                               axis-angle with Jacobian), hand PCA, pose mean, shape vector
//   SMPL-X forward, fused contact loss, weighted sum with stage gates (:215-223)
//   backward of all of it, then torch.optim.Adam's update for the variables the stage's optimiser owns (:77-84).
//
// The reference does this with ~1,400 eager kernel launches and >= 14 host syncs per iteration at batch size 1.
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "saga_common.h"
#include "rot6d.h"

void* saga_prof_begin(int id, void* stream);
void saga_prof_end(void* scope);
int saga_grasp_loss_bwd_tagged(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* g_terms, const float* verts,
                               const float* obj, const float* markers_target, const float* c_markers, const float* c_object,
                               const float* markers_rec, const float* hand_rec, const float* h2o_signed,
                               const float* o2h_signed, const int32_t* idx_m2o, const int32_t* idx_h2o,
                               const int32_t* idx_o2h, float* g_verts, int B, int O, int32_t* zero_tag, int32_t tag, int compact, void* stream);
int saga_smplx_fwd_sel(const saga_model_t* m, const saga_model_t* sub, const float* shape, const float* full_pose, const float* transl,
                       float* vertices, float* joints, float* feat, float* rot, float* jrest, float* chain, float* A, float* v_posed,
                       float* sel_rows, int B, void* stream, float* sel_verts = nullptr);
int saga_launch_hand_ring(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* verts, float* uniq_rows,
                          float* hand_normals, int B, void* stream);
int saga_grasp_loss_fast_ok(const saga_grasp_tables_t* tb, int O);
int saga_grasp_loss_fwd_pre(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* obj, const float* obj_normals,
                            const float* markers_target, const float* c_markers, const float* c_object, float* terms,
                            float* markers_rec, float* hand_rec, float* hand_normals, const float* uniq_rows, float* h2o_marker,
                            float* h2o_signed, float* o2h_signed, int32_t* idx_m2o, int32_t* idx_h2o, int32_t* idx_o2h, int B, int O,
                            void* stream);
int saga_grasp_loss_fwd_rec(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* verts, const float* markers_target,
                            float* terms, float* markers_rec, int32_t* idx_m2o, int B, void* stream);
int saga_launch_gemm_nt(int epi, const float* A, int lda, const float* Bt, int ldb, float* C, int ldc, const float* bias,
                        const float* aux, int M, int N, int K, hipStream_t st);

#define VS SAGA_VAR_STRIDE
#define X6_LD 128            // decoder output row: n_body_joints*6 (126) padded to one GEMM tile
#define NB_MAX 21
#define NTERM SAGA_GRASP_NTERMS

struct saga_decoder {
    int latent, hidden, nb, n_pca, device;
    float *w1, *b1, *w2, *b2, *w3, *b3;   // forward B operands [N][K]: w1 [hidden][latent], w2 [hidden][hidden], w3 [128][hidden]
    float *w1t, *w2t, *w3t;               // backward B operands: w1t [128][hidden] (rows >= latent 0), w2t, w3t [hidden][128]
    float* hand_comp;                      // [2][n_pca][45]
    float* pose_mean;                      // [165]
};

template <typename T>
static int dec_upload(T** dst, const std::vector<T>& src) {
    SAGA_HIP(hipMalloc((void**)dst, src.size() * sizeof(T)));
    SAGA_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

extern "C" void saga_decoder_destroy(saga_decoder_t* d) {
    if (!d) return;
    void* ptrs[] = {d->w1, d->b1, d->w2, d->b2, d->w3, d->b3, d->w1t, d->w2t, d->w3t, d->hand_comp, d->pose_mean};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete d;
}

extern "C" int saga_decoder_create(saga_decoder_t** out, const saga_decoder_desc_t* desc) {
    SAGA_CHECK_ARG(out && desc, "saga_decoder_create: null argument");
    *out = nullptr;
    const int L = desc->latent, H = desc->hidden, nb = desc->n_body_joints, P = desc->n_pca;
    SAGA_CHECK_ARG(L == 32 && nb == NB_MAX && P == 24,
                   "saga_decoder_create: the packed variable row is laid out for latent 32 / 21 body joints / 24 hand PCA "
                   "components (got %d / %d / %d)", L, nb, P);
    SAGA_CHECK_ARG(H > 0 && H % 128 == 0, "saga_decoder_create: hidden=%d must be a multiple of 128", H);
    SAGA_CHECK_ARG(desc->w1_host && desc->b1_host && desc->w2_host && desc->b2_host && desc->w3_host && desc->b3_host &&
                       desc->lhand_components_host && desc->rhand_components_host && desc->pose_mean_host,
                   "saga_decoder_create: null array");
    const int n_out = nb * 6;
    saga_decoder* d = new saga_decoder();
    memset(d, 0, sizeof(*d));
    d->latent = L; d->hidden = H; d->nb = nb; d->n_pca = P;
    (void)hipGetDevice(&d->device);
    std::vector<float> w1(desc->w1_host, desc->w1_host + (size_t)H * L), b1(desc->b1_host, desc->b1_host + H);
    std::vector<float> w2(desc->w2_host, desc->w2_host + (size_t)H * H), b2(desc->b2_host, desc->b2_host + H);
    std::vector<float> w3((size_t)X6_LD * H, 0.f), b3(X6_LD, 0.f);
    for (int o = 0; o < n_out; ++o) {
        b3[o] = desc->b3_host[o];
        for (int k = 0; k < H; ++k) w3[(size_t)o * H + k] = desc->w3_host[(size_t)o * H + k];
    }
    std::vector<float> w1t((size_t)128 * H, 0.f), w2t((size_t)H * H), w3t((size_t)H * X6_LD, 0.f);
    for (int o = 0; o < H; ++o)
        for (int k = 0; k < L; ++k) w1t[(size_t)k * H + o] = w1[(size_t)o * L + k];
    for (int o = 0; o < H; ++o)
        for (int k = 0; k < H; ++k) w2t[(size_t)k * H + o] = w2[(size_t)o * H + k];
    for (int o = 0; o < n_out; ++o)
        for (int k = 0; k < H; ++k) w3t[(size_t)k * X6_LD + o] = w3[(size_t)o * H + k];
    std::vector<float> hc((size_t)2 * P * 45), pm(desc->pose_mean_host, desc->pose_mean_host + 165);
    for (int i = 0; i < P * 45; ++i) { hc[i] = desc->lhand_components_host[i]; hc[P * 45 + i] = desc->rhand_components_host[i]; }
    int rc;
    if ((rc = dec_upload(&d->w1, w1)) || (rc = dec_upload(&d->b1, b1)) || (rc = dec_upload(&d->w2, w2)) ||
        (rc = dec_upload(&d->b2, b2)) || (rc = dec_upload(&d->w3, w3)) || (rc = dec_upload(&d->b3, b3)) ||
        (rc = dec_upload(&d->w1t, w1t)) || (rc = dec_upload(&d->w2t, w2t)) || (rc = dec_upload(&d->w3t, w3t)) ||
        (rc = dec_upload(&d->hand_comp, hc)) || (rc = dec_upload(&d->pose_mean, pm))) {
        saga_decoder_destroy(d);
        return rc;
    }
    *out = d;
    return 0;
}

// ---- workspace layout ------------------------------------------------------------------------------------------
struct FitWs {
    float *h1, *h2, *x6, *jac, *shape, *full_pose, *transl;               // parameter maps
    float *feat, *rot, *jrest, *chain, *A, *v_posed, *verts, *joints;     // SMPL-X forward
    float *terms, *g_terms, *hand_rec, *hand_nrm, *h2o_marker, *h2o_signed, *o2h_signed, *uniq_rows;
    int32_t *idx_m2o, *idx_h2o, *idx_o2h;
    float *g_verts, *gvp, *slab, *dA, *g_shape, *g_pose, *g_transl;       // SMPL-X backward
    float *g_x6, *g_h, *g_z, *grad;                                       // reverse parameter maps
    int32_t* zero_tag;                                                    // [B] rows of g_verts already zero outside the touched vertices
    int32_t* hdr;                                                         // word 0: layout hash of the last saga_fit_step on this workspace
    size_t total;
};

static FitWs fit_ws_carve(float* base, const saga_model* m, const saga_grasp_tables_t* tb, int hidden, int B, int O) {
    FitWs w;
    size_t off = 0;
    auto take = [&](size_t n) {
        float* p = base ? base + off : nullptr;
        off += (n + 63) / 64 * 64;       // 256-B aligned sub-buffers
        return p;
    };
    const size_t b = (size_t)B;
    const int M = tb ? tb->n_markers : 0, H = tb ? tb->n_hand : 0;
    const int V = m ? m->V : 0, Npad = m ? m->Npad : 0, S = m ? m->S : 0, njo = m ? NJ + m->E + m->L : 0;
    w.hdr = (int32_t*)take(64);          // first, so that every layout of one workspace finds it at the same address
    w.h1 = take(b * hidden); w.h2 = take(b * hidden); w.x6 = take(b * X6_LD); w.jac = take(b * (NB_MAX + 1) * 18);
    w.shape = take(b * (S > 0 ? S : 1)); w.full_pose = take(b * 165); w.transl = take(b * 3);
    w.feat = take(b * FEAT_K); w.rot = take(b * NJ * 9); w.jrest = take(b * NJ * 3); w.chain = take(b * NJ * 12);
    w.A = take(b * NJ * 12); w.v_posed = take(b * Npad); w.verts = take(b * V * 3); w.joints = take(b * njo * 3);
    w.terms = take(b * NTERM); w.g_terms = take(b * NTERM); w.hand_rec = take(b * H * 3); w.hand_nrm = take(b * H * 3);
    w.h2o_marker = take(b * M); w.h2o_signed = take(b * H); w.o2h_signed = take(b * O);
    w.uniq_rows = take(b * (tb ? tb->n_unique : 0) * 3);
    w.idx_m2o = (int32_t*)take(b * M); w.idx_h2o = (int32_t*)take(b * H); w.idx_o2h = (int32_t*)take(b * O);
    w.g_verts = take(b * V * 3); w.gvp = take(b * Npad); w.slab = take((size_t)SAGA_BWD_SPLITK * b * FEAT_K);
    w.dA = take(b * SAGA_DA_STRIDE); w.g_shape = take(b * (S > 0 ? S : 1)); w.g_pose = take(b * 165); w.g_transl = take(b * 3);
    w.g_x6 = take(b * X6_LD); w.g_h = take(b * hidden); w.g_z = take(b * 128); w.grad = take(b * VS);
    w.zero_tag = (int32_t*)take(b);
    w.total = off;
    return w;
}

extern "C" size_t saga_fit_workspace_floats(const saga_model_t* m, const saga_grasp_tables_t* tb, int B, int O) {
    if (!m || !tb || B <= 0 || O <= 0) return 0;
    return fit_ws_carve(nullptr, m, tb, 512 /* upper bound checked in saga_fit_step */, B, O).total;
}

extern "C" long long saga_fit_workspace_offset(const saga_model_t* m, const saga_grasp_tables_t* tb, int B, int O, const char* name) {
    if (!m || !tb || B <= 0 || O <= 0 || !name) return -1;
    float* base = (float*)(uintptr_t)256;         // carve against a fake (never dereferenced) base to read the offsets
    FitWs w = fit_ws_carve(base, m, tb, 512, B, O);
    struct { const char* n; const void* p; } tab[] = {
        {"full_pose", w.full_pose}, {"shape", w.shape}, {"vertices", w.verts}, {"joints", w.joints}, {"terms", w.terms},
        {"hand_rec", w.hand_rec}, {"hand_normals", w.hand_nrm}, {"h2o_marker", w.h2o_marker}, {"h2o_signed", w.h2o_signed},
        {"o2h_signed", w.o2h_signed}, {"idx_m2o", w.idx_m2o}, {"idx_h2o", w.idx_h2o}, {"idx_o2h", w.idx_o2h},
        {"g_vertices", w.g_verts}, {"g_full_pose", w.g_pose}, {"g_shape", w.g_shape}, {"g_transl", w.g_transl},
        {"grad", w.grad}, {"g_z", w.g_z}, {"x6", w.x6}};
    for (auto& e : tab)
        if (!strcmp(e.n, name)) return (long long)((const float*)e.p - base);
    return -1;
}

// ---- kernels ---------------------------------------------------------------------------------------------------
#define PK_THREADS 128

// One block per candidate.  threads 0..21: 6D -> axis-angle (0 = global orientation from the variable row, 1..21 = the
// decoder's joints); 32..121: hand PCA (left 45 | right 45) + mean; 122: jaw; 123: eyes; 124: shape; 125: transl.
__global__ __launch_bounds__(PK_THREADS) void params_fwd_kernel(const float* __restrict__ vars, const float* __restrict__ x6,
                                                                const float* __restrict__ hand_comp,
                                                                const float* __restrict__ pose_mean, float* __restrict__ full_pose,
                                                                float* __restrict__ shape, float* __restrict__ transl,
                                                                float* __restrict__ jac, float* __restrict__ body_params, int S,
                                                                int n_pca, int32_t* __restrict__ hdr_clear) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (hdr_clear && b == 0 && tid == 126) hdr_clear[0] = 0;      // saga_fit_decode: this workspace's last layout is void
    const float* v = vars + (size_t)b * VS;
    float* fp = full_pose ? full_pose + (size_t)b * 165 : nullptr;
    float* bp = body_params ? body_params + (size_t)b * SAGA_BODY_PARAM_COUNT : nullptr;
    if (tid <= NB_MAX) {
        float x[6], a[3], J[18];
        const float* src = tid == 0 ? v + SAGA_VAR_ROT6D : x6 + (size_t)b * X6_LD + (tid - 1) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) x[k] = src[k];
        cont6d_to_aa_dual(x, a, J);
        if (fp) {
#pragma unroll
            for (int c = 0; c < 3; ++c) fp[tid * 3 + c] = a[c] + pose_mean[tid * 3 + c];
        }
        if (jac) {
            float* Jo = jac + ((size_t)b * (NB_MAX + 1) + tid) * 18;
#pragma unroll
            for (int k = 0; k < 18; ++k) Jo[k] = J[k];
        }
        if (bp) {
            // transl 0:3 | global_orient 3:6 | betas 6:16 | body_pose 16:79 | lh 79:103 | rh 103:127 | leye 127:130 | reye 130:133
            float* o = tid == 0 ? bp + 3 : bp + 16 + (tid - 1) * 3;
            o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
        }
    } else if (tid >= 32 && tid < 122) {
        if (fp) {
            const int h = tid - 32, side = h / 45, c = h % 45;
            const float* coef = v + SAGA_VAR_HAND + side * n_pca;
            const float* comp = hand_comp + (size_t)side * n_pca * 45 + c;
            float acc = 0.f;
            for (int i = 0; i < n_pca; ++i) acc = fmaf(coef[i], comp[i * 45], acc);
            fp[75 + h] = acc + pose_mean[75 + h];
        }
    } else if (tid == 122) {
        if (fp) { fp[66] = pose_mean[66]; fp[67] = pose_mean[67]; fp[68] = pose_mean[68]; }
    } else if (tid == 123) {
        if (fp)
            for (int k = 0; k < 6; ++k) fp[69 + k] = v[SAGA_VAR_EYE + k] + pose_mean[69 + k];
        if (bp)
            for (int k = 0; k < 6; ++k) bp[127 + k] = v[SAGA_VAR_EYE + k];
    } else if (tid == 124) {
        if (shape)
            for (int k = 0; k < S; ++k) shape[(size_t)b * S + k] = k < 10 ? v[SAGA_VAR_BETAS + k] : 0.f;   // expression = 0
        if (bp) {
            for (int k = 0; k < 3; ++k) bp[k] = v[SAGA_VAR_TRANSL + k];
            for (int k = 0; k < 10; ++k) bp[6 + k] = v[SAGA_VAR_BETAS + k];
            for (int k = 0; k < 48; ++k) bp[79 + k] = v[SAGA_VAR_HAND + k];
        }
    } else if (tid == 125) {
        if (transl)
            for (int k = 0; k < 3; ++k) transl[(size_t)b * 3 + k] = v[SAGA_VAR_TRANSL + k];
    }
}

struct LossCfg { int stage, only_rec, prior, custom; float w_rec, w_foot, reg_vpose, reg_hand, reg_eye, reg_shape; };
// gradient coefficients of the regularisers, d/dx of w * mean(x^2) = (2 w / n) x  (0: the term is off in this stage)
struct RegCfg { float c_vpose, c_hand, c_eye, c_shape; };     // c_shape: custom_weights only (reg_shape * mean(betas^2))

// weighted sum of fittingop.py:215-223 (same order of additions) + the gradient of it w.r.t. the eight fused terms
// Also the workspace's layout check: when the last saga_fit_step on this workspace carved it differently (another B, O,
// mesh or table set), that call wrote into what is g_verts now, so every "row already zero" mark of this layout is void.
__global__ void loss_sum_kernel(const float* __restrict__ terms, const float* __restrict__ vars, LossCfg cfg,
                                float* __restrict__ g_terms, float* __restrict__ loss, float* __restrict__ out, int B,
                                const int32_t* __restrict__ hdr, int32_t layout, int32_t* __restrict__ zero_tag) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (hdr[0] != layout) zero_tag[b] = 0;
    const float* t = terms + (size_t)b * NTERM;
    const float* v = vars + (size_t)b * VS;
    float sv = 0.f, sh = 0.f, se = 0.f;
    for (int k = 0; k < 32; ++k) sv = fmaf(v[SAGA_VAR_VPOSE + k], v[SAGA_VAR_VPOSE + k], sv);
    for (int k = 0; k < 48; ++k) sh = fmaf(v[SAGA_VAR_HAND + k], v[SAGA_VAR_HAND + k], sh);
    for (int k = 0; k < 6; ++k) se = fmaf(v[SAGA_VAR_EYE + k], v[SAGA_VAR_EYE + k], se);
    if (cfg.custom) {        // marker-fit objective with the caller's weights (GraspMotion-Opt stage 1, :309-325, in its order)
        float sb = 0.f;
        for (int k = 0; k < 10; ++k) sb = fmaf(v[SAGA_VAR_BETAS + k], v[SAGA_VAR_BETAS + k], sb);
        const float rv = cfg.reg_vpose * (sv / 32.f), rs = cfg.reg_shape * (sb / 10.f), rh = cfg.reg_hand * (sh / 48.f),
                    re = cfg.reg_eye * (se / 6.f);
        float total = cfg.w_rec * t[0];
        total += rv;
        total += rs;
        total += rh;
        total += cfg.w_foot * t[2];
        total += re;
        loss[b] = total;
        float* g = g_terms + (size_t)b * NTERM;
#pragma unroll
        for (int k = 0; k < NTERM; ++k) g[k] = 0.f;
        g[0] = cfg.w_rec; g[2] = cfg.w_foot;
        if (out) {
            float* o = out + (size_t)b * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k) o[k] = 0.f;
            o[0] = total; o[1] = t[0]; o[2] = t[1]; o[8] = t[2]; o[9] = ((rv + rs) + rh) + re;
        }
        return;
    }
    const float reg_v = 0.0001f * (sv / 32.f), reg_h = 0.0005f * (sh / 48.f), reg_e = 0.0001f * (se / 6.f);
    const float on = cfg.only_rec ? 0.f : 1.f;
    const float s2 = cfg.stage > 1 ? 1.f : 0.f, s1 = cfg.stage > 0 ? 1.f : 0.f;
    const float w_contact = on * (cfg.prior ? 0.f : 1.f) * s2 * 15.f;
    const float w_prior = on * (cfg.prior ? 1.f : 0.f) * s2 * 15.f;
    const float w_coll = on * s2 * 10.f;
    float total = 1.f * t[0];
    total += w_contact * (t[3] + t[4]);
    total += w_prior * t[5];
    total += w_coll * (t[6] + t[7]);
    total += s2 * reg_h;
    total += s2 * reg_e;
    total += s1 * reg_v;
    total += 0.1f * t[2];
    loss[b] = total;
    float* g = g_terms + (size_t)b * NTERM;
    g[0] = 1.f; g[1] = 0.f; g[2] = 0.1f; g[3] = w_contact; g[4] = w_contact; g[5] = w_prior; g[6] = w_coll; g[7] = w_coll;
#pragma unroll
    for (int k = 8; k < NTERM; ++k) g[k] = 0.f;
    if (out) {
        float* o = out + (size_t)b * 16;
        o[0] = total; o[1] = t[0]; o[2] = t[1]; o[3] = t[3]; o[4] = t[4]; o[5] = t[5]; o[6] = t[6]; o[7] = t[7]; o[8] = t[2];
        o[9] = (reg_v + reg_h) + reg_e;
        o[10] = t[8]; o[11] = t[9]; o[12] = t[10]; o[13] = t[11]; o[14] = 0.f; o[15] = 0.f;
    }
}

// reverse of params_fwd_kernel: gradient rows grad[B,VS] (every group but vpose, which comes out of the decoder's
// backward) and g_x6[B,128] for the decoder.
__global__ __launch_bounds__(PK_THREADS) void params_bwd_kernel(const float* __restrict__ vars, const float* __restrict__ jac,
                                                                const float* __restrict__ hand_comp,
                                                                const float* __restrict__ g_pose, const float* __restrict__ g_shape,
                                                                const float* __restrict__ g_transl, float* __restrict__ grad,
                                                                float* __restrict__ g_x6, int S, int n_pca, RegCfg reg) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* v = vars + (size_t)b * VS;
    const float* gp = g_pose + (size_t)b * 165;
    float* g = grad + (size_t)b * VS;
    if (tid <= NB_MAX) {
        const float* J = jac + ((size_t)b * (NB_MAX + 1) + tid) * 18;
        const float g0 = gp[tid * 3], g1 = gp[tid * 3 + 1], g2 = gp[tid * 3 + 2];
        float* o = tid == 0 ? g + SAGA_VAR_ROT6D : g_x6 + (size_t)b * X6_LD + (tid - 1) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = (J[k] * g0 + J[6 + k] * g1) + J[12 + k] * g2;
    } else if (tid >= 32 && tid < 32 + 48) {
        const int i = tid - 32, side = i / n_pca, ii = i % n_pca;
        const float* comp = hand_comp + ((size_t)side * n_pca + ii) * 45;
        const float* gh = gp + 75 + side * 45;
        float acc = 0.f;
        for (int c = 0; c < 45; ++c) acc = fmaf(comp[c], gh[c], acc);
        if (reg.c_hand != 0.f) acc += reg.c_hand * v[SAGA_VAR_HAND + i];           // d/d hand of 5e-4 * mean(hand^2) (stage 2)
        g[SAGA_VAR_HAND + i] = acc;
    } else if (tid >= 80 && tid < 86) {
        const int k = tid - 80;
        float acc = gp[69 + k];
        if (reg.c_eye != 0.f) acc += reg.c_eye * v[SAGA_VAR_EYE + k];
        g[SAGA_VAR_EYE + k] = acc;
    } else if (tid >= 86 && tid < 96) {
        const int k = tid - 86;
        float acc = k < S ? g_shape[(size_t)b * S + k] : 0.f;
        if (reg.c_shape != 0.f) acc += reg.c_shape * v[SAGA_VAR_BETAS + k];
        g[SAGA_VAR_BETAS + k] = acc;
    } else if (tid >= 96 && tid < 99) {
        g[SAGA_VAR_TRANSL + (tid - 96)] = g_transl ? g_transl[(size_t)b * 3 + (tid - 96)] : 0.f;
    } else if (tid >= 126) {
        g_x6[(size_t)b * X6_LD + tid] = 0.f;      // columns 126, 127 of the decoder-output gradient
    }
}

struct AdamCfg { int mask; float c_vpose, step_size, beta1, beta2, bc2_sqrt, eps; };

__device__ __forceinline__ int var_group(int col) {
    return col < SAGA_VAR_HAND ? SAGA_GROUP_VPOSE
         : col < SAGA_VAR_EYE ? SAGA_GROUP_HAND
         : col < SAGA_VAR_BETAS ? SAGA_GROUP_EYE
         : col < SAGA_VAR_TRANSL ? SAGA_GROUP_BETAS
         : col < SAGA_VAR_ROT6D ? SAGA_GROUP_TRANSL
         : SAGA_GROUP_ROT6D;
}

// torch.optim.Adam (no amsgrad, no weight decay): exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2*v + (1-b2) g g;
// denom = sqrt(v) / sqrt(1 - b2^t) + eps; p -= lr / (1 - b1^t) * m / denom
// Last kernel of an iteration: it also records the layout this iteration carved the workspace with (read by the next
// call's loss_sum_kernel; every block of that kernel has finished before this one starts).
__global__ void adam_kernel(float* __restrict__ vars, float* __restrict__ m, float* __restrict__ vv,
                            const float* __restrict__ grad, const float* __restrict__ g_z, AdamCfg c, int B,
                            int32_t* __restrict__ hdr, int32_t layout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * SAGA_VAR_COUNT) return;
    if (i == 0 && hdr) hdr[0] = layout;
    const int b = i / SAGA_VAR_COUNT, col = i % SAGA_VAR_COUNT;
    if (!(var_group(col) & c.mask)) return;
    const size_t o = (size_t)b * VS + col;
    float g;
    if (col < SAGA_VAR_HAND && g_z) {                  // (g_z NULL: complete gradient rows, saga_adam_rows)
        g = g_z[(size_t)b * 128 + col];
        if (c.c_vpose != 0.f) g += c.c_vpose * vars[o];                            // d/dz of 1e-4 * mean(z^2) (stages 1, 2)
    } else {
        g = grad[o];
    }
    float mm = m[o], v2 = vv[o];
    mm = mm + (g - mm) * (1.f - c.beta1);
    v2 = v2 * c.beta2 + (1.f - c.beta2) * g * g;
    m[o] = mm; vv[o] = v2;
    const float denom = sqrtf(v2) / c.bc2_sqrt + c.eps;
    vars[o] = vars[o] - c.step_size * (mm / denom);
}

__global__ __launch_bounds__(PK_THREADS) void best_update_kernel(const float* __restrict__ loss, const float* __restrict__ vars,
                                                                 const float* __restrict__ markers, float* __restrict__ best_loss,
                                                                 float* __restrict__ best_vars, float* __restrict__ best_markers,
                                                                 int M) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float l = loss[b];
    const bool better = l < best_loss[b];
    __syncthreads();                          // every thread has read best_loss[b] before thread 0 lowers it
    if (!better) return;
    for (int k = tid; k < VS; k += PK_THREADS) best_vars[(size_t)b * VS + k] = vars[(size_t)b * VS + k];
    for (int k = tid; k < M * 3; k += PK_THREADS) best_markers[(size_t)b * M * 3 + k] = markers[(size_t)b * M * 3 + k];
    if (tid == 0) best_loss[b] = l;
}

// ---- host orchestration ------------------------------------------------------------------------------------------
struct Prof {
    void* p;
    Prof(int id, void* st) : p(saga_prof_begin(id, st)) {}
    ~Prof() { saga_prof_end(p); }
};

static int decoder_fwd(const saga_decoder* d, const float* vars, const FitWs& w, int B, hipStream_t st) {
    int rc;
    if ((rc = saga_launch_gemm_nt(2, vars + SAGA_VAR_VPOSE, VS, d->w1, d->latent, w.h1, d->hidden, d->b1, nullptr, B, d->hidden,
                                  d->latent, st)))
        return rc;
    if ((rc = saga_launch_gemm_nt(2, w.h1, d->hidden, d->w2, d->hidden, w.h2, d->hidden, d->b2, nullptr, B, d->hidden, d->hidden, st)))
        return rc;
    return saga_launch_gemm_nt(0, w.h2, d->hidden, d->w3, d->hidden, w.x6, X6_LD, d->b3, nullptr, B, X6_LD, d->hidden, st);
}

extern "C" int saga_fit_decode(const saga_decoder_t* d, const float* vars, float* body_params, float* ws, int B, void* stream) {
    SAGA_CHECK_ARG(d && vars && body_params && ws, "saga_fit_decode: null argument");
    SAGA_CHECK_DEVICE(d, "saga_fit_decode");
    SAGA_CHECK_ARG(B >= 0 && B <= 65535 && d->hidden <= 512, "saga_fit_decode: bad sizes");
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    FitWs w = fit_ws_carve(ws, nullptr, nullptr, 512, B, 1);      // only h1 / h2 / x6 are touched (they come first)
    int rc;
    if ((rc = decoder_fwd(d, vars, w, B, st))) return rc;
    params_fwd_kernel<<<B, PK_THREADS, 0, st>>>(vars, w.x6, d->hand_comp, d->pose_mean, nullptr, nullptr, nullptr, nullptr, body_params,
                                                0, d->n_pca, w.hdr);
    SAGA_LAUNCH_CHECK();
    return 0;
}

// ---- the parameter maps as a stand-alone operator (autograd callers: GraspMotion-Opt stage 2) --------------------------------
struct ParamsWs { float *h1, *h2, *x6, *jac, *g_x6, *g_h, *g_h1, *g_z; size_t total; };
static ParamsWs params_ws_carve(float* base, int hidden, int B) {
    ParamsWs w;
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return p; };
    const size_t b = (size_t)B;
    w.h1 = take(b * hidden); w.h2 = take(b * hidden); w.x6 = take(b * X6_LD); w.jac = take(b * (NB_MAX + 1) * 18);
    w.g_x6 = take(b * X6_LD); w.g_h = take(b * hidden); w.g_z = take(b * 128);
    w.g_h1 = take(b * hidden);      // (last: the backward leaves the forward's activations h1 / h2 intact, so it can run again)
    w.total = off;
    return w;
}

// grad rows: the latent columns come out of the decoder's backward (g_z), the pad columns are zero
__global__ void params_grad_finish_kernel(const float* __restrict__ g_z, const float* __restrict__ vars, float c_vpose,
                                          float* __restrict__ grad, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * VS) return;
    const int b = i / VS, col = i - b * VS;
    if (col < SAGA_VAR_HAND) {
        float g = g_z[(size_t)b * 128 + col];
        if (c_vpose != 0.f) g += c_vpose * vars[i];          // a caller's latent regulariser (saga_params_bwd_reg)
        grad[i] = g;
    } else if (col >= SAGA_VAR_COUNT) grad[i] = 0.f;
}

extern "C" size_t saga_params_workspace_floats(const saga_decoder_t* d, int B) {
    if (!d || B <= 0) return 0;
    return params_ws_carve(nullptr, d->hidden, B).total;
}

extern "C" int saga_params_fwd(const saga_decoder_t* d, const float* vars, float* full_pose, float* shape, float* transl, int n_shape,
                               float* ws, int B, void* stream) {
    SAGA_CHECK_ARG(d, "saga_params_fwd: null decoder");
    SAGA_CHECK_ARG(B >= 0 && B <= 65535 && d->hidden <= 512 && n_shape >= 10, "saga_params_fwd: bad sizes B=%d n_shape=%d", B, n_shape);
    if (B == 0) return 0;
    SAGA_CHECK_ARG(vars && full_pose && shape && transl && ws, "saga_params_fwd: null argument");
    SAGA_CHECK_DEVICE(d, "saga_params_fwd");
    hipStream_t st = (hipStream_t)stream;
    ParamsWs p = params_ws_carve(ws, d->hidden, B);
    FitWs w;
    memset(&w, 0, sizeof(w));
    w.h1 = p.h1; w.h2 = p.h2; w.x6 = p.x6;
    Prof prof(SAGA_PROF_PARAMS_FWD, stream);
    if (int rc = decoder_fwd(d, vars, w, B, st)) return rc;
    params_fwd_kernel<<<B, PK_THREADS, 0, st>>>(vars, p.x6, d->hand_comp, d->pose_mean, full_pose, shape, transl, p.jac, nullptr, n_shape,
                                                d->n_pca, nullptr);
    SAGA_LAUNCH_CHECK();
    return 0;
}

// saga_params_bwd with the gradient of a caller's regularisers c_vpose/2 * |latent|^2 + c_hand/2 * |hand|^2 folded into the rows
// (internal: saga_motion_step)
int saga_params_bwd_reg(const saga_decoder_t* d, const float* vars, const float* g_full_pose, const float* g_shape,
                        const float* g_transl, float* grad_vars, int n_shape, float* ws, int B, float c_vpose, float c_hand,
                        void* stream) {
    SAGA_CHECK_ARG(d, "saga_params_bwd: null decoder");
    SAGA_CHECK_ARG(B >= 0 && B <= 65535 && d->hidden <= 512 && n_shape >= 10, "saga_params_bwd: bad sizes B=%d n_shape=%d", B, n_shape);
    if (B == 0) return 0;
    SAGA_CHECK_ARG(vars && g_full_pose && grad_vars && ws, "saga_params_bwd: null argument");
    SAGA_CHECK_DEVICE(d, "saga_params_bwd");
    hipStream_t st = (hipStream_t)stream;
    ParamsWs p = params_ws_carve(ws, d->hidden, B);     // h1 / h2 / jac: what saga_params_fwd left there for these vars
    Prof prof(SAGA_PROF_PARAMS_BWD, stream);
    params_bwd_kernel<<<B, PK_THREADS, 0, st>>>(vars, p.jac, d->hand_comp, g_full_pose, g_shape, g_transl, grad_vars, p.g_x6,
                                                g_shape ? n_shape : 0, d->n_pca, RegCfg{0.f, c_hand, 0.f, 0.f});
    SAGA_LAUNCH_CHECK();
    int rc;
    if ((rc = saga_launch_gemm_nt(3, p.g_x6, X6_LD, d->w3t, X6_LD, p.g_h, d->hidden, nullptr, p.h2, B, d->hidden, X6_LD, st))) return rc;
    if ((rc = saga_launch_gemm_nt(3, p.g_h, d->hidden, d->w2t, d->hidden, p.g_h1, d->hidden, nullptr, p.h1, B, d->hidden, d->hidden, st)))
        return rc;
    if ((rc = saga_launch_gemm_nt(0, p.g_h1, d->hidden, d->w1t, d->hidden, p.g_z, 128, nullptr, nullptr, B, 128, d->hidden, st))) return rc;
    const int n = B * VS;
    params_grad_finish_kernel<<<(n + 255) / 256, 256, 0, st>>>(p.g_z, vars, c_vpose, grad_vars, B);
    SAGA_LAUNCH_CHECK();
    return 0;
}

extern "C" int saga_params_bwd(const saga_decoder_t* d, const float* vars, const float* g_full_pose, const float* g_shape,
                               const float* g_transl, float* grad_vars, int n_shape, float* ws, int B, void* stream) {
    return saga_params_bwd_reg(d, vars, g_full_pose, g_shape, g_transl, grad_vars, n_shape, ws, B, 0.f, 0.f, stream);
}

// torch.optim.Adam's update of the variable groups in `mask` from complete gradient rows grad [B,VS] (internal: saga_motion_step)
int saga_adam_rows(float* vars, float* adam_m, float* adam_v, const float* grad, int mask, int step, float lr, float beta1,
                   float beta2, float eps, int B, void* stream) {
    SAGA_CHECK_ARG(vars && adam_m && adam_v && grad && step >= 1 && B >= 0, "saga_adam_rows: bad arguments");
    if (B == 0) return 0;
    Prof p(SAGA_PROF_ADAM, stream);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamCfg ac{mask, 0.f, (float)((double)lr / bc1), beta1, beta2, (float)sqrt(bc2), eps};
    const int n = B * SAGA_VAR_COUNT;
    adam_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(vars, adam_m, adam_v, grad, nullptr, ac, B, nullptr, 0);
    SAGA_LAUNCH_CHECK();
    return 0;
}

extern "C" int saga_fit_best_update(const float* loss, const float* vars, const float* markers_rec, float* best_loss,
                                    float* best_vars, float* best_markers, int B, int M, void* stream) {
    SAGA_CHECK_ARG(loss && vars && markers_rec && best_loss && best_vars && best_markers && B >= 0 && M > 0,
                   "saga_fit_best_update: bad arguments");
    if (B == 0) return 0;
    best_update_kernel<<<B, PK_THREADS, 0, (hipStream_t)stream>>>(loss, vars, markers_rec, best_loss, best_vars, best_markers, M);
    SAGA_LAUNCH_CHECK();
    return 0;
}

// mb: NULL (dense backward over the whole mesh) or the subset model over tb->unique_ids (structurally sparse backward)
static int fit_step_impl(const saga_model_t* m, const saga_model_t* mb, const saga_decoder_t* d, const saga_grasp_tables_t* tb,
                         const saga_fit_problem_t* prob, const saga_fit_step_cfg_t* cfg, float* vars, float* adam_m,
                         float* adam_v, float* ws, float* loss, float* loss_terms, float* markers_rec, int B, void* stream) {
    SAGA_CHECK_ARG(m && d && tb && prob && cfg && vars && adam_m && adam_v && ws && loss && markers_rec,
                   "saga_fit_step: null argument");
    SAGA_CHECK_DEVICE(m, "saga_fit_step (model)");
    SAGA_CHECK_DEVICE(d, "saga_fit_step (decoder)");
    SAGA_CHECK_ARG(B >= 0 && B <= 65535, "saga_fit_step: batch %d outside [0, 65535]", B);
    SAGA_CHECK_ARG(d->hidden <= 512, "saga_fit_step: decoder hidden size %d > 512", d->hidden);
    SAGA_CHECK_ARG(m->S >= 10 || m->S == 0, "saga_fit_step: model has %d shape coefficients (< 10 betas)", m->S);
    SAGA_CHECK_ARG(cfg->stage >= 0 && cfg->stage <= 2 && cfg->adam_step >= 1, "saga_fit_step: stage %d / adam step %d",
                   cfg->stage, cfg->adam_step);
    SAGA_CHECK_ARG(!cfg->custom_weights || (cfg->stage == 0 && cfg->skip_gated_terms),
                   "saga_fit_step: custom_weights is a marker-fit objective: stage 0 with skip_gated_terms");
    SAGA_CHECK_ARG(prob->n_obj > 0 && prob->obj && prob->obj_normals && prob->markers_target && prob->c_markers && prob->c_object,
                   "saga_fit_step: incomplete problem");
    if (mb) {
        SAGA_CHECK_ARG(mb->sel_ids && mb->VF == m->V && mb->V == tb->n_unique && mb->S == m->S && !m->sel_ids,
                       "saga_fit_step_sparse: the backward model must be the subset of the forward model's mesh over the tables' "
                       "unique_ids (subset of %d vertices over a mesh of %d; tables list %d, mesh has %d)",
                       mb->V, mb->VF, tb->n_unique, m->V);
        SAGA_CHECK_ARG(2 * (size_t)mb->Npad + 3 * (size_t)mb->V <= (size_t)m->Npad,
                       "saga_fit_step_sparse: %d loss vertices of %d leave no room for the compact buffers", mb->V, m->V);
    }
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int O = prob->n_obj;
    FitWs w = fit_ws_carve(ws, m, tb, 512, B, O);
    // what the carve above depends on (+ the table set the zero-fill marks are for); never 0, the value a fresh workspace holds
    uint64_t lh = 0xCBF29CE484222325ull;
    for (uint64_t x : {(uint64_t)B, (uint64_t)O, (uint64_t)m->V, (uint64_t)m->Npad, (uint64_t)m->S, (uint64_t)(m->E + m->L),
                       (uint64_t)tb->n_markers, (uint64_t)tb->n_hand, (uint64_t)tb->n_foot, (uint64_t)tb->n_unique, (uint64_t)(uintptr_t)tb->unique_ids,
                       (uint64_t)(uintptr_t)ws})
        lh = (lh ^ x) * 0x100000001B3ull;
    int32_t layout = (int32_t)(lh ^ (lh >> 32));
    if (layout == 0) layout = 1;
    int rc;
    {   // optimiser variables -> shape / full_pose / transl
        Prof p(SAGA_PROF_PARAMS_FWD, stream);
        if ((rc = decoder_fwd(d, vars, w, B, st))) return rc;
        params_fwd_kernel<<<B, PK_THREADS, 0, st>>>(vars, w.x6, d->hand_comp, d->pose_mean, w.full_pose, w.shape, w.transl, w.jac,
                                                    nullptr, m->S, d->n_pca, nullptr);
        SAGA_LAUNCH_CHECK();
    }
    // sparse backward: the skinning forward also writes the v_posed rows of the loss vertices, compact, for the subset backward
    float* vps = mb ? w.gvp + (size_t)B * mb->Npad : nullptr;
    // every contact / collision term has weight 0 in stages 0 and 1 (and with only_rec): the reference still evaluates and logs
    // them; with skip_gated_terms they are neither evaluated nor logged (written as 0) -- no NN scan, no hand normals
    const bool gated_off = cfg->skip_gated_terms && (cfg->stage < 2 || cfg->only_rec);
    if ((rc = saga_smplx_fwd_sel(m, mb, w.shape, w.full_pose, w.transl, w.verts, w.joints, w.feat, w.rot, w.jrest, w.chain, w.A,
                                 w.v_posed, vps, B, stream)))
        return rc;
    // One small kernel stages the hand's face ring in LDS, takes the hand normals from there and writes the vertices the loss reads
    // as compact rows; the loss kernel then touches neither the mesh nor its faces (tables with the optional ring fields).
    // SAGA_FIT_NO_PRE=1: the loss kernel gathers and builds the normals by itself -- the cross-check path, same results bit for bit.
    int pre_done = 0;
    // Only for meshes whose faces join far-apart vertices (m->scattered_faces: the SURVEY synthetic model): on a mesh-like face
    // list the corner reads hit L2 and hide under the other blocks' scans -- measured, B = 4096: ring kernel 0.09 + loss 0.86 ms
    // against 0.92 ms for the loss kernel alone there; 0.16 + 0.87 against 1.10 on random faces (SAGA_FIT_PRE=1 forces the ring path).
    if (!gated_off && saga_grasp_loss_fast_ok(tb, O) && m->F > 0 && getenv("SAGA_FIT_NO_PRE") == nullptr &&
        (m->scattered_faces || getenv("SAGA_FIT_PRE") != nullptr)) {
        rc = saga_launch_hand_ring(m, tb, w.verts, w.uniq_rows, w.hand_nrm, B, stream);
        if (rc != 0 && rc != 1) return rc;
        pre_done = rc == 0;
    }
    if (gated_off) {
        if ((rc = saga_grasp_loss_fwd_rec(m, tb, w.verts, prob->markers_target, w.terms, markers_rec, w.idx_m2o, B, stream))) return rc;
    } else if (pre_done) {
        if ((rc = saga_grasp_loss_fwd_pre(m, tb, prob->obj, prob->obj_normals, prob->markers_target, prob->c_markers, prob->c_object,
                                          w.terms, markers_rec, w.hand_rec, w.hand_nrm, w.uniq_rows, w.h2o_marker, w.h2o_signed,
                                          w.o2h_signed, w.idx_m2o, w.idx_h2o, w.idx_o2h, B, O, stream)))
            return rc;
    } else if ((rc = saga_grasp_loss_fwd(m, tb, w.verts, prob->obj, prob->obj_normals, prob->markers_target, prob->c_markers,
                                         prob->c_object, w.terms, markers_rec, w.hand_rec, w.hand_nrm, w.h2o_marker, w.h2o_signed,
                                         w.o2h_signed, w.idx_m2o, w.idx_h2o, w.idx_o2h, B, O, stream)))
        return rc;
    {
        Prof p(SAGA_PROF_LOSS_SUM, stream);
        LossCfg lc{cfg->stage, cfg->only_rec, cfg->prior_contact, cfg->custom_weights, cfg->w_rec, cfg->w_foot, cfg->reg_vpose,
                   cfg->reg_hand,  cfg->reg_eye, cfg->reg_shape};
        loss_sum_kernel<<<(B + 127) / 128, 128, 0, st>>>(w.terms, vars, lc, w.g_terms, loss, loss_terms, B, w.hdr, layout, w.zero_tag);
        SAGA_LAUNCH_CHECK();
    }
    if (mb) {
        // Structurally sparse backward: the loss reads only the tables' vertices, so g_vertices is zero everywhere else and
        // every product the dense backward forms with those rows is an exact zero.  The loss backward writes the compact
        // gradient [B,U,3], the skinning forward has written the v_posed rows of those vertices, and the SMPL-X backward runs on the
        // subset model (same joint regression, same chain; blend operand and weight tables of the U vertices only).  The
        // three compact buffers live in the dense g_v_posed workspace (contents undefined between calls anyway).
        float* gvp_s = w.gvp;
        float* g_compact = vps + (size_t)B * mb->Npad;
        if ((rc = saga_grasp_loss_bwd_tagged(m, tb, w.g_terms, w.verts, prob->obj, prob->markers_target, prob->c_markers,
                                             prob->c_object, markers_rec, w.hand_rec, w.h2o_signed, w.o2h_signed, w.idx_m2o,
                                             w.idx_h2o, w.idx_o2h, g_compact, B, O, nullptr, 0, 1, stream)))
            return rc;
        if ((rc = saga_smplx_bwd(mb, g_compact, nullptr, w.full_pose, w.rot, w.jrest, w.chain, w.A, vps, w.g_shape, w.g_pose,
                                 w.g_transl, gvp_s, w.slab, w.dA, nullptr, B, stream)))
            return rc;
    } else {
        // the dense gradient rows of the workspace are zero-filled once: every call rewrites the same (tables') vertices.
        // The tag ties the "already zero" mark to this table set, mesh and workspace address; it is never 0, so a
        // zero-initialised workspace (the contract) always fills on first use.  A call that carved the workspace differently
        // in between (another B / O / mesh) has had its marks cleared by loss_sum_kernel above (header word != layout).
        uint64_t h = (uint64_t)(uintptr_t)tb->unique_ids * 0x9E3779B97F4A7C15ull ^ (uint64_t)(uintptr_t)w.g_verts * 0xC2B2AE3D27D4EB4Full ^
                     ((uint64_t)tb->n_unique << 32) ^ (uint64_t)m->V;
        int32_t tag = (int32_t)(h ^ (h >> 32));
        if (tag == 0) tag = 1;
        if ((rc = saga_grasp_loss_bwd_tagged(m, tb, w.g_terms, w.verts, prob->obj, prob->markers_target, prob->c_markers,
                                             prob->c_object, markers_rec, w.hand_rec, w.h2o_signed, w.o2h_signed, w.idx_m2o,
                                             w.idx_h2o, w.idx_o2h, w.g_verts, B, O, w.zero_tag, tag, 0, stream)))
            return rc;
        if ((rc = saga_smplx_bwd(m, w.g_verts, nullptr, w.full_pose, w.rot, w.jrest, w.chain, w.A, w.v_posed, w.g_shape, w.g_pose,
                                 w.g_transl, w.gvp, w.slab, w.dA, nullptr, B, stream)))
            return rc;
    }
    // regulariser gradients: fittingop.py:177-181,215-223 (vpose from stage 1 on, hand and eye in stage 2), or the caller's weights
    RegCfg reg{cfg->stage > 0 ? 2.f * 0.0001f / 32.f : 0.f, cfg->stage > 1 ? 2.f * 0.0005f / 48.f : 0.f,
               cfg->stage > 1 ? 2.f * 0.0001f / 6.f : 0.f, 0.f};
    if (cfg->custom_weights)
        reg = RegCfg{2.f * cfg->reg_vpose / 32.f, 2.f * cfg->reg_hand / 48.f, 2.f * cfg->reg_eye / 6.f, 2.f * cfg->reg_shape / 10.f};
    {
        Prof p(SAGA_PROF_PARAMS_BWD, stream);
        params_bwd_kernel<<<B, PK_THREADS, 0, st>>>(vars, w.jac, d->hand_comp, w.g_pose, w.g_shape, w.g_transl, w.grad, w.g_x6, m->S,
                                                    d->n_pca, reg);
        SAGA_LAUNCH_CHECK();
        if (cfg->update_mask & SAGA_GROUP_VPOSE) {
            // decoder backward: g_h2 = (g_x6 . W3) * lrelu'(h2); g_h1 = (g_h2 . W2) * lrelu'(h1); g_z = g_h1 . W1
            if ((rc = saga_launch_gemm_nt(3, w.g_x6, X6_LD, d->w3t, X6_LD, w.g_h, d->hidden, nullptr, w.h2, B, d->hidden, X6_LD, st)))
                return rc;
            // in place is not possible (A operand = C); ping-pong through h2, whose values are no longer needed
            if ((rc = saga_launch_gemm_nt(3, w.g_h, d->hidden, d->w2t, d->hidden, w.h2, d->hidden, nullptr, w.h1, B, d->hidden,
                                          d->hidden, st)))
                return rc;
            if ((rc = saga_launch_gemm_nt(0, w.h2, d->hidden, d->w1t, d->hidden, w.g_z, 128, nullptr, nullptr, B, 128, d->hidden, st)))
                return rc;
        }
    }
    {
        Prof p(SAGA_PROF_ADAM, stream);
        const double bc1 = 1.0 - pow((double)cfg->beta1, (double)cfg->adam_step);
        const double bc2 = 1.0 - pow((double)cfg->beta2, (double)cfg->adam_step);
        AdamCfg ac{cfg->update_mask, reg.c_vpose, (float)((double)cfg->lr / bc1), cfg->beta1, cfg->beta2, (float)sqrt(bc2), cfg->eps};
        const int n = B * SAGA_VAR_COUNT;
        adam_kernel<<<(n + 255) / 256, 256, 0, st>>>(vars, adam_m, adam_v, w.grad, w.g_z, ac, B, w.hdr, layout);
        SAGA_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int saga_fit_step(const saga_model_t* m, const saga_decoder_t* d, const saga_grasp_tables_t* tb,
                             const saga_fit_problem_t* prob, const saga_fit_step_cfg_t* cfg, float* vars, float* adam_m,
                             float* adam_v, float* ws, float* loss, float* loss_terms, float* markers_rec, int B, void* stream) {
    return fit_step_impl(m, nullptr, d, tb, prob, cfg, vars, adam_m, adam_v, ws, loss, loss_terms, markers_rec, B, stream);
}

extern "C" int saga_fit_step_sparse(const saga_model_t* m, const saga_model_t* m_loss, const saga_decoder_t* d,
                                    const saga_grasp_tables_t* tb, const saga_fit_problem_t* prob, const saga_fit_step_cfg_t* cfg,
                                    float* vars, float* adam_m, float* adam_v, float* ws, float* loss, float* loss_terms,
                                    float* markers_rec, int B, void* stream) {
    SAGA_CHECK_ARG(m_loss, "saga_fit_step_sparse: null subset model");
    return fit_step_impl(m, m_loss, d, tb, prob, cfg, vars, adam_m, adam_v, ws, loss, loss_terms, markers_rec, B, stream);
}
