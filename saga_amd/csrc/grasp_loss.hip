// Fused contact / penetration loss of GraspPose-Opt: everything WholeGraspPose/models/fittingop.py:152-212 does
// between the SMPL-X forward and the weighted sum -- marker / foot / hand gathers, right-hand vertex normals
// (Open3D on the host in the reference, :158-164), the two point2point_signed calls (utils/train_helper.py:17-86:
// bidirectional nearest neighbours, gathered difference vectors, norms, normal-dot signs) and the per-candidate
// means -- in ONE kernel per direction, one workgroup per candidate.
//
// Forward: object cloud tiles, the 778 hand vertices + 143 markers and the hand normals live in LDS; lanes own
// query points and scan the other cloud with broadcast ds_read_b128 (same exact-rounding distance and strict '<'
// as saga_chamfer_fwd, so the indices are identical to it).  FP32-VALU bound.
// Backward: analytic gradient of the eight loss terms w.r.t. the gathered vertices, accumulated per candidate in
// LDS (object points scatter-add into their nearest hand vertex -> LDS float atomics) and written as the dense
// grad_vertices [B,V,3] the SMPL-X backward consumes.
#include <math.h>

#include "saga_common.h"

void* saga_prof_begin(int id, void* stream);
void saga_prof_end(void* scope);

#define GL_THREADS 512
#define GL_TILE 2048            // object points staged per LDS tile
#define GL_MAX_HAND 1024
#define GL_MAX_MARK 256
#define GL_MAX_UNIQUE 2048
#define NT SAGA_GRASP_NTERMS

__device__ __forceinline__ float block_sum(float v, float* red) {
    // all GL_THREADS threads call; returns the total in every thread
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < GL_THREADS / 64; ++w) s += red[w];
    return s;
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(GL_THREADS) void grasp_loss_fwd_kernel(
    saga_grasp_tables_t tb, const float* __restrict__ verts, const int32_t* __restrict__ faces,
    const int32_t* __restrict__ vf_ptr, const int32_t* __restrict__ vf_idx, const float* __restrict__ obj,
    const float* __restrict__ obj_n, const float* __restrict__ mk_target, const float* __restrict__ c_mk,
    const float* __restrict__ c_obj, float* __restrict__ terms, float* __restrict__ markers_rec,
    float* __restrict__ hand_rec, float* __restrict__ hand_normals, float* __restrict__ h2o_marker,
    float* __restrict__ h2o_signed, float* __restrict__ o2h_signed, int32_t* __restrict__ idx_m2o,
    int32_t* __restrict__ idx_h2o, int32_t* __restrict__ idx_o2h, int V, int O) {
#pragma clang fp contract(off)
    __shared__ float4 sObj[GL_TILE];
    __shared__ float4 sQ[GL_MAX_HAND + GL_MAX_MARK];     // hand vertices first, then markers
    __shared__ float4 sNrm[GL_MAX_HAND];
    __shared__ float red[GL_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int H = tb.n_hand, M = tb.n_markers, NQ = H + M;
    const float* vb = verts + (size_t)b * V * 3;
    const float* ob = obj + (size_t)b * O * 3;
    const float* onb = obj_n + (size_t)b * O * 3;

    // ---- gather hand vertices + markers; hand normals (area-weighted, ascending face order) ----------------
    for (int i = tid; i < NQ; i += GL_THREADS) {
        const int v = i < H ? tb.hand_ids[i] : tb.marker_ids[i - H];
        const float x = vb[v * 3], y = vb[v * 3 + 1], z = vb[v * 3 + 2];
        sQ[i] = make_float4(x, y, z, 0.f);
        float* dst = i < H ? hand_rec + ((size_t)b * H + i) * 3 : markers_rec + ((size_t)b * M + (i - H)) * 3;
        dst[0] = x; dst[1] = y; dst[2] = z;
    }
    for (int i = tid; i < H; i += GL_THREADS) {
        const int v = tb.hand_ids[i];
        float nx = 0.f, ny = 0.f, nz = 0.f;
        for (int e = vf_ptr[v]; e < vf_ptr[v + 1]; ++e) {
            const int f = vf_idx[e];
            const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
            const float ax = vb[i1 * 3] - vb[i0 * 3], ay = vb[i1 * 3 + 1] - vb[i0 * 3 + 1], az = vb[i1 * 3 + 2] - vb[i0 * 3 + 2];
            const float bx = vb[i2 * 3] - vb[i0 * 3], by = vb[i2 * 3 + 1] - vb[i0 * 3 + 1], bz = vb[i2 * 3 + 2] - vb[i0 * 3 + 2];
            nx += ay * bz - az * by;
            ny += az * bx - ax * bz;
            nz += ax * by - ay * bx;
        }
        const float sq = nx * nx + ny * ny + nz * nz;
        if (sq > 0.f) { const float inv = 1.0f / sqrtf(sq); nx *= inv; ny *= inv; nz *= inv; }
        if (isnan(nx)) { nx = 0.f; ny = 0.f; nz = 1.f; }
        sNrm[i] = make_float4(nx, ny, nz, 0.f);
        float* dst = hand_normals + ((size_t)b * H + i) * 3;
        dst[0] = nx; dst[1] = ny; dst[2] = nz;
    }

    // ---- x -> y : every hand vertex / marker finds its nearest object point -------------------------------
    float qx[2], qy[2], qz[2], best[2];
    int bi[2];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = min(tid + i * GL_THREADS, NQ - 1);
        qx[i] = sQ[q].x; qy[i] = sQ[q].y; qz[i] = sQ[q].z;
        best[i] = INFINITY; bi[i] = 0;
    }
    for (int t0 = 0; t0 < O; t0 += GL_TILE) {
        const int cnt = min(GL_TILE, O - t0);
        __syncthreads();
        for (int k = tid; k < cnt; k += GL_THREADS) {
            const float* p = ob + (size_t)(t0 + k) * 3;
            sObj[k] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < cnt; ++k) {
            const float4 p = sObj[k];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float dx = qx[i] - p.x, dy = qy[i] - p.y, dz = qz[i] - p.z;
                const float d = (dx * dx + dy * dy) + dz * dz;
                const bool lt = d < best[i];
                best[i] = lt ? d : best[i];
                bi[i] = lt ? (t0 + k) : bi[i];
            }
        }
    }
    // per-query terms
    float t_rec = 0.f, t_brec = 0.f, t_mc = 0.f, t_prior = 0.f, t_hcol = 0.f;
    float n_hcol = 0.f, n_contact = 0.f, n_mcol = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + i * GL_THREADS;
        if (q >= NQ) continue;
        const int k = bi[i];
        const float dx = qx[i] - ob[k * 3], dy = qy[i] - ob[k * 3 + 1], dz = qz[i] - ob[k * 3 + 2];
        const float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
        if (q < H) {
            // h2o_signed: sign from the OBJECT normal at the neighbour (train_helper.py:76-79)
            const float dot = (onb[k * 3] * dx + onb[k * 3 + 1] * dy) + onb[k * 3 + 2] * dz;
            const float d = dist * sgn(dot);
            h2o_signed[(size_t)b * H + q] = d;
            idx_h2o[(size_t)b * H + q] = k;
            const float vw = tb.hand_weights[q];
            const bool near = (d < 0.02f) && (d > 0.f) && (powf(vw, 1.0f / 2.5f) > 0.7f);   // fittingop.py:195
            t_prior += fabsf(d) * (near ? vw : 0.f);                                         // :199,210
            t_hcol += fabsf(d) * (d < 0.f ? 10.f : 0.f);                                     // :200,211
            n_hcol += d < 0.f ? 1.f : 0.f;
            n_contact += (d < 0.001f && d > -0.001f) ? 1.f : 0.f;
        } else {
            const int p = q - H;
            // marker call (:183): y_normals is None -> unsigned distance
            h2o_marker[(size_t)b * M + p] = dist;
            idx_m2o[(size_t)b * M + p] = k;
            t_mc += dist * c_mk[(size_t)b * M + p];                                          // :190
            const float* tg = mk_target + ((size_t)b * M + p) * 3;
            const float e = (fabsf(qx[i] - tg[0]) + fabsf(qy[i] - tg[1])) + fabsf(qz[i] - tg[2]);
            t_rec += e;                                                                       // :168
            if (p < tb.n_body_markers) t_brec += e;                                           // :169
            n_mcol += dist < 0.f ? 1.f : 0.f;
        }
    }
    // foot term: 0.1 * mean |z| (:173)
    float t_foot = 0.f;
    for (int i = tid; i < tb.n_foot; i += GL_THREADS) t_foot += fabsf(vb[tb.foot_ids[i] * 3 + 2]);

    // ---- y -> x : every object point finds its nearest HAND vertex (the marker direction is never used) ------
    float t_oc = 0.f, t_ocol = 0.f, n_ocol = 0.f;
    for (int k0 = 0; k0 < O; k0 += GL_THREADS * 4) {
        float ox[4], oy[4], oz[4], ob_best[4];
        int oj[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = min(k0 + tid + i * GL_THREADS, O - 1);
            ox[i] = ob[k * 3]; oy[i] = ob[k * 3 + 1]; oz[i] = ob[k * 3 + 2];
            ob_best[i] = INFINITY; oj[i] = 0;
        }
#pragma unroll 2
        for (int j = 0; j < H; ++j) {
            const float4 p = sQ[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dx = ox[i] - p.x, dy = oy[i] - p.y, dz = oz[i] - p.z;
                const float d = (dx * dx + dy * dy) + dz * dz;
                const bool lt = d < ob_best[i];
                ob_best[i] = lt ? d : ob_best[i];
                oj[i] = lt ? j : oj[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + tid + i * GL_THREADS;
            if (k >= O) continue;
            const int j = oj[i];
            const float dx = ox[i] - sQ[j].x, dy = oy[i] - sQ[j].y, dz = oz[i] - sQ[j].z;
            const float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
            const float dot = (sNrm[j].x * dx + sNrm[j].y * dy) + sNrm[j].z * dz;           // HAND normal (:68-71)
            const float d = dist * sgn(dot);
            o2h_signed[(size_t)b * O + k] = d;
            idx_o2h[(size_t)b * O + k] = j;
            t_oc += fabsf(d) * c_obj[(size_t)b * O + k];                                      // :191
            t_ocol += fabsf(d) * (d < 0.f ? 20.f : 0.f);                                      // :206,212
            n_ocol += d < 0.f ? 1.f : 0.f;
        }
    }

    // ---- per-candidate means ---------------------------------------------------------------------------------
    const float s_rec = block_sum(t_rec, red), s_brec = block_sum(t_brec, red), s_foot = block_sum(t_foot, red);
    const float s_mc = block_sum(t_mc, red), s_oc = block_sum(t_oc, red), s_prior = block_sum(t_prior, red);
    const float s_hcol = block_sum(t_hcol, red), s_ocol = block_sum(t_ocol, red);
    const float c_hcol = block_sum(n_hcol, red), c_ocol = block_sum(n_ocol, red), c_cont = block_sum(n_contact, red);
    const float c_mcol = block_sum(n_mcol, red);
    if (tid == 0) {
        float* t = terms + (size_t)b * NT;
        t[0] = s_rec / (float)(M * 3);
        t[1] = s_brec / (float)(tb.n_body_markers * 3);
        t[2] = 0.1f * (s_foot / (float)tb.n_foot);
        t[3] = s_mc / (float)M;
        t[4] = s_oc / (float)O;
        t[5] = s_prior / (float)H;
        t[6] = s_hcol / (float)H;
        t[7] = s_ocol / (float)O;
        t[8] = c_hcol; t[9] = c_ocol; t[10] = c_cont; t[11] = c_mcol;
        t[12] = 0.f; t[13] = 0.f; t[14] = 0.f; t[15] = 0.f;
    }
}

// grad_vertices [B,V,3] (pre-zeroed) from the upstream gradients of the eight differentiable terms
__global__ __launch_bounds__(GL_THREADS) void grasp_loss_bwd_kernel(
    saga_grasp_tables_t tb, const float* __restrict__ g_terms, const float* __restrict__ verts,
    const float* __restrict__ obj, const float* __restrict__ mk_target, const float* __restrict__ c_mk,
    const float* __restrict__ c_obj, const float* __restrict__ markers_rec, const float* __restrict__ hand_rec,
    const float* __restrict__ h2o_signed, const float* __restrict__ o2h_signed, const int32_t* __restrict__ idx_m2o,
    const int32_t* __restrict__ idx_h2o, const int32_t* __restrict__ idx_o2h, float* __restrict__ g_verts, int V, int O) {
    __shared__ float acc[GL_MAX_UNIQUE * 3];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int H = tb.n_hand, M = tb.n_markers, U = tb.n_unique;
    const float* gt = g_terms + (size_t)b * NT;
    const float g_rec = gt[0], g_brec = gt[1], g_foot = gt[2], g_mc = gt[3], g_oc = gt[4], g_prior = gt[5], g_hcol = gt[6],
                g_ocol = gt[7];
    const float* vb = verts + (size_t)b * V * 3;
    const float* ob = obj + (size_t)b * O * 3;
    for (int i = tid; i < U * 3; i += GL_THREADS) acc[i] = 0.f;
    __syncthreads();
    // markers: L1 reconstruction + contact-weighted distance to the nearest object point
    for (int p = tid; p < M; p += GL_THREADS) {
        const float* m = markers_rec + ((size_t)b * M + p) * 3;
        const float* tg = mk_target + ((size_t)b * M + p) * 3;
        const int k = idx_m2o[(size_t)b * M + p];
        const float dx = m[0] - ob[k * 3], dy = m[1] - ob[k * 3 + 1], dz = m[2] - ob[k * 3 + 2];
        const float n = sqrtf(dx * dx + dy * dy + dz * dz);
        const float w = n > 0.f ? g_mc * c_mk[(size_t)b * M + p] / ((float)M * n) : 0.f;
        const float wr = g_rec / (float)(M * 3) + (p < tb.n_body_markers ? g_brec / (float)(tb.n_body_markers * 3) : 0.f);
        float* a = acc + tb.marker_slot[p] * 3;
        atomicAdd(a + 0, wr * sgn(m[0] - tg[0]) + w * dx);
        atomicAdd(a + 1, wr * sgn(m[1] - tg[1]) + w * dy);
        atomicAdd(a + 2, wr * sgn(m[2] - tg[2]) + w * dz);
    }
    // foot: 0.1 * mean |z|
    for (int f = tid; f < tb.n_foot; f += GL_THREADS) {
        const float z = vb[tb.foot_ids[f] * 3 + 2];
        atomicAdd(acc + tb.foot_slot[f] * 3 + 2, g_foot * 0.1f * sgn(z) / (float)tb.n_foot);
    }
    // hand -> object: |d_h| weighted by the prior-contact / collision masks (masks carry no gradient)
    for (int p = tid; p < H; p += GL_THREADS) {
        const float d = h2o_signed[(size_t)b * H + p];
        const float vw = tb.hand_weights[p];
        const bool near = (d < 0.02f) && (d > 0.f) && (powf(vw, 1.0f / 2.5f) > 0.7f);
        const float coef = (g_prior * (near ? vw : 0.f) + g_hcol * (d < 0.f ? 10.f : 0.f)) / (float)H;
        // d|d|/dh = sign(d)^2 * unit(h - y): zero where the sign is zero
        if (coef != 0.f && d != 0.f) {
            const float* h = hand_rec + ((size_t)b * H + p) * 3;
            const int k = idx_h2o[(size_t)b * H + p];
            const float dx = h[0] - ob[k * 3], dy = h[1] - ob[k * 3 + 1], dz = h[2] - ob[k * 3 + 2];
            const float w = coef / sqrtf(dx * dx + dy * dy + dz * dz);
            float* a = acc + tb.hand_slot[p] * 3;
            atomicAdd(a + 0, w * dx); atomicAdd(a + 1, w * dy); atomicAdd(a + 2, w * dz);
        }
    }
    // object -> hand: gradient lands on the nearest hand vertex (scatter-add; many object points share one vertex)
    for (int k = tid; k < O; k += GL_THREADS) {
        const float d = o2h_signed[(size_t)b * O + k];
        const float coef = (g_oc * c_obj[(size_t)b * O + k] + g_ocol * (d < 0.f ? 20.f : 0.f)) / (float)O;
        if (coef != 0.f && d != 0.f) {
            const int j = idx_o2h[(size_t)b * O + k];
            const float* h = hand_rec + ((size_t)b * H + j) * 3;
            const float dx = ob[k * 3] - h[0], dy = ob[k * 3 + 1] - h[1], dz = ob[k * 3 + 2] - h[2];
            const float w = -coef / sqrtf(dx * dx + dy * dy + dz * dz);
            float* a = acc + tb.hand_slot[j] * 3;
            atomicAdd(a + 0, w * dx); atomicAdd(a + 1, w * dy); atomicAdd(a + 2, w * dz);
        }
    }
    __syncthreads();
    float* gb = g_verts + (size_t)b * V * 3;
    for (int i = tid; i < U * 3; i += GL_THREADS) gb[tb.unique_ids[i / 3] * 3 + (i % 3)] = acc[i];
}

static int check_tables(const saga_grasp_tables_t* tb) {
    SAGA_CHECK_ARG(tb, "grasp loss: null tables");
    SAGA_CHECK_ARG(tb->n_hand > 0 && tb->n_hand <= GL_MAX_HAND && tb->n_markers > 0 && tb->n_markers <= GL_MAX_MARK &&
                       tb->n_hand + tb->n_markers <= 2 * GL_THREADS,
                   "grasp loss: n_hand=%d (max %d), n_markers=%d (max %d)", tb->n_hand, GL_MAX_HAND, tb->n_markers, GL_MAX_MARK);
    SAGA_CHECK_ARG(tb->n_foot > 0 && tb->n_unique > 0 && tb->n_unique <= GL_MAX_UNIQUE && tb->n_body_markers > 0 &&
                       tb->n_body_markers <= tb->n_markers,
                   "grasp loss: bad table sizes (foot %d, unique %d, body markers %d)", tb->n_foot, tb->n_unique, tb->n_body_markers);
    SAGA_CHECK_ARG(tb->marker_ids && tb->hand_ids && tb->foot_ids && tb->hand_weights && tb->unique_ids && tb->marker_slot &&
                       tb->hand_slot && tb->foot_slot,
                   "grasp loss: null table pointer");
    return 0;
}

extern "C" int saga_grasp_loss_fwd(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* verts, const float* obj,
                                   const float* obj_normals, const float* markers_target, const float* c_markers,
                                   const float* c_object, float* terms, float* markers_rec, float* hand_rec,
                                   float* hand_normals, float* h2o_marker, float* h2o_signed, float* o2h_signed,
                                   int32_t* idx_m2o, int32_t* idx_h2o, int32_t* idx_o2h, int B, int O, void* stream) {
    SAGA_CHECK_ARG(m && m->F > 0, "saga_grasp_loss_fwd: model without faces");
    int rc = check_tables(tb);
    if (rc) return rc;
    SAGA_CHECK_ARG(B >= 0 && O > 0, "saga_grasp_loss_fwd: bad sizes B=%d O=%d", B, O);
    if (B == 0) return 0;
    SAGA_CHECK_ARG(verts && obj && obj_normals && markers_target && c_markers && c_object && terms && markers_rec && hand_rec &&
                       hand_normals && h2o_marker && h2o_signed && o2h_signed && idx_m2o && idx_h2o && idx_o2h,
                   "saga_grasp_loss_fwd: null buffer");
    void* prof = saga_prof_begin(SAGA_PROF_GRASP_FWD, stream);
    grasp_loss_fwd_kernel<<<B, GL_THREADS, 0, (hipStream_t)stream>>>(*tb, verts, m->faces, m->vf_ptr, m->vf_idx, obj, obj_normals,
                                                                     markers_target, c_markers, c_object, terms, markers_rec,
                                                                     hand_rec, hand_normals, h2o_marker, h2o_signed, o2h_signed,
                                                                     idx_m2o, idx_h2o, idx_o2h, m->V, O);
    saga_prof_end(prof);
    SAGA_LAUNCH_CHECK();
    return 0;
}

extern "C" int saga_grasp_loss_bwd(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* g_terms, const float* verts,
                                   const float* obj, const float* markers_target, const float* c_markers, const float* c_object,
                                   const float* markers_rec, const float* hand_rec, const float* h2o_signed,
                                   const float* o2h_signed, const int32_t* idx_m2o, const int32_t* idx_h2o,
                                   const int32_t* idx_o2h, float* g_verts, int B, int O, void* stream) {
    SAGA_CHECK_ARG(m, "saga_grasp_loss_bwd: null model");
    int rc = check_tables(tb);
    if (rc) return rc;
    SAGA_CHECK_ARG(B >= 0 && O > 0, "saga_grasp_loss_bwd: bad sizes B=%d O=%d", B, O);
    if (B == 0) return 0;
    SAGA_CHECK_ARG(g_terms && verts && obj && markers_target && c_markers && c_object && markers_rec && hand_rec && h2o_signed &&
                       o2h_signed && idx_m2o && idx_h2o && idx_o2h && g_verts,
                   "saga_grasp_loss_bwd: null buffer");
    hipStream_t st = (hipStream_t)stream;
    void* prof = saga_prof_begin(SAGA_PROF_GRASP_BWD, stream);
    hipError_t e = hipMemsetAsync(g_verts, 0, (size_t)B * m->V * 3 * sizeof(float), st);
    if (e == hipSuccess)
        grasp_loss_bwd_kernel<<<B, GL_THREADS, 0, st>>>(*tb, g_terms, verts, obj, markers_target, c_markers, c_object, markers_rec,
                                                        hand_rec, h2o_signed, o2h_signed, idx_m2o, idx_h2o, idx_o2h, g_verts, m->V, O);
    saga_prof_end(prof);
    SAGA_HIP(e);
    SAGA_LAUNCH_CHECK();
    return 0;
}
