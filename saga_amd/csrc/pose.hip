// Pose path of the SMPL-X forward/backward: Rodrigues, folded joint regression, the 55-joint
// rigid-transform chain and its reverse.  One 64-lane wavefront per candidate: lane j owns joint j,
// transforms live in LDS, the tree is walked level by level (depth <= 10) and, in the backward pass,
// parents GATHER from their children in a fixed order (deterministic, no atomics).
// Replaces smplx.lbs.{batch_rodrigues, vertices2joints, batch_rigid_transform} (SURVEY.md section 3.2;
// reference call site WholeGraspPose/models/fittingop.py:149).
#include "saga_common.h"

#define WAVES_PER_BLOCK 4

// ---- Rodrigues: R = I + sin(t) K + (1-cos(t)) K^2, t = ||r + 1e-8||, K = skew(r / t) -----------------
__device__ __forceinline__ void rodrigues(const float r[3], float R[9]) {
    const float ex = r[0] + 1e-8f, ey = r[1] + 1e-8f, ez = r[2] + 1e-8f;
    const float t = sqrtf(ex * ex + ey * ey + ez * ez);
    const float inv = 1.0f / t;
    const float x = r[0] * inv, y = r[1] * inv, z = r[2] * inv;
    float s, c;
    sincosf(t, &s, &c);
    const float oc = 1.0f - c;
    // K^2 = n n^T - (n.n) I  (n.n is 1 only up to the 1e-8 offset, keep it explicit)
    const float xx = x * x, yy = y * y, zz = z * z;
    R[0] = 1.0f + oc * (-(yy + zz));
    R[1] = -s * z + oc * (x * y);
    R[2] = s * y + oc * (x * z);
    R[3] = s * z + oc * (x * y);
    R[4] = 1.0f + oc * (-(xx + zz));
    R[5] = -s * x + oc * (y * z);
    R[6] = -s * y + oc * (x * z);
    R[7] = s * x + oc * (y * z);
    R[8] = 1.0f + oc * (-(xx + yy));
}

// gR [9] -> gr [3]
__device__ __forceinline__ void rodrigues_bwd(const float r[3], const float g[9], float gr[3]) {
    const float e[3] = {r[0] + 1e-8f, r[1] + 1e-8f, r[2] + 1e-8f};
    const float t = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    const float inv = 1.0f / t;
    const float n[3] = {r[0] * inv, r[1] * inv, r[2] * inv};
    float s, c;
    sincosf(t, &s, &c);
    const float oc = 1.0f - c;
    const float x = n[0], y = n[1], z = n[2];
    // <g, K> and <g, K^2>
    const float gK = g[1] * (-z) + g[2] * y + g[3] * z + g[5] * (-x) + g[6] * (-y) + g[7] * x;
    const float gK2 = g[0] * (-(y * y + z * z)) + g[4] * (-(x * x + z * z)) + g[8] * (-(x * x + y * y)) +
                      (g[1] + g[3]) * x * y + (g[2] + g[6]) * x * z + (g[5] + g[7]) * y * z;
    const float g_t = c * gK + s * gK2;        // dL/dtheta through sin / (1-cos)
    // dL/dn: from sin*K and (1-cos)*K^2
    float gn[3];
    gn[0] = s * (g[7] - g[5]) + oc * (-2.f * x * (g[4] + g[8]) + (g[1] + g[3]) * y + (g[2] + g[6]) * z);
    gn[1] = s * (g[2] - g[6]) + oc * (-2.f * y * (g[0] + g[8]) + (g[1] + g[3]) * x + (g[5] + g[7]) * z);
    gn[2] = s * (g[3] - g[1]) + oc * (-2.f * z * (g[0] + g[4]) + (g[2] + g[6]) * x + (g[5] + g[7]) * y);
    // n = r / t ; t = ||r + eps||  ->  dr = gn / t + (g_t - <gn, r>/t^2) * (r+eps)/t
    const float gn_r = gn[0] * r[0] + gn[1] * r[1] + gn[2] * r[2];
    const float coef = (g_t - gn_r * inv * inv) * inv;
#pragma unroll
    for (int i = 0; i < 3; ++i) gr[i] = gn[i] * inv + coef * e[i];
}

__global__ void rodrigues_fwd_kernel(const float* __restrict__ rv, float* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r[3] = {rv[i * 3], rv[i * 3 + 1], rv[i * 3 + 2]}, R[9];
    rodrigues(r, R);
#pragma unroll
    for (int k = 0; k < 9; ++k) out[(size_t)i * 9 + k] = R[k];
}

__global__ void rodrigues_bwd_kernel(const float* __restrict__ rv, const float* __restrict__ g, float* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r[3] = {rv[i * 3], rv[i * 3 + 1], rv[i * 3 + 2]}, gR[9], gr[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) gR[k] = g[(size_t)i * 9 + k];
    rodrigues_bwd(r, gR, gr);
    out[i * 3] = gr[0]; out[i * 3 + 1] = gr[1]; out[i * 3 + 2] = gr[2];
}

extern "C" int saga_rodrigues_fwd(const float* rot_vecs, float* rot_mats, int N, void* stream) {
    SAGA_CHECK_ARG(N >= 0 && (N == 0 || (rot_vecs && rot_mats)), "saga_rodrigues_fwd: bad arguments");
    if (N == 0) return 0;
    rodrigues_fwd_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(rot_vecs, rot_mats, N);
    SAGA_LAUNCH_CHECK();
    return 0;
}

extern "C" int saga_rodrigues_bwd(const float* rot_vecs, const float* g_rot_mats, float* g_rot_vecs, int N, void* stream) {
    SAGA_CHECK_ARG(N >= 0 && (N == 0 || (rot_vecs && g_rot_mats && g_rot_vecs)), "saga_rodrigues_bwd: bad arguments");
    if (N == 0) return 0;
    rodrigues_bwd_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(rot_vecs, g_rot_mats, g_rot_vecs, N);
    SAGA_LAUNCH_CHECK();
    return 0;
}

// ---- forward: feat, R, J, chain G, skinning transforms A -------------------------------------------
// LDS per wave: G[55][12]
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void pose_fwd_kernel(
    const float* __restrict__ shape, const float* __restrict__ full_pose, const float* __restrict__ J_template,
    const float* __restrict__ J_dirs, const int32_t* __restrict__ parents, const int32_t* __restrict__ depth,
    float* __restrict__ feat, float* __restrict__ rot, float* __restrict__ jrest, float* __restrict__ chain,
    float* __restrict__ Aout, int B, int S, int max_depth, int* __restrict__ zero8) {
    __shared__ float sG[WAVES_PER_BLOCK][NJ * 12];
    if (zero8 && blockIdx.x == 0 && threadIdx.x < 8) zero8[threadIdx.x] = 0;      // the tile queues of the blend contraction that follows
    __shared__ float sShape[WAVES_PER_BLOCK][32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * WAVES_PER_BLOCK + wave;
    const bool live = b < B;             // whole wave shares b
    const int j = lane;
    const bool own = live && j < NJ;
    float* G = sG[wave];

    if (live && lane < 32) sShape[wave][lane] = lane < S ? shape[(size_t)b * S + lane] : 0.f;
    __syncthreads();

    float R[9], J[3] = {0, 0, 0}, Jp[3] = {0, 0, 0};
    int par = -1, dep = 0;
    if (own) {
        float r[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) r[i] = full_pose[(size_t)b * (NJ * 3) + j * 3 + i];
        rodrigues(r, R);
        par = parents[j];
        dep = depth[j];
        // J_j = J_template + shape . J_dirs   (folded J_regressor . (v_template + shapedirs.shape))
        const int pj = par < 0 ? j : par;
#pragma unroll
        for (int c = 0; c < 3; ++c) { J[c] = J_template[j * 3 + c]; Jp[c] = J_template[pj * 3 + c]; }
        for (int s = 0; s < S; ++s) {
            const float bs = sShape[wave][s];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                J[c] = fmaf(bs, J_dirs[s * (NJ * 3) + j * 3 + c], J[c]);
                Jp[c] = fmaf(bs, J_dirs[s * (NJ * 3) + pj * 3 + c], Jp[c]);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) rot[(size_t)b * (NJ * 9) + j * 9 + k] = R[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) jrest[(size_t)b * (NJ * 3) + j * 3 + c] = J[c];
        // pose feature: vec(R_j - I) for j >= 1, after the S shape coefficients
        if (j >= 1) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                feat[(size_t)b * FEAT_K + S + (j - 1) * 9 + k] = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.0f : 0.0f);
        }
    }
    if (live) {
        // shape coefficients and zero tail of the K=512 row
        for (int k = lane; k < S; k += 64) feat[(size_t)b * FEAT_K + k] = sShape[wave][k];
        for (int k = S + NPF + lane; k < FEAT_K; k += 64) feat[(size_t)b * FEAT_K + k] = 0.f;
    }

    // chain, level by level
    float Gj[12];
    for (int d = 0; d <= max_depth; ++d) {
        if (own && dep == d) {
            if (par < 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    Gj[r * 4 + 0] = R[r * 3 + 0]; Gj[r * 4 + 1] = R[r * 3 + 1]; Gj[r * 4 + 2] = R[r * 3 + 2];
                    Gj[r * 4 + 3] = J[r];
                }
            } else {
                const float rel[3] = {J[0] - Jp[0], J[1] - Jp[1], J[2] - Jp[2]};
                const float* P = G + par * 12;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float p0 = P[r * 4 + 0], p1 = P[r * 4 + 1], p2 = P[r * 4 + 2], p3 = P[r * 4 + 3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) Gj[r * 4 + c] = p0 * R[c] + p1 * R[3 + c] + p2 * R[6 + c];
                    Gj[r * 4 + 3] = p0 * rel[0] + p1 * rel[1] + p2 * rel[2] + p3;
                }
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) G[j * 12 + k] = Gj[k];
        }
        __syncthreads();
    }
    if (own) {
#pragma unroll
        for (int k = 0; k < 12; ++k) chain[(size_t)b * (NJ * 12) + j * 12 + k] = Gj[k];
        // A_j = [Rg | t - Rg J]
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float t = Gj[r * 4 + 3] - (Gj[r * 4 + 0] * J[0] + Gj[r * 4 + 1] * J[1] + Gj[r * 4 + 2] * J[2]);
            float* a = Aout + (size_t)b * (NJ * 12) + j * 12 + r * 4;
            a[0] = Gj[r * 4 + 0]; a[1] = Gj[r * 4 + 1]; a[2] = Gj[r * 4 + 2]; a[3] = t;
        }
    }
}

// ---- backward ----------------------------------------------------------------------------------------
// inputs : dA [B,56,12] (row 55 col 3/7/11 = sum_v g -> d transl), g_pj [B,55,3] (grad of posed joints, may be NULL),
//          slabs [SPLIT,B,512] partial grads of feat, saved R/J/G.
// outputs: g_shape [B,S], g_full_pose [B,165], g_transl [B,3]
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void pose_bwd_kernel(
    const float* __restrict__ dA, const float* __restrict__ g_pj, const float* __restrict__ g_jsum,
    const float* __restrict__ slabs, int n_slabs,
    int small_stride, const float* __restrict__ rot, const float* __restrict__ jrest,
    const float* __restrict__ chain, const float* __restrict__ J_dirs, const int32_t* __restrict__ parents,
    const int32_t* __restrict__ depth, const int32_t* __restrict__ child_ptr, const int32_t* __restrict__ child_idx,
    const float* __restrict__ full_pose, float* __restrict__ g_shape, float* __restrict__ g_full_pose,
    float* __restrict__ g_transl, int B, int S, int max_depth, int has_verts) {
    const int dA_stride = small_stride;
    // per wave: dRg[55][9], dt[55][3], R[55][9], rel[55][3], dJ[55][3]
    __shared__ float s_dRg[WAVES_PER_BLOCK][NJ * 9];
    __shared__ float s_dt[WAVES_PER_BLOCK][NJ * 3];
    __shared__ float s_R[WAVES_PER_BLOCK][NJ * 9];
    __shared__ float s_rel[WAVES_PER_BLOCK][NJ * 3];
    __shared__ float s_dJ[WAVES_PER_BLOCK][NJ * 3];
    __shared__ float s_drel[WAVES_PER_BLOCK][NJ * 3];
    __shared__ __attribute__((aligned(16))) float s_gfeat[WAVES_PER_BLOCK][FEAT_K];   // split-K slabs summed (slab order)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * WAVES_PER_BLOCK + wave;
    const bool live = b < B;
    if (live && has_verts) {
        // grad of feat = sum of the split-K slabs: the wave reads each slab row as 2 KB of coalesced float4s
        // (lane owns columns 4*lane..+3 and 256+4*lane..+3); every lane then picks its 9 columns from LDS
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        for (int s = 0; s < n_slabs; ++s) {
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            const f32x4_t* row = reinterpret_cast<const f32x4_t*>(slabs + ((size_t)s * B + b) * FEAT_K);
            const f32x4_t u = __builtin_nontemporal_load(row + lane), v = __builtin_nontemporal_load(row + 64 + lane);   // read once
            a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
            a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
        }
        reinterpret_cast<float4*>(s_gfeat[wave])[lane] = a0;
        reinterpret_cast<float4*>(s_gfeat[wave])[64 + lane] = a1;
    }
    const int j = lane;
    const bool own = live && j < NJ;

    float R[9], J[3], Gp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Gj[12];
    float dRg[9], dt[3], dJ[3] = {0, 0, 0};
    int par = -1, dep = 0;
    if (own) {
        par = parents[j];
        dep = depth[j];
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = rot[(size_t)b * (NJ * 9) + j * 9 + k];
#pragma unroll
        for (int c = 0; c < 3; ++c) J[c] = jrest[(size_t)b * (NJ * 3) + j * 3 + c];
#pragma unroll
        for (int k = 0; k < 12; ++k) Gj[k] = chain[(size_t)b * (NJ * 12) + j * 12 + k];
        float rel[3] = {J[0], J[1], J[2]};
        if (par >= 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Gp[r * 3 + c] = chain[(size_t)b * (NJ * 12) + par * 12 + r * 4 + c];
#pragma unroll
            for (int c = 0; c < 3; ++c) rel[c] -= jrest[(size_t)b * (NJ * 3) + par * 3 + c];
        }
        // own terms from A_j = [Rg_j | t_j - Rg_j J_j] and posed_j = t_j
        float dAr[12];
        if (has_verts) {
#pragma unroll
            for (int k = 0; k < 12; ++k) dAr[k] = dA[(size_t)b * dA_stride + j * 12 + k];
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) dAr[k] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float at = dAr[r * 4 + 3];
            dt[r] = at + (g_pj ? g_pj[(size_t)b * small_stride + j * 3 + r] : 0.f);
#pragma unroll
            for (int c = 0; c < 3; ++c) dRg[r * 3 + c] = dAr[r * 4 + c] - at * J[c];
        }
        // dJ_j += -Rg_j^T dAt
#pragma unroll
        for (int c = 0; c < 3; ++c)
            dJ[c] = -(Gj[0 * 4 + c] * dAr[3] + Gj[1 * 4 + c] * dAr[7] + Gj[2 * 4 + c] * dAr[11]);
#pragma unroll
        for (int k = 0; k < 9; ++k) s_R[wave][j * 9 + k] = R[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) s_rel[wave][j * 3 + c] = rel[c];
    }
    __syncthreads();

    // reverse levels: a joint at depth d first gathers from its children (depth d+1, already final)
    float dR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, drel[3] = {0, 0, 0};
    for (int d = max_depth; d >= 0; --d) {
        if (own && dep == d) {
            const int c0 = child_ptr[j], c1 = child_ptr[j + 1];
            for (int ci = c0; ci < c1; ++ci) {
                const int ch = child_idx[ci];
                const float* cR = s_R[wave] + ch * 9;
                const float* cdRg = s_dRg[wave] + ch * 9;
                const float* cdt = s_dt[wave] + ch * 3;
                const float* crel = s_rel[wave] + ch * 3;
                // Rg_c = Rg_j R_c -> dRg_j += dRg_c R_c^T ; t_c = Rg_j rel_c + t_j -> dRg_j += dt_c (x) rel_c ; dt_j += dt_c
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        dRg[r * 3 + c] += cdRg[r * 3 + 0] * cR[c * 3 + 0] + cdRg[r * 3 + 1] * cR[c * 3 + 1] +
                                          cdRg[r * 3 + 2] * cR[c * 3 + 2] + cdt[r] * crel[c];
                    dt[r] += cdt[r];
                }
                // rel_c = J_c - J_j -> dJ_j -= drel_c
#pragma unroll
                for (int c = 0; c < 3; ++c) dJ[c] -= s_drel[wave][ch * 3 + c];
            }
            // own local grads: dR_j = Rg_p^T dRg_j ; drel_j = Rg_p^T dt_j  (Rg_p = I for the root)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    dR[r * 3 + c] = Gp[0 * 3 + r] * dRg[0 * 3 + c] + Gp[1 * 3 + r] * dRg[1 * 3 + c] + Gp[2 * 3 + r] * dRg[2 * 3 + c];
                drel[r] = Gp[0 * 3 + r] * dt[0] + Gp[1 * 3 + r] * dt[1] + Gp[2 * 3 + r] * dt[2];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) dJ[c] += drel[c];
#pragma unroll
            for (int k = 0; k < 9; ++k) s_dRg[wave][j * 9 + k] = dRg[k];
#pragma unroll
            for (int c = 0; c < 3; ++c) { s_dt[wave][j * 3 + c] = dt[c]; s_drel[wave][j * 3 + c] = drel[c]; }
        }
        __syncthreads();
    }

    if (own) {
        // add the direct pose-feature gradient (pose_feat = vec(R_j - I), j >= 1), summed over the split-K slabs
        if (j >= 1 && has_verts) {
#pragma unroll
            for (int k = 0; k < 9; ++k) dR[k] += s_gfeat[wave][S + (j - 1) * 9 + k];
        }
        float r[3], gr[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) r[i] = full_pose[(size_t)b * (NJ * 3) + j * 3 + i];
        rodrigues_bwd(r, dR, gr);
#pragma unroll
        for (int i = 0; i < 3; ++i) g_full_pose[(size_t)b * (NJ * 3) + j * 3 + i] = gr[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) s_dJ[wave][j * 3 + c] = dJ[c];
    }
    __syncthreads();
    if (live) {
        // g_shape[s] = slab-sum(feat grad)[s] + sum_{j,c} dJ[j,c] * J_dirs[s, j*3+c]
        if (lane < S) {
            float acc = has_verts ? s_gfeat[wave][lane] : 0.f;
            for (int q = 0; q < NJ * 3; ++q) acc = fmaf(s_dJ[wave][q], J_dirs[lane * (NJ * 3) + q], acc);
            g_shape[(size_t)b * S + lane] = acc;
        }
        if (lane < 3) {
            float acc = has_verts ? dA[(size_t)b * dA_stride + NJ * 12 + lane * 4 + 3] : 0.f;
            if (g_jsum) acc += g_jsum[(size_t)b * small_stride + lane];
            g_transl[(size_t)b * 3 + lane] = acc;
        }
    }
}

int saga_launch_pose_fwd(const saga_model* m, const float* shape, const float* full_pose, float* feat, float* rot,
                         float* jrest, float* chain, float* A, int B, hipStream_t st, int* zero8) {
    dim3 grid((B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(64 * WAVES_PER_BLOCK);
    pose_fwd_kernel<<<grid, block, 0, st>>>(shape, full_pose, m->J_template, m->J_dirs, m->parents, m->depth, feat, rot,
                                            jrest, chain, A, B, m->S, m->max_depth, zero8);
    SAGA_LAUNCH_CHECK();
    return 0;
}

int saga_launch_pose_bwd(const saga_model* m, const float* dA, const float* g_pj, const float* g_jsum, const float* slabs,
                         int n_slabs, const float* rot, const float* jrest, const float* chain, const float* full_pose,
                         float* g_shape, float* g_full_pose, float* g_transl, int B, int has_verts, hipStream_t st) {
    dim3 grid((B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(64 * WAVES_PER_BLOCK);
    pose_bwd_kernel<<<grid, block, 0, st>>>(dA, g_pj, g_jsum, slabs, n_slabs, SAGA_DA_STRIDE, rot, jrest, chain, m->J_dirs,
                                            m->parents, m->depth, m->child_ptr, m->child_idx, full_pose, g_shape,
                                            g_full_pose, g_transl, B, m->S, m->max_depth, has_verts);
    SAGA_LAUNCH_CHECK();
    return 0;
}
