// Linear-blend skinning forward/backward, extra joints + landmarks, and the dA reduction.
// Replaces the skinning tail of smplx.lbs.lbs and VertexJointSelector / vertices2landmarks
// (reference call site WholeGraspPose/models/fittingop.py:149).
//
// HBM-bound kernels: per candidate the 55 skinning transforms (55x12 floats = 2.6 KB) are staged in LDS,
// LBS weights come from a compressed (joint, weight) table (model constant, L2-resident), v_posed is read
// and vertices are written exactly once.  The backward reduction dA[b,j] = sum_v W[v,j] g[b,v] (x) [v_posed;1]
// is shaped as a GEMM over the vertex index and runs on v_mfma_f32_16x16x4_f32 (exact f32, deterministic).
#include "saga_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SKIN_THREADS 256

// ---- forward -------------------------------------------------------------------------------------------
// grid (ceil(V / (256*SKIN_VPT)), B). verts[b,v] = (sum_k w_k A[b, j_k]) . [v_posed[b,v]; 1] + transl[b]
// One block = one candidate x 1024 vertices: A[b] (2.6 KB) is staged in LDS once, every lane owns SKIN_VPT vertices
// (tile i: vertex base + i*256 + tid, so the 12-B per-lane accesses of a wave are one contiguous 768-B run) and issues
// all of its loads before the barrier.  KW4: the compressed weight table has exactly 4 entries per vertex and is read
// as one int4 + one float4.
#define SKIN_VPT 4
template <bool KW4>
__global__ __launch_bounds__(SKIN_THREADS) void skin_fwd_kernel(const float* __restrict__ A, const float* __restrict__ v_posed,
                                                                const int32_t* __restrict__ w_idx,
                                                                const float* __restrict__ w_val,
                                                                const float* __restrict__ transl, float* __restrict__ verts,
                                                                int V, int Npad, int KW) {
    __shared__ __attribute__((aligned(16))) float sA[NJ * 12];
    const int b = blockIdx.y;
    const int vbase = blockIdx.x * (SKIN_THREADS * SKIN_VPT);
    const int tid = threadIdx.x;
    for (int i = tid; i < NJ * 12; i += SKIN_THREADS) sA[i] = A[(size_t)b * (NJ * 12) + i];
    const float* src = v_posed + (size_t)b * Npad;
    float px[SKIN_VPT], py[SKIN_VPT], pz[SKIN_VPT];
    int4 ji[SKIN_VPT];
    float4 wv[SKIN_VPT];
#pragma unroll
    for (int i = 0; i < SKIN_VPT; ++i) {
        const int v = min(vbase + i * SKIN_THREADS + tid, V - 1);
        px[i] = src[v * 3]; py[i] = src[v * 3 + 1]; pz[i] = src[v * 3 + 2];
        if (KW4) {
            ji[i] = reinterpret_cast<const int4*>(w_idx)[v];
            wv[i] = reinterpret_cast<const float4*>(w_val)[v];
        }
    }
    const float tx = transl ? transl[b * 3] : 0.f, ty = transl ? transl[b * 3 + 1] : 0.f, tz = transl ? transl[b * 3 + 2] : 0.f;
    __syncthreads();
    float* dst = verts + (size_t)b * V * 3;
#pragma unroll
    for (int i = 0; i < SKIN_VPT; ++i) {
        const int v = vbase + i * SKIN_THREADS + tid;
        if (v >= V) continue;
        float T[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = 0.f;
        if (KW4) {
            const int jj[4] = {ji[i].x, ji[i].y, ji[i].z, ji[i].w};
            const float ww[4] = {wv[i].x, wv[i].y, wv[i].z, wv[i].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4* a = reinterpret_cast<const float4*>(sA + jj[k] * 12);
                const float4 a0 = a[0], a1 = a[1], a2 = a[2];
                const float w = ww[k];
                T[0] = fmaf(w, a0.x, T[0]); T[1] = fmaf(w, a0.y, T[1]); T[2] = fmaf(w, a0.z, T[2]); T[3] = fmaf(w, a0.w, T[3]);
                T[4] = fmaf(w, a1.x, T[4]); T[5] = fmaf(w, a1.y, T[5]); T[6] = fmaf(w, a1.z, T[6]); T[7] = fmaf(w, a1.w, T[7]);
                T[8] = fmaf(w, a2.x, T[8]); T[9] = fmaf(w, a2.y, T[9]); T[10] = fmaf(w, a2.z, T[10]); T[11] = fmaf(w, a2.w, T[11]);
            }
        } else {
            for (int k = 0; k < KW; ++k) {
                const float w = w_val[(size_t)v * KW + k];
                if (w == 0.f) continue;
                const float* a = sA + w_idx[(size_t)v * KW + k] * 12;
#pragma unroll
                for (int q = 0; q < 12; ++q) T[q] = fmaf(w, a[q], T[q]);
            }
        }
        dst[v * 3] = (T[0] * px[i] + T[1] * py[i] + T[2] * pz[i] + T[3]) + tx;
        dst[v * 3 + 1] = (T[4] * px[i] + T[5] * py[i] + T[6] * pz[i] + T[7]) + ty;
        dst[v * 3 + 2] = (T[8] * px[i] + T[9] * py[i] + T[10] * pz[i] + T[11]) + tz;
    }
}

// joints[b, 0:55] = posed joints (chain translation) ; [55:55+E] = picked vertices ; [55+E:] = landmarks ; + transl
__global__ void joints_fwd_kernel(const float* __restrict__ chain, const float* __restrict__ verts,
                                  const int32_t* __restrict__ extra_verts, const int32_t* __restrict__ lmk_verts,
                                  const float* __restrict__ lmk_bary, const float* __restrict__ transl,
                                  float* __restrict__ joints, int B, int V, int E, int L) {
    const int NJO = NJ + E + L;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * NJO) return;
    const int b = i / NJO, j = i % NJO;
    const float* vb = verts + (size_t)b * V * 3;
    float o[3];
    if (j < NJ) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = chain[(size_t)b * (NJ * 12) + j * 12 + c * 4 + 3] + (transl ? transl[b * 3 + c] : 0.f);
    } else if (j < NJ + E) {
        const int v = extra_verts[j - NJ];
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = vb[v * 3 + c];     // vertices already carry transl
    } else {
        const int l = j - NJ - E;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // smplx sums bary * (vertex) BEFORE adding transl; bary rows sum to 1 only approximately, so
            // remove transl from the corners and add it once, as the reference order does
            const float t = transl ? transl[b * 3 + c] : 0.f;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc += (vb[lmk_verts[l * 3 + k] * 3 + c] - t) * lmk_bary[l * 3 + k];
            o[c] = acc + t;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) joints[(size_t)i * 3 + c] = o[c];
}

// ---- backward of the joints output -----------------------------------------------------------------------
// g_joints [B,NJO,3] -> g_pj [B,55,3] (compact), g_jsum [B,3] (direct d transl of the joints output), and the
// vertex-picked / landmark gradients scatter-added into gv [B,V,3] (gv already holds g_vertices or zeros).
// One block of 64 threads per candidate; landmarks share vertices, so the adds use (agent-scope) float atomics.
__global__ void joints_bwd_kernel(const float* __restrict__ g_joints, const int32_t* __restrict__ extra_verts,
                                  const int32_t* __restrict__ lmk_verts, const float* __restrict__ lmk_bary,
                                  float* __restrict__ g_pj, float* __restrict__ g_jsum, float* __restrict__ gv, int V, int E,
                                  int L, int small_stride) {
    const int NJO = NJ + E + L;
    const int b = blockIdx.x, t = threadIdx.x;
    const float* g = g_joints + (size_t)b * NJO * 3;
    float s[3] = {0, 0, 0};
    // d transl: the 55 posed joints get `+ transl` directly; picked vertices inherit it through the vertices
    // (their gradient reaches transl via sum_v gv); a landmark is sum_k bary_k (v_k - t) + t, so besides the
    // vertex path it contributes g * (1 - sum_k bary_k) directly.
    for (int j = t; j < NJO; j += 64) {
        const float gx = g[j * 3], gy = g[j * 3 + 1], gz = g[j * 3 + 2];
        float direct = j < NJ ? 1.f : 0.f;
        if (j >= NJ + E) {
            const int l = j - NJ - E;
            direct = 1.f - ((lmk_bary[l * 3] + lmk_bary[l * 3 + 1]) + lmk_bary[l * 3 + 2]);
        }
        s[0] += gx * direct; s[1] += gy * direct; s[2] += gz * direct;
        if (j < NJ) {
            g_pj[(size_t)b * small_stride + j * 3] = gx;
            g_pj[(size_t)b * small_stride + j * 3 + 1] = gy;
            g_pj[(size_t)b * small_stride + j * 3 + 2] = gz;
        } else if (j < NJ + E) {
            float* d = gv + ((size_t)b * V + extra_verts[j - NJ]) * 3;
            atomicAdd(d, gx); atomicAdd(d + 1, gy); atomicAdd(d + 2, gz);
        } else {
            const int l = j - NJ - E;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float* d = gv + ((size_t)b * V + lmk_verts[l * 3 + k]) * 3;
                const float w = lmk_bary[l * 3 + k];
                atomicAdd(d, gx * w); atomicAdd(d + 1, gy * w); atomicAdd(d + 2, gz * w);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float x = s[c];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
        if (t == 0) g_jsum[(size_t)b * small_stride + c] = x;
    }
}

// ---- backward of skinning, part 1: g_v_posed = T^T g  (elementwise, HBM-bound) -------------------------------
// Same decomposition as the forward; grid x covers Npad/3 (+1) vertices so the pad columns of g_v_posed (read by the
// split-K GEMM) are zero-filled by the lanes past V.
template <bool KW4>
__global__ __launch_bounds__(SKIN_THREADS) void skin_bwd_vposed_kernel(const float* __restrict__ A, const float* __restrict__ g_verts,
                                                                       const int32_t* __restrict__ w_idx,
                                                                       const float* __restrict__ w_val,
                                                                       float* __restrict__ g_vposed, int V, int Npad, int KW) {
    __shared__ __attribute__((aligned(16))) float sA[NJ * 12];
    const int b = blockIdx.y;
    const int vbase = blockIdx.x * (SKIN_THREADS * SKIN_VPT);
    const int tid = threadIdx.x;
    for (int i = tid; i < NJ * 12; i += SKIN_THREADS) sA[i] = A[(size_t)b * (NJ * 12) + i];
    const float* src = g_verts + (size_t)b * V * 3;
    float gx[SKIN_VPT], gy[SKIN_VPT], gz[SKIN_VPT];
    int4 ji[SKIN_VPT];
    float4 wv[SKIN_VPT];
#pragma unroll
    for (int i = 0; i < SKIN_VPT; ++i) {
        const int v = min(vbase + i * SKIN_THREADS + tid, V - 1);
        gx[i] = src[v * 3]; gy[i] = src[v * 3 + 1]; gz[i] = src[v * 3 + 2];
        if (KW4) {
            ji[i] = reinterpret_cast<const int4*>(w_idx)[v];
            wv[i] = reinterpret_cast<const float4*>(w_val)[v];
        }
    }
    __syncthreads();
    float* dst = g_vposed + (size_t)b * Npad;
#pragma unroll
    for (int i = 0; i < SKIN_VPT; ++i) {
        const int v = vbase + i * SKIN_THREADS + tid;
        if (v >= V) {
            // pad columns [3V, Npad)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (v * 3 + c < Npad) dst[v * 3 + c] = 0.f;
            continue;
        }
        float T[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) T[k] = 0.f;
        if (KW4) {
            const int jj[4] = {ji[i].x, ji[i].y, ji[i].z, ji[i].w};
            const float ww[4] = {wv[i].x, wv[i].y, wv[i].z, wv[i].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4* a = reinterpret_cast<const float4*>(sA + jj[k] * 12);
                const float4 a0 = a[0], a1 = a[1], a2 = a[2];
                const float w = ww[k];
                T[0] = fmaf(w, a0.x, T[0]); T[1] = fmaf(w, a0.y, T[1]); T[2] = fmaf(w, a0.z, T[2]);
                T[3] = fmaf(w, a1.x, T[3]); T[4] = fmaf(w, a1.y, T[4]); T[5] = fmaf(w, a1.z, T[5]);
                T[6] = fmaf(w, a2.x, T[6]); T[7] = fmaf(w, a2.y, T[7]); T[8] = fmaf(w, a2.z, T[8]);
            }
        } else {
            for (int k = 0; k < KW; ++k) {
                const float w = w_val[(size_t)v * KW + k];
                if (w == 0.f) continue;
                const float* a = sA + w_idx[(size_t)v * KW + k] * 12;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) T[r * 3 + c] = fmaf(w, a[r * 4 + c], T[r * 3 + c]);
            }
        }
        dst[v * 3] = T[0] * gx[i] + T[3] * gy[i] + T[6] * gz[i];
        dst[v * 3 + 1] = T[1] * gx[i] + T[4] * gy[i] + T[7] * gz[i];
        dst[v * 3 + 2] = T[2] * gx[i] + T[5] * gy[i] + T[8] * gz[i];
    }
}

// ---- backward of skinning, part 2: dA[b, j, (r,c)] = sum_v Wd[v][j] * g[b,v,r] * [v_posed[b,v];1][c] ---------------
// GEMM view: rows = (candidate, rc) pairs, 4 candidates x 12 = 48 = three 16-row MFMA tiles; columns = 64 padded
// joints (column 55 of Wd is all ones -> sum_v g = d transl); contraction over vertices, 4 per v_mfma_f32_16x16x4_f32.
// Block = 4 waves, each wave reduces a quarter of the vertex range; partials are combined through LDS in wave order.
#define DA_CANDS 4
#define DA_CHUNK 64     // vertices staged per wave per step

__global__ __launch_bounds__(256) void skin_bwd_dA_kernel(const float* __restrict__ g_verts, const float* __restrict__ v_posed,
                                                          const float* __restrict__ Wd, float* __restrict__ dA, int B, int V,
                                                          int Npad, int dA_stride) {
    // one 48 KB LDS block, used first as per-wave staging and then (after a barrier) as the reduction buffer
    __shared__ __attribute__((aligned(16))) float smem[4 * 3 * 4 * 64 * 4];
    // per wave staging: [wave][cand][vertex][8]: g(3), vp(3), 1, pad
    float (*sX)[DA_CANDS][DA_CHUNK][8] = reinterpret_cast<float (*)[DA_CANDS][DA_CHUNK][8]>(smem);
    float (*sRed)[3][4][64][4] = reinterpret_cast<float (*)[3][4][64][4]>(smem);   // [wave][mt][nt][lane][4 regs]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b0 = blockIdx.x * DA_CANDS;
    // this wave's vertex range (multiples of DA_CHUNK)
    const int n_chunks = (V + DA_CHUNK - 1) / DA_CHUNK;
    const int per_wave = (n_chunks + 3) / 4;
    const int c_begin = wave * per_wave, c_end = min(n_chunks, c_begin + per_wave);

    // A-operand role of this lane: row i = lane&15 of M-tile mt -> (cand, rc) ; k = lane>>4 -> vertex within the 4-group
    const int kq = lane >> 4, i16 = lane & 15;
    int a_cand[3], a_r[3], a_c[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
        const int row = mt * 16 + i16;          // 0..47
        a_cand[mt] = row / 12;
        const int rc = row % 12;
        a_r[mt] = rc >> 2;
        a_c[mt] = rc & 3;
    }
    f32x4 acc[3][4];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int ch = c_begin; ch < c_end; ++ch) {
        const int v0 = ch * DA_CHUNK;
        // stage 4 candidates x 64 vertices: lane = vertex
        const int v = v0 + lane;
#pragma unroll
        for (int cd = 0; cd < DA_CANDS; ++cd) {
            const int b = b0 + cd;
            float g0 = 0.f, g1 = 0.f, g2 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f;
            if (b < B && v < V) {
                const float* gp = g_verts + ((size_t)b * V + v) * 3;
                const float* pp = v_posed + (size_t)b * Npad + (size_t)v * 3;
                g0 = gp[0]; g1 = gp[1]; g2 = gp[2];
                p0 = pp[0]; p1 = pp[1]; p2 = pp[2];
            }
            float* x = sX[wave][cd][lane];
            *reinterpret_cast<float4*>(x) = make_float4(g0, g1, g2, p0);
            *reinterpret_cast<float4*>(x + 4) = make_float4(p1, p2, 1.0f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the wave's own LDS writes have landed
#pragma unroll 4
        for (int ks = 0; ks < DA_CHUNK / 4; ++ks) {
            const int vl = ks * 4 + kq;          // vertex within chunk for this lane's k
            // B operand: Wd[v][16*nt + i16]
            const float* wrow = Wd + (size_t)(v0 + vl) * WT_COLS + i16;
            float bw[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bw[nt] = wrow[nt * 16];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const float* x = sX[wave][a_cand[mt]][vl];
                const float a = x[a_r[mt]] * x[3 + a_c[mt]];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // cross-wave reduction in fixed wave order (sRed aliases sX: wait until every wave has left the main loop)
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            *reinterpret_cast<f32x4*>(&sRed[wave][mt][nt][lane][0]) = acc[mt][nt];
    __syncthreads();
    // C/D layout 16x16: col = lane&15 (joint within N-tile), row = (lane>>4)*4 + reg
    for (int e = threadIdx.x; e < 3 * 4 * 64 * 4; e += 256) {
        const int reg = e & 3, ln = (e >> 2) & 63, nt = (e >> 8) & 3, mt = e >> 10;
        const float s = ((sRed[0][mt][nt][ln][reg] + sRed[1][mt][nt][ln][reg]) + sRed[2][mt][nt][ln][reg]) + sRed[3][mt][nt][ln][reg];
        const int row = mt * 16 + (ln >> 4) * 4 + reg;   // (cand, rc)
        const int col = nt * 16 + (ln & 15);             // joint
        const int b = b0 + row / 12, rc = row % 12;
        if (b < B && col < DA_ROWS) dA[(size_t)b * dA_stride + col * 12 + rc] = s;
    }
}

int saga_launch_skin_fwd(const saga_model* m, const float* A, const float* v_posed, const float* transl, float* verts, int B,
                         hipStream_t st) {
    const int per_block = SKIN_THREADS * SKIN_VPT;
    dim3 grid((m->V + per_block - 1) / per_block, B);
    if (m->KW == 4)
        skin_fwd_kernel<true><<<grid, SKIN_THREADS, 0, st>>>(A, v_posed, m->w_idx, m->w_val, transl, verts, m->V, m->Npad, m->KW);
    else
        skin_fwd_kernel<false><<<grid, SKIN_THREADS, 0, st>>>(A, v_posed, m->w_idx, m->w_val, transl, verts, m->V, m->Npad, m->KW);
    SAGA_LAUNCH_CHECK();
    return 0;
}

int saga_launch_joints_fwd(const saga_model* m, const float* chain, const float* verts, const float* transl, float* joints, int B,
                           hipStream_t st) {
    const int total = B * (NJ + m->E + m->L);
    joints_fwd_kernel<<<(total + 255) / 256, 256, 0, st>>>(chain, verts, m->extra_verts, m->lmk_verts, m->lmk_bary, transl, joints,
                                                           B, m->V, m->E, m->L);
    SAGA_LAUNCH_CHECK();
    return 0;
}

int saga_launch_joints_bwd(const saga_model* m, const float* g_joints, float* g_pj, float* g_jsum, float* gv, int B,
                           int small_stride, hipStream_t st) {
    joints_bwd_kernel<<<B, 64, 0, st>>>(g_joints, m->extra_verts, m->lmk_verts, m->lmk_bary, g_pj, g_jsum, gv, m->V, m->E, m->L,
                                        small_stride);
    SAGA_LAUNCH_CHECK();
    return 0;
}

int saga_launch_skin_bwd(const saga_model* m, const float* A, const float* g_verts, const float* v_posed, float* g_vposed,
                         float* dA, int dA_stride, int B, hipStream_t st) {
    const int per_block = SKIN_THREADS * SKIN_VPT;
    dim3 grid1((m->Npad / 3 + 1 + per_block - 1) / per_block, B);     // covers every column of [0, Npad)
    if (m->KW == 4)
        skin_bwd_vposed_kernel<true><<<grid1, SKIN_THREADS, 0, st>>>(A, g_verts, m->w_idx, m->w_val, g_vposed, m->V, m->Npad, m->KW);
    else
        skin_bwd_vposed_kernel<false><<<grid1, SKIN_THREADS, 0, st>>>(A, g_verts, m->w_idx, m->w_val, g_vposed, m->V, m->Npad, m->KW);
    SAGA_LAUNCH_CHECK();
    skin_bwd_dA_kernel<<<(B + DA_CANDS - 1) / DA_CANDS, 256, 0, st>>>(g_verts, v_posed, m->Wd, dA, B, m->V, m->Npad, dA_stride);
    SAGA_LAUNCH_CHECK();
    return 0;
}
