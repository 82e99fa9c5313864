// Linear-blend skinning forward/backward, extra joints + landmarks, and the dA reduction.
// Replaces the skinning tail of smplx.lbs.lbs and VertexJointSelector / vertices2landmarks
// (reference call site WholeGraspPose/models/fittingop.py:149).
//
// HBM-bound kernels: per candidate the 55 skinning transforms (55x12 floats = 2.6 KB) are staged in LDS,
// LBS weights come from a compressed (joint, weight) table (model constant, L2-resident), v_posed is read
// and vertices are written exactly once.  The backward reduction dA[b,j] = sum_v W[v,j] g[b,v] (x) [v_posed;1]
// is shaped as a GEMM over the vertex index and runs on v_mfma_f32_16x16x4_f32 (exact f32, deterministic).
#include <stdlib.h>

#include "saga_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SKIN_THREADS 256

// ---- forward -------------------------------------------------------------------------------------------
// grid (ceil(V / (256*SKIN_VPT)), B). verts[b,v] = (sum_k w_k A[b, j_k]) . [v_posed[b,v]; 1] + transl[b]
// One block = one candidate x 1024 vertices: A[b] (2.6 KB) is staged in LDS once, every lane owns SKIN_VPT vertices
// (tile i: vertex base + i*256 + tid, so the 12-B per-lane accesses of a wave are one contiguous 768-B run) and issues
// all of its loads before the barrier.  KW4: the compressed weight table has exactly 4 entries per vertex and is read
// as one int4 + one float4.
#ifndef SKIN_VPT
#define SKIN_VPT 4
#endif
// SEL (saga_fit_step_sparse): the v_posed rows of the vertices a subset model keeps are also written, compact, in that model's
// layout (sel_rows [B, npad_s], pad columns 0; sel_slot[v] = row of full-mesh vertex v in the subset or -1) while they are in
// registers -- a separate gather would touch nearly every 128-B line of v_posed again (0.07 ms at B = 4096).
template <bool KW4, bool SEL = false>
__global__ __launch_bounds__(SKIN_THREADS) void skin_fwd_kernel(const float* __restrict__ A, const float* __restrict__ v_posed,
                                                                const int32_t* __restrict__ w_idx,
                                                                const float* __restrict__ w_val,
                                                                const float* __restrict__ transl, float* __restrict__ verts,
                                                                int V, int Npad, int KW,
                                                                const int32_t* __restrict__ sel_slot = nullptr,
                                                                float* __restrict__ sel_rows = nullptr, int n_sel3 = 0, int npad_s = 0,
                                                                const int32_t* __restrict__ parent_ids = nullptr,
                                                                float* __restrict__ sel_verts = nullptr) {
    __shared__ __attribute__((aligned(16))) float sA[NJ * 12];
    const int b = blockIdx.y;
    const int vbase = blockIdx.x * (SKIN_THREADS * SKIN_VPT);
    const int tid = threadIdx.x;
    for (int i = tid; i < NJ * 12; i += SKIN_THREADS) sA[i] = A[(size_t)b * (NJ * 12) + i];
    const float* src = v_posed + (size_t)b * Npad;
    float px[SKIN_VPT], py[SKIN_VPT], pz[SKIN_VPT];
    int4 ji[SKIN_VPT];
    float4 wv[SKIN_VPT];
    int slot[SKIN_VPT];
#pragma unroll
    for (int i = 0; i < SKIN_VPT; ++i) {
        const int v = min(vbase + i * SKIN_THREADS + tid, V - 1);
        if (SEL) slot[i] = sel_slot[parent_ids ? parent_ids[v] : v];     // (this model may itself be a subset: full-mesh id first)
        typedef float f32x3_t __attribute__((ext_vector_type(3)));
        const f32x3_t pv = __builtin_nontemporal_load(reinterpret_cast<const f32x3_t*>(src + v * 3));   // one 12-B load, read once
        const float3 p3 = make_float3(pv.x, pv.y, pv.z);
        px[i] = p3.x; py[i] = p3.y; pz[i] = p3.z;
        if (KW4) {
            ji[i] = reinterpret_cast<const int4*>(w_idx)[v];
            wv[i] = reinterpret_cast<const float4*>(w_val)[v];
        }
    }
    const float tx = transl ? transl[b * 3] : 0.f, ty = transl ? transl[b * 3 + 1] : 0.f, tz = transl ? transl[b * 3 + 2] : 0.f;
    __syncthreads();
    float* dst = verts + (size_t)b * V * 3;
    if (SEL && blockIdx.x == 0)
        for (int c = n_sel3 + tid; c < npad_s; c += SKIN_THREADS) sel_rows[(size_t)b * npad_s + c] = 0.f;
#pragma unroll
    for (int i = 0; i < SKIN_VPT; ++i) {
        const int v = vbase + i * SKIN_THREADS + tid;
        if (v >= V) continue;
        if (SEL && slot[i] >= 0) {
            float* r = sel_rows + (size_t)b * npad_s + slot[i] * 3;
            r[0] = px[i]; r[1] = py[i]; r[2] = pz[i];
        }
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
        float3 o;
        o.x = (T[0] * px[i] + T[1] * py[i] + T[2] * pz[i] + T[3]) + tx;
        o.y = (T[4] * px[i] + T[5] * py[i] + T[6] * pz[i] + T[7]) + ty;
        o.z = (T[8] * px[i] + T[9] * py[i] + T[10] * pz[i] + T[11]) + tz;
        {   // written once, read by other kernels much later: non-temporal (0.245 -> 0.232 ms)
            typedef float f32x3_t __attribute__((ext_vector_type(3)));
            const f32x3_t oo = {o.x, o.y, o.z};
            __builtin_nontemporal_store(oo, reinterpret_cast<f32x3_t*>(dst + v * 3));   // one 12-B store
        }
        if (SEL && sel_verts && slot[i] >= 0) {      // the selected vertices again, compact [B, n_sel, 3]
            float* r = sel_verts + ((size_t)b * (n_sel3 / 3) + slot[i]) * 3;
            r[0] = o.x; r[1] = o.y; r[2] = o.z;
        }
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
#ifdef DA_PHASE_TIMING
__device__ unsigned long long da_phase[8];
extern "C" int saga_debug_read_da_phases(unsigned long long* out) {
    unsigned long long z[8] = {0};
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(da_phase), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(da_phase), z, sizeof(z));
    return (int)e;
}
#define DA_T(i) do { const unsigned long long t_ = clock64(); tacc[i] += t_ - t_prev; t_prev = t_; } while (0)
#else
#define DA_T(i)
#endif
#define DA_CANDS 4
#define DA_STEP 16      // vertices per wave per iteration: 4 k-steps of v_mfma_f32_16x16x4_f32
#define DA_XSTRIDE (DA_STEP * 8 + 16)  // floats per candidate plane of a staging tile: = 16 mod 32, so the A-operand reads of the two
                                       // candidates of an MFMA tile (offsets kq*8 + 0..6) fall on disjoint banks

// FUSE (compressed weight table with exactly 4 entries per vertex): the same pass also produces
// g_v_posed[b,v] = T_v^T g[b,v] (part 1 above) from the rows it has already fetched -- its VALU work hides under the MFMAs
// and g_verts is read from HBM once instead of twice.
template <bool FUSE>
__global__ __launch_bounds__(256) void skin_bwd_dA_kernel(const float* __restrict__ g_verts, const float* __restrict__ v_posed,
                                                          const float* __restrict__ Wd, float* __restrict__ dA,
                                                          const float* __restrict__ A, const int32_t* __restrict__ w_idx,
                                                          const float* __restrict__ w_val, float* __restrict__ g_vposed, int B,
                                                          int V, int Npad, int dA_stride, size_t part_stride) {
    // 24 KB of LDS (+ 10.6 KB of skinning transforms when fused; 4 blocks = 16 waves per CU): per-wave staging tiles
    // first, then (after a barrier) the two 12 KB slots of the cross-wave reduction
    __shared__ __attribute__((aligned(16))) float smem[2 * 12 * 64 * 4 + (FUSE ? DA_CANDS * NJ * 12 : 0)];
    float* sA = smem + 2 * 12 * 64 * 4;      // [cand][joint][12]
    float (*sRed)[12][64][4] = reinterpret_cast<float (*)[12][64][4]>(smem);   // [slot][mt*4+nt][lane][4 regs]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // per wave staging: [cand][vertex][8]: g(3), vp(3), 1, pad
    float* sXw = smem + wave * (DA_CANDS * DA_XSTRIDE);
    const int b0 = blockIdx.x * DA_CANDS;
    // this block's share of the 16-vertex steps (blockIdx.y of gridDim.y vertex ranges: partial sums go to slab
    // blockIdx.y of dA and are added in slab order by da_sum_kernel), and this wave's quarter of it
    const int n_steps = (V + DA_STEP - 1) / DA_STEP;
    const int per_block = (n_steps + gridDim.y - 1) / gridDim.y;
    const int blk_begin = blockIdx.y * per_block, blk_end = min(n_steps, blk_begin + per_block);
    const int per_wave = (max(blk_end - blk_begin, 0) + 3) / 4;
    const int s_begin = blk_begin + wave * per_wave, s_end = min(blk_end, s_begin + per_wave);
    dA += blockIdx.y * part_stride;

    // A-operand role of this lane: row i = lane&15 of M-tile mt -> (cand, rc) ; k = lane>>4 -> vertex within the 4-group
    const int kq = lane >> 4, i16 = lane & 15;
    int a_off_r[3], a_off_c[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
        const int row = mt * 16 + i16;          // 0..47
        const int cand = row / 12, rc = row % 12;
        a_off_r[mt] = cand * DA_XSTRIDE + (rc >> 2);
        a_off_c[mt] = cand * DA_XSTRIDE + 3 + (rc & 3);
    }
    f32x4 acc[3][4];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging role of this lane: candidate lane>>4, vertex lane&15 of the step.  Software pipeline: the next step's
    // (g, v_posed) rows are in flight while this step's MFMAs run; the step's Wd rows are requested BEFORE that
    // prefetch, so waiting for them (vmcnt retires in order) does not wait for the prefetch.
    const int st_cd = lane >> 4, st_v = lane & 15;
    const int st_b = min(b0 + st_cd, B - 1);
    const float* gbase = g_verts + (size_t)st_b * V * 3;
    const float* pbase = v_posed + (size_t)st_b * Npad;
    float* xw = sXw + st_cd * DA_XSTRIDE + st_v * 8;
    float3 rg, rp;
    int4 rj;
    float4 rw;
    auto fetch = [&](int step) {
        const int vc = min(step * DA_STEP + st_v, V - 1);
        rg = *reinterpret_cast<const float3*>(gbase + (size_t)vc * 3);
        rp = *reinterpret_cast<const float3*>(pbase + (size_t)vc * 3);
        if (FUSE) {
            rj = reinterpret_cast<const int4*>(w_idx)[vc];
            rw = reinterpret_cast<const float4*>(w_val)[vc];
        }
    };
    if (FUSE) {
        for (int i = threadIdx.x; i < DA_CANDS * NJ * 12; i += 256) {
            const int cd = i / (NJ * 12);
            sA[i] = A[(size_t)min(b0 + cd, B - 1) * (NJ * 12) + (i - cd * NJ * 12)];
        }
        // pad columns [3V, Npad) of g_v_posed (the split-K GEMM reads them)
        if (wave == 0 && blockIdx.y == 0 && b0 + st_cd < B)
            for (int c = 3 * V + st_v; c < Npad; c += 16) g_vposed[(size_t)(b0 + st_cd) * Npad + c] = 0.f;
        __syncthreads();
    }
    float* gvp_base = FUSE ? g_vposed + (size_t)st_b * Npad : nullptr;
    if (s_begin < s_end) fetch(s_begin);
#ifdef DA_PHASE_TIMING
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, t_prev = clock64();
#endif
    for (int step = s_begin; step < s_end; ++step) {
        const int v0 = step * DA_STEP;
        // B operands of the four k-steps: joints i16 + 16*nt of vertex v0 + 4*ks + kq, one 16-B load per k-step (the table
        // stores them side by side; rows >= V are zero).  Requested FIRST: what limits this kernel is the vector-memory
        // issue rate (16 dword loads per step cost 2.4k cycles of issue per wave against 1.5k cycles of MFMA), and the
        // staging and g_v_posed arithmetic below run while these are in flight.  The rows fetched one step ago are older
        // in vmcnt order, so consuming them does not wait for these.
        float bw[4][4];
        {
            const float4* wlane = reinterpret_cast<const float4*>(Wd + (size_t)v0 * WT_COLS + kq * WT_COLS) + i16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float4 w4 = wlane[ks * 4 * (WT_COLS / 4)];
                bw[ks][0] = w4.x; bw[ks][1] = w4.y; bw[ks][2] = w4.z; bw[ks][3] = w4.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        DA_T(0);
        {
            const bool ok = (b0 + st_cd < B) && (v0 + st_v < V);
            const float z = ok ? 1.f : 0.f;
            *reinterpret_cast<float4*>(xw) = make_float4(ok ? rg.x : 0.f, ok ? rg.y : 0.f, ok ? rg.z : 0.f, ok ? rp.x : 0.f);
            *reinterpret_cast<float4*>(xw + 4) = make_float4(ok ? rp.y : 0.f, ok ? rp.z : 0.f, z, 0.f);
        }
        DA_T(1);
        float3 gvp;
        bool gvp_ok = false;
        if (FUSE) {
            // g_v_posed of this lane's (candidate, vertex): T^T g with T = sum_k w_k A[j_k][:3,:3]
            const int jj[4] = {rj.x, rj.y, rj.z, rj.w};
            const float ww[4] = {rw.x, rw.y, rw.z, rw.w};
            float T[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) T[k] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4* a = reinterpret_cast<const float4*>(sA + (st_cd * NJ + jj[k]) * 12);
                const float4 a0 = a[0], a1 = a[1], a2 = a[2];
                const float w = ww[k];
                T[0] = fmaf(w, a0.x, T[0]); T[1] = fmaf(w, a0.y, T[1]); T[2] = fmaf(w, a0.z, T[2]);
                T[3] = fmaf(w, a1.x, T[3]); T[4] = fmaf(w, a1.y, T[4]); T[5] = fmaf(w, a1.z, T[5]);
                T[6] = fmaf(w, a2.x, T[6]); T[7] = fmaf(w, a2.y, T[7]); T[8] = fmaf(w, a2.z, T[8]);
            }
            gvp.x = T[0] * rg.x + T[3] * rg.y + T[6] * rg.z;
            gvp.y = T[1] * rg.x + T[4] * rg.y + T[7] * rg.z;
            gvp.z = T[2] * rg.x + T[5] * rg.y + T[8] * rg.z;
            gvp_ok = (b0 + st_cd < B) && (v0 + st_v < V);
        }
        DA_T(2);
        fetch(min(step + 1, s_end - 1));        // unconditional (a branch here makes hipcc wait for the loads at the join)
        if (FUSE && gvp_ok) *reinterpret_cast<float3*>(gvp_base + (size_t)(v0 + st_v) * 3) = gvp;
        __builtin_amdgcn_wave_barrier();
        DA_T(3);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int vl = ks * 4 + kq;          // vertex within the step for this lane's k
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const float a = sXw[a_off_r[mt] + vl * 8] * sXw[a_off_c[mt] + vl * 8];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[ks][nt], acc[mt][nt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        DA_T(4);
    }
#ifdef DA_PHASE_TIMING
    if (lane == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(&da_phase[i], tacc[i]);
#endif
    // cross-wave reduction, fixed order (w0 + w2) + (w1 + w3), in the accumulators' own register layout
    // (sRed aliases the staging tiles: wait until every wave has left the main loop)
    __syncthreads();
    if (wave >= 2) {
#pragma unroll
        for (int t = 0; t < 12; ++t) *reinterpret_cast<f32x4*>(&sRed[wave - 2][t][lane][0]) = acc[t >> 2][t & 3];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int t = 0; t < 12; ++t) acc[t >> 2][t & 3] += *reinterpret_cast<const f32x4*>(&sRed[wave][t][lane][0]);
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int t = 0; t < 12; ++t) *reinterpret_cast<f32x4*>(&sRed[0][t][lane][0]) = acc[t >> 2][t & 3];
    }
    __syncthreads();
    if (wave != 0) return;
    // C/D layout 16x16: col = lane&15 (joint within N-tile), row = (lane>>4)*4 + reg
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        const int mt = t >> 2, nt = t & 3;
        const f32x4 s = acc[mt][nt] + *reinterpret_cast<const f32x4*>(&sRed[0][t][lane][0]);
        const int col = nt * 16 + (lane & 15);             // joint
        if (col >= DA_ROWS) continue;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = mt * 16 + (lane >> 4) * 4 + reg;   // (cand, rc)
            const int b = b0 + row / 12, rc = row % 12;
            if (b < B) dA[(size_t)b * dA_stride + col * 12 + rc] = s[reg];
        }
    }
}


// ---- backward of skinning in one streaming pass, sparse weights ------------------------------------------------------
// The skinning weights have a handful of non-zeros per vertex (4 in SMPL-family models), so the dense GEMM view above multiplies
// mostly zeros (13.8 MFLOP per candidate against 1 MFLOP of useful work).  Here a block owns one candidate x one vertex range:
//  phase 1  streams the range's g_vertices / v_posed rows once (coalesced 12-B accesses), writes g_v_posed = T^T g, and leaves
//           (g, v_posed) in LDS (6 floats per vertex);
//  phase 2  one lane per item (= a chunk of one joint's (vertex, weight) list inside the range, model constant, built by
//           saga_model_create): the lane walks its chunk in ascending vertex order and accumulates its 12 sums in registers --
//           no atomics, no cross-lane traffic;
//  phase 3  the items of a joint are added in item order.  Every order is fixed: results are bit-reproducible and do not
//           depend on the batch.  Partial sums per range go to slab blockIdx.y; da_sum_kernel adds the slabs in order.
// Row 55 (sum_v g -> d transl): per-lane sums in vertex order, wave tree, waves in order.
#define DS_VPT ((DS_VR + DS_THREADS - 1) / DS_THREADS)
#define DS_OPT ((DA_ROWS * 12 + DS_THREADS - 1) / DS_THREADS)        // outputs per thread
// A block walks `rpb` consecutive ranges of its candidate (grid.y blocks per candidate; each writes its own slab).  The next
// range's rows are requested as soon as phase 1 has consumed the current ones, so they are in flight during phases 2 and 3.
template <bool KW4>
__global__ __launch_bounds__(DS_THREADS, 4) void skin_bwd_sparse_kernel(
    const float* __restrict__ g_verts, const float* __restrict__ v_posed, const float* __restrict__ A,
    const int32_t* __restrict__ w_idx, const float* __restrict__ w_val, const int2* __restrict__ ds_ent,
    const int32_t* __restrict__ ds_meta, const int32_t* __restrict__ ds_item_ptr, float* __restrict__ g_vposed,
    float* __restrict__ dA_part, int V, int Npad, int KW, int Vr, int n_ranges, int rpb, int dA_stride, size_t part_stride,
    int* __restrict__ zero8) {
    if (zero8 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 8) zero8[threadIdx.x] = 0;      // the tile queues of the blend contraction that follows
    __shared__ __attribute__((aligned(16))) float sX[DS_VR * 6];     // phase 1/2: [vertex][g.xyz, p.xyz]; phase 3: [12][DS_THREADS]
    __shared__ __attribute__((aligned(16))) float sA[NJ * 12];
    __shared__ float sG[DS_THREADS / 64][4];
#if defined(SAGA_EXPERIMENT_BUILD) && defined(DS_EXP_PAD)       // timing experiment: extra LDS per block (occupancy of a replicated joint table)
    __shared__ float sPad[DS_EXP_PAD];
    if (V < 0) sPad[threadIdx.x] = (float)Npad;
    if (V < -1) g_vposed[threadIdx.x] = sPad[threadIdx.x + 1];
#endif
    const int b = blockIdx.x, tid = threadIdx.x;
    const int r_begin = blockIdx.y * rpb, r_end = min(n_ranges, r_begin + rpb);
    for (int i = tid; i < NJ * 12; i += DS_THREADS) sA[i] = A[(size_t)b * (NJ * 12) + i];
    const float* gsrc = g_verts + (size_t)b * V * 3;
    const float* psrc = v_posed + (size_t)b * Npad;
    float* dst = g_vposed + (size_t)b * Npad;
    float3 g[DS_VPT], p[DS_VPT];
    int4 ji[DS_VPT];
    float4 wv[DS_VPT];
    int2 en[4];
    auto fetch_rows = [&](int r) {
        const int v_lo = r * Vr, n_here = min(Vr, V - v_lo);
#pragma unroll
        for (int i = 0; i < DS_VPT; ++i) {
            const int v = v_lo + min(i * DS_THREADS + tid, n_here - 1);
            // read once: keep the streamed rows from evicting the weight / entry tables from L2
            typedef float f32x3_t __attribute__((ext_vector_type(3)));
            const f32x3_t gg = __builtin_nontemporal_load(reinterpret_cast<const f32x3_t*>(gsrc + (size_t)v * 3));
            const f32x3_t pp = __builtin_nontemporal_load(reinterpret_cast<const f32x3_t*>(psrc + (size_t)v * 3));
            g[i] = make_float3(gg.x, gg.y, gg.z);
            p[i] = make_float3(pp.x, pp.y, pp.z);
        }
    };
    auto fetch_weights = [&](int r) {       // model constants (L2-resident): fetched when phase 1 starts, not a range ahead
        const int v_lo = r * Vr, n_here = min(Vr, V - v_lo);
#pragma unroll
        for (int i = 0; i < DS_VPT; ++i) {
            const int v = v_lo + min(i * DS_THREADS + tid, n_here - 1);
            if (KW4) {
                ji[i] = reinterpret_cast<const int4*>(w_idx)[v];
                wv[i] = reinterpret_cast<const float4*>(w_val)[v];
            }
        }
    };
    auto fetch_first_entries = [&](int r) {
        const int2* e = ds_ent + ds_meta[2 * r] + tid;
#pragma unroll
        for (int u = 0; u < 4; ++u) en[u] = e[(size_t)u * DS_THREADS];
    };
    fetch_first_entries(r_begin);
    fetch_rows(r_begin);
    // pad columns [3V, Npad) of g_v_posed (the split-K GEMM reads them)
    if (blockIdx.y == gridDim.y - 1)
        for (int c = 3 * V + tid; c < Npad; c += DS_THREADS) dst[c] = 0.f;
    float sgx = 0.f, sgy = 0.f, sgz = 0.f;
    float outv[DS_OPT];
#pragma unroll
    for (int q = 0; q < DS_OPT; ++q) outv[q] = 0.f;

    for (int r = r_begin; r < r_end; ++r) {
        const int v_lo = r * Vr, n_here = min(Vr, V - v_lo);
        fetch_weights(r);
        __syncthreads();        // sA staged (first range) / phase 3 of the previous range has left sX
#pragma unroll
        for (int i = 0; i < DS_VPT; ++i) {
            const int vl = i * DS_THREADS + tid;
            if (vl >= n_here) continue;
            const int v = v_lo + vl;
            float T[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) T[k] = 0.f;
            if (KW4) {
#if defined(SAGA_EXPERIMENT_BUILD) && defined(DS_EXP_P1)        // timing experiment: conflict-free joint reads (wrong results)
                const int jj[4] = {(tid & 15) + (ji[i].x & 16), (tid & 15) + (ji[i].y & 16), (tid & 15) + (ji[i].z & 16), (tid & 15) + (ji[i].w & 16)};
#else
                const int jj[4] = {ji[i].x, ji[i].y, ji[i].z, ji[i].w};
#endif
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
                    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                        for (int c = 0; c < 3; ++c) T[rr * 3 + c] = fmaf(w, a[rr * 4 + c], T[rr * 3 + c]);
                }
            }
            float3 o;
            o.x = T[0] * g[i].x + T[3] * g[i].y + T[6] * g[i].z;
            o.y = T[1] * g[i].x + T[4] * g[i].y + T[7] * g[i].z;
            o.z = T[2] * g[i].x + T[5] * g[i].y + T[8] * g[i].z;
            *reinterpret_cast<float3*>(dst + (size_t)v * 3) = o;
            float2* row = reinterpret_cast<float2*>(sX + vl * 6);
            row[0] = make_float2(g[i].x, g[i].y);
            row[1] = make_float2(g[i].z, p[i].x);
            row[2] = make_float2(p[i].y, p[i].z);
            sgx += g[i].x; sgy += g[i].y; sgz += g[i].z;
            __builtin_amdgcn_sched_barrier(0);      // one vertex at a time: interleaving them costs 40+ registers
        }
        const int C = ds_meta[2 * r + 1];
        const int2* e = ds_ent + ds_meta[2 * r] + tid;
        const int rn = min(r + 1, r_end - 1);
        if (r + 1 < r_end) fetch_rows(rn);
        __syncthreads();

        // phase 2 (the lane's first entries were requested a range ago; each batch of 4 is requested one batch ahead)
        float acc[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[k] = 0.f;
        for (int c = 0; c < C; c += 4) {
            int2 cur[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) cur[u] = en[u];
            const int cn = min(c + 4, C - 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) en[u] = e[(size_t)(cn + u) * DS_THREADS];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#if defined(SAGA_EXPERIMENT_BUILD) && defined(DS_EXP_P2)        // timing experiment: conflict-free row reads (wrong results)
                const float2* row = reinterpret_cast<const float2*>(sX + (((cur[u].x & 480) + tid) & 511) * 6);
#else
                const float2* row = reinterpret_cast<const float2*>(sX + cur[u].x * 6);
#endif
                const float2 r0 = row[0], r1 = row[1], r2 = row[2];
                const float w = __int_as_float(cur[u].y);
                const float wg0 = w * r0.x, wg1 = w * r0.y, wg2 = w * r1.x;
                const float px = r1.y, py = r2.x, pz = r2.y;
                acc[0] = fmaf(wg0, px, acc[0]); acc[1] = fmaf(wg0, py, acc[1]); acc[2] = fmaf(wg0, pz, acc[2]); acc[3] += wg0;
                acc[4] = fmaf(wg1, px, acc[4]); acc[5] = fmaf(wg1, py, acc[5]); acc[6] = fmaf(wg1, pz, acc[6]); acc[7] += wg1;
                acc[8] = fmaf(wg2, px, acc[8]); acc[9] = fmaf(wg2, py, acc[9]); acc[10] = fmaf(wg2, pz, acc[10]); acc[11] += wg2;
            }
        }
        fetch_first_entries(rn);
        __syncthreads();        // every lane has left sX
        float* sP = sX;
#pragma unroll
        for (int k = 0; k < 12; ++k) sP[k * DS_THREADS + tid] = acc[k];
        __syncthreads();
        // phase 3: this thread's outputs += the items of their joint, in item order
        const int32_t* iptr = ds_item_ptr + r * (NJ + 1);
#pragma unroll
        for (int q = 0; q < DS_OPT; ++q) {
            const int o = q * DS_THREADS + tid;
            const int j = o / 12, k = o - j * 12;
            if (j < NJ) {
                const int i0 = iptr[j], i1 = iptr[j + 1];
                float s = outv[q];
                int it = i0;
                for (; it + 4 <= i1; it += 4) {     // loads issued together, added in item order
                    const float a0 = sP[k * DS_THREADS + it], a1 = sP[k * DS_THREADS + it + 1], a2 = sP[k * DS_THREADS + it + 2],
                                a3 = sP[k * DS_THREADS + it + 3];
                    s += a0; s += a1; s += a2; s += a3;
                }
                for (; it < i1; ++it) s += sP[k * DS_THREADS + it];
                outv[q] = s;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        sgx += __shfl_down(sgx, o, 64); sgy += __shfl_down(sgy, o, 64); sgz += __shfl_down(sgz, o, 64);
    }
    if ((tid & 63) == 0) { sG[tid >> 6][0] = sgx; sG[tid >> 6][1] = sgy; sG[tid >> 6][2] = sgz; }
    __syncthreads();
    float* out = dA_part + (size_t)blockIdx.y * part_stride + (size_t)b * dA_stride;
#pragma unroll
    for (int q = 0; q < DS_OPT; ++q) {
        const int o = q * DS_THREADS + tid;
        if (o >= DA_ROWS * 12) continue;
        const int j = o / 12, k = o - j * 12;
        float s = outv[q];
        if (j == NJ) {
            s = 0.f;
            if ((k & 3) == 3) {
                s = sG[0][k >> 2];
                for (int w = 1; w < DS_THREADS / 64; ++w) s += sG[w][k >> 2];
            }
        }
        out[o] = s;
    }
}

// sub (may be NULL): subset model over this mesh; its rows of v_posed are also written to sel_rows [B, sub->Npad]
// m may itself be a subset model (of the same full mesh as sub: sub's vertices must all be vertices of m -- the caller's contract);
// sel_verts (may be NULL): the selected rows of verts, compact
int saga_launch_skin_fwd_sel(const saga_model* m, const saga_model* sub, const float* A, const float* v_posed, const float* transl,
                             float* verts, float* sel_rows, float* sel_verts, int B, hipStream_t st) {
    const int per_block = SKIN_THREADS * SKIN_VPT;
    dim3 grid((m->V + per_block - 1) / per_block, B);
    SAGA_CHECK_ARG(sub && sub->full_slot && sub->VF == m->VF && sub->V <= m->V && sel_rows, "skin_fwd_sel: not a subset model of this mesh");
    if (m->KW == 4)
        skin_fwd_kernel<true, true><<<grid, SKIN_THREADS, 0, st>>>(A, v_posed, m->w_idx, m->w_val, transl, verts, m->V, m->Npad, m->KW,
                                                                   sub->full_slot, sel_rows, sub->V * 3, sub->Npad, m->sel_ids, sel_verts);
    else
        skin_fwd_kernel<false, true><<<grid, SKIN_THREADS, 0, st>>>(A, v_posed, m->w_idx, m->w_val, transl, verts, m->V, m->Npad, m->KW,
                                                                    sub->full_slot, sel_rows, sub->V * 3, sub->Npad, m->sel_ids, sel_verts);
    SAGA_LAUNCH_CHECK();
    return 0;
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

// dA[b][i] = sum of the vertex-range partials, in slab order (fixed -> bit-reproducible); i < DA_ROWS*12 only: the rest of a
// dA row belongs to the joint gradients
__global__ void da_sum_kernel(const float* __restrict__ part, float* __restrict__ dA, int B, int dA_stride, int n_parts, size_t part_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (DA_ROWS * 12)) return;
    const int b = i / (DA_ROWS * 12), c = i - b * (DA_ROWS * 12);
    const size_t off = (size_t)b * dA_stride + c;
    float acc = part[off];
    for (int p = 1; p < n_parts; ++p) acc += part[p * part_stride + off];
    dA[off] = acc;
}

// scratch: >= 3 * B * dA_stride floats, free for the duration of the call (the split-K slabs, written later), or NULL
// zero8 (optional): 8 ints this launch sets to zero for the kernel after it; *zeroed says whether it did (only the streaming kernel does)
int saga_launch_skin_bwd(const saga_model* m, const float* A, const float* g_verts, const float* v_posed, float* g_vposed,
                         float* dA, int dA_stride, float* scratch, int B, hipStream_t st, int* zero8, bool* zeroed) {
    if (zeroed) *zeroed = false;
    static const int force_mfma = getenv("SAGA_DA_MFMA") != nullptr;      // cross-check switch: the dense MFMA reduction
    if (m->ds_parts > 0 && scratch && !force_mfma) {
        // one block per candidate walks every range (B = 4096: exactly 4 rounds of 4 blocks per CU; two blocks per candidate with
        // their partial sums added by da_sum_kernel measured 0.02 ms slower).  The kernel supports grid.y > 1 (a fixed split,
        // whatever the batch: a candidate's sums must not depend on the batch around it).
        const int halves = 1;
        const int rpb = (m->ds_parts + halves - 1) / halves;
        const size_t part_stride = (size_t)B * dA_stride;
        float* out = halves > 1 ? scratch : dA;
        dim3 grid(B, halves);
        if (m->KW == 4)
            skin_bwd_sparse_kernel<true><<<grid, DS_THREADS, 0, st>>>(g_verts, v_posed, A, m->w_idx, m->w_val, m->ds_ent, m->ds_meta,
                                                                      m->ds_item_ptr, g_vposed, out, m->V, m->Npad, m->KW, m->ds_vr,
                                                                      m->ds_parts, rpb, dA_stride, part_stride, zero8);
        else
            skin_bwd_sparse_kernel<false><<<grid, DS_THREADS, 0, st>>>(g_verts, v_posed, A, m->w_idx, m->w_val, m->ds_ent, m->ds_meta,
                                                                       m->ds_item_ptr, g_vposed, out, m->V, m->Npad, m->KW, m->ds_vr,
                                                                       m->ds_parts, rpb, dA_stride, part_stride, zero8);
        SAGA_LAUNCH_CHECK();
        if (zeroed) *zeroed = zero8 != nullptr;
        if (halves > 1) {
            const int n = B * (DA_ROWS * 12);
            da_sum_kernel<<<(n + 255) / 256, 256, 0, st>>>(scratch, dA, B, dA_stride, halves, part_stride);
            SAGA_LAUNCH_CHECK();
        }
        return 0;
    }
    const int per_block = SKIN_THREADS * SKIN_VPT;
    dim3 grid1((m->Npad / 3 + 1 + per_block - 1) / per_block, B);     // covers every column of [0, Npad)
    const int blocks = (B + DA_CANDS - 1) / DA_CANDS;
    // three vertex ranges per candidate group: 3 x (B/4) blocks of a third of the length fill the 3 blocks/CU evenly (B = 4096:
    // 3072 blocks = 4 full rounds instead of 1 1/3) and keep small batches from leaving the chip idle
    const int n_steps = (m->V + DA_STEP - 1) / DA_STEP;
    // (a fixed number of ranges, whatever the batch: a candidate's sums must not depend on the batch around it)
    const int parts = (scratch && n_steps >= 48) ? 3 : 1;
    const size_t part_stride = (size_t)B * dA_stride;
    float* out = parts > 1 ? scratch : dA;
    dim3 grid(blocks, parts);
    if (m->KW == 4) {
        // one pass: dA reduction on the matrix cores + g_v_posed on the vector units
        skin_bwd_dA_kernel<true><<<grid, 256, 0, st>>>(g_verts, v_posed, m->Wd, out, A, m->w_idx, m->w_val, g_vposed, B, m->V, m->Npad,
                                                       dA_stride, part_stride);
        SAGA_LAUNCH_CHECK();
    } else {
        skin_bwd_vposed_kernel<false><<<grid1, SKIN_THREADS, 0, st>>>(A, g_verts, m->w_idx, m->w_val, g_vposed, m->V, m->Npad, m->KW);
        SAGA_LAUNCH_CHECK();
        skin_bwd_dA_kernel<false><<<grid, 256, 0, st>>>(g_verts, v_posed, m->Wd, out, nullptr, nullptr, nullptr, nullptr, B, m->V, m->Npad,
                                                        dA_stride, part_stride);
        SAGA_LAUNCH_CHECK();
    }
    if (parts > 1) {
        const int n = B * (DA_ROWS * 12);
        da_sum_kernel<<<(n + 255) / 256, 256, 0, st>>>(scratch, dA, B, dA_stride, parts, part_stride);
        SAGA_LAUNCH_CHECK();
    }
    return 0;
}
