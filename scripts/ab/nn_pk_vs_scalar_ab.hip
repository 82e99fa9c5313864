// A/B micro-benchmark: the approximate NN pass with v_pk_fma_f32 (nnf_scan of nn_fast.h) against the same pass with plain
// v_fma_f32 (same LDS tile, same chunk bookkeeping).  On gfx950 a wave64 v_fma_f32 issues in 2 cycles when the SIMD has other
// waves to interleave; the question is what v_pk_fma_f32 costs.  Scan only; random data.  Not part of the library.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas -fno-slp-vectorize -I saga_amd/csrc scripts/ab/nn_pk_vs_scalar_ab.hip -o build/nn_pk_ab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "nn_fast.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define NPTS 2048

template <int NQ>
__device__ __forceinline__ void scan_scalar(const float* __restrict__ s, int n_chunks, const float (&qx)[NQ], const float (&qy)[NQ],
                                            const float (&qz)[NQ], float (&m1)[NQ], float (&m2)[NQ], float (&m3)[NQ], int (&c1)[NQ],
                                            int (&c2)[NQ]) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) { m1[i] = INFINITY; m2[i] = INFINITY; m3[i] = INFINITY; c1[i] = -1; c2[i] = -1; }
    auto load4 = [&](float4 (&r)[4], int c, int g) {
        const float4* t = reinterpret_cast<const float4*>(s + c * NNF_CSTRIDE) + 4 * g;
        r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; r[3] = t[3];
    };
    float m[NQ];
    auto step = [&](const float4 (&r)[4]) {
        // r[0] = x0 x1 y0 y1, r[1] = z0 z1 w0 w1 (points 0,1); r[2], r[3]: points 2,3
        float v[NQ][4];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            v[i][0] = __builtin_fmaf(qx[i], r[0].x, r[1].z); v[i][1] = __builtin_fmaf(qx[i], r[0].y, r[1].w);
            v[i][2] = __builtin_fmaf(qx[i], r[2].x, r[3].z); v[i][3] = __builtin_fmaf(qx[i], r[2].y, r[3].w);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            v[i][0] = __builtin_fmaf(qy[i], r[0].z, v[i][0]); v[i][1] = __builtin_fmaf(qy[i], r[0].w, v[i][1]);
            v[i][2] = __builtin_fmaf(qy[i], r[2].z, v[i][2]); v[i][3] = __builtin_fmaf(qy[i], r[2].w, v[i][3]);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            v[i][0] = __builtin_fmaf(qz[i], r[1].x, v[i][0]); v[i][1] = __builtin_fmaf(qz[i], r[1].y, v[i][1]);
            v[i][2] = __builtin_fmaf(qz[i], r[3].x, v[i][2]); v[i][3] = __builtin_fmaf(qz[i], r[3].y, v[i][3]);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            m[i] = __builtin_fminf(__builtin_fminf(m[i], v[i][0]), v[i][1]);
            m[i] = __builtin_fminf(__builtin_fminf(m[i], v[i][2]), v[i][3]);
        }
    };
    float4 A[4], B[4];
    load4(A, 0, 0);
    for (int c = 0; c < n_chunks; ++c) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) m[i] = INFINITY;
#pragma unroll
        for (int g = 0; g < NNF_CH / 4; g += 2) {
            load4(B, c, g + 1);
            step(A);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < NNF_CH / 4) load4(A, c, g + 2);
            else load4(A, min(c + 1, n_chunks - 1), 0);
            step(B);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const bool lt1 = m[i] < m1[i], lt2 = m[i] < m2[i];
            m3[i] = __builtin_amdgcn_fmed3f(m[i], m2[i], m3[i]);
            m2[i] = __builtin_amdgcn_fmed3f(m[i], m1[i], m2[i]);
            m1[i] = __builtin_fminf(m[i], m1[i]);
            int n2 = lt2 ? c : c2[i];
            n2 = lt1 ? c1[i] : n2;
            c2[i] = n2;
            c1[i] = lt1 ? c : c1[i];
        }
    }
}

template <bool PK, int NQ, int OCC>
__global__ __launch_bounds__(256, OCC) void scan_kernel(const float* __restrict__ pts, const float* __restrict__ qs, float* __restrict__ out, int reps) {
    __shared__ __attribute__((aligned(16))) float sdb[NNF_TILE_FLOATS(NPTS)];
    const int tid = threadIdx.x, b = blockIdx.x;
    NnfFrame f; f.cx = f.cy = f.cz = 0.f; f.rp = 1.f;
    for (int k = tid; k < NPTS; k += 256) nnf_store_point(sdb, k, pts[k * 3], pts[k * 3 + 1], pts[k * 3 + 2], f);
    __syncthreads();
    float qx[NQ], qy[NQ], qz[NQ], m1[NQ], m2[NQ], m3[NQ];
    int c1[NQ], c2[NQ];
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const float* q = qs + ((size_t)(b * 1024 + i * 256 + tid) % 65536) * 3;
            qx[i] = q[0] + r * 1e-3f; qy[i] = q[1]; qz[i] = q[2];
        }
        if (PK) nnf_scan<NQ>(sdb, NPTS / NNF_CH, qx, qy, qz, m1, m2, m3, c1, c2);
        else scan_scalar<NQ>(sdb, NPTS / NNF_CH, qx, qy, qz, m1, m2, m3, c1, c2);
#pragma unroll
        for (int i = 0; i < NQ; ++i) acc += m1[i] + m2[i] + m3[i] + c1[i] + c2[i];
    }
    out[(size_t)b * 256 + tid] = acc;
}

int main() {
    const int NB = 4096, REPS = 2;
    std::vector<float> hp(NPTS * 3), hq(65536 * 3);
    srand(1);
    for (auto& v : hp) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    for (auto& v : hq) v = (rand() / (float)RAND_MAX - 0.5f) * 0.4f;
    float *p, *q, *o;
    CK(hipMalloc(&p, hp.size() * 4)); CK(hipMalloc(&q, hq.size() * 4)); CK(hipMalloc(&o, (size_t)NB * 256 * 4));
    CK(hipMemcpy(p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char* name; void (*k)(const float*, const float*, float*, int); int nq; };
    V vs[] = {{"pk     NQ=4 occ 3", scan_kernel<true, 4, 3>, 4},  {"scalar NQ=4 occ 3", scan_kernel<false, 4, 3>, 4},
              {"pk     NQ=4 occ 2", scan_kernel<true, 4, 2>, 4},  {"scalar NQ=4 occ 2", scan_kernel<false, 4, 2>, 4},
              {"pk     NQ=4 occ 1", scan_kernel<true, 4, 1>, 4},  {"scalar NQ=4 occ 1", scan_kernel<false, 4, 1>, 4},
              {"pk     NQ=2 occ 4", scan_kernel<true, 2, 4>, 2},  {"scalar NQ=2 occ 4", scan_kernel<false, 2, 4>, 2},
              {"scalar NQ=6 occ 2", scan_kernel<false, 6, 2>, 6}, {"scalar NQ=8 occ 2", scan_kernel<false, 8, 2>, 8}};
    for (int round = 0; round < 2; ++round)
        for (auto& v : vs) {
            v.k<<<NB, 256>>>(p, q, o, REPS);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 5; ++i) v.k<<<NB, 256>>>(p, q, o, REPS);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 5;
            const double pairs = (double)NB * 256 * v.nq * NPTS * REPS;
            printf("%s  %.3f ms  %.2f Tpairs/s\n", v.name, ms, pairs / (ms * 1e-3) / 1e12);
        }
    CK(hipGetLastError());
    return 0;
}
