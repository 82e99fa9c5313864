// A/B micro-benchmark: approximate pass of the two-level NN scan on the vector units (nnf_scan of nn_fast.h) against the
// same pass on the matrix cores (v_mfma_f32_16x16x4_f32: 16 points x 16 queries x (x,y,z,1) per instruction, per-lane
// chunk minima).  Scan only (no exact pass); random data.  Not part of the library.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas -I saga_amd/csrc scripts/ab/nn_mfma_ab.hip -o build/nn_mfma_ab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "nn_fast.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define NPTS 2048
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- A: vector units -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void scan_valu(const float* __restrict__ pts, const float* __restrict__ qs, float* __restrict__ out, int reps) {
    __shared__ __attribute__((aligned(16))) float sdb[NNF_TILE_FLOATS(NPTS)];
    const int tid = threadIdx.x, b = blockIdx.x;
    NnfFrame f; f.cx = f.cy = f.cz = 0.f; f.rp = 1.f;
    for (int k = tid; k < NPTS; k += 256) nnf_store_point(sdb, k, pts[k * 3], pts[k * 3 + 1], pts[k * 3 + 2], f);
    __syncthreads();
    float qx[4], qy[4], qz[4], m1[4], m2[4], m3[4];
    int c1[4], c2[4];
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* q = qs + ((size_t)(b * 1024 + i * 256 + tid) % 65536) * 3;
            qx[i] = q[0] + r * 1e-3f; qy[i] = q[1]; qz[i] = q[2];
        }
        nnf_scan<4>(sdb, NPTS / NNF_CH, qx, qy, qz, m1, m2, m3, c1, c2);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc += m1[i] + m2[i] + m3[i] + c1[i] + c2[i];
    }
    out[(size_t)b * 256 + tid] = acc;
}

// ---- B: matrix cores -------------------------------------------------------------------------------------------------------
// tile layout: group g (16 points) = 64 floats [coord k = 0..3][point i = 0..15] with (-2x, -2y, -2z, |p|^2): the A operand of
// lane l (row i = l&15, k = l>>4) is s[g*64 + l].  NT query tiles of 16 per wave; lane (j = l&15, rg = l>>4) sees, for query j
// of every tile, points 4*rg .. 4*rg+3 of each group; its chunk = CG consecutive groups.
template <int NT, int CG>
__global__ __launch_bounds__(256, 3) void scan_mfma(const float* __restrict__ pts, const float* __restrict__ qs, float* __restrict__ out, int reps) {
    __shared__ __attribute__((aligned(16))) float sdb[NPTS * 4];
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < NPTS; k += 256) {
        const float x = pts[k * 3], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
        float* d = sdb + (k >> 4) * 64 + (k & 15);
        d[0] = -2.f * x; d[16] = -2.f * y; d[32] = -2.f * z; d[48] = fmaf(z, z, fmaf(y, y, x * x));
    }
    __syncthreads();
    const int kk = lane >> 4, j = lane & 15;
    float acc = 0.f;
    const f32x4 Z = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
        float bq[NT], m1[NT], m2[NT], m3[NT];
        int c1[NT], c2[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float* q = qs + ((size_t)(b * 1024 + wave * (16 * NT) + t * 16 + j) % 65536) * 3;
            bq[t] = kk < 3 ? q[kk] + (kk == 0 ? r * 1e-3f : 0.f) : 1.f;
            m1[t] = m2[t] = m3[t] = INFINITY; c1[t] = c2[t] = -1;
        }
        float a_next = sdb[lane];
        for (int c = 0; c < NPTS / 16 / CG; ++c) {
            float m[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) m[t] = INFINITY;
#pragma unroll
            for (int g = 0; g < CG; ++g) {
                const float a = a_next;
                a_next = sdb[min(c * CG + g + 1, NPTS / 16 - 1) * 64 + lane];
#pragma unroll
                for (int t0 = 0; t0 < NT; t0 += 4) {
                    f32x4 d[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq[t0 + t], Z, 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        m[t0 + t] = __builtin_fminf(__builtin_fminf(m[t0 + t], d[t][0]), d[t][1]);
                        m[t0 + t] = __builtin_fminf(__builtin_fminf(m[t0 + t], d[t][2]), d[t][3]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bool lt1 = m[t] < m1[t], lt2 = m[t] < m2[t];
                m3[t] = __builtin_amdgcn_fmed3f(m[t], m2[t], m3[t]);
                m2[t] = __builtin_amdgcn_fmed3f(m[t], m1[t], m2[t]);
                m1[t] = __builtin_fminf(m[t], m1[t]);
                int n2 = lt2 ? c : c2[t];
                n2 = lt1 ? c1[t] : n2;
                c2[t] = n2;
                c1[t] = lt1 ? c : c1[t];
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc += m1[t] + m2[t] + m3[t] + c1[t] + c2[t];
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
    struct V { const char* name; void (*k)(const float*, const float*, float*, int); int qpb; };
    V vs[] = {{"valu nnf_scan<4>      ", scan_valu, 1024},
              {"mfma NT=16 chunk 4 grp", scan_mfma<16, 4>, 1024},
              {"mfma NT=8  chunk 4 grp", scan_mfma<8, 4>, 512},
              {"mfma NT=16 chunk 8 grp", scan_mfma<16, 8>, 1024},
              {"mfma NT=12 chunk 4 grp", scan_mfma<12, 4>, 768}};
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
            const double pairs = (double)NB * v.qpb * NPTS * REPS;
            printf("%s  %.3f ms  %.2f Tpairs/s\n", v.name, ms, pairs / (ms * 1e-3) / 1e12);
        }
    CK(hipGetLastError());
    return 0;
}
