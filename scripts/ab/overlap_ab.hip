// Experiment: can an MFMA-bound kernel (blend GEMM, forward shape) and a VALU-bound kernel (approximate NN scan) share the
// chip?  Times each alone, both back to back on one stream, and both on two streams, for GEMM variants whose LDS
// footprint does (3 stages = 48 KB) or does not (4 stages = 64 KB, two blocks per CU) leave room for scan blocks (35 KB).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas -I saga_amd/csrc scripts/ab/overlap_ab.hip -o build/overlap_ab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
void saga_set_error(const char*, ...) {}
#include "blend_gemm.hip"
#include "nn_fast.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define NPTS 2048

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

template <int NST>
void gemm(const float* A, const float* Bt, float* C, const float* bias, int M, int N, int K, hipStream_t st) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    gemm_nt_f32_kernel<0, 16, NST><<<dim3(tiles_m * tiles_n), 256, 0, st>>>(A, K, Bt, K, C, N, bias, nullptr, M, N, K, K, tiles_m);
}

int main() {
    const int M = 4096, NF = 31488, KF = 512, NB = 4096;
    std::vector<float> hA((size_t)M * KF), hB((size_t)NF * KF), hp(NPTS * 3), hq(65536 * 3);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hB) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hp) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    for (auto& v : hq) v = (rand() / (float)RAND_MAX - 0.5f) * 0.4f;
    float *A, *Bt, *C, *p, *q, *o;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&Bt, hB.size() * 4)); CK(hipMalloc(&C, (size_t)M * NF * 4));
    CK(hipMalloc(&p, hp.size() * 4)); CK(hipMalloc(&q, hq.size() * 4)); CK(hipMalloc(&o, (size_t)NB * 256 * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Bt, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    const int IT = 10;
    auto timeit = [&](auto fn) {
        fn(); CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < IT; ++i) fn();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / IT;
    };
    for (int nst = 4; nst >= 3; --nst) {
        auto g = [&](hipStream_t s) { if (nst == 4) gemm<4>(A, Bt, C, nullptr, M, NF, KF, s); else gemm<3>(A, Bt, C, nullptr, M, NF, KF, s); };
        auto v = [&](hipStream_t s) { scan_valu<<<NB, 256, 0, s>>>(p, q, o, 1); };
        const float tg = timeit([&] { g(s0); }), tv = timeit([&] { v(s0); });
        const float serial = timeit([&] { g(s0); v(s0); });
        const float both = timeit([&] { g(s0); v(s1); });
        printf("GEMM %d stages (%d KB LDS): gemm %.3f ms, scan %.3f ms, one stream %.3f ms, two streams %.3f ms\n", nst, nst * 16, tg, tv, serial, both);
    }
    CK(hipGetLastError());
    return 0;
}
