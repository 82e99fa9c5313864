// A/B harness for the blend GEMM variants (one process, interleaved rounds, random data). Not part of the library.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I saga_amd/csrc scripts/ab/gemm_ab.hip -o build/gemm_ab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
void saga_set_error(const char*, ...) {}
#include "blend_gemm.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef void (*launch_t)(const float*, const float*, float*, const float*, int, int, int, int, hipStream_t);

template <int EPI, int BKT, int NST, bool LEPI, int XCD>
void launch2(const float* A, const float* Bt, float* C, const float* bias, int M, int N, int K, int splitk, hipStream_t st) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    const int kchunk = splitk == 1 ? K : ((K + splitk - 1) / splitk + 31) / 32 * 32;
    dim3 grid(tiles_m * tiles_n, 1, splitk);
    gemm_nt_f32_kernel<EPI, BKT, NST, 2, true, LEPI><<<grid, 256, 0, st>>>(A, K, Bt, K, C, N, bias, nullptr, M, N, K, kchunk, tiles_m,
                                                                         splitk == 1 ? XCD : (XCD ? 2 : 0));
}

template <int EPI, int BKT, int NST, bool TRS = true>
void launch(const float* A, const float* Bt, float* C, const float* bias, int M, int N, int K, int splitk, hipStream_t st) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    const int kchunk = splitk == 1 ? K : ((K + splitk - 1) / splitk + 31) / 32 * 32;
    dim3 grid(tiles_m * tiles_n, 1, splitk);
    gemm_nt_f32_kernel<EPI, BKT, NST, 2, TRS><<<grid, 256, 0, st>>>(A, K, Bt, K, C, N, bias, nullptr, M, N, K, kchunk, tiles_m);
}

struct Var { const char* name; launch_t f0; launch_t f1; };

int main() {
    const int M = 4096, NF = 31488, KF = 512;
    std::vector<float> hA((size_t)M * KF), hB((size_t)NF * KF), hbias(NF);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hB) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hbias) v = (rand() / (float)RAND_MAX);
    float *A, *Bt, *C, *bias, *G, *Bk, *slab, *Cref;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&Bt, hB.size() * 4)); CK(hipMalloc(&bias, NF * 4));
    CK(hipMalloc(&C, (size_t)M * NF * 4)); CK(hipMalloc(&Cref, (size_t)M * NF * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Bt, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hbias.data(), NF * 4, hipMemcpyHostToDevice));
    // backward shape: G [M, NF] (reuse Cref after check) , Bk [512, NF] (reuse Bt memory reinterpretation: same size)
    G = Cref; Bk = Bt;
    CK(hipMalloc(&slab, (size_t)16 * M * 512 * 4));
    Var vars[] = {
        {"st4_dword", launch<0, 16, 4, false>, launch<1, 16, 4, false>},
        {"st4_x4", launch<0, 16, 4, true>, launch<1, 16, 4, true>},
        {"st4_xcd", launch2<0, 16, 4, false, 1>, launch2<1, 16, 5, false, 1>},
        {"st4_xcd_lepi", launch2<0, 16, 4, true, 1>, launch2<1, 16, 5, false, 1>},
        {"st4_lepi", launch2<0, 16, 4, true, 0>, launch2<1, 16, 5, false, 0>},
        {"st5_dword", launch<0, 16, 5, false>, launch<1, 16, 5, false>},
        {"st5_x4", launch<0, 16, 5, true>, launch<1, 16, 5, true>},
        {"st3_x4", launch<0, 16, 3, true>, launch<1, 16, 3, true>},
        {"st6_x4", launch<0, 16, 6, true>, launch<1, 16, 6, true>},
    };
    const int NV = sizeof(vars) / sizeof(vars[0]);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // correctness: every variant equals variant 0 bitwise (same k order)
    vars[0].f0(A, Bt, Cref, bias, M, NF, KF, 1, 0); CK(hipDeviceSynchronize());
    std::vector<float> r0((size_t)M * 64), r1((size_t)M * 64);
    // spot check vs host for a few entries
    {
        std::vector<float> row(NF);
        CK(hipMemcpy(row.data(), Cref + (size_t)1234 * NF, NF * 4, hipMemcpyDeviceToHost));
        double maxe = 0;
        for (int n = 0; n < NF; n += 997) {
            double s = hbias[n];
            for (int k = 0; k < KF; ++k) s += (double)hA[(size_t)1234 * KF + k] * hB[(size_t)n * KF + k];
            maxe = std::max(maxe, fabs(s - row[n]));
        }
        printf("host check max err %.3e\n", maxe);
    }
    for (int v = 1; v < NV; ++v) {
        CK(hipMemset(C, 0, (size_t)M * NF * 4));
        vars[v].f0(A, Bt, C, bias, M, NF, KF, 1, 0); CK(hipDeviceSynchronize());
        // compare a strided sample on the host
        size_t bad = 0;
        std::vector<float> a(NF), b(NF);
        for (int r = 0; r < M; r += 257) {
            CK(hipMemcpy(a.data(), Cref + (size_t)r * NF, NF * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), C + (size_t)r * NF, NF * 4, hipMemcpyDeviceToHost));
            for (int n = 0; n < NF; ++n) bad += a[n] != b[n];
        }
        printf("%s fwd mismatches vs %s: %zu\n", vars[v].name, vars[0].name, bad);
    }
    const int R = 3, IT = 10;
    for (int pass = 0; pass < 2; ++pass) {
        printf("== %s ==\n", pass == 0 ? "forward  M4096 N31488 K512" : "backward M4096 N512 K31488 splitK16");
        std::vector<std::vector<float>> ms(NV);
        for (int r = 0; r < R; ++r)
            for (int v = 0; v < NV; ++v) {
                auto run = [&]() {
                    if (pass == 0) vars[v].f0(A, Bt, C, bias, M, NF, KF, 1, 0);
                    else vars[v].f1(G, Bk, slab, nullptr, M, 512, NF, 16, 0);
                };
                run(); run();
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < IT; ++i) run();
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                ms[v].push_back(t / IT);
            }
        for (int v = 0; v < NV; ++v) {
            std::sort(ms[v].begin(), ms[v].end());
            const double fl = 2.0 * M * NF * KF;
            printf("%-10s median %.4f ms  min %.4f ms  %.1f TFLOP/s\n", vars[v].name, ms[v][R / 2], ms[v][0], fl / (ms[v][R / 2] * 1e-3) / 1e12);
        }
    }
    CK(hipGetLastError());
    return 0;
}
