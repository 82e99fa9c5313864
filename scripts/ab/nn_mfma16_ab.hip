// A/B micro-benchmark + error check: approximate NN pass on the vector units (nnf_scan) against the f16-split pass on the
// matrix cores (nnf_scan_mfma, v_mfma_f32_32x32x16_f16).  Reports Tpairs/s of each, how often the best chunk agrees, and the
// largest |m1_mfma - min_p t(p)| (float64 on the host) in units of Rp (Rp + 2|q|), to be compared with NNF_MFMA_KAPPA.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas -I saga_amd/csrc scripts/ab/nn_mfma16_ab.hip -o build/nn_mfma16_ab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include "nn_fast.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define NPTS 2048
#define NQL 4

template <bool MFMA, int OCC>
__global__ __launch_bounds__(256, OCC) void scan_kernel(const float* __restrict__ pts, const float* __restrict__ qs, float* __restrict__ out_m,
                                                        int* __restrict__ out_c, float rp, int reps) {
    __shared__ __attribute__((aligned(16))) float sdb[NNF_TILE_FLOATS(NPTS)];
    const int tid = threadIdx.x, b = blockIdx.x;
    NnfFrame f; f.cx = f.cy = f.cz = 0.f; f.rp = rp;
    for (int k = tid; k < NPTS; k += 256) nnf_store_point(sdb, k, pts[k * 3], pts[k * 3 + 1], pts[k * 3 + 2], f);
    __syncthreads();
    float qx[NQL], qy[NQL], qz[NQL], m1[NQL], m2[NQL], m3[NQL];
    int c1[NQL], c2[NQL];
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < NQL; ++i) {
            const float* q = qs + ((size_t)(b * 1024 + i * 256 + tid) % 65536) * 3;
            qx[i] = q[0]; qy[i] = q[1]; qz[i] = q[2];
        }
        if (MFMA) nnf_scan_mfma<NQL>(sdb, NPTS / NNF_CH, f, qx, qy, qz, m1, m2, m3, c1, c2);
        else nnf_scan<NQL>(sdb, NPTS / NNF_CH, qx, qy, qz, m1, m2, m3, c1, c2);
    }
#pragma unroll
    for (int i = 0; i < NQL; ++i) {
        out_m[((size_t)b * 1024 + i * 256 + tid) * 3] = m1[i];
        out_m[((size_t)b * 1024 + i * 256 + tid) * 3 + 1] = m2[i];
        out_m[((size_t)b * 1024 + i * 256 + tid) * 3 + 2] = m3[i];
        out_c[((size_t)b * 1024 + i * 256 + tid) * 2] = c1[i];
        out_c[((size_t)b * 1024 + i * 256 + tid) * 2 + 1] = c2[i];
    }
}

int main() {
    const int NB = 4096, REPS = 2;
    std::vector<float> hp(NPTS * 3), hq(65536 * 3);
    srand(1);
    for (auto& v : hp) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    for (auto& v : hq) v = (rand() / (float)RAND_MAX - 0.5f) * 0.4f;
    const float rp = 0.0867f;       // >= sqrt(3) * 0.05
    float *p, *q, *om[2];
    int* oc[2];
    CK(hipMalloc(&p, hp.size() * 4)); CK(hipMalloc(&q, hq.size() * 4));
    for (int k = 0; k < 2; ++k) { CK(hipMalloc(&om[k], (size_t)NB * 1024 * 3 * 4)); CK(hipMalloc(&oc[k], (size_t)NB * 1024 * 2 * 4)); }
    CK(hipMemcpy(p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char* name; void (*k)(const float*, const float*, float*, int*, float, int); int slot; };
    V vs[] = {{"valu  occ 3", scan_kernel<false, 3>, 0}, {"mfma16 occ 3", scan_kernel<true, 3>, 1}, {"mfma16 occ 2", scan_kernel<true, 2>, 1},
              {"mfma16 occ 1", scan_kernel<true, 1>, 1}};
    for (int round = 0; round < 2; ++round)
        for (auto& v : vs) {
            v.k<<<NB, 256>>>(p, q, om[v.slot], oc[v.slot], rp, REPS);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 5; ++i) v.k<<<NB, 256>>>(p, q, om[v.slot], oc[v.slot], rp, REPS);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 5;
            const double pairs = (double)NB * 1024 * NPTS * REPS;
            printf("%s  %.3f ms  %.2f Tpairs/s\n", v.name, ms, pairs / (ms * 1e-3) / 1e12);
        }
    CK(hipGetLastError());
    // agreement + error on the first 65536 queries (block b, slot i, lane tid -> query (b*1024 + i*256 + tid) % 65536)
    const size_t NCHK = 65536;
    std::vector<float> m0(NCHK * 3), m1v(NCHK * 3);
    std::vector<int> c0(NCHK * 2), c1v(NCHK * 2);
    CK(hipMemcpy(m0.data(), om[0], NCHK * 3 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(m1v.data(), om[1], NCHK * 3 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c0.data(), oc[0], NCHK * 2 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1v.data(), oc[1], NCHK * 2 * 4, hipMemcpyDeviceToHost));
    size_t same1 = 0, same2 = 0;
    double worst = 0.0, worst_valu = 0.0;
    for (size_t n = 0; n < NCHK; n += 7) {
        const float* qq = &hq[n * 3];
        double tmin = 1e300;
        for (int k = 0; k < NPTS; ++k) {
            const double dx = (double)qq[0] - hp[k * 3], dy = (double)qq[1] - hp[k * 3 + 1], dz = (double)qq[2] - hp[k * 3 + 2];
            const double t = dx * dx + dy * dy + dz * dz - ((double)qq[0] * qq[0] + (double)qq[1] * qq[1] + (double)qq[2] * qq[2]);
            if (t < tmin) tmin = t;
        }
        const double nq = sqrt((double)qq[0] * qq[0] + (double)qq[1] * qq[1] + (double)qq[2] * qq[2]);
        const double unit = (double)rp * (rp + 2.0 * nq);
        worst = fmax(worst, fabs((double)m1v[n * 3] - tmin) / unit);
        worst_valu = fmax(worst_valu, fabs((double)m0[n * 3] - tmin) / unit);
    }
    for (size_t n = 0; n < NCHK; ++n) { same1 += c0[n * 2] == c1v[n * 2]; same2 += c0[n * 2 + 1] == c1v[n * 2 + 1]; }
    printf("best chunk agrees for %zu / %zu queries, second for %zu\n", same1, NCHK, same2);
    printf("max |m1 - t_min| / (Rp (Rp + 2|q|)):  mfma16 %.3e  (kappa %.3e)   valu %.3e\n", worst, (double)NNF_MFMA_KAPPA, worst_valu);
    return 0;
}
