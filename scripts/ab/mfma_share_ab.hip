// Does v_mfma_f32_32x32x2_f32 keep its 64-cycle issue interval when TWO waves of a SIMD issue it?  One block = 4 waves (one per
// SIMD); 64 KB or 128 KB of LDS per block decides whether one or two blocks share a CU.  Each wave runs `steps` groups of 32
// MFMAs on four accumulators (the k-step of blend_gemm.hip), optionally with a few scalar / vector instructions between groups.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ab/mfma_share_ab.hip -o build/mfma_share_ab
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int EXTRA>
__global__ __launch_bounds__(256) void k(float* out, int steps, long long* cyc) {
    extern __shared__ float lds[];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    int x = threadIdx.x;
    long long t0 = clock64();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        if (EXTRA >= 1) {       // ~28 dependent-free vector instructions per group, as the DMA address arithmetic
#pragma unroll
            for (int e = 0; e < 28; ++e) x = x * 3 + e;
        }
        if (EXTRA >= 2) __builtin_amdgcn_s_barrier();
    }
    long long t1 = clock64();
    float r = 0.f;
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) r += acc[i][q];
    if (r == 123.f || x == 7) out[threadIdx.x] = r + lds[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int EXTRA>
static void run(const char* name, int blocks, int lds_kb, int steps, float* out, long long* cyc) {
    hipFuncSetAttribute((const void*)k<EXTRA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<EXTRA><<<blocks, 256, lds_kb * 1024>>>(out, 8, cyc);
    hipEventRecord(e0);
    k<EXTRA><<<blocks, 256, lds_kb * 1024>>>(out, steps, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double per = (double)h[0] / (steps * 32.0);
    double tf = (double)blocks * 4 * steps * 32 * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-34s blocks %4d lds %3d KB: %.1f ticks per MFMA and wave, %.3f ms, %.1f TFLOP/s\n", name, blocks, lds_kb, per, ms, tf);
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8192 * 8);
    const int steps = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("pure, one block per CU", 256, 128, steps, out, cyc);
        run<0>("pure, two blocks per CU", 512, 64, steps, out, cyc);
        run<1>("+28 VALU, one block per CU", 256, 128, steps, out, cyc);
        run<1>("+28 VALU, two blocks per CU", 512, 64, steps, out, cyc);
        run<2>("+28 VALU + barrier, one per CU", 256, 128, steps, out, cyc);
        run<2>("+28 VALU + barrier, two per CU", 512, 64, steps, out, cyc);
    }
    return 0;
}
