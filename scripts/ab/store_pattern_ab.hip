// How long does the blend GEMM's epilogue store pattern take by itself?  512 blocks x 4 waves; per "tile" a wave issues 16
// dwordx4 stores.  Pattern 0 (the kernel's): lane&31 -> row (stride ldc floats), lane>>5 -> +4 floats, quad q -> +8 floats:
// 32 rows x 32 B per instruction.  Pattern 1 (whole lines): lane>>4 -> row, lane&15 -> +4 floats: 4 rows x 256 B per instruction.
// Optional MFMA work between tiles (steps x 32 MFMAs) to see whether the stores' issue time hides behind the partner's compute.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ab/store_pattern_ab.hip -o build/store_pattern_ab
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int PAT>
__global__ __launch_bounds__(256) void k(float* C, int ldc, int tiles_m, int n_tiles, int steps, long long* cyc) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = threadIdx.x;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    long long in_epi = 0, t_begin = clock64();
    for (int bx = blockIdx.x; bx < n_tiles; bx += gridDim.x) {
        for (int s = 0; s < steps; ++s)
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        const int m0 = (bx % tiles_m) * 128, n0 = (bx / tiles_m) * 128;
        long long t0 = clock64();
#ifdef EPI_PRIO
        __builtin_amdgcn_s_setprio(EPI_PRIO);
#endif
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    size_t off;
                    if (PAT == 2) {     // the kernel's pattern with NO per-tile vector arithmetic: uniform tile base + per-lane 32-bit offset + immediate
                        const float* base = C + (size_t)m0 * ldc + n0;      // uniform
                        const unsigned voff = ((wm * 64 + (lane & 31)) * ldc + wn * 64 + 4 * (lane >> 5)) * 4 + (unsigned)mt * 32 * ldc * 4;   // loop-invariant
                        *reinterpret_cast<float4*>((char*)base + voff + (nt * 32 + 8 * q) * 4) =
                            make_float4(acc[mt * 2 + nt][4 * q], acc[mt * 2 + nt][4 * q + 1], acc[mt * 2 + nt][4 * q + 2], acc[mt * 2 + nt][4 * q + 3]);
                        continue;
                    }
                    if (PAT == 0) off = (size_t)(m0 + wm * 64 + mt * 32 + (lane & 31)) * ldc + n0 + wn * 64 + nt * 32 + 8 * q + 4 * (lane >> 5);
                    else off = (size_t)(m0 + wm * 64 + mt * 32 + q * 8 + nt * 4 + (lane >> 4)) * ldc + n0 + wn * 64 + (lane & 15) * 4;
                    *reinterpret_cast<float4*>(C + off) = make_float4(acc[mt * 2 + nt][4 * q], acc[mt * 2 + nt][4 * q + 1], acc[mt * 2 + nt][4 * q + 2], acc[mt * 2 + nt][4 * q + 3]);
                }
#ifdef EPI_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        in_epi += clock64() - t0;
    }
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = in_epi; cyc[2 * blockIdx.x + 1] = clock64() - t_begin; }
    if (lds[0] == 123.f) C[0] = 1.f;
}
template <int PAT>
static void run(const char* name, float* C, int steps, long long* cyc, int blocks = 512) {
    const int B = 4096, N = 31488, tiles_m = B / 128, n_tiles = tiles_m * (N / 128);
    hipFuncSetAttribute((const void*)k<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const int lds = blocks > 256 ? 64 * 1024 : 128 * 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<PAT><<<blocks, 256, lds>>>(C, N, tiles_m, n_tiles, steps, cyc);
    hipEventRecord(e0);
    k<PAT><<<blocks, 256, lds>>>(C, N, tiles_m, n_tiles, steps, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double e = 0, t = 0; for (int i = 0; i < blocks; ++i) { e += h[2 * i]; t += h[2 * i + 1]; }
    printf("%-44s steps %2d: %.3f ms (%.2f TB/s written), epilogue %.0f cycles per tile of %.0f per tile\n", name, steps, ms,
           (double)B * N * 4 / (ms * 1e-3) / 1e12, e / blocks / (n_tiles / (double)blocks), t / blocks / (n_tiles / (double)blocks));
}
int main() {
    float* C; long long* cyc;
    hipMalloc(&C, (size_t)4096 * 31488 * 4); hipMalloc(&cyc, 1024 * 8);
    for (int rep = 0; rep < 2; ++rep)
        for (int steps : {0, 8, 32}) {
            run<0>("32 rows x 32 B per instruction (the kernel's)", C, steps, cyc);
            run<1>("4 rows x 256 B per instruction (whole lines)", C, steps, cyc);
            run<2>("32 rows x 32 B, no per-tile vector arithmetic", C, steps, cyc);
            run<2>("the same, ONE block per CU", C, steps, cyc, 256);
        }
    return 0;
}
