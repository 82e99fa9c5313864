// Does v_mfma_f32_32x32x16_f16 keep f16 SUBNORMAL inputs?  (nn_fast.h relies on it for the low parts of the split operands;
// if they were flushed, the error bound of the approximate pass would need an absolute term.)  Also: is a product of two
// subnormal-adjacent values accumulated exactly, and does the accumulation of 16 products round like fp32 adds?
// build: hipcc -O3 --offload-arch=gfx950 scripts/ab/mfma_f16_denorm.hip -o build/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k(const float* av, const float* bv, float* out) {
    // A row = lane&31, k = 8*(lane>>5)+j ; every row / column gets the same 16 values av[k], bv[k]
    const int h = threadIdx.x >> 5;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)av[8 * h + j]; b[j] = (_Float16)bv[8 * h + j]; }
    f16v z = {0};
    f16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = d[0];
}

int main() {
    float *a, *b, *o;
    hipMalloc(&a, 64); hipMalloc(&b, 64); hipMalloc(&o, 4);
    auto run = [&](const float* ha, const float* hb) {
        hipMemcpy(a, ha, 64, hipMemcpyHostToDevice); hipMemcpy(b, hb, 64, hipMemcpyHostToDevice);
        k<<<1, 64>>>(a, b, o);
        float r; hipMemcpy(&r, o, 4, hipMemcpyDeviceToHost); return r;
    };
    float ha[16] = {0}, hb[16] = {0};
    ha[0] = 3.0e-5f; hb[0] = 1.0f;                      // subnormal f16 (min normal 6.1e-5) times one
    printf("subnormal A x 1      : %.9g (expected ~2.998e-05 if kept, 0 if flushed)\n", run(ha, hb));
    ha[0] = 1.0f; hb[0] = 3.0e-5f;
    printf("1 x subnormal B      : %.9g\n", run(ha, hb));
    ha[0] = 6.0e-8f; hb[0] = 1.0f;                      // the smallest subnormal
    printf("min subnormal A x 1  : %.9g (5.96e-08 if kept)\n", run(ha, hb));
    // accumulation: 1024 + 15 x 2^-14 (each far below half an ulp of 1024 = 6.1e-5): fp32 adds one at a time would lose every
    // one of them; a wide accumulator keeps 15 x 2^-14 = 9.16e-4
    for (int j = 0; j < 16; ++j) { ha[j] = 0.0078125f; hb[j] = 0.0078125f; }   // 2^-7 x 2^-7 = 2^-14
    ha[0] = 32.f; hb[0] = 32.f;                          // 1024
    printf("1024 + 15 x 2^-14    : %.9g (1024.00092 with a wide accumulator, 1024 with sequential fp32 rounding)\n", run(ha, hb));
    return 0;
}
