// Row picks for losses that read several (overlapping) vertex sets of one cloud: src [N,U,3] -> out [N,P,3] with
// out[n][i] = src[n][pos[i]], and the backward as a gather over the inverse map -- g_src[n][u] = sum of g_out[n][i] over the
// picks i of u in ascending i: no atomics, no zero-filled full-size temporaries, one pass.  (The autograd form of
// MotionFill/models/fittingop.py:356-409 is one index_select per set: each backward zero-fills and index_adds a full-size
// gradient and autograd then adds them pairwise.)
#include "saga_common.h"

#define PICK_THREADS 256

__global__ __launch_bounds__(PICK_THREADS) void pick_rows_fwd_kernel(const float* __restrict__ src, const int32_t* __restrict__ pos,
                                                                     float* __restrict__ out, int U, int P) {
    const int n = blockIdx.y;
    const float* s = src + (size_t)n * U * 3;
    float* o = out + (size_t)n * P * 3;
    for (int e = blockIdx.x * PICK_THREADS + threadIdx.x; e < P * 3; e += gridDim.x * PICK_THREADS) {
        const int i = e / 3, c = e - i * 3;
        o[e] = s[pos[i] * 3 + c];
    }
}

__global__ __launch_bounds__(PICK_THREADS) void pick_rows_bwd_kernel(const float* __restrict__ g_out, const int32_t* __restrict__ inv_ptr,
                                                                     const int32_t* __restrict__ inv_idx, float* __restrict__ g_src,
                                                                     int U, int P) {
    const int n = blockIdx.y;
    const float* g = g_out + (size_t)n * P * 3;
    float* o = g_src + (size_t)n * U * 3;
    for (int e = blockIdx.x * PICK_THREADS + threadIdx.x; e < U * 3; e += gridDim.x * PICK_THREADS) {
        const int u = e / 3, c = e - u * 3;
        float acc = 0.f;
        for (int k = inv_ptr[u]; k < inv_ptr[u + 1]; ++k) acc += g[inv_idx[k] * 3 + c];
        o[e] = acc;
    }
}

extern "C" int saga_pick_rows_fwd(const float* src, const int32_t* pos, float* out, int N, int U, int P, void* stream) {
    SAGA_CHECK_ARG(src && pos && out, "saga_pick_rows_fwd: null argument");
    SAGA_CHECK_ARG(N >= 0 && U > 0 && P > 0 && N <= 65535, "saga_pick_rows_fwd: bad sizes N=%d U=%d P=%d", N, U, P);
    if (N == 0) return 0;
    dim3 grid(std::min(8, (P * 3 + PICK_THREADS - 1) / PICK_THREADS), N);
    pick_rows_fwd_kernel<<<grid, PICK_THREADS, 0, (hipStream_t)stream>>>(src, pos, out, U, P);
    SAGA_LAUNCH_CHECK();
    return 0;
}

extern "C" int saga_pick_rows_bwd(const float* g_out, const int32_t* inv_ptr, const int32_t* inv_idx, float* g_src, int N, int U, int P,
                                  void* stream) {
    SAGA_CHECK_ARG(g_out && inv_ptr && inv_idx && g_src, "saga_pick_rows_bwd: null argument");
    SAGA_CHECK_ARG(N >= 0 && U > 0 && P > 0 && N <= 65535, "saga_pick_rows_bwd: bad sizes N=%d U=%d P=%d", N, U, P);
    if (N == 0) return 0;
    dim3 grid(std::min(8, (U * 3 + PICK_THREADS - 1) / PICK_THREADS), N);
    pick_rows_bwd_kernel<<<grid, PICK_THREADS, 0, (hipStream_t)stream>>>(g_out, inv_ptr, inv_idx, g_src, U, P);
    SAGA_LAUNCH_CHECK();
    return 0;
}
