// 6D continuous rotation -> axis-angle, forward value + exact Jacobian in one pass.
// Replaces RotConverter.rotmat2aa(RotConverter.cont2rotmat(x)) (utils/utils.py:385-417 -> :223-370), which the
// reference evaluates every iteration for the global orientation (WholeGraspPose/models/fittingop.py:140) and inside
// VPoser's decoder for the 21 body joints (:142).  In eager PyTorch that chain is ~90 tiny kernels forward and ~200
// backward; here it is one kernel that carries forward-mode derivatives (dual numbers, 6 tangents) through the same
// arithmetic -- Gram-Schmidt, the 4-branch matrix->quaternion map on the transposed matrix, the atan2 quaternion->
// axis-angle map -- and stores d(aa)/d(x6) [3x6] per row; the backward is Jt.g.
#include <math.h>

#include "saga_common.h"

#include "rot6d.h"

__global__ void cont6d_to_aa_kernel(const float* __restrict__ x6, float* __restrict__ aa, float* __restrict__ jac, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[6], a[3], J[18];
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = x6[(size_t)i * 6 + k];
    cont6d_to_aa_dual(x, a, J);
#pragma unroll
    for (int c = 0; c < 3; ++c) aa[(size_t)i * 3 + c] = a[c];
#pragma unroll
    for (int k = 0; k < 18; ++k) jac[(size_t)i * 18 + k] = J[k];
}

__global__ void cont6d_to_aa_bwd_kernel(const float* __restrict__ jac, const float* __restrict__ g_aa, float* __restrict__ g_x6,
                                        int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g0 = g_aa[(size_t)i * 3], g1 = g_aa[(size_t)i * 3 + 1], g2 = g_aa[(size_t)i * 3 + 2];
    const float* J = jac + (size_t)i * 18;
#pragma unroll
    for (int k = 0; k < 6; ++k) g_x6[(size_t)i * 6 + k] = (J[k] * g0 + J[6 + k] * g1) + J[12 + k] * g2;
}

extern "C" int saga_cont6d_to_aa_fwd(const float* x6, float* aa, float* jac, int N, void* stream) {
    SAGA_CHECK_ARG(N >= 0 && (N == 0 || (x6 && aa && jac)), "saga_cont6d_to_aa_fwd: bad arguments");
    if (N == 0) return 0;
    cont6d_to_aa_kernel<<<(N + 127) / 128, 128, 0, (hipStream_t)stream>>>(x6, aa, jac, N);
    SAGA_LAUNCH_CHECK();
    return 0;
}

extern "C" int saga_cont6d_to_aa_bwd(const float* jac, const float* g_aa, float* g_x6, int N, void* stream) {
    SAGA_CHECK_ARG(N >= 0 && (N == 0 || (jac && g_aa && g_x6)), "saga_cont6d_to_aa_bwd: bad arguments");
    if (N == 0) return 0;
    cont6d_to_aa_bwd_kernel<<<(N + 127) / 128, 128, 0, (hipStream_t)stream>>>(jac, g_aa, g_x6, N);
    SAGA_LAUNCH_CHECK();
    return 0;
}
