// 6D continuous rotation -> axis-angle with its exact Jacobian (forward-mode duals); shared by rot6d.hip and fit_step.hip.
// Follows RotConverter.rotmat2aa(RotConverter.cont2rotmat(x)) of the reference (utils/utils.py:385-417 -> :223-370).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

struct Dual {
    float v;
    float d[6];
};

__device__ __forceinline__ Dual dconst(float c) {
    Dual r; r.v = c;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = 0.f;
    return r;
}
__device__ __forceinline__ Dual dvar(float x, int k) { Dual r = dconst(x); r.d[k] = 1.f; return r; }
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) {
    Dual r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) {
    Dual r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
__device__ __forceinline__ Dual operator-(const Dual& a) {
    Dual r; r.v = -a.v;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = -a.d[i];
    return r;
}
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) {
    Dual r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
__device__ __forceinline__ Dual operator*(const Dual& a, float s) {
    Dual r; r.v = a.v * s;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * s;
    return r;
}
__device__ __forceinline__ Dual operator/(const Dual& a, const Dual& b) {
    Dual r; const float inv = 1.0f / b.v; r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
__device__ __forceinline__ Dual dsqrt(const Dual& a) {
    Dual r; r.v = sqrtf(a.v); const float h = 0.5f / r.v;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * h;
    return r;
}
// F.normalize: v / max(||v||, 1e-12)
__device__ __forceinline__ void dnormalize(const Dual a[3], Dual out[3]) {
    Dual n = dsqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (n.v < 1e-12f) n = dconst(1e-12f);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = a[i] / n;
}
__device__ __forceinline__ Dual datan2(const Dual& y, const Dual& x) {
    Dual r; r.v = atan2f(y.v, x.v);
    const float inv = 1.0f / (x.v * x.v + y.v * y.v);
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * inv;
    return r;
}

// x[6] -> aa[3], jac[3][6] = d aa / d x
__device__ __forceinline__ void cont6d_to_aa_dual(const float (&x)[6], float (&aa)[3], float (&jac)[18]) {
    // view(-1,3,2): a = column 0 = (x0,x2,x4), b = column 1 = (x1,x3,x5)
    Dual a[3], b[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        a[r] = dvar(x[2 * r], 2 * r);
        b[r] = dvar(x[2 * r + 1], 2 * r + 1);
    }
    Dual b1[3], b2[3], b3[3], u[3];
    dnormalize(a, b1);
    const Dual dot = b1[0] * b[0] + b1[1] * b[1] + b1[2] * b[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) u[r] = b[r] - dot * b1[r];
    dnormalize(u, b2);
    b3[0] = b1[1] * b2[2] - b1[2] * b2[1];
    b3[1] = b1[2] * b2[0] - b1[0] * b2[2];
    b3[2] = b1[0] * b2[1] - b1[1] * b2[0];
    // R[r][c]: columns b1,b2,b3.  The reference's quaternion code indexes t = R^T: t[i][j] = R[j][i]
#define T_(i, j) ((i) == 0 ? b1[j] : ((i) == 1 ? b2[j] : b3[j]))
    const Dual m00 = T_(0, 0), m11 = T_(1, 1), m22 = T_(2, 2);
    const Dual one = dconst(1.0f);
    Dual q[4], t;
    if (m22.v < 1e-6f) {
        if (m00.v > m11.v) {
            t = one + m00 - m11 - m22;
            q[0] = T_(1, 2) - T_(2, 1); q[1] = t; q[2] = T_(0, 1) + T_(1, 0); q[3] = T_(2, 0) + T_(0, 2);
        } else {
            t = one - m00 + m11 - m22;
            q[0] = T_(2, 0) - T_(0, 2); q[1] = T_(0, 1) + T_(1, 0); q[2] = t; q[3] = T_(1, 2) + T_(2, 1);
        }
    } else {
        if (m00.v < -m11.v) {
            t = one - m00 - m11 + m22;
            q[0] = T_(0, 1) - T_(1, 0); q[1] = T_(2, 0) + T_(0, 2); q[2] = T_(1, 2) + T_(2, 1); q[3] = t;
        } else {
            t = one + m00 + m11 + m22;
            q[0] = t; q[1] = T_(1, 2) - T_(2, 1); q[2] = T_(2, 0) - T_(0, 2); q[3] = T_(0, 1) - T_(1, 0);
        }
    }
#undef T_
    const Dual st = dsqrt(t);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = (q[k] / st) * 0.5f;
    // quaternion -> axis-angle (utils.py:350-369)
    const Dual s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    Dual kk;
    if (s2.v > 0.f) {
        const Dual s = dsqrt(s2);
        const Dual two_theta = (q[0].v < 0.f ? datan2(-s, -q[0]) : datan2(s, q[0])) * 2.0f;
        kk = two_theta / s;
    } else {
        kk = dconst(2.0f);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const Dual o = q[c + 1] * kk;
        aa[c] = o.v;
#pragma unroll
        for (int k = 0; k < 6; ++k) jac[c * 6 + k] = o.d[k];
    }
}
