// Brute-force nearest-neighbour scan primitives shared by chamfer.hip and grasp_loss.hip (gfx950).
//
// FP32-VALU bound, so the inner loop is built to issue as few vector instructions per (query, point) pair as the
// exact result allows:
//   * the database cloud sits in LDS in PAIRS, pair p = {x0,x1,y0,y1,z0,z1} (6 floats), so that three broadcast
//     ds_read_b128 bring four points with every dword used, and one v_pk_{add,mul,fma}_f32 evaluates a query
//     against two points (the query coordinate is splat by op_sel, no moves);
//   * no per-pair index bookkeeping: points are visited in chunks of NN_CH, a chunk only keeps its running
//     minimum (v_min3_f32, two points per instruction) and one compare/select pair per chunk records the first
//     chunk that holds the overall minimum (strict '<', chunks in index order);
//   * the winning chunk is re-scanned once per tile to recover the lowest index whose distance equals the minimum.
// The result is exactly what a sequential `if (d < best) {best = d; idx = k;}` scan returns (lowest index on
// ties), at 3.5 instead of 11 vector instructions per pair.
//
// Squared distance = fma(dz, dz, fma(dy, dy, dx*dx)) -- the contraction a CUDA compiler applies to the
// extension's `(x2-x1)*(x2-x1)+(y2-y1)*(y2-y1)+(z2-z1)*(z2-z1)`; oracle/csrc/nn_ref.c evaluates the same chain
// with fmaf(), so indices and squared distances are bit-exact against it.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

typedef float nn_f2 __attribute__((ext_vector_type(2)));

#define NN_CH 16          // points per chunk (LDS tiles are padded to a multiple of this with NN_FAR points)
#define NN_FAR 1.0e18f    // pad coordinate: squared distance ~1e36 is finite and never wins against a real point

// floats of LDS needed for a tile of n points (n a multiple of NN_CH)
#define NN_TILE_FLOATS(n) ((n) * 3)

__device__ __forceinline__ nn_f2 nn_dist2_pk(float qx, float qy, float qz, nn_f2 px, nn_f2 py, nn_f2 pz) {
#pragma clang fp contract(off)
    const nn_f2 dx = qx - px, dy = qy - py, dz = qz - pz;
    return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}

__device__ __forceinline__ float nn_dist2(float qx, float qy, float qz, float px, float py, float pz) {
#pragma clang fp contract(off)
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// coordinates of point j of a paired tile
__device__ __forceinline__ void nn_tile_point(const float* __restrict__ s, int j, float& x, float& y, float& z) {
    const float* p = s + (j >> 1) * 6 + (j & 1);
    x = p[0]; y = p[2]; z = p[4];
}

// Scan the LDS tile (points [0, cnt_pad), cnt_pad a multiple of NN_CH; global index of point 0 is `base`) for NQ
// queries held in registers.  best[] / bc[] carry (minimum so far, global index of the first chunk attaining it)
// across tiles.
template <int NQ>
__device__ __forceinline__ void nn_scan_tile(const float* __restrict__ s, int cnt_pad, int base, const float (&qx)[NQ],
                                             const float (&qy)[NQ], const float (&qz)[NQ], float (&best)[NQ], int (&bc)[NQ]) {
    for (int c = 0; c < cnt_pad; c += NN_CH) {
        const float4* t = reinterpret_cast<const float4*>(s + (c >> 1) * 6);    // 16 points = 12 float4
        float m[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) m[i] = INFINITY;
#pragma unroll
        for (int g = 0; g < NN_CH / 4; ++g) {
            const float4 a = t[3 * g], b = t[3 * g + 1], e = t[3 * g + 2];
            const nn_f2 x01 = {a.x, a.y}, y01 = {a.z, a.w}, z01 = {b.x, b.y};
            const nn_f2 x23 = {b.z, b.w}, y23 = {e.x, e.y}, z23 = {e.z, e.w};
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const nn_f2 d01 = nn_dist2_pk(qx[i], qy[i], qz[i], x01, y01, z01);
                const nn_f2 d23 = nn_dist2_pk(qx[i], qy[i], qz[i], x23, y23, z23);
                m[i] = __builtin_fminf(__builtin_fminf(m[i], d01.x), d01.y);
                m[i] = __builtin_fminf(__builtin_fminf(m[i], d23.x), d23.y);
            }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const bool lt = m[i] < best[i];
            best[i] = lt ? m[i] : best[i];
            bc[i] = lt ? (base + c) : bc[i];
        }
    }
}

// For every query whose minimum moved into this tile (bc >= base): lowest index in the winning chunk whose distance
// equals the minimum.  Same arithmetic as the packed scan (v_fma_f32 == one half of v_pk_fma_f32), so a match exists.
template <int NQ>
__device__ __forceinline__ void nn_resolve_tile(const float* __restrict__ s, int base, const float (&qx)[NQ],
                                                const float (&qy)[NQ], const float (&qz)[NQ], const float (&best)[NQ],
                                                const int (&bc)[NQ], int (&bi)[NQ]) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        if (bc[i] < base) continue;
        const float4* t = reinterpret_cast<const float4*>(s + ((bc[i] - base) >> 1) * 6);
        int r = bc[i];
#pragma unroll
        for (int g = NN_CH / 4 - 1; g >= 0; --g) {
            const float4 a = t[3 * g], b = t[3 * g + 1], e = t[3 * g + 2];
            r = (nn_dist2(qx[i], qy[i], qz[i], b.w, e.y, e.w) == best[i]) ? (bc[i] + 4 * g + 3) : r;
            r = (nn_dist2(qx[i], qy[i], qz[i], b.z, e.x, e.z) == best[i]) ? (bc[i] + 4 * g + 2) : r;
            r = (nn_dist2(qx[i], qy[i], qz[i], a.y, a.w, b.y) == best[i]) ? (bc[i] + 4 * g + 1) : r;
            r = (nn_dist2(qx[i], qy[i], qz[i], a.x, a.z, b.x) == best[i]) ? (bc[i] + 4 * g + 0) : r;
        }
        bi[i] = r;
    }
}

// Stage `cnt` points (xyz rows of `src`) into a paired LDS tile, padded to a multiple of NN_CH with far points.
// Returns the padded count.  Caller synchronises.
__device__ __forceinline__ int nn_stage_tile(float* __restrict__ s, const float* __restrict__ src, int cnt, int tid,
                                             int nthreads) {
    const int cnt_pad = (cnt + NN_CH - 1) / NN_CH * NN_CH;
    for (int k = tid; k < cnt_pad; k += nthreads) {
        float x = NN_FAR, y = NN_FAR, z = NN_FAR;
        if (k < cnt) {
            const float* p = src + (size_t)k * 3;
            x = p[0]; y = p[1]; z = p[2];
        }
        float* d = s + (k >> 1) * 6 + (k & 1);
        d[0] = x; d[2] = y; d[4] = z;
    }
    return cnt_pad;
}
