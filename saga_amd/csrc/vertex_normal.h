// Area-weighted normal of vertex v (Open3D compute_vertex_normals convention, fittingop.py:158-164): sum of the
// un-normalised cross products of its incident faces in ascending face order, normalised; zero stays zero, NaN -> (0,0,1).
// The incidence list carries the three vertex ids of every face (one 16-B load), and the faces are fetched four at a
// time (ids first, then all 36 coordinates) so that a vertex costs two memory round trips per four faces instead of
// three per face; the sums are still taken face by face, in order.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

__device__ __forceinline__ void saga_vertex_normal(const float* __restrict__ vb, const int32_t* __restrict__ vf_ptr,
                                                   const int32_t* __restrict__ vf_tri, int v, float& ox, float& oy, float& oz) {
#pragma clang fp contract(off)
    float nx = 0.f, ny = 0.f, nz = 0.f;
    const int e1 = vf_ptr[v + 1];
    for (int e = vf_ptr[v]; e < e1; e += 4) {
        int4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = reinterpret_cast<const int4*>(vf_tri)[min(e + u, e1 - 1)];
        float p[4][9];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id[3] = {t[u].x, t[u].y, t[u].z};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                p[u][c * 3] = vb[id[c] * 3]; p[u][c * 3 + 1] = vb[id[c] * 3 + 1]; p[u][c * 3 + 2] = vb[id[c] * 3 + 2];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (e + u >= e1) break;
            const float ax = p[u][3] - p[u][0], ay = p[u][4] - p[u][1], az = p[u][5] - p[u][2];
            const float bx = p[u][6] - p[u][0], by = p[u][7] - p[u][1], bz = p[u][8] - p[u][2];
            nx += ay * bz - az * by;
            ny += az * bx - ax * bz;
            nz += ax * by - ay * bx;
        }
    }
    const float sq = nx * nx + ny * ny + nz * nz;
    if (sq > 0.f) { const float inv = 1.0f / sqrtf(sq); nx *= inv; ny *= inv; nz *= inv; }
    if (isnan(nx)) { nx = 0.f; ny = 0.f; nz = 1.f; }
    ox = nx; oy = ny; oz = nz;
}

// The same sums in pieces, for a caller that holds the mesh (or the part of it the faces touch) in a flat [n,3] array such as LDS
// and walks the incident faces itself, in ascending order: same arithmetic -> same bits as saga_vertex_normal.
__device__ __forceinline__ void saga_face_normal_add(const float* __restrict__ sv, int i0, int i1, int i2, float (&n)[3]) {
#pragma clang fp contract(off)
    const float* p0 = sv + i0 * 3;
    const float* p1 = sv + i1 * 3;
    const float* p2 = sv + i2 * 3;
    const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
    const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
    n[0] += ay * bz - az * by;
    n[1] += az * bx - ax * bz;
    n[2] += ax * by - ay * bx;
}
__device__ __forceinline__ void saga_normal_finish(float (&n)[3]) {
#pragma clang fp contract(off)
    float nx = n[0], ny = n[1], nz = n[2];
    const float sq = nx * nx + ny * ny + nz * nz;
    if (sq > 0.f) { const float inv = 1.0f / sqrtf(sq); nx *= inv; ny *= inv; nz *= inv; }
    if (isnan(nx)) { nx = 0.f; ny = 0.f; nz = 1.f; }
    n[0] = nx; n[1] = ny; n[2] = nz;
}

// NV vertices per lane at once (ids v[k], k < NV; v[k] < 0 = none): the same sums in the same order as saga_vertex_normal, but the
// NV chains of dependent reads (incidence range -> face ids -> corner coordinates) advance together, two faces per vertex and
// step, so a lane waits for 1/NV as many memory round trips.
template <int NV, int FPS = 2>
__device__ __forceinline__ void saga_vertex_normals_multi(const float* __restrict__ vb, const int32_t* __restrict__ vf_ptr,
                                                          const int32_t* __restrict__ vf_tri, const int (&v)[NV], float (&out)[NV][3]) {
#pragma clang fp contract(off)
    int e0[NV], e1[NV];
    float n[NV][3];
    int steps = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int vk = max(v[k], 0);
        e0[k] = vf_ptr[vk]; e1[k] = v[k] >= 0 ? vf_ptr[vk + 1] : e0[k];
        n[k][0] = 0.f; n[k][1] = 0.f; n[k][2] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) steps = max(steps, (e1[k] - e0[k] + FPS - 1) / FPS);
    for (int g = 0; g < steps; ++g) {
        int4 t[NV][FPS];
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int u = 0; u < FPS; ++u) t[k][u] = reinterpret_cast<const int4*>(vf_tri)[max(min(e0[k] + FPS * g + u, e1[k] - 1), 0)];
        float3 p[NV][FPS][3];
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int u = 0; u < FPS; ++u) {
                p[k][u][0] = *reinterpret_cast<const float3*>(vb + (size_t)t[k][u].x * 3);
                p[k][u][1] = *reinterpret_cast<const float3*>(vb + (size_t)t[k][u].y * 3);
                p[k][u][2] = *reinterpret_cast<const float3*>(vb + (size_t)t[k][u].z * 3);
            }
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int u = 0; u < FPS; ++u) {
                if (e0[k] + FPS * g + u >= e1[k]) continue;
                const float ax = p[k][u][1].x - p[k][u][0].x, ay = p[k][u][1].y - p[k][u][0].y, az = p[k][u][1].z - p[k][u][0].z;
                const float bx = p[k][u][2].x - p[k][u][0].x, by = p[k][u][2].y - p[k][u][0].y, bz = p[k][u][2].z - p[k][u][0].z;
                n[k][0] += ay * bz - az * by;
                n[k][1] += az * bx - ax * bz;
                n[k][2] += ax * by - ay * bx;
            }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float nx = n[k][0], ny = n[k][1], nz = n[k][2];
        const float sq = nx * nx + ny * ny + nz * nz;
        if (sq > 0.f) { const float inv = 1.0f / sqrtf(sq); nx *= inv; ny *= inv; nz *= inv; }
        if (isnan(nx)) { nx = 0.f; ny = 0.f; nz = 1.f; }
        out[k][0] = nx; out[k][1] = ny; out[k][2] = nz;
    }
}
