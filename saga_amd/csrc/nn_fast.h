// Two-level exact nearest-neighbour scan (gfx950): a cheap approximate pass picks the candidates, an exact pass decides.
//
// The exact squared distance of nn_scan.h costs 6 vector operations per (query, point) pair (3 subtractions, 1 multiply,
// 2 fmas); the scan is FP32-VALU bound, so that count IS the run time.  Here the database tile is moved into a frame
// centred on its own bounding box and stored as (-2x, -2y, -2z, |p|^2); then
//        t(p) = fma(qz, -2z, fma(qy, -2y, fma(qx, -2x, |p|^2)))  =  |q - p|^2 - |q|^2  + rounding
// costs 3 fmas per pair (two pairs per v_pk_fma_f32), and |q|^2 is the same for every p so argmin t = argmin distance
// up to the rounding error, for which a rigorous bound E is carried (below).  Points are visited in chunks of NNF_CH; a
// query keeps the three smallest chunk minima (m1 <= m2 <= m3) and the chunks (c1, c2) of the first two.
//   * every point whose EXACT distance could be the minimum lies in a chunk with chunk-minimum <= m1 + margin;
//   * if m3 > m1 + margin those chunks are c1 and c2 only: their 2 x NNF_CH points are re-evaluated with the exact arithmetic
//     of nn_scan.h (original coordinates, index order, strict '<') -- bit-identical to the sequential exact scan;
//   * otherwise (three or more chunks inside the margin: ~1e-3 of the queries) the query takes a slow path that
//     re-evaluates every chunk inside the margin exactly.
// So the result is ALWAYS that of the exact scan (lowest index among the exact minima), never an approximation.
// Cost split measured on the fused loss kernel (B = 4096, 1.46 ms): approximate pass 1.24 ms, exact pass of the two
// candidate chunks 0.16 ms, slow path 0.06 ms (a wave runs it when any of its 64 lanes needs it).
//
// Frame: per axis c = the bounding box's midpoint if the box satisfies hi <= 2 lo (same sign), else 0.  Then p/2 <= c <=
// 2p for every point, so p - c is exact (Sterbenz) and p = fma(-0.5, -2(p - c), c) recovers the original coordinate
// bit-for-bit from the LDS tile -- the exact pass needs no second copy of the cloud.
//
// Error bound, u = 2^-24, pc = p - c (exact), qc = fl(q - c), Rp >= |pc|, Rq = |qc|:
//   |p|^2 chain: 3u Rp^2;  qc vs q - c: 2u Rq Rp;  three fma roundings: 3u (Rp^2 + 2 Rq Rp)   =>  |t - S| <= 8u Rp (Rp + Rq)
// with S = |q - p|^2 - |q - c|^2 in real arithmetic; E = 10u Rp (Rp + Rq) leaves slack for evaluating E itself.
// The oracle's fp32 distance d_o = D (1 + theta), |theta| <= 6u.  If p* is the exact-scan winner and p1 the approximate
// one: t(p*) <= t(p1) + 2E + 13u D(p1), D(p1) <= t(p1) + E + |q - c|^2  =>  margin = 2E + 1e-6 (max(m1 + |qc|^2, 0) + E).
#pragma once
#include "nn_scan.h"

#ifndef NNF_CH
#define NNF_CH 16         // points per chunk (a multiple of 8).  Measured on the fused loss kernel: 16 -> 1.47 ms, 32 -> 1.64 ms (the
                          // sorted insert is amortised over more pairs, but the exact pass re-evaluates 2 x NNF_CH points)
#endif
#define NNF_CSTRIDE (NNF_CH * 4 + 4)   // floats per chunk: NNF_CH/2 point pairs x {x0,x1,y0,y1,z0,z1,w0,w1} + 4 pad (per-lane
                          // chunk reads of the exact pass then start on different bank groups instead of one)
#define NNF_TILE_FLOATS(n) (((n) + NNF_CH - 1) / NNF_CH * NNF_CSTRIDE)

struct NnfFrame { float cx, cy, cz, rp; };

__device__ __forceinline__ float nnf_axis_centre(float lo, float hi) {
    if (lo > 0.f && hi <= 2.f * lo) return 0.5f * (lo + hi);
    if (hi < 0.f && lo >= 2.f * hi) return 0.5f * (lo + hi);
    return 0.f;
}

// Block-wide bounding box -> frame.  lo/hi: this thread's partial box (+INF/-INF when it holds no point).
// sred: >= 6 * (nthreads/64) floats of LDS.  Contains barriers: every thread of the block calls it.
__device__ __forceinline__ NnfFrame nnf_frame_from_box(float (&lo)[3], float (&hi)[3], float* sred, int tid, int nthreads) {
    const int wave = tid >> 6, lane = tid & 63, nw = nthreads >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64));
        }
    }
    __syncthreads();                     // sred may still be read by an earlier call
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { sred[wave * 6 + k] = lo[k]; sred[wave * 6 + 3 + k] = hi[k]; }
    }
    __syncthreads();
    float c[3], r2 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float l = sred[k], h = sred[3 + k];
        for (int w = 1; w < nw; ++w) { l = fminf(l, sred[w * 6 + k]); h = fmaxf(h, sred[w * 6 + 3 + k]); }
        c[k] = nnf_axis_centre(l, h);
        const float r = fmaxf(h - c[k], c[k] - l);
        r2 = fmaf(r, r, r2);
    }
    NnfFrame f;
    f.cx = c[0]; f.cy = c[1]; f.cz = c[2];
    f.rp = sqrtf(r2) * 1.0001f;
    return f;
}

// bounding box of cnt points of a global cloud (thread tid of nthreads takes every nthreads-th point)
__device__ __forceinline__ NnfFrame nnf_frame(const float* __restrict__ src, int cnt, float* sred, int tid, int nthreads) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = tid; k < cnt; k += nthreads) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = src[(size_t)k * 3 + a];
            lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v);
        }
    }
    return nnf_frame_from_box(lo, hi, sred, tid, nthreads);
}

__device__ __forceinline__ void nnf_store_point(float* __restrict__ s, int k, float x, float y, float z, const NnfFrame& f) {
    const float px = x - f.cx, py = y - f.cy, pz = z - f.cz;       // exact for points inside the frame's box
    const float w = __builtin_fmaf(pz, pz, __builtin_fmaf(py, py, px * px));
    float* d = s + (k / NNF_CH) * NNF_CSTRIDE + ((k % NNF_CH) >> 1) * 8 + (k & 1);
    d[0] = -2.f * px; d[2] = -2.f * py; d[4] = -2.f * pz; d[6] = w;
}

// Stage cnt points (xyz rows) as a framed tile padded to a multiple of NNF_CH with far points.  Caller synchronises.
__device__ __forceinline__ int nnf_stage_tile(float* __restrict__ s, const float* __restrict__ src, int cnt, const NnfFrame& f,
                                              int tid, int nthreads) {
    const int cnt_pad = (cnt + NNF_CH - 1) / NNF_CH * NNF_CH;
    for (int k = tid; k < cnt_pad; k += nthreads) {
        float x = NN_FAR, y = NN_FAR, z = NN_FAR;
        if (k < cnt) { x = src[(size_t)k * 3]; y = src[(size_t)k * 3 + 1]; z = src[(size_t)k * 3 + 2]; }
        nnf_store_point(s, k, x, y, z, f);
    }
    return cnt_pad;
}

// original coordinates of point j of a framed tile (bit-exact)
__device__ __forceinline__ void nnf_tile_point(const float* __restrict__ s, const NnfFrame& f, int j, float& x, float& y, float& z) {
    const float* p = s + (j / NNF_CH) * NNF_CSTRIDE + ((j % NNF_CH) >> 1) * 8 + (j & 1);
    x = __builtin_fmaf(-0.5f, p[0], f.cx); y = __builtin_fmaf(-0.5f, p[2], f.cy); z = __builtin_fmaf(-0.5f, p[4], f.cz);
}

// Approximate pass: NQ framed queries (qc = q - c) against n_chunks chunks.
template <int NQ>
__device__ __forceinline__ void nnf_scan(const float* __restrict__ s, int n_chunks, const float (&qx)[NQ], const float (&qy)[NQ],
                                         const float (&qz)[NQ], float (&m1)[NQ], float (&m2)[NQ], float (&m3)[NQ], int (&c1)[NQ],
                                         int (&c2)[NQ]) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) { m1[i] = INFINITY; m2[i] = INFINITY; m3[i] = INFINITY; c1[i] = -1; c2[i] = -1; }
    // One step = two point pairs (4 points, 4 broadcast ds_read_b128) against the NQ queries: 2*NQ independent chains of
    // three v_pk_fma_f32, issued stage by stage.  Steps are software-pipelined through two register sets (A, B): the
    // reads of step g+1 are in flight while step g computes; a chunk is an even number of steps (A,B,...,A,B), so the set that holds the
    // next chunk's first step at the loop's back edge is A again and no register moves are needed.
    auto load4 = [&](float4 (&r)[4], int c, int g) {
        const float4* t = reinterpret_cast<const float4*>(s + c * NNF_CSTRIDE) + 4 * g;
        r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; r[3] = t[3];
    };
    float m[NQ];
    auto step = [&](const float4 (&r)[4]) {
        const nn_f2 X0 = {r[0].x, r[0].y}, Y0 = {r[0].z, r[0].w}, Z0 = {r[1].x, r[1].y}, W0 = {r[1].z, r[1].w};
        const nn_f2 X1 = {r[2].x, r[2].y}, Y1 = {r[2].z, r[2].w}, Z1 = {r[3].x, r[3].y}, W1 = {r[3].z, r[3].w};
        nn_f2 v0[NQ], v1[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const nn_f2 sx = {qx[i], qx[i]};
            v0[i] = __builtin_elementwise_fma(sx, X0, W0);
            v1[i] = __builtin_elementwise_fma(sx, X1, W1);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const nn_f2 sy = {qy[i], qy[i]};
            v0[i] = __builtin_elementwise_fma(sy, Y0, v0[i]);
            v1[i] = __builtin_elementwise_fma(sy, Y1, v1[i]);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const nn_f2 sz = {qz[i], qz[i]};
            v0[i] = __builtin_elementwise_fma(sz, Z0, v0[i]);
            v1[i] = __builtin_elementwise_fma(sz, Z1, v1[i]);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            m[i] = __builtin_fminf(__builtin_fminf(m[i], v0[i].x), v0[i].y);
            m[i] = __builtin_fminf(__builtin_fminf(m[i], v1[i].x), v1[i].y);
        }
    };
    float4 A[4], B[4];
    load4(A, 0, 0);
    for (int c = 0; c < n_chunks; ++c) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) m[i] = INFINITY;
#pragma unroll
        for (int g = 0; g < NNF_CH / 4; g += 2) {
            load4(B, c, g + 1);
            step(A);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < NNF_CH / 4) load4(A, c, g + 2);
            else load4(A, min(c + 1, n_chunks - 1), 0);
            step(B);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            // sorted insert of m into (m1 <= m2 <= m3), chunks of the first two (selects, no branches)
            const bool lt1 = m[i] < m1[i], lt2 = m[i] < m2[i];
            m3[i] = __builtin_amdgcn_fmed3f(m[i], m2[i], m3[i]);
            m2[i] = __builtin_amdgcn_fmed3f(m[i], m1[i], m2[i]);
            m1[i] = __builtin_fminf(m[i], m1[i]);
            int n2 = lt2 ? c : c2[i];
            n2 = lt1 ? c1[i] : n2;
            c2[i] = n2;
            c1[i] = lt1 ? c : c1[i];
        }
    }
}

// exact evaluation of one chunk for one query (original coordinates o*), index order, strict '<'.  Two points per packed
// instruction; every packed lane performs the IEEE operation of nn_dist2, so the values are those of the scalar chain.
__device__ __forceinline__ void nnf_exact_chunk(const float* __restrict__ s, const NnfFrame& f, int ch, int base, float ox, float oy,
                                                float oz, float& best, int& bi) {
#pragma clang fp contract(off)
    const float4* t = reinterpret_cast<const float4*>(s + ch * NNF_CSTRIDE);
    const nn_f2 mh = {-0.5f, -0.5f}, cx = {f.cx, f.cx}, cy = {f.cy, f.cy}, cz = {f.cz, f.cz};
    const nn_f2 qx = {ox, ox}, qy = {oy, oy}, qz = {oz, oz};
#pragma unroll
    for (int g = 0; g < NNF_CH / 2; ++g) {
        const float4 a = t[2 * g], b = t[2 * g + 1];
        const nn_f2 X = {a.x, a.y}, Y = {a.z, a.w}, Z = {b.x, b.y};
        const nn_f2 px = __builtin_elementwise_fma(mh, X, cx), py = __builtin_elementwise_fma(mh, Y, cy),
                    pz = __builtin_elementwise_fma(mh, Z, cz);                          // original coordinates, bit-exact
        const nn_f2 dx = qx - px, dy = qy - py, dz = qz - pz;
        const nn_f2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        const int k = base + ch * NNF_CH + 2 * g;
        if (d.x < best) { best = d.x; bi = k; }
        if (d.y < best) { best = d.y; bi = k + 1; }
    }
}

// Exact pass.  qc*: framed queries (for the margin), o*: original query coordinates.  best/bi (IN/OUT; the caller
// starts them at +INF / 0): exact squared distance and index of the sequential exact scan over every tile so far
// (global index of point 0 of this tile = base; tiles must come in index order).  Every lane of the wave must call it
// together (the slow path is a wave-wide loop).
template <int NQ>
__device__ __forceinline__ void nnf_resolve(const float* __restrict__ s, int n_chunks, const NnfFrame& f, int base,
                                            const float (&qx)[NQ], const float (&qy)[NQ], const float (&qz)[NQ],
                                            const float (&ox)[NQ], const float (&oy)[NQ], const float (&oz)[NQ],
                                            const float (&m1)[NQ], const float (&m2)[NQ], const float (&m3)[NQ],
                                            const int (&c1)[NQ], const int (&c2)[NQ],
                                            float (&best)[NQ], int (&bi)[NQ]) {
    const int lane = __lane_id();
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const float qq = __builtin_fmaf(qz[i], qz[i], __builtin_fmaf(qy[i], qy[i], qx[i] * qx[i])) * 1.001f;
        const float E = 5.9604645e-7f * f.rp * (f.rp + sqrtf(qq));                       // 10 u Rp (Rp + Rq)
        const float thr = m1[i] + (2.f * E + 1.0e-6f * (fmaxf(m1[i] + qq, 0.f) + E));
        // a tile whose best possible distance cannot beat the exact minimum carried in from earlier tiles is skipped:
        // D(p) >= t(p) - E + |q - c|^2 (1 - 3u) for every point of this tile, and its fp32 distance is >= D (1 - 6u)
        const bool live = !(m1[i] - E + qq * 0.998f > best[i] * 1.000001f);
        const bool fast = live && m3[i] > thr;
        if (fast) {
            // the exact winner is in chunk c1 or c2: lower chunk first so that ties keep the lowest index
            // (c2 only matters when its minimum is inside the margin)
            const bool two = c2[i] >= 0 && m2[i] <= thr;
            const int a = two ? min(c1[i], c2[i]) : c1[i];
            const int b = two ? max(c1[i], c2[i]) : c1[i];
            nnf_exact_chunk(s, f, a, base, ox[i], oy[i], oz[i], best[i], bi[i]);
            if (b != a) nnf_exact_chunk(s, f, b, base, ox[i], oy[i], oz[i], best[i], bi[i]);
        }
        // slow path (three or more chunks inside the margin), one query at a time with the whole wave: lane l takes
        // chunks l, l+64, ..., re-evaluates those whose approximate minimum is inside the margin exactly (ascending, strict
        // '<'), and the lanes' (distance, index) pairs are reduced lexicographically -- the lowest index among the exact
        // minima of the tile, as a sequential scan finds it.  (Per lane this loop would cost a third of a whole scan each
        // time one of 64 lanes needs it.)
        unsigned long long todo = __ballot(live && !fast);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const float sqx = __shfl(qx[i], src, 64), sqy = __shfl(qy[i], src, 64), sqz = __shfl(qz[i], src, 64);
            const float sox = __shfl(ox[i], src, 64), soy = __shfl(oy[i], src, 64), soz = __shfl(oz[i], src, 64);
            const float sthr = __shfl(thr, src, 64);
            float cb = INFINITY;
            int ci = 0x7fffffff;
            for (int c = lane; c < n_chunks; c += 64) {
                const float4* t = reinterpret_cast<const float4*>(s + c * NNF_CSTRIDE);
                float m = INFINITY;
#pragma unroll
                for (int g = 0; g < NNF_CH / 2; ++g) {
                    const float4 a = t[2 * g], b = t[2 * g + 1];
                    const float v0 = __builtin_fmaf(sqz, b.x, __builtin_fmaf(sqy, a.z, __builtin_fmaf(sqx, a.x, b.z)));
                    const float v1 = __builtin_fmaf(sqz, b.y, __builtin_fmaf(sqy, a.w, __builtin_fmaf(sqx, a.y, b.w)));
                    m = __builtin_fminf(__builtin_fminf(m, v0), v1);
                }
                if (m <= sthr) nnf_exact_chunk(s, f, c, base, sox, soy, soz, cb, ci);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(cb, o, 64);
                const int oi = __shfl_xor(ci, o, 64);
                const bool take = ob < cb || (ob == cb && oi < ci);
                cb = take ? ob : cb;
                ci = take ? oi : ci;
            }
            if (lane == src && cb < best[i]) { best[i] = cb; bi[i] = ci; }
        }
    }
}
