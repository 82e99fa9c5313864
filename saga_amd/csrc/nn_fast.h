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

// An integer again, as a value the compiler cannot connect to its earlier uses: index arithmetic formed from it (tid >> 6, tid & 63,
// chunk offsets) is recomputed where it is needed -- one or two integer operations -- instead of being kept live, and spilled to
// scratch, across a matrix-core pass that needs every register.
__device__ __forceinline__ int nnf_again(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// a value every lane of the wave holds identically, moved to a scalar register (gfx950 has no scalar float ALU: whatever is computed
// from LDS reads or other uniform floats otherwise lives in a vector register per lane for as long as it is live)
__device__ __forceinline__ float nnf_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

__device__ __forceinline__ float nnf_axis_centre(float lo, float hi) {
    if (lo > 0.f && hi <= 2.f * lo) return 0.5f * (lo + hi);
    if (hi < 0.f && lo >= 2.f * hi) return 0.5f * (lo + hi);
    return 0.f;
}

// wave-wide minimum / maximum in every lane: five immediate swizzle patterns and one half-wave swap (no index registers, no LDS
// traffic; min / max are exact, so the pattern does not matter for the result)
#define NNF_SWZ(v, o) __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), ((o) << 10) | 0x1f))
__device__ __forceinline__ float nnf_wave_min(float v) {
    v = fminf(v, NNF_SWZ(v, 16)); v = fminf(v, NNF_SWZ(v, 8)); v = fminf(v, NNF_SWZ(v, 4)); v = fminf(v, NNF_SWZ(v, 2));
    v = fminf(v, NNF_SWZ(v, 1));
    const unsigned u = (unsigned)__float_as_int(v);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // one of the two is the lane's own value, the other its partner's
    return fminf(__int_as_float((int)sw[0]), __int_as_float((int)sw[1]));
}
__device__ __forceinline__ float nnf_wave_max(float v) {
    v = fmaxf(v, NNF_SWZ(v, 16)); v = fmaxf(v, NNF_SWZ(v, 8)); v = fmaxf(v, NNF_SWZ(v, 4)); v = fmaxf(v, NNF_SWZ(v, 2));
    v = fmaxf(v, NNF_SWZ(v, 1));
    const unsigned u = (unsigned)__float_as_int(v);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__int_as_float((int)sw[0]), __int_as_float((int)sw[1]));
}

// Block-wide bounding boxes -> frames, K boxes in one pass (two barriers for all of them).  lo/hi: this thread's partial boxes
// (+INF/-INF when it holds no point).  sred: >= 6 * K * (nthreads/64) floats of LDS.  Contains barriers: every thread of the block calls it.
template <int K>
__device__ __forceinline__ void nnf_frames_from_boxes(float (&lo)[K][3], float (&hi)[K][3], float* sred, int tid, int nthreads,
                                                      NnfFrame (&f)[K]) {
    const int wave = tid >> 6, lane = tid & 63, nw = nthreads >> 6;
#pragma unroll
    for (int b = 0; b < K; ++b)
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[b][k] = nnf_wave_min(lo[b][k]); hi[b][k] = nnf_wave_max(hi[b][k]); }
    __syncthreads();                     // sred may still be read by an earlier call
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < K; ++b)
#pragma unroll
            for (int k = 0; k < 3; ++k) { sred[(wave * K + b) * 6 + k] = lo[b][k]; sred[(wave * K + b) * 6 + 3 + k] = hi[b][k]; }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < K; ++b) {
        float c[3], r2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float l = sred[b * 6 + k], h = sred[b * 6 + 3 + k];
            for (int w = 1; w < nw; ++w) { l = fminf(l, sred[(w * K + b) * 6 + k]); h = fmaxf(h, sred[(w * K + b) * 6 + 3 + k]); }
            c[k] = nnf_axis_centre(l, h);
            const float r = fmaxf(h - c[k], c[k] - l);
            r2 = fmaf(r, r, r2);
        }
        f[b].cx = nnf_uniform(c[0]); f[b].cy = nnf_uniform(c[1]); f[b].cz = nnf_uniform(c[2]);
        f[b].rp = nnf_uniform(sqrtf(r2) * 1.0001f);
    }
}

__device__ __forceinline__ NnfFrame nnf_frame_from_box(float (&lo)[3], float (&hi)[3], float* sred, int tid, int nthreads) {
    float l[1][3] = {{lo[0], lo[1], lo[2]}}, h[1][3] = {{hi[0], hi[1], hi[2]}};
    NnfFrame f[1];
    nnf_frames_from_boxes<1>(l, h, sred, tid, nthreads, f);
    return f[0];
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

__device__ __forceinline__ void nnf_store_point(float* __restrict__ s, int k, float x, float y, float z, const NnfFrame f) {
    const float px = x - f.cx, py = y - f.cy, pz = z - f.cz;       // exact for points inside the frame's box
    const float w = __builtin_fmaf(pz, pz, __builtin_fmaf(py, py, px * px));
    float* d = s + (k / NNF_CH) * NNF_CSTRIDE + ((k % NNF_CH) >> 1) * 8 + (k & 1);
    d[0] = -2.f * px; d[2] = -2.f * py; d[4] = -2.f * pz; d[6] = w;
}

// Stage cnt points (xyz rows) as a framed tile padded to a multiple of NNF_CH with far points.  Caller synchronises.
__device__ __forceinline__ int nnf_stage_tile(float* __restrict__ s, const float* __restrict__ src, int cnt, const NnfFrame f,
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
__device__ __forceinline__ void nnf_tile_point(const float* __restrict__ s, const NnfFrame f, int j, float& x, float& y, float& z) {
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

// ---- approximate pass on the matrix cores ---------------------------------------------------------------------------------
// Same contract as nnf_scan (per lane: NQ framed queries in, the three smallest chunk minima and the chunks of the first two
// out, chunks = NNF_CH consecutive points), but t(p) for 32 points x 32 queries comes from ONE v_mfma_f32_32x32x16_f16:
// every f32 operand is split into two f16 parts (x = xh + xl, |x - xh - xl| <= 2^-22 |x|) after scaling by a power of two,
// and the 16 k-slots carry every partial product plus |p|^2 in three parts:
//     A (point)  [ahx ahy ahz ahx ahy ahz alx aly | alz wh wm wl alx aly alz P]        a = -2 (p - c) sa,  w = |p - c|^2 sa sq
//     B (query)  [qhx qhy qhz qlx qly qlz qhx qhy | qhz 1  1  1  qlx qly qlz 1]        q = (q - c) sq
// (P = 0; pad points: a = 0, wh = wm = wl = P = 65504 -> larger than any real t).  f16 x f16 products are exact in f32; the
// matrix core accumulates in f32.  A wave64 holds output column (lane & 31) = query and rows {(r&3) + 8(r>>2) + 4(lane>>5)};
// tile row rho carries point 16((rho>>2)&1) + 4(rho>>3) + (rho&3) of the 32, so the 16 values a lane sees are exactly chunk
// 2*tile + (lane>>5): the lane-local minimum of the 16 accumulator registers IS the chunk minimum, and the exact pass
// (nnf_resolve) is unchanged.  The two lanes that share a query (l, l^32) merge their sorted triples at the end.
// Error: the matrix core keeps f16 subnormals and adds the 16 exact products in a wide accumulator with one final rounding
// (scripts/ab/mfma_f16_denorm.hip), so |t_mfma / S - t| <= 2^-21 |a||q| (the residuals of the two-part splits) + 2^-24 |t|
// (the rounding) + O(2^-25) absolute per subnormal low part <= 2^-20.8 Rp (Rp + 2 |q - c|).  NNF_MFMA_KAPPA = 2^-19 leaves 3.5x
// on that and 13x on the largest error scripts/ab/nn_mfma16_ab.hip measures against float64 (2^-22.7); a sequential fp32
// accumulation with truncation at every step would give 2^-18.8.
// That bound needs the low parts to be f16 NORMALS, i.e. a lane's own |q - c| and |p - c| not much smaller than what the wave's
// scales were chosen for.  In general a split's residual is max(2^-22 |x|, 2^-25) in SCALED units (f16 subnormal spacing 2^-24), so
//     |error of t| <= 3 [ 2 Rp max(2^-22 |q|, 2^-25 / sq) + |q| max(2^-21 Rp, 2^-25 / sa) ]
//                  <= 3 2^-20 Rp |q|  +  3 2^-23 Rp / sq                (1 / sa <= 2 Rp 2^-7, 1 / sq = max(Rq_wave, Rp) / [128, 256))
// The first term is inside KAPPA Rp (Rp + 2 |q|); the second is not once the scale basis Rq_wave exceeds ~1,400 Rp -- one far
// query in a wave (a sentinel point at 1e7 among points of size 1), or a small tile seen from far away -- and the queries NEXT to
// that outlier then lost neighbours (found with scripts/exp/nn_prune_nonfinite.py; round 3).  nnf_mfma_error_floor = 2^-20 Rp / sq
// (2.7 x the term) is added to E by every caller of the matrix-core pass: nothing in the usual case (5e-10 against 1.4e-7 in the
// fused loss kernel), and with an outlier the whole wave falls back to the exact slow path -- slow, but right.
// The A operand is rebuilt per wave from the f32 tile (no second LDS image: the block keeps its occupancy).
typedef _Float16 nn_h8 __attribute__((ext_vector_type(8)));
typedef float nn_f16v __attribute__((ext_vector_type(16)));
#define NNF_MFMA_KAPPA 1.9073486328125e-6f    // 2^-19

__device__ __forceinline__ float nnf_pow2_scale(float r) {      // 2^(7-e) for r = m 2^e, m in [1,2): r * scale in [128, 256)
    int e = (__float_as_int(r) >> 23) & 0xff;
    e = min(max(e, 127 - 40), 127 + 40);
    return __int_as_float((261 - e) << 23);
}

__device__ __forceinline__ void nnf_insert3(float m, int c, float& m1, float& m2, float& m3, int& c1, int& c2) {
    const bool lt1 = m < m1, lt2 = m < m2;
    m3 = __builtin_amdgcn_fmed3f(m, m2, m3);
    m2 = __builtin_amdgcn_fmed3f(m, m1, m2);
    m1 = __builtin_elementwise_minimum(m, m1);
    int n2 = lt2 ? c : c2;
    n2 = lt1 ? c1 : n2;
    c2 = n2;
    c1 = lt1 ? c : c1;
}

__device__ __forceinline__ void nnf_split16(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// scales (powers of two) of the matrix-core pass: 2 Rp sa in [128, 256); max|q| sq < 256 and Rp sq < 256, so that
// w = |p|^2 sa sq < 32768.  rq: any upper bound of the framed queries' largest |coordinate| the pass will see.
struct NnfScales { float sa, sq, S, inv_S; };
__device__ __forceinline__ NnfScales nnf_scales(float rp, float rq) {
    NnfScales k;
    // rp, rq are wave-uniform wherever this is called (frames, block- or wave-wide maxima): keep the scales in scalar registers
    k.sa = nnf_uniform(nnf_pow2_scale(2.f * rp));
    k.sq = nnf_uniform(fminf(nnf_pow2_scale(rq), nnf_pow2_scale(rp)));
    k.S = nnf_uniform(k.sa * k.sq);
    k.inv_S = nnf_uniform(1.f / k.S);
    return k;
}

__device__ __forceinline__ float nnf_mfma_error_floor(const NnfFrame f, const NnfScales& k) {
    return nnf_uniform(f.rp * 9.5367431640625e-7f / k.sq);      // 2^-20 Rp / sq (see the error analysis above)
}

// A operand of this lane for 32-point tile `tile` of the framed f32 tile s (n_pts points, a multiple of NNF_CH)
__device__ __forceinline__ nn_h8 nnf_build_A(const float* __restrict__ s, int n_pts, int tile, int lane, const NnfScales& k) {
    const int col = lane & 31, h = lane >> 5;
    const int prow = 16 * ((col >> 2) & 1) + 4 * (col >> 3) + (col & 3);      // point of the tile this lane's A row carries
    const _Float16 big = (_Float16)65504.f, hz = (_Float16)0.f;
    const int p = min(tile * 32 + prow, n_pts - 1);
    const float* d = s + (p >> 4) * NNF_CSTRIDE + ((p & 15) >> 1) * 8 + (p & 1);
    const float X = d[0], Y = d[2], Z = d[4], W = d[6];
    // pad points (NN_FAR), and any point with a non-finite or absurdly large coordinate: it can never be a nearest neighbour, and a NaN
    // / Inf in its operand row would turn the whole chunk's minimum into NaN (v_minimum3 propagates it) and hide its 15 neighbours
    const bool pad = !(fabsf(X) < 1.0e17f && fabsf(Y) < 1.0e17f && fabsf(Z) < 1.0e17f && fabsf(W) < 3.0e38f);
    _Float16 axh, axl, ayh, ayl, azh, azl, wh, wm, wl;
    nnf_split16(pad ? 0.f : X * k.sa, axh, axl);
    nnf_split16(pad ? 0.f : Y * k.sa, ayh, ayl);
    nnf_split16(pad ? 0.f : Z * k.sa, azh, azl);
    const float w = pad ? 0.f : W * k.S;
    wh = (_Float16)w;
    const float r1 = w - (float)wh;
    wm = (_Float16)r1;
    wl = (_Float16)(r1 - (float)wm);
    if (pad) { wh = big; wm = big; wl = big; }
    const nn_h8 alo = {axh, ayh, azh, axh, ayh, azh, axl, ayl}, ahi = {azl, wh, wm, wl, axl, ayl, azl, pad ? big : hz};
    return h ? ahi : alo;
}

// The scan proper, in three steps so that a caller can feed the tiles in rounds (e.g. an LDS image of the A operands that only
// holds a part of the cloud at a time): nnf_mfma_begin (B operands of the lane's queries, empty sorted triples), nnf_mfma_tiles
// (any number of calls, tiles in any order: chunk ids are global), nnf_mfma_end (the two lanes that share a query merge).
// IMAGE = false: s is the framed f32 tile, every wave builds its A operands as it goes.  IMAGE = true: s is an LDS image of
// the A operands ([tile - img_tile0][lane] nn_h8, built with nnf_build_A and THESE scales) -- for callers whose waves scan
// the same tile and have the LDS for it.  EVEN: the cloud is padded with far points to a whole number of 32-point tiles
// (n_chunks even), so no half tile has to be masked.  act: bit (2 i + hh) = query tile (i, hh) holds at least one real
// query (wave-uniform); the matrix instruction and the bookkeeping of an all-padding query tile are skipped.
template <int NQ>
struct NnfMfmaState {
    nn_h8 Bq[NQ][2];
    float M1[NQ][2], M2[NQ][2], M3[NQ][2];
    int C1[NQ][2], C2[NQ][2];
};

template <int NQ>
__device__ __forceinline__ void nnf_mfma_begin(NnfMfmaState<NQ>& st, const NnfScales& k, const float (&qx)[NQ], const float (&qy)[NQ],
                                               const float (&qz)[NQ]) {
    const int lane = __lane_id(), h = lane >> 5;
    const float sq = k.sq;
    // B operands: tile (i, hh) = query i of lanes hh*32 .. hh*32+31 -- every lane needs the query of lane (lane & 31) + 32 hh: its
    // own and its half-wave partner's (one v_permlane32_swap per coordinate hands both halves what they need)
    const _Float16 one = (_Float16)1.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const auto sx = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(qx[i]), (unsigned)__float_as_int(qx[i]), false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(qy[i]), (unsigned)__float_as_int(qy[i]), false, false);
        const auto sz = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(qz[i]), (unsigned)__float_as_int(qz[i]), false, false);
        // (swap(v, v): element 0 = the value of the lane's counterpart in the LOWER half, element 1 = in the UPPER half)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            _Float16 xh, xl, yh, yl, zh, zl;
            nnf_split16(__int_as_float((int)sx[hh]) * sq, xh, xl);
            nnf_split16(__int_as_float((int)sy[hh]) * sq, yh, yl);
            nnf_split16(__int_as_float((int)sz[hh]) * sq, zh, zl);
            const nn_h8 lo = {xh, yh, zh, xl, yl, zl, xh, yh}, hi = {zh, one, one, one, xl, yl, zl, one};
            st.Bq[i][hh] = h ? hi : lo;
            st.M1[i][hh] = INFINITY; st.M2[i][hh] = INFINITY; st.M3[i][hh] = INFINITY; st.C1[i][hh] = -1; st.C2[i][hh] = -1;
        }
    }
}

// PAIR: two consecutive tiles (2k, 2k+1; tile_lo even) share one sorted insert -- the unit the triples rank is then the "pair
// chunk" 2k + (lane>>5) = chunks {4k + h, 4k + 2 + h} (32 points), half as many inserts per matrix instruction, and the exact
// pass re-evaluates up to four 16-point chunks instead of two (nnf_resolve<.., PAIR>).  Pays for long scans (the 64 object tiles
// of the fused loss kernel), not for short ones.
template <int NQ, bool IMAGE, bool EVEN, bool PAIR = false>
__device__ __forceinline__ void nnf_mfma_tiles(NnfMfmaState<NQ>& st, const float* __restrict__ s, int n_chunks, int tile_lo, int tile_hi,
                                               int img_tile0, const NnfScales& k, unsigned act) {
    const int lane = __lane_id(), h = lane >> 5;
    const int n_pts = n_chunks * NNF_CH;
    const nn_f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (PAIR) {
        auto tile_min = [&](const nn_f16v& D, float m) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) m = __builtin_elementwise_minimum(__builtin_elementwise_minimum(m, D[r]), D[r + 1]);
            return m;
        };
        for (int tile = tile_lo; tile < tile_hi; tile += 2) {
            const bool has2 = tile + 1 < tile_hi;                        // (wave-uniform)
            const int tb = has2 ? tile + 1 : tile;
            const nn_h8 A0 = IMAGE ? reinterpret_cast<const nn_h8*>(s)[(tile - img_tile0) * 64 + lane] : nnf_build_A(s, n_pts, tile, lane, k);
            const nn_h8 A1 = IMAGE ? reinterpret_cast<const nn_h8*>(s)[(tb - img_tile0) * 64 + lane] : nnf_build_A(s, n_pts, tb, lane, k);
            const int c = tile + h;                                       // pair chunk 2 (tile / 2) + h
            const bool v0 = EVEN || 2 * tile + h < n_chunks, v1 = has2 && (EVEN || 2 * tile + 2 + h < n_chunks);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const bool a0 = (act >> (2 * i)) & 1u, a1 = (act >> (2 * i + 1)) & 1u;
                if (a0 && a1) {
                    float ma = INFINITY, mb = INFINITY;
                    {
                        const nn_f16v D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, st.Bq[i][0], zero16, 0, 0, 0);
                        const nn_f16v D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, st.Bq[i][1], zero16, 0, 0, 0);
                        ma = tile_min(D0, ma); mb = tile_min(D1, mb);
                        if (!EVEN) { ma = v0 ? ma : INFINITY; mb = v0 ? mb : INFINITY; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (has2) {
                        const nn_f16v D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, st.Bq[i][0], zero16, 0, 0, 0);
                        const nn_f16v D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, st.Bq[i][1], zero16, 0, 0, 0);
                        const float na = tile_min(D0, ma), nb = tile_min(D1, mb);
                        ma = (EVEN || v1) ? na : ma; mb = (EVEN || v1) ? nb : mb;
                    }
                    nnf_insert3(ma, c, st.M1[i][0], st.M2[i][0], st.M3[i][0], st.C1[i][0], st.C2[i][0]);
                    nnf_insert3(mb, c, st.M1[i][1], st.M2[i][1], st.M3[i][1], st.C1[i][1], st.C2[i][1]);
                } else if (a0 || a1) {
                    const nn_h8 Bq = a0 ? st.Bq[i][0] : st.Bq[i][1];
                    float ma = tile_min(__builtin_amdgcn_mfma_f32_32x32x16_f16(A0, Bq, zero16, 0, 0, 0), INFINITY);
                    if (!EVEN) ma = v0 ? ma : INFINITY;
                    if (has2) {
                        const float na = tile_min(__builtin_amdgcn_mfma_f32_32x32x16_f16(A1, Bq, zero16, 0, 0, 0), ma);
                        ma = (EVEN || v1) ? na : ma;
                    }
                    if (a0) nnf_insert3(ma, c, st.M1[i][0], st.M2[i][0], st.M3[i][0], st.C1[i][0], st.C2[i][0]);
                    else nnf_insert3(ma, c, st.M1[i][1], st.M2[i][1], st.M3[i][1], st.C1[i][1], st.C2[i][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const nn_h8 A = IMAGE ? reinterpret_cast<const nn_h8*>(s)[(tile - img_tile0) * 64 + lane] : nnf_build_A(s, n_pts, tile, lane, k);
        const int c = 2 * tile + h;
        const bool valid = EVEN || c < n_chunks;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            // two MFMAs in flight, then their two reductions; the scheduling fence keeps the compiler from issuing all 2 NQ
            // MFMAs first (16 accumulator registers each)
            const bool a0 = (act >> (2 * i)) & 1u, a1 = (act >> (2 * i + 1)) & 1u;
            if (a0 && a1) {
                const nn_f16v D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, st.Bq[i][0], zero16, 0, 0, 0);
                const nn_f16v D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, st.Bq[i][1], zero16, 0, 0, 0);
                // v_minimum3_f32 (IEEE-754-2019 minimum: no canonicalisation of the accumulator registers needed)
                float ma = __builtin_elementwise_minimum(D0[0], D0[1]), mb = __builtin_elementwise_minimum(D1[0], D1[1]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) ma = __builtin_elementwise_minimum(__builtin_elementwise_minimum(ma, D0[r]), D0[r + 1]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) mb = __builtin_elementwise_minimum(__builtin_elementwise_minimum(mb, D1[r]), D1[r + 1]);
                if (!EVEN) { ma = valid ? ma : INFINITY; mb = valid ? mb : INFINITY; }
                nnf_insert3(ma, c, st.M1[i][0], st.M2[i][0], st.M3[i][0], st.C1[i][0], st.C2[i][0]);
                nnf_insert3(mb, c, st.M1[i][1], st.M2[i][1], st.M3[i][1], st.C1[i][1], st.C2[i][1]);
            } else if (a0 || a1) {
                const int hh = a0 ? 0 : 1;
                const nn_f16v D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, a0 ? st.Bq[i][0] : st.Bq[i][1], zero16, 0, 0, 0);
                float ma = __builtin_elementwise_minimum(D0[0], D0[1]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) ma = __builtin_elementwise_minimum(__builtin_elementwise_minimum(ma, D0[r]), D0[r + 1]);
                if (!EVEN) ma = valid ? ma : INFINITY;
                if (hh == 0) nnf_insert3(ma, c, st.M1[i][0], st.M2[i][0], st.M3[i][0], st.C1[i][0], st.C2[i][0]);
                else nnf_insert3(ma, c, st.M1[i][1], st.M2[i][1], st.M3[i][1], st.C1[i][1], st.C2[i][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NQ>
__device__ __forceinline__ void nnf_mfma_end(const NnfMfmaState<NQ>& st, const NnfScales& k, float (&m1)[NQ], float (&m2)[NQ],
                                             float (&m3)[NQ], int (&c1)[NQ], int (&c2)[NQ]) {
    const int h = __lane_id() >> 5;
    const float inv_S = k.inv_S;
    // the lane's own query i was scanned in tile (i, h): its rows by this lane, the other 16 of every 32 by lane ^ 32
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        float a1 = h ? st.M1[i][1] : st.M1[i][0], a2 = h ? st.M2[i][1] : st.M2[i][0], a3 = h ? st.M3[i][1] : st.M3[i][0];
        int k1 = h ? st.C1[i][1] : st.C1[i][0], k2 = h ? st.C2[i][1] : st.C2[i][0];
        const float b1 = __shfl_xor(h ? st.M1[i][0] : st.M1[i][1], 32, 64), b2 = __shfl_xor(h ? st.M2[i][0] : st.M2[i][1], 32, 64),
                    b3 = __shfl_xor(h ? st.M3[i][0] : st.M3[i][1], 32, 64);
        const int j1 = __shfl_xor(h ? st.C1[i][0] : st.C1[i][1], 32, 64), j2 = __shfl_xor(h ? st.C2[i][0] : st.C2[i][1], 32, 64);
        nnf_insert3(b1, j1, a1, a2, a3, k1, k2);
        nnf_insert3(b2, j2, a1, a2, a3, k1, k2);
        nnf_insert3(b3, -1, a1, a2, a3, k1, k2);       // only its value can matter (as m3)
        m1[i] = a1 * inv_S; m2[i] = a2 * inv_S; m3[i] = a3 * inv_S; c1[i] = k1; c2[i] = k2;
    }
}

// One call over the whole cloud (every query tile active).
template <int NQ, bool IMAGE>
__device__ __forceinline__ void nnf_scan_mfma_core(const float* __restrict__ s, int n_chunks, const NnfScales& k, const float (&qx)[NQ],
                                                   const float (&qy)[NQ], const float (&qz)[NQ], float (&m1)[NQ], float (&m2)[NQ],
                                                   float (&m3)[NQ], int (&c1)[NQ], int (&c2)[NQ]) {
    NnfMfmaState<NQ> st;
    nnf_mfma_begin<NQ>(st, k, qx, qy, qz);
    nnf_mfma_tiles<NQ, IMAGE, false>(st, s, n_chunks, 0, (n_chunks + 1) >> 1, 0, k, ~0u);
    nnf_mfma_end<NQ>(st, k, m1, m2, m3, c1, c2);
}

// Stand-alone form: scales from this wave's own queries, A operands built on the fly.
template <int NQ>
__device__ __forceinline__ void nnf_scan_mfma(const float* __restrict__ s, int n_chunks, const NnfFrame f, const float (&qx)[NQ],
                                              const float (&qy)[NQ], const float (&qz)[NQ], float (&m1)[NQ], float (&m2)[NQ],
                                              float (&m3)[NQ], int (&c1)[NQ], int (&c2)[NQ], float& e_floor) {
    float rq = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) rq = fmaxf(rq, fmaxf(fabsf(qx[i]), fmaxf(fabsf(qy[i]), fabsf(qz[i]))));
    for (int o = 32; o > 0; o >>= 1) rq = fmaxf(rq, __shfl_xor(rq, o, 64));
    const NnfScales k = nnf_scales(f.rp, rq);
    e_floor = nnf_mfma_error_floor(f, k);           // what nnf_resolve adds to its error bound for THIS wave's scales
    nnf_scan_mfma_core<NQ, false>(s, n_chunks, k, qx, qy, qz, m1, m2, m3, c1, c2);
}

// exact evaluation of one chunk for one query (original coordinates o*), index order, strict '<'.  Two points per packed
// instruction; every packed lane performs the IEEE operation of nn_dist2, so the values are those of the scalar chain.
// ORIG (nn_prune.h: the tile holds SORTED points): sorig[k] is the original index of tile point k, bi carries original indices and
// the comparison is lexicographic on (distance, original index) -- the result no longer depends on the order points are visited in.
template <bool ORIG = false>
__device__ __forceinline__ void nnf_exact_chunk(const float* __restrict__ s, const NnfFrame f, int ch, int base, float ox, float oy,
                                                float oz, float& best, int& bi, const int* __restrict__ sorig = nullptr) {
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
        if (ORIG) {
            const int2 o = *reinterpret_cast<const int2*>(sorig + ch * NNF_CH + 2 * g);
            if (d.x < best || (d.x == best && o.x < bi)) { best = d.x; bi = o.x; }
            if (d.y < best || (d.y == best && o.y < bi)) { best = d.y; bi = o.y; }
        } else {
            const int k = base + ch * NNF_CH + 2 * g;
            if (d.x < best) { best = d.x; bi = k; }
            if (d.y < best) { best = d.y; bi = k + 1; }
        }
    }
}

// Exact pass.  qc*: framed queries (for the margin), o*: original query coordinates.  best/bi (IN/OUT; the caller
// starts them at +INF / 0): exact squared distance and index of the sequential exact scan over every tile so far
// (global index of point 0 of this tile = base; tiles must come in index order).  Every lane of the wave must call it
// together (the slow path is a wave-wide loop).
// MFMA_PASS: the chunk minima come from nnf_scan_mfma (its error bound replaces the vector pass's).
// PAIR: c1 / c2 are pair chunks of nnf_mfma_tiles<.., PAIR> (pair chunk c = chunks 4 (c>>1) + (c&1) and that + 2).
template <int NQ, bool MFMA_PASS = false, bool ORIG = false, bool PAIR = false>
__device__ __forceinline__ void nnf_resolve(const float* __restrict__ s, int n_chunks, const NnfFrame f, int base,
                                            const float (&qx)[NQ], const float (&qy)[NQ], const float (&qz)[NQ],
                                            const float (&ox)[NQ], const float (&oy)[NQ], const float (&oz)[NQ],
                                            const float (&m1)[NQ], const float (&m2)[NQ], const float (&m3)[NQ],
                                            const int (&c1)[NQ], const int (&c2)[NQ],
                                            float (&best)[NQ], int (&bi)[NQ], const int* __restrict__ sorig = nullptr,
                                            float e_floor = 0.f) {
    const int lane = __lane_id();
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const float qq = __builtin_fmaf(qz[i], qz[i], __builtin_fmaf(qy[i], qy[i], qx[i] * qx[i])) * 1.001f;
        // (e_floor: nnf_mfma_error_floor of the scales the matrix-core pass ran with -- wave-uniform; 0 for the vector pass)
        const float E = MFMA_PASS ? NNF_MFMA_KAPPA * f.rp * (f.rp + 2.f * sqrtf(qq)) + e_floor
                                  : 5.9604645e-7f * f.rp * (f.rp + sqrtf(qq));           // 10 u Rp (Rp + Rq)
        const float thr = m1[i] + (2.f * E + 1.0e-6f * (fmaxf(m1[i] + qq, 0.f) + E));
        // a tile whose best possible distance cannot beat the exact minimum carried in from earlier tiles is skipped:
        // D(p) >= t(p) - E + |q - c|^2 (1 - 3u) for every point of this tile, and its fp32 distance is >= D (1 - 6u)
        // (m1 = +INF: this query was not scanned at all -- a padding query of a skipped query tile -- nothing to resolve)
        const bool live = m1[i] < INFINITY && !(m1[i] - E + qq * 0.998f > best[i] * 1.000001f);
        const bool fast = live && m3[i] > thr;
        if (fast) {
            // the exact winner is in chunk c1 or c2: lower chunk first so that ties keep the lowest index
            // (c2 only matters when its minimum is inside the margin)
            const bool two = c2[i] >= 0 && m2[i] <= thr;
            const int a = two ? min(c1[i], c2[i]) : c1[i];
            const int b = two ? max(c1[i], c2[i]) : c1[i];
            if (PAIR) {
                // up to four 16-point chunks, ascending: pair chunks of different tile pairs do not interleave, the two of one
                // tile pair (a = 2k, b = 2k + 1) are chunks 4k, 4k+2 and 4k+1, 4k+3
                const int e0 = 4 * (a >> 1) + (a & 1), e3 = 4 * (b >> 1) + (b & 1) + 2;
                const bool same = b != a && (a >> 1) == (b >> 1);
                const int e1 = same ? e0 + 1 : e0 + 2, e2 = same ? e0 + 2 : e3 - 2;
                nnf_exact_chunk<ORIG>(s, f, e0, base, ox[i], oy[i], oz[i], best[i], bi[i], sorig);
                if (e1 < n_chunks) nnf_exact_chunk<ORIG>(s, f, e1, base, ox[i], oy[i], oz[i], best[i], bi[i], sorig);
                if (b != a) {
                    if (e2 < n_chunks) nnf_exact_chunk<ORIG>(s, f, e2, base, ox[i], oy[i], oz[i], best[i], bi[i], sorig);
                    if (e3 < n_chunks) nnf_exact_chunk<ORIG>(s, f, e3, base, ox[i], oy[i], oz[i], best[i], bi[i], sorig);
                }
            } else {
                nnf_exact_chunk<ORIG>(s, f, a, base, ox[i], oy[i], oz[i], best[i], bi[i], sorig);
                if (b != a) nnf_exact_chunk<ORIG>(s, f, b, base, ox[i], oy[i], oz[i], best[i], bi[i], sorig);
            }
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
                if (m <= sthr) nnf_exact_chunk<ORIG>(s, f, c, base, sox, soy, soz, cb, ci, sorig);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(cb, o, 64);
                const int oi = __shfl_xor(ci, o, 64);
                const bool take = ob < cb || (ob == cb && oi < ci);
                cb = take ? ob : cb;
                ci = take ? oi : ci;
            }
            if (lane == src && (cb < best[i] || (ORIG && cb == best[i] && ci < bi[i]))) { best[i] = cb; bi[i] = ci; }
        }
    }
}
