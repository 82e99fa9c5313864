// Pruned exact nearest-neighbour search for large clouds (row f4 of SURVEY.md section 8: BASELINE config 5, 10,475 body vertices
// against 100,000 scene points per batch element; reference call shape utils/train_helper.py:57, WholeGraspPose/trainer.py:280-281).
//
// Same contract as the brute-force scan (chamfer.hip): for every query the squared distance fma(dz,dz, fma(dy,dy, dx*dx)) to, and
// the ORIGINAL index of, its nearest database point, lowest index among equal distances -- bit for bit.  What changes is which
// pairs are evaluated:
//   1. nnp_sort_kernel: each cloud of a batch element is counting-sorted by the Morton code of its cell in a 32^3 grid over the
//      cloud's bounding box (one block per batch element, the 32,768 counters in LDS); a sorted point carries its original index.
//      The order inside a cell is whatever the LDS atomics give -- the search result does not depend on it (below).
//   2. nnp_boxes_kernel: bounding boxes of every 32 consecutive sorted points ("group": one matrix-core tile of the scan) and of
//      every 2,048 ("tile": what a block stages in LDS).
//   3. nnp_scan_kernel: a block owns 1,024 consecutive SORTED queries (a wave 256, a matrix-core query tile 32: all spatially
//      compact) and visits the database tiles NEAREST FIRST by the box-to-box lower bound of the squared distance, until that
//      bound exceeds the largest upper bound any of its queries still has.  Inside a staged tile every wave evaluates only the
//      (32-query tile, 32-point group) pairs whose box-to-box lower bound does not exceed the query tile's upper bound = the
//      largest exact minimum its 32 queries hold so far, seeded before anything is scanned by min over groups of the box-to-box
//      MAXIMUM distance (a non-empty group has a point at most that far from every query of the box).
//   The pairs that survive run the two-level scan of nn_fast.h unchanged (matrix-core approximate pass, exact pass over the
//   candidate chunks), except that the exact pass compares (distance, original index) lexicographically instead of relying on the
//   visiting order.  A skipped group cannot hold the minimum or tie with it: lower bounds are rounded down and upper bounds up by
//   1e-5 relative (the bounds themselves carry < 1e-6), and a comparison that involves a NaN never skips.  So the result is the
//   unique lexicographic minimum over ALL points -- independent of the visiting order and of the order inside a grid cell.
#pragma once
#include <limits.h>

#include "nn_fast.h"

#define NNP_CELLS 32768                            // grid cells (2^NNP_KEY_BITS), dealt to the axes per cloud
#define NNP_SORT_THREADS 1024
#define NNP_TILE 512                               // sorted database points a wave stages at a time
#define NNP_GROUP 32                               // points per matrix-core tile
#define NNP_GPT (NNP_TILE / NNP_GROUP)             // groups per tile: 16
#define NNP_MAX_TILES 512                          // per batch element (262,144 database points); beyond: brute force
#define NNP_THREADS 64                             // one wave per block
#define NNP_NQ 4                                   // queries per lane
#define NNP_QPB (NNP_THREADS * NNP_NQ)             // queries per block: 256
#ifndef NNP_WAVES_PER_EU
#define NNP_WAVES_PER_EU 3                         // register budget of the scan kernel: 512 / this many per lane (measured on config 5:
                                                   // 3 waves per SIMD 20.5 ms, 2 waves per SIMD with 216 registers 22.7 ms)
#endif

// ---- lane exchanges with immediate patterns (no per-lane addresses kept live) ---------------------------------------------
#define NNP_SWZ(v, o) __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), ((o) << 10) | 0x1f))
__device__ __forceinline__ float nnp_min32(float v) {      // minimum over the lane's 32-lane half, in every lane of it
    v = fminf(v, NNP_SWZ(v, 16)); v = fminf(v, NNP_SWZ(v, 8)); v = fminf(v, NNP_SWZ(v, 4)); v = fminf(v, NNP_SWZ(v, 2));
    return fminf(v, NNP_SWZ(v, 1));
}
__device__ __forceinline__ float nnp_max32(float v) {
    v = fmaxf(v, NNP_SWZ(v, 16)); v = fmaxf(v, NNP_SWZ(v, 8)); v = fmaxf(v, NNP_SWZ(v, 4)); v = fmaxf(v, NNP_SWZ(v, 2));
    return fmaxf(v, NNP_SWZ(v, 1));
}
__device__ __forceinline__ float nnp_min64(float v) {      // over the wave, in every lane
    v = nnp_min32(v);
    const unsigned u = (unsigned)__float_as_int(v);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // one of the two is the lane's own value, the other its partner's
    return fminf(__int_as_float((int)sw[0]), __int_as_float((int)sw[1]));
}
__device__ __forceinline__ float nnp_max64(float v) {
    v = nnp_max32(v);
    const unsigned u = (unsigned)__float_as_int(v);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__int_as_float((int)sw[0]), __int_as_float((int)sw[1]));
}

// ---- 1. counting sort by grid cell ------------------------------------------------------------------------------------------
// The 15 key bits are dealt to the axes so that the cells come out as close to cubes as powers of two allow (a body's bounding box
// is 3 x 1.5 x 8: equal bits per axis would make cells 1 : 0.5 : 2.8): NnpGrid.seq lists, from the most significant key bit down,
// which axis the bit halves -- always the axis whose cells are currently the longest -- so the key is a Morton code with unequal
// bit counts (a kd-style interleave).  32 consecutive sorted points then fill a compact box (measured on config 5: the share of
// (query tile, group) pairs the box test keeps falls from 15 % to 7 % against equal bits).
#define NNP_KEY_BITS 15
struct NnpGrid { float lo[3], sc[3]; int bits[3]; unsigned seq; };     // seq: 2 bits per key bit, MSB first at the low end
__device__ __forceinline__ NnpGrid nnp_make_grid(const float (&lo)[3], const float (&hi)[3]) {
    NnpGrid g;
    float cell[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float e = hi[a] - lo[a];
        const bool ok = (e > 0.f) && (e <= 3.0e38f) && (fabsf(lo[a]) <= 3.0e38f);
        cell[a] = ok ? e : 0.f;                                   // degenerate / non-finite axis: no bits, every point in cell 0
        g.lo[a] = ok ? lo[a] : 0.f;
        g.bits[a] = 0;
    }
    g.seq = 0;
    for (int k = 0; k < NNP_KEY_BITS; ++k) {
        int a = 0;
        if (cell[1] > cell[a]) a = 1;
        if (cell[2] > cell[a]) a = 2;
        g.seq |= (unsigned)a << (2 * k);
        if (a == 0) { cell[0] *= 0.5f; g.bits[0]++; } else if (a == 1) { cell[1] *= 0.5f; g.bits[1]++; } else { cell[2] *= 0.5f; g.bits[2]++; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float e = hi[a] - lo[a];
        g.sc[a] = (cell[a] > 0.f || g.bits[a] > 0) && e > 0.f && e <= 3.0e38f ? (float)(1 << g.bits[a]) / e * 0.999999f : 0.f;
    }
    return g;
}
__device__ __forceinline__ unsigned nnp_cell(float x, float y, float z, const NnpGrid& g) {
    // fmaxf / fminf drop a NaN operand: a non-finite coordinate lands in a border cell, never out of range
    const unsigned cx = (unsigned)fminf(fmaxf((x - g.lo[0]) * g.sc[0], 0.f), (float)((1 << g.bits[0]) - 1));
    const unsigned cy = (unsigned)fminf(fmaxf((y - g.lo[1]) * g.sc[1], 0.f), (float)((1 << g.bits[1]) - 1));
    const unsigned cz = (unsigned)fminf(fmaxf((z - g.lo[2]) * g.sc[2], 0.f), (float)((1 << g.bits[2]) - 1));
    int rx = g.bits[0], ry = g.bits[1], rz = g.bits[2];
    unsigned key = 0;
#pragma unroll
    for (int k = 0; k < NNP_KEY_BITS; ++k) {
        const unsigned a = (g.seq >> (2 * k)) & 3u;              // (uniform)
        unsigned bit;
        if (a == 0) { --rx; bit = (cx >> rx) & 1u; } else if (a == 1) { --ry; bit = (cy >> ry) & 1u; } else { --rz; bit = (cz >> rz) & 1u; }
        key = (key << 1) | bit;
    }
    return key;
}

// pts [N,P,3] -> sorted [N,P] (x, y, z, original index as int bits).  One block per batch element; dynamic LDS: NNP_CELLS counters.
__global__ __launch_bounds__(NNP_SORT_THREADS) void nnp_sort_kernel(const float* __restrict__ pts, float4* __restrict__ sorted, int P) {
    extern __shared__ unsigned int nnp_bins[];                  // [NNP_CELLS]
    __shared__ float sred[6 * (NNP_SORT_THREADS / 64)];
    __shared__ unsigned int swsum[NNP_SORT_THREADS / 64];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pn = pts + (size_t)n * P * 3;
    // A: bounding box of the finite coordinates
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = tid; k < P; k += NNP_SORT_THREADS) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pn[(size_t)k * 3 + a];
            if (fabsf(v) <= 1.0e30f) { lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = nnp_min64(lo[a]); hi[a] = nnp_max64(hi[a]); }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { sred[wave * 6 + a] = lo[a]; sred[wave * 6 + 3 + a] = hi[a]; }
    }
    for (int c = tid; c < NNP_CELLS; c += NNP_SORT_THREADS) nnp_bins[c] = 0u;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = sred[a], h = sred[3 + a];
        for (int w = 1; w < NNP_SORT_THREADS / 64; ++w) { l = fminf(l, sred[w * 6 + a]); h = fmaxf(h, sred[w * 6 + 3 + a]); }
        lo[a] = nnf_uniform(l); hi[a] = nnf_uniform(h);
    }
    const NnpGrid grid = nnp_make_grid(lo, hi);
    // B: histogram of the cells
    for (int k = tid; k < P; k += NNP_SORT_THREADS) {
        const float* p = pn + (size_t)k * 3;
        atomicAdd(&nnp_bins[nnp_cell(p[0], p[1], p[2], grid)], 1u);
    }
    __syncthreads();
    // C: exclusive scan of the counters (thread t owns cells [32 t, 32 t + 32))
    constexpr int CPT = NNP_CELLS / NNP_SORT_THREADS;
    unsigned int local = 0;
#pragma unroll 8
    for (int c = 0; c < CPT; ++c) local += nnp_bins[tid * CPT + c];
    unsigned int incl = local;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) swsum[wave] = incl;
    __syncthreads();
    unsigned int base = 0;
    for (int w = 0; w < wave; ++w) base += swsum[w];
    unsigned int run = base + incl - local;
#pragma unroll 8
    for (int c = 0; c < CPT; ++c) {
        const unsigned int cnt = nnp_bins[tid * CPT + c];
        nnp_bins[tid * CPT + c] = run;
        run += cnt;
    }
    __syncthreads();
    // D: scatter (the position inside a cell is the order the atomics land in: the search result does not depend on it)
    float4* sn = sorted + (size_t)n * P;
    for (int k = tid; k < P; k += NNP_SORT_THREADS) {
        const float* p = pn + (size_t)k * 3;
        const float x = p[0], y = p[1], z = p[2];
        const unsigned int pos = atomicAdd(&nnp_bins[nnp_cell(x, y, z, grid)], 1u);
        sn[pos] = make_float4(x, y, z, __int_as_float(k));
    }
}

// ---- 2. boxes of the sorted groups and tiles --------------------------------------------------------------------------------
// gbox [N, n_tiles * 16, 6], tbox [N, n_tiles, 6] (lo xyz, hi xyz; an empty group / tile: lo = +INF, hi = -INF).
// grid ((n_tiles + 3) / 4, N), 256 threads: a wave = one 512-point tile, four lanes = one group.
__global__ __launch_bounds__(256) void nnp_boxes_kernel(const float4* __restrict__ sorted, float* __restrict__ gbox,
                                                        float* __restrict__ tbox, int P, int n_tiles) {
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;
    const float4* sn = sorted + (size_t)n * P + (size_t)t * NNP_TILE;
    const int cnt = min(NNP_TILE, P - t * NNP_TILE);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int j = 0; j < 8; ++j) {                           // 8 consecutive points per thread, 4 threads per group
        const int k = lane * 8 + j;
        if (k < cnt) {
            const float4 v = sn[k];
            lo[0] = fminf(lo[0], v.x); hi[0] = fmaxf(hi[0], v.x);
            lo[1] = fminf(lo[1], v.y); hi[1] = fmaxf(hi[1], v.y);
            lo[2] = fminf(lo[2], v.z); hi[2] = fmaxf(hi[2], v.z);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = fminf(lo[a], NNP_SWZ(lo[a], 1)); lo[a] = fminf(lo[a], NNP_SWZ(lo[a], 2));
        hi[a] = fmaxf(hi[a], NNP_SWZ(hi[a], 1)); hi[a] = fmaxf(hi[a], NNP_SWZ(hi[a], 2));
    }
    if ((lane & 3) == 0) {
        float* g = gbox + (((size_t)n * n_tiles + t) * NNP_GPT + (lane >> 2)) * 6;
        g[0] = lo[0]; g[1] = lo[1]; g[2] = lo[2]; g[3] = hi[0]; g[4] = hi[1]; g[5] = hi[2];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = nnp_min64(lo[a]); hi[a] = nnp_max64(hi[a]); }
    if (lane == 0) {
        float* tb = tbox + ((size_t)n * n_tiles + t) * 6;
        tb[0] = lo[0]; tb[1] = lo[1]; tb[2] = lo[2]; tb[3] = hi[0]; tb[4] = hi[1]; tb[5] = hi[2];
    }
}

// ---- 3. the pruned scan -------------------------------------------------------------------------------------------------------
// squared box-to-box distances: lower bound (0 when the boxes overlap) and the largest distance between any two of their points
__device__ __forceinline__ void nnp_box_bounds(const float (&qlo)[3], const float (&qhi)[3], const float (&glo)[3], const float (&ghi)[3],
                                               float& lb2, float& md2) {
    lb2 = 0.f; md2 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float gap = fmaxf(fmaxf(qlo[a] - ghi[a], glo[a] - qhi[a]), 0.f);
        const float far_ = fmaxf(qhi[a] - glo[a], ghi[a] - qlo[a]);
        lb2 = __builtin_fmaf(gap, gap, lb2);
        md2 = __builtin_fmaf(far_, far_, md2);
    }
}
#if defined(SAGA_EXPERIMENT_BUILD) && defined(NNP_EXP_STATS)      // scripts/exp/nn_prune_stats.py: how much the bounds prune
__device__ unsigned long long nnp_stats[16];    // + cycles (lane 0 of every wave): 8 top..B1, 9 staging + tests, 10 B2 wait, 11 matrix-core pass, 12 exact pass, 13 whole kernel      // blocks, tiles staged, wave-tile visits, wave-tile skips, active (group, query tile) pairs, groups with any
extern "C" int saga_debug_nnp_stats(unsigned long long* out, int reset) {
    unsigned long long z[16] = {0};
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(nnp_stats), sizeof(z));
    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(nnp_stats), z, sizeof(z));
    return (int)e;
}
// (accumulated in registers, one atomic per counter and wave at the end: atomics inside the loop would be what is measured)
#define NNP_STAT(i, v) do { st_acc[i] += (unsigned long long)(v); } while (0)
#define NNP_CLK(i) do { const unsigned long long t_ = clock64(); st_acc[i] += t_ - t_prev; t_prev = t_; } while (0)
#define NNP_CLK0() unsigned long long t_prev = clock64(); const unsigned long long t_begin_ = t_prev
#define NNP_STAT_DECL() unsigned long long st_acc[16] = {0}
#define NNP_CLK_END() do { st_acc[13] = clock64() - t_begin_; if (lane == 0) { for (int i_ = 0; i_ < 14; ++i_) if (st_acc[i_]) atomicAdd(&nnp_stats[i_], st_acc[i_]); \
    atomicMax(&nnp_stats[14], st_acc[13]); if (st_acc[13] > 4 * 1000000ull) atomicAdd(&nnp_stats[15], 1ull); } } while (0)
#else
#define NNP_STAT(i, v)
#define NNP_CLK(i)
#define NNP_CLK0()
#define NNP_STAT_DECL()
#define NNP_CLK_END()
#endif
#define NNP_DOWN 0.99999f
#define NNP_UP 1.00001f

// qs / dbs: sorted clouds [N,Pq] / [N,Pd]; dist / idx [N,Pq] in ORIGINAL query order, idx = original database index.
// grid ((Pq + 255) / 256, N); ONE WAVE per block: 256 consecutive sorted queries (eight matrix-core query tiles of 32), its own
// 512-point LDS tile, its own visiting order -- no barrier, no wave ever waits for a tile that only matters to another one (the
// first version had four waves share 2,048-point tiles: 36-47 % of a wave's time went to the barrier).
__global__ __launch_bounds__(NNP_THREADS, NNP_WAVES_PER_EU) void nnp_scan_kernel(const float4* __restrict__ qs, const float4* __restrict__ dbs,
                                                                  const float* __restrict__ gbox, const float* __restrict__ tbox,
                                                                  float* __restrict__ dist, int32_t* __restrict__ idx, int Pq, int Pd,
                                                                  int n_tiles) {
    constexpr int NQ = NNP_NQ;
    __shared__ __attribute__((aligned(16))) float sdb[NNF_TILE_FLOATS(NNP_TILE)];
    __shared__ __attribute__((aligned(8))) int sorig[NNP_TILE];
    __shared__ float sqbox[2 * NQ][6];
    __shared__ float sgb[NNP_GPT][6];                 // the staged tile's group boxes (for the per-query refinement)
    const int n = blockIdx.y, lane = threadIdx.x;
    const int q0 = blockIdx.x * NNP_QPB;
    const float4* qn = qs + (size_t)n * Pq;
    const float4* dn = dbs + (size_t)n * Pd;
    const float* gbn = gbox + (size_t)n * n_tiles * NNP_GPT * 6;
    const float* tbn = tbox + (size_t)n * n_tiles * 6;
    NNP_STAT_DECL();
    NNP_STAT(0, 1);

    float ox[NQ], oy[NQ], oz[NQ], best[NQ];
    int bi[NQ];
    float wlo[3] = {INFINITY, INFINITY, INFINITY}, whi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const float4 v = qn[min(q0 + i * 64 + lane, Pq - 1)];
        ox[i] = v.x; oy[i] = v.y; oz[i] = v.z;
        best[i] = INFINITY; bi[i] = 0;
        // box of the 32 queries of matrix-core query tile j = 2 i + (lane >> 5)
        const float l0 = nnp_min32(v.x), l1 = nnp_min32(v.y), l2 = nnp_min32(v.z);
        const float h0 = nnp_max32(v.x), h1 = nnp_max32(v.y), h2 = nnp_max32(v.z);
        if ((lane & 31) == 0) {
            float* b = sqbox[2 * i + (lane >> 5)];
            b[0] = l0; b[1] = l1; b[2] = l2; b[3] = h0; b[4] = h1; b[5] = h2;
        }
        wlo[0] = fminf(wlo[0], l0); wlo[1] = fminf(wlo[1], l1); wlo[2] = fminf(wlo[2], l2);
        whi[0] = fmaxf(whi[0], h0); whi[1] = fmaxf(whi[1], h1); whi[2] = fmaxf(whi[2], h2);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { wlo[a] = nnf_uniform(nnp_min64(wlo[a])); whi[a] = nnf_uniform(nnp_max64(whi[a])); }
    // Visiting order: start at the tile whose box is nearest to the wave's centre (one argmin), then outward in sorted order,
    // t*, t* + 1, t* - 1, t* + 2, ... -- neighbours in Morton order are neighbours in space more often than not, every query
    // already holds a real candidate (below), and a tile that cannot matter costs one box test (a nearest-first order by repeated
    // argmin cost 30 % of the kernel).
    // With at most 64 tiles (one per lane: small databases, where a badly chosen early tile costs a seventh of the work) the
    // order is nearest first throughout: tiles whose box touches the wave's box by centre distance, then the others by ascending
    // lower bound -- the first of those out of every query's reach ends the search.
    int t_start;
    const bool few = n_tiles <= 64;
    float key_own = INFINITY;                       // (few) visiting key of tile `lane`; +INF: visited / none
    {
        const float ctr[3] = {0.5f * (wlo[0] + whi[0]), 0.5f * (wlo[1] + whi[1]), 0.5f * (wlo[2] + whi[2])};
        float m = INFINITY;
        int mt = 0x7fffffff;
        for (int t = lane; t < n_tiles; t += 64) {
            const float* tb = tbn + (size_t)t * 6;
            const float tlo[3] = {tb[0], tb[1], tb[2]}, thi[3] = {tb[3], tb[4], tb[5]};
            float c2, md2, lb2;
            nnp_box_bounds(ctr, ctr, tlo, thi, c2, md2);
            nnp_box_bounds(wlo, whi, tlo, thi, lb2, md2);
            c2 = (c2 == c2) ? fminf(c2, 3.0e38f) : 3.0e38f;
            if (c2 < m) { m = c2; mt = t; }
            const float key = lb2 > 0.f ? fminf(lb2, 3.0e38f) : -1.f / (1.f + c2);
            key_own = (key == key) ? key : -2.f;    // (few: this loop runs once per lane) a NaN bound: visit first, never skip
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float om = __shfl_xor(m, o, 64);
            const int ot = __shfl_xor(mt, o, 64);
            if (om < m || (om == m && ot < mt)) { m = om; mt = ot; }
        }
        t_start = __builtin_amdgcn_readfirstlane(mt);
        if (t_start >= n_tiles) t_start = 0;
    }
    // a lane's part in the group tests of a tile: group g = lane & 15 of 16, query tiles 2 r and 2 r + 1 (r = lane >> 4: its row of 16 lanes)
    // Every query starts from a real candidate: the nearest of one representative per tile (its first sorted point), evaluated with
    // the exact arithmetic and the (distance, original index) order -- so `best` is a finite upper bound of the query's minimum from
    // the first tile on, and a far query (whose box-level bounds are hopeless: its 32-query tile spans a metre) prunes by its OWN
    // distance.  n_tiles x 4 evaluations per lane: ~3 % of what a wave spends otherwise.
    for (int t = 0; t < n_tiles; ++t) {
        const float4 p = dn[(size_t)t * NNP_TILE];
        const int o = __float_as_int(p.w);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const float d = nn_dist2(ox[i], oy[i], oz[i], p.x, p.y, p.z);
            if (d < best[i] || (d == best[i] && o < bi[i])) { best[i] = d; bi[i] = o; }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    NNP_CLK0();
    // upper bounds: the largest exact minimum the 32 queries of a query tile hold so far (refreshed after every exact pass)
    float ub0, ub1, wub;
    auto refresh_bounds = [&]() {
        float ubb[2 * NQ];
        wub = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const float mx = nnp_max32(best[i]);
            ubb[2 * i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mx), 0));
            ubb[2 * i + 1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mx), 32));
            wub = fmaxf(wub, fmaxf(ubb[2 * i], ubb[2 * i + 1]));
        }
        wub = nnf_uniform(wub);
        const int r_own = nnf_again(lane) >> 4;
        ub0 = r_own == 0 ? ubb[0] : (r_own == 1 ? ubb[2] : (r_own == 2 ? ubb[4] : ubb[6]));
        ub1 = r_own == 0 ? ubb[1] : (r_own == 1 ? ubb[3] : (r_own == 2 ? ubb[5] : ubb[7]));
    };
    refresh_bounds();
    for (int step = 0; step < 2 * n_tiles; ++step) {
        int t;
        float key_t = -1.f;
        if (few) {                                      // nearest first: the unvisited tile with the smallest key
            float m = key_own;
            int mt = lane;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float om = __shfl_xor(m, o, 64);
                const int ot = __shfl_xor(mt, o, 64);
                if (om < m || (om == m && ot < mt)) { m = om; mt = ot; }
            }
            t = __builtin_amdgcn_readfirstlane(mt);
            key_t = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m)));
            if (!(key_t < INFINITY)) break;             // every tile has been looked at
            if (lane == t) key_own = INFINITY;
        } else {
            // t* , t* + 1, t* - 1, t* + 2, ...: offset (step + 1) / 2, sign by parity; steps outside [0, n_tiles) are passed over
            t = t_start + ((step & 1) ? ((step + 1) >> 1) : -(step >> 1));
            if (t < 0 || t >= n_tiles) continue;
        }
        const float* tb = tbn + (size_t)t * 6;
        const float tlo[3] = {tb[0], tb[1], tb[2]}, thi[3] = {tb[3], tb[4], tb[5]};
        {
            float lbt, mdt;
            nnp_box_bounds(wlo, whi, tlo, thi, lbt, mdt);
            if (lbt * NNP_DOWN > wub) {                  // (uniform) out of every query's reach
                if (few && key_t > 0.f) break;           // ... and so is every tile after it in ascending lower-bound order
                continue;
            }
            // per query: can any point of the tile's box be, or tie with, this query's minimum?
            bool any = false;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const float q[3] = {ox[i], oy[i], oz[i]};
                nnp_box_bounds(q, q, tlo, thi, lbt, mdt);
                any = any || !(lbt * NNP_DOWN > best[i]);
            }
            NNP_CLK(8);
            if (__ballot(any) == 0ull) continue;
        }
        NNP_STAT(1, 1);
        // stage the tile: framed f32 points + original indices, padded with far points to 512
        const int t0 = t * NNP_TILE, cnt = min(NNP_TILE, Pd - t0);
        NnfFrame f;
        {
            float r2 = 0.f, c[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                c[a] = nnf_axis_centre(tlo[a], thi[a]);
                const float r = fmaxf(thi[a] - c[a], c[a] - tlo[a]);
                r2 = __builtin_fmaf(r, r, r2);
            }
            f.cx = nnf_uniform(c[0]); f.cy = nnf_uniform(c[1]); f.cz = nnf_uniform(c[2]);
            f.rp = nnf_uniform(sqrtf(r2) * 1.0001f);
        }
        __builtin_amdgcn_wave_barrier();                // (LDS: every lane is done reading the previous tile)
        const int lane_s = nnf_again(lane);             // (index arithmetic formed here, not kept live -- and spilled -- across the loop)
#pragma unroll
        for (int j = 0; j < NNP_TILE / 64; ++j) {
            const int k = lane_s + j * 64;
            float x = NN_FAR, y = NN_FAR, z = NN_FAR;
            int o = INT_MAX;
            if (k < cnt) {
                const float4 v = dn[t0 + k];
                x = v.x; y = v.y; z = v.z; o = __float_as_int(v.w);
            }
            nnf_store_point(sdb, k, x, y, z, f);
            sorig[k] = o;
        }
        // group g_own of the tile against query tiles 2 r_own, 2 r_own + 1 (box against box: 128 tests, two per lane)
        bool a0, a1;
        {
            const int g_own = lane_s & 15, r_own = lane_s >> 4;
            const float* gb = gbn + ((size_t)t * NNP_GPT + g_own) * 6;
            const float glo[3] = {gb[0], gb[1], gb[2]}, ghi[3] = {gb[3], gb[4], gb[5]};
            if (lane < NNP_GPT) {
#pragma unroll
                for (int a = 0; a < 3; ++a) { sgb[lane][a] = glo[a]; sgb[lane][3 + a] = ghi[a]; }
            }
            const float* qb0 = sqbox[2 * r_own];
            const float* qb1 = sqbox[2 * r_own + 1];
            const float q0lo[3] = {qb0[0], qb0[1], qb0[2]}, q0hi[3] = {qb0[3], qb0[4], qb0[5]};
            const float q1lo[3] = {qb1[0], qb1[1], qb1[2]}, q1hi[3] = {qb1[3], qb1[4], qb1[5]};
            float lb0, md0, lb1, md1;
            nnp_box_bounds(q0lo, q0hi, glo, ghi, lb0, md0);
            nnp_box_bounds(q1lo, q1hi, glo, ghi, lb1, md1);
            const bool nonempty = glo[0] <= ghi[0];
            a0 = nonempty && !(lb0 * NNP_DOWN > ub0);          // (a NaN bound never skips)
            a1 = nonempty && !(lb1 * NNP_DOWN > ub1);
        }
        const unsigned long long b0 = __ballot(a0), b1 = __ballot(a1);
        unsigned todo = 0;
        {
            const unsigned long long u = b0 | b1;
            todo = (unsigned)((u | (u >> 16) | (u >> 32) | (u >> 48)) & 0xffffull);
        }
        NNP_STAT(todo ? 2 : 3, 1);
        NNP_STAT(5, __popc(todo));
        NNP_STAT(4, __popcll(b0) + __popcll(b1));
        NNP_CLK(9);
        if (todo == 0u) continue;                       // nothing in this tile can matter
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();                // the tile is staged
        float cx[NQ], cy[NQ], cz[NQ], m1[NQ], m2[NQ], m3[NQ];
        int c1[NQ], c2[NQ];
        float rq = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            cx[i] = ox[i] - f.cx; cy[i] = oy[i] - f.cy; cz[i] = oz[i] - f.cz;
            rq = fmaxf(rq, fmaxf(fabsf(cx[i]), fmaxf(fabsf(cy[i]), fabsf(cz[i]))));
        }
        const NnfScales ks = nnf_scales(f.rp, nnp_max64(rq));
        {
            NnfMfmaState<NQ> st;
            nnf_mfma_begin<NQ>(st, ks, cx, cy, cz);
            while (todo) {
                const int g = __ffs((int)todo) - 1;
                todo &= todo - 1;
                unsigned act = 0;                       // bit j: query tile j against group g
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    act |= (unsigned)((b0 >> (16 * r + g)) & 1ull) << (2 * r) | (unsigned)((b1 >> (16 * r + g)) & 1ull) << (2 * r + 1);
                {   // refine per query: a (query tile, group) pair is scanned only if the group's box is within reach of at least
                    // one of the tile's 32 queries by that query's OWN upper bound
                    const float bl[3] = {sgb[g][0], sgb[g][1], sgb[g][2]}, bh[3] = {sgb[g][3], sgb[g][4], sgb[g][5]};
                    unsigned ract = 0;
#pragma unroll
                    for (int i = 0; i < NQ; ++i) {
                        if (!(act & (3u << (2 * i)))) continue;         // (uniform)
                        const float q[3] = {ox[i], oy[i], oz[i]};
                        float lb2, md2;
                        nnp_box_bounds(q, q, bl, bh, lb2, md2);
                        const unsigned long long bb = __ballot(!(lb2 * NNP_DOWN > best[i]));
                        ract |= ((bb & 0xffffffffull) ? 1u : 0u) << (2 * i) | ((bb >> 32) ? 2u : 0u) << (2 * i);
                    }
                    act &= ract;
                    NNP_STAT(6, __popc(act));
                    if (act == 0u) continue;
                    NNP_STAT(7, 1);
                }
                nnf_mfma_tiles<NQ, false, true>(st, sdb, NNP_TILE / NNF_CH, g, g + 1, 0, ks, act);
            }
            nnf_mfma_end<NQ>(st, ks, m1, m2, m3, c1, c2);
        }
        NNP_CLK(11);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {       // framed queries again (same bits): not kept live across the matrix-core pass
            asm volatile("" : "+v"(ox[i]), "+v"(oy[i]), "+v"(oz[i]));
            cx[i] = ox[i] - f.cx; cy[i] = oy[i] - f.cy; cz[i] = oz[i] - f.cz;
        }
        nnf_resolve<NQ, true, true>(sdb, NNP_TILE / NNF_CH, f, 0, cx, cy, cz, ox, oy, oz, m1, m2, m3, c1, c2, best, bi, sorig,
                                    nnf_mfma_error_floor(f, ks));
        refresh_bounds();
        NNP_CLK(12);
    }
    NNP_CLK_END();
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = q0 + i * 64 + nnf_again(lane);
        if (q < Pq) {
            const int oq = __float_as_int(qn[q].w);
            if (dist) dist[(size_t)n * Pq + oq] = best[i];
            if (idx) idx[(size_t)n * Pq + oq] = bi[i];
        }
    }
}
