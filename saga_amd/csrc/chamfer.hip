// chamfer_distance.ChamferDistance(x, y) -> (dist1, dist2, idx1, idx2): brute-force bidirectional nearest neighbour.
// Replaces the third-party CUDA extension called at utils/train_helper.py:57.
//
// Tiled scan (nn_scan.h): the "database" cloud is staged through LDS in paired tiles and read back with broadcast
// ds_read_b128; each lane owns NQ query points in registers; packed FP32 math, per-chunk minima and one re-scan of the
// winning chunk give exactly the (best, index) of a sequential strict-'<' scan in index order (lowest index wins
// ties).  Squared distance = fma(dz,dz, fma(dy,dy, dx*dx)), bit-exact against oracle/csrc/nn_ref.c.  FP32 VALU bound.
#include <math.h>
#include <stdlib.h>

#include "saga_common.h"
#include "nn_scan.h"
#include "nn_fast.h"
#include "nn_prune.h"

void* saga_prof_begin(int id, void* stream);
void saga_prof_end(void* scope);

#define NN_TILE 2048

template <int NQ, int NTH>
__global__ __launch_bounds__(NTH) void nn_scan_kernel(const float* __restrict__ q, const float* __restrict__ db,
                                                      float* __restrict__ dist, int32_t* __restrict__ idx, int Pq, int Pd) {
    __shared__ __attribute__((aligned(16))) float sdb[NN_TILE_FLOATS(NN_TILE)];
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * (NTH * NQ);
    const float* qn = q + (size_t)n * Pq * 3;
    const float* dn = db + (size_t)n * Pd * 3;

    float qx[NQ], qy[NQ], qz[NQ], best[NQ];
    int bc[NQ], bi[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int p = min(q0 + tid + i * NTH, Pq - 1);
        qx[i] = qn[p * 3]; qy[i] = qn[p * 3 + 1]; qz[i] = qn[p * 3 + 2];
        best[i] = INFINITY;
        bc[i] = -1;
        bi[i] = 0;
    }
    for (int t0 = 0; t0 < Pd; t0 += NN_TILE) {
        const int cnt = min(NN_TILE, Pd - t0);
        if (t0) __syncthreads();
        const int cnt_pad = nn_stage_tile(sdb, dn + (size_t)t0 * 3, cnt, tid, NTH);
        __syncthreads();
        nn_scan_tile<NQ>(sdb, cnt_pad, t0, qx, qy, qz, best, bc);
        nn_resolve_tile<NQ>(sdb, t0, qx, qy, qz, best, bc, bi);
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int p = q0 + tid + i * NTH;
        if (p < Pq) {
            if (dist) dist[(size_t)n * Pq + p] = best[i];
            if (idx) idx[(size_t)n * Pq + p] = bi[i];
        }
    }
}

// Two-level scan of nn_fast.h, tile by tile: approximate expanded-form pass (3 fmas per pair) + exact re-evaluation of
// the candidate chunks of every tile that can still hold the minimum (each tile has its own centred frame; the exact
// minimum and its index are carried across tiles with a strict '<', tiles in index order).  Same outputs as
// nn_scan_kernel, bit for bit.  One tile for every call SAGA makes (2,048 object points / 778 hand vertices); dense
// clouds (BASELINE config 5: 100,000 points) stream through LDS 2,048 points at a time.
// MF: approximate pass on the matrix cores (nnf_scan_mfma) instead of the vector units (nnf_scan); same results.
template <int NQ, int NTH, bool MF>
__global__ __launch_bounds__(NTH, NTH == 256 ? (MF ? 3 : 2) : 1) void nn_fast_kernel(     // (the vector-unit pass is the cross-check path: two blocks per CU, no spills)
const float* __restrict__ q, const float* __restrict__ db,
                                                      float* __restrict__ dist, int32_t* __restrict__ idx, int Pq, int Pd) {
    __shared__ __attribute__((aligned(16))) float sdb[NNF_TILE_FLOATS(NN_TILE)];
    __shared__ float sred[6 * (NTH / 64)];
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * (NTH * NQ);
    const float* qn = q + (size_t)n * Pq * 3;
    const float* dn = db + (size_t)n * Pd * 3;
    float ox[NQ], oy[NQ], oz[NQ], best[NQ];
    int bi[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int p = min(q0 + tid + i * NTH, Pq - 1);
        ox[i] = qn[p * 3]; oy[i] = qn[p * 3 + 1]; oz[i] = qn[p * 3 + 2];
        best[i] = INFINITY; bi[i] = 0;
    }
    for (int t0 = 0; t0 < Pd; t0 += NN_TILE) {
        const int cnt = min(NN_TILE, Pd - t0);
        const NnfFrame f = nnf_frame(dn + (size_t)t0 * 3, cnt, sred, tid, NTH);      // (barriers: the previous tile is done)
        const int cnt_pad = nnf_stage_tile(sdb, dn + (size_t)t0 * 3, cnt, f, tid, NTH);
        float qx[NQ], qy[NQ], qz[NQ], m1[NQ], m2[NQ], m3[NQ];
        int c1[NQ], c2[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) { qx[i] = ox[i] - f.cx; qy[i] = oy[i] - f.cy; qz[i] = oz[i] - f.cz; }
        __syncthreads();
        float e_floor = 0.f;
        if (MF) {
            nnf_scan_mfma<NQ>(sdb, cnt_pad / NNF_CH, f, qx, qy, qz, m1, m2, m3, c1, c2, e_floor);
            // the framed queries are formed again (same subtraction, same bits) instead of staying live across the matrix-core pass
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                asm volatile("" : "+v"(ox[i]), "+v"(oy[i]), "+v"(oz[i]));
                qx[i] = ox[i] - f.cx; qy[i] = oy[i] - f.cy; qz[i] = oz[i] - f.cz;
            }
        } else nnf_scan<NQ>(sdb, cnt_pad / NNF_CH, qx, qy, qz, m1, m2, m3, c1, c2);
        nnf_resolve<NQ, MF>(sdb, cnt_pad / NNF_CH, f, t0, qx, qy, qz, ox, oy, oz, m1, m2, m3, c1, c2, best, bi, nullptr, e_floor);
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int p = q0 + tid + i * NTH;
        if (p < Pq) {
            if (dist) dist[(size_t)n * Pq + p] = best[i];
            if (idx) idx[(size_t)n * Pq + p] = bi[i];
        }
    }
}

static int launch_scan(const float* q, const float* db, float* dist, int32_t* idx, int N, int Pq, int Pd, hipStream_t st) {
    if (!dist && !idx) return 0;
    // SAGA_NN_EXACT_SCAN=1 forces the plain 6-op exact scan (cross-checks of the two code paths)
    const char* force = getenv("SAGA_NN_EXACT_SCAN");
    // SAGA_NN_VALU_PASS=1 keeps the two-level scan but runs its approximate pass on the vector units (cross-check of the two passes)
    static const bool valu_pass = getenv("SAGA_NN_VALU_PASS") != nullptr;
    if (!(force && force[0] == '1')) {
        // 4 queries per lane for large query clouds (8 would need > 128 VGPRs: the scan wants 3-4 waves per SIMD more than it
        // wants fewer broadcast LDS reads); clouds of more than 1,024 queries take several blocks per batch element
        const dim3 grid4((Pq + 256 * 4 - 1) / (256 * 4), N);
#define NNF_LAUNCH(MF)                                                                                        \
        if (Pq <= 128) nn_fast_kernel<2, 64, MF><<<dim3(1, N), 64, 0, st>>>(q, db, dist, idx, Pq, Pd);           \
        else if (Pq <= 256) nn_fast_kernel<2, 128, MF><<<dim3(1, N), 128, 0, st>>>(q, db, dist, idx, Pq, Pd);    \
        else if (Pq <= 512) nn_fast_kernel<2, 256, MF><<<dim3(1, N), 256, 0, st>>>(q, db, dist, idx, Pq, Pd);    \
        else nn_fast_kernel<4, 256, MF><<<grid4, 256, 0, st>>>(q, db, dist, idx, Pq, Pd);
        if (valu_pass) { NNF_LAUNCH(false) } else { NNF_LAUNCH(true) }
#undef NNF_LAUNCH
        SAGA_LAUNCH_CHECK();
        return 0;
    }
    if (Pq <= 128) {
        nn_scan_kernel<2, 64><<<dim3(1, N), 64, 0, st>>>(q, db, dist, idx, Pq, Pd);
    } else if (Pq <= 256) {
        nn_scan_kernel<2, 128><<<dim3(1, N), 128, 0, st>>>(q, db, dist, idx, Pq, Pd);
    } else if (Pq <= 512) {
        nn_scan_kernel<2, 256><<<dim3(1, N), 256, 0, st>>>(q, db, dist, idx, Pq, Pd);
    } else if (Pq <= 1024) {
        nn_scan_kernel<4, 256><<<dim3(1, N), 256, 0, st>>>(q, db, dist, idx, Pq, Pd);
    } else {
        dim3 grid((Pq + 256 * 8 - 1) / (256 * 8), N);
        nn_scan_kernel<8, 256><<<grid, 256, 0, st>>>(q, db, dist, idx, Pq, Pd);
    }
    SAGA_LAUNCH_CHECK();
    return 0;
}

extern "C" int saga_chamfer_fwd(const float* x, const float* y, float* dist1, float* dist2, int32_t* idx1, int32_t* idx2,
                                int N, int P1, int P2, void* stream) {
    SAGA_CHECK_ARG(N >= 0 && P1 >= 0 && P2 >= 0, "saga_chamfer_fwd: negative size");
    if (N == 0 || (P1 == 0 && P2 == 0)) return 0;
    SAGA_CHECK_ARG(P1 > 0 && P2 > 0, "saga_chamfer_fwd: empty cloud (P1=%d, P2=%d) has no nearest neighbour", P1, P2);
    SAGA_CHECK_ARG(x && y, "saga_chamfer_fwd: null cloud");
    SAGA_CHECK_ARG(N <= 65535, "saga_chamfer_fwd: batch %d exceeds 65535", N);
    void* prof = saga_prof_begin(SAGA_PROF_CHAMFER, stream);
    int rc = launch_scan(x, y, dist1, idx1, N, P1, P2, (hipStream_t)stream);
    if (!rc) rc = launch_scan(y, x, dist2, idx2, N, P2, P1, (hipStream_t)stream);
    saga_prof_end(prof);
    return rc;
}

// ---- large clouds: the pruned search of nn_prune.h ------------------------------------------------------------------------
// A direction (queries -> database) is pruned when the database has at least NNP_MIN_DB points (below that the brute-force scan's
// two or three tiles cost less than sorting) and at most NNP_MAX_TILES tiles.  SAGA_NN_PRUNE=0 switches it off (cross-check).
#define NNP_MIN_DB 4096
static bool nnp_direction(int Pd) {
    const char* e = getenv("SAGA_NN_PRUNE");
    if (e && e[0] == '0') return false;
    const char* f = getenv("SAGA_NN_EXACT_SCAN");
    if (f && f[0] == '1') return false;
    return Pd >= NNP_MIN_DB && (Pd + NNP_TILE - 1) / NNP_TILE <= NNP_MAX_TILES;
}
struct NnpLayout { size_t sx, sy, gbx, tbx, gby, tby, total; int tx, ty; bool px, py; };      // byte offsets; px: database x is pruned (y -> x)
static NnpLayout nnp_layout(int N, int P1, int P2) {
    NnpLayout L;
    memset(&L, 0, sizeof(L));
    L.px = nnp_direction(P1); L.py = nnp_direction(P2);
    if (!L.px && !L.py) return L;
    L.tx = (P1 + NNP_TILE - 1) / NNP_TILE; L.ty = (P2 + NNP_TILE - 1) / NNP_TILE;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    L.sx = take((size_t)N * P1 * sizeof(float4));
    L.sy = take((size_t)N * P2 * sizeof(float4));
    if (L.px) { L.gbx = take((size_t)N * L.tx * NNP_GPT * 6 * sizeof(float)); L.tbx = take((size_t)N * L.tx * 6 * sizeof(float)); }
    if (L.py) { L.gby = take((size_t)N * L.ty * NNP_GPT * 6 * sizeof(float)); L.tby = take((size_t)N * L.ty * 6 * sizeof(float)); }
    L.total = off;
    return L;
}

extern "C" size_t saga_chamfer_workspace_bytes(int N, int P1, int P2) {
    if (N <= 0 || P1 <= 0 || P2 <= 0) return 0;
    return nnp_layout(N, P1, P2).total;
}

extern "C" int saga_chamfer_fwd_ws(const float* x, const float* y, float* dist1, float* dist2, int32_t* idx1, int32_t* idx2, int N,
                                   int P1, int P2, void* ws, size_t ws_bytes, void* stream) {
    SAGA_CHECK_ARG(N >= 0 && P1 >= 0 && P2 >= 0, "saga_chamfer_fwd_ws: negative size");
    if (N == 0 || (P1 == 0 && P2 == 0)) return 0;
    SAGA_CHECK_ARG(P1 > 0 && P2 > 0, "saga_chamfer_fwd_ws: empty cloud (P1=%d, P2=%d) has no nearest neighbour", P1, P2);
    const NnpLayout L = nnp_layout(N, P1, P2);
    if (L.total == 0 || !ws) return saga_chamfer_fwd(x, y, dist1, dist2, idx1, idx2, N, P1, P2, stream);
    SAGA_CHECK_ARG(ws_bytes >= L.total, "saga_chamfer_fwd_ws: workspace of %zu bytes, saga_chamfer_workspace_bytes says %zu", ws_bytes, L.total);
    SAGA_CHECK_ARG(((uintptr_t)ws & 15) == 0, "saga_chamfer_fwd_ws: the workspace must be 16-byte aligned");
    SAGA_CHECK_ARG(x && y, "saga_chamfer_fwd_ws: null cloud");
    SAGA_CHECK_ARG(N <= 65535, "saga_chamfer_fwd_ws: batch %d exceeds 65535", N);
    hipStream_t st = (hipStream_t)stream;
    {
        int dev = 0;
        SAGA_HIP(hipGetDevice(&dev));
        static bool attr_set[64] = {};
        if (!attr_set[dev & 63]) {
            SAGA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&nnp_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         NNP_CELLS * (int)sizeof(unsigned int)));
            attr_set[dev & 63] = true;
        }
    }
    char* base = (char*)ws;
    float4* sx = (float4*)(base + L.sx);
    float4* sy = (float4*)(base + L.sy);
    void* prof = saga_prof_begin(SAGA_PROF_CHAMFER, stream);
    nnp_sort_kernel<<<N, NNP_SORT_THREADS, NNP_CELLS * sizeof(unsigned int), st>>>(x, sx, P1);
    nnp_sort_kernel<<<N, NNP_SORT_THREADS, NNP_CELLS * sizeof(unsigned int), st>>>(y, sy, P2);
    if (L.px) nnp_boxes_kernel<<<dim3((L.tx + 3) / 4, N), 256, 0, st>>>(sx, (float*)(base + L.gbx), (float*)(base + L.tbx), P1, L.tx);
    if (L.py) nnp_boxes_kernel<<<dim3((L.ty + 3) / 4, N), 256, 0, st>>>(sy, (float*)(base + L.gby), (float*)(base + L.tby), P2, L.ty);
    int rc = 0;
    if (dist1 || idx1) {
        if (L.py)
            nnp_scan_kernel<<<dim3((P1 + NNP_QPB - 1) / NNP_QPB, N), NNP_THREADS, 0, st>>>(sx, sy, (const float*)(base + L.gby),
                                                                                         (const float*)(base + L.tby), dist1, idx1, P1, P2, L.ty);
        else
            rc = launch_scan(x, y, dist1, idx1, N, P1, P2, st);
    }
    if (!rc && (dist2 || idx2)) {
        if (L.px)
            nnp_scan_kernel<<<dim3((P2 + NNP_QPB - 1) / NNP_QPB, N), NNP_THREADS, 0, st>>>(sy, sx, (const float*)(base + L.gbx),
                                                                                         (const float*)(base + L.tbx), dist2, idx2, P2, P1, L.tx);
        else
            rc = launch_scan(y, x, dist2, idx2, N, P2, P1, st);
    }
    saga_prof_end(prof);
    if (rc) return rc;
    SAGA_LAUNCH_CHECK();
    return 0;
}

// Backward of the two squared-distance outputs (the extension's NmDistanceGrad): for every point p of a with neighbour
// k in b: g = 2 * g_dist[p] * (a_p - b_k); g_a[p] += g; g_b[k] -= g -- and the same with the roles swapped.  The CUDA
// extension scatters with float atomicAdd (many p share one k), whose result depends on the order the atomics land in.
// Here it is a GATHER: one lane owns one point p of a cloud and adds, in a fixed order, its own term and then the terms of
// the other cloud's points q whose neighbour is p, in ascending q (the other cloud's neighbour indices stream through LDS,
// four per broadcast read; a match is rare, so the loop is one compare per pair).  No atomics: bit-reproducible, and a
// batch element's gradient does not depend on the batch around it.
#define NG_THREADS 256
#define NG_TILE 4096
__global__ __launch_bounds__(NG_THREADS) void nn_grad_gather_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                    const int32_t* __restrict__ idx_ab, const float* __restrict__ g_dist_a,
                                                                    const int32_t* __restrict__ idx_ba, const float* __restrict__ g_dist_b,
                                                                    float* __restrict__ g_a, int Pa, int Pb) {
    __shared__ __attribute__((aligned(16))) int32_t sidx[NG_TILE];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int p = blockIdx.x * NG_THREADS + tid;
    const bool live = p < Pa;
    const float* an = a + (size_t)n * Pa * 3;
    const float* bn = b + (size_t)n * Pb * 3;
    float ax = 0.f, ay = 0.f, az = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
    if (live) {
        ax = an[p * 3]; ay = an[p * 3 + 1]; az = an[p * 3 + 2];
        if (g_dist_a) {
            const int k = idx_ab[(size_t)n * Pa + p];
            const float g = 2.0f * g_dist_a[(size_t)n * Pa + p];
            sx = g * (ax - bn[k * 3]); sy = g * (ay - bn[k * 3 + 1]); sz = g * (az - bn[k * 3 + 2]);
        }
    }
    if (g_dist_b) {
        const int32_t* ib = idx_ba + (size_t)n * Pb;
        const float* gb = g_dist_b + (size_t)n * Pb;
        const int me = live ? p : -2;
        for (int t0 = 0; t0 < Pb; t0 += NG_TILE) {
            const int cnt = min(NG_TILE, Pb - t0);
            if (t0) __syncthreads();
            for (int i = tid; i < NG_TILE; i += NG_THREADS) sidx[i] = i < cnt ? ib[t0 + i] : -1;
            __syncthreads();
            const int cnt4 = (cnt + 3) & ~3;
            for (int j = 0; j < cnt4; j += 4) {
                const int4 v = *reinterpret_cast<const int4*>(sidx + j);
                if (v.x == me || v.y == me || v.z == me || v.w == me) {
                    const int vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (vv[c] == me) {
                            const int q = t0 + j + c;
                            const float g = 2.0f * gb[q];
                            sx -= g * (bn[q * 3] - ax); sy -= g * (bn[q * 3 + 1] - ay); sz -= g * (bn[q * 3 + 2] - az);
                        }
                }
            }
        }
    }
    if (live) {
        float* o = g_a + ((size_t)n * Pa + p) * 3;
        o[0] += sx; o[1] += sy; o[2] += sz;
    }
}

extern "C" int saga_chamfer_bwd(const float* x, const float* y, const int32_t* idx1, const int32_t* idx2, const float* g_dist1,
                                const float* g_dist2, float* g_x, float* g_y, int N, int P1, int P2, void* stream) {
    SAGA_CHECK_ARG(N >= 0 && P1 >= 0 && P2 >= 0, "saga_chamfer_bwd: negative size");
    if (N == 0 || P1 == 0 || P2 == 0) return 0;
    SAGA_CHECK_ARG(x && y && g_x && g_y, "saga_chamfer_bwd: null argument");
    SAGA_CHECK_ARG(N <= 65535, "saga_chamfer_bwd: batch %d exceeds 65535", N);
    SAGA_CHECK_ARG(!g_dist1 || idx1, "saga_chamfer_bwd: idx1 missing");
    SAGA_CHECK_ARG(!g_dist2 || idx2, "saga_chamfer_bwd: idx2 missing");
    if (!g_dist1 && !g_dist2) return 0;
    hipStream_t st = (hipStream_t)stream;
    nn_grad_gather_kernel<<<dim3((P1 + NG_THREADS - 1) / NG_THREADS, N), NG_THREADS, 0, st>>>(x, y, idx1, g_dist1, idx2, g_dist2, g_x, P1, P2);
    SAGA_LAUNCH_CHECK();
    nn_grad_gather_kernel<<<dim3((P2 + NG_THREADS - 1) / NG_THREADS, N), NG_THREADS, 0, st>>>(y, x, idx2, g_dist2, idx1, g_dist1, g_y, P2, P1);
    SAGA_LAUNCH_CHECK();
    return 0;
}

// ---- point2point_signed (utils/train_helper.py:17-86) as ONE operator ------------------------------------------------------
// Forward: the bidirectional nearest-neighbour search above, then per point the difference vector to its neighbour, its norm
// and the sign of its dot product with the OTHER cloud's normal at the neighbour (:65-81; no normals given: unsigned).
// Same operation order as the reference's torch code: norm = sqrt((dx dx + dy dy) + dz dz), dot = (nx dx + ny dy) + nz dz.
__global__ void p2p_signed_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ b_normals,
                                  const int32_t* __restrict__ idx, float* __restrict__ signed_dist, float* __restrict__ vec, int N, int Pa,
                                  int Pb) {
#pragma clang fp contract(off)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * Pa) return;
    const int n = (int)(i / Pa);
    const size_t k = (size_t)n * Pb + idx[i];
    const float dx = a[i * 3] - b[k * 3], dy = a[i * 3 + 1] - b[k * 3 + 1], dz = a[i * 3 + 2] - b[k * 3 + 2];
    float d = sqrtf((dx * dx + dy * dy) + dz * dz);
    if (b_normals) {
        const float dot = (b_normals[k * 3] * dx + b_normals[k * 3 + 1] * dy) + b_normals[k * 3 + 2] * dz;
        d *= (dot > 0.f) ? 1.f : ((dot < 0.f) ? -1.f : 0.f);
    }
    signed_dist[i] = d;
    if (vec) { vec[i * 3] = dx; vec[i * 3 + 1] = dy; vec[i * 3 + 2] = dz; }
}

extern "C" int saga_p2p_signed_fwd(const float* x, const float* y, const float* x_normals, const float* y_normals, float* y2x_signed,
                                   float* x2y_signed, int32_t* yidx, int32_t* xidx, float* y2x_vec, float* x2y_vec, int N, int P1, int P2,
                                   void* stream) {
    return saga_p2p_signed_fwd_ws(x, y, x_normals, y_normals, y2x_signed, x2y_signed, yidx, xidx, y2x_vec, x2y_vec, N, P1, P2, nullptr, 0,
                                  stream);
}

// ws: NULL, or saga_chamfer_workspace_bytes(N, P1, P2) bytes -- a direction whose database cloud is large is then searched with the
// pruned search of saga_chamfer_fwd_ws (same indices, bit for bit)
extern "C" int saga_p2p_signed_fwd_ws(const float* x, const float* y, const float* x_normals, const float* y_normals, float* y2x_signed,
                                      float* x2y_signed, int32_t* yidx, int32_t* xidx, float* y2x_vec, float* x2y_vec, int N, int P1,
                                      int P2, void* ws, size_t ws_bytes, void* stream) {
    SAGA_CHECK_ARG(N >= 0 && P1 >= 0 && P2 >= 0, "saga_p2p_signed_fwd: negative size");
    if (N == 0 || (P1 == 0 && P2 == 0)) return 0;
    SAGA_CHECK_ARG(P1 > 0 && P2 > 0, "saga_p2p_signed_fwd: empty cloud (P1=%d, P2=%d) has no nearest neighbour", P1, P2);
    SAGA_CHECK_ARG(x && y, "saga_p2p_signed_fwd: null argument");
    // a caller that reads one direction only passes the other direction's outputs (signed, idx, vec) as NULL: that search is
    // not run.  (GraspMotion-Opt discards the object -> marker direction of its marker call and never reads the hand -> object
    // direction of its hand call: MotionFill/models/fittingop.py:154-158,178-190.)
    const bool do_y2x = y2x_signed != nullptr, do_x2y = x2y_signed != nullptr;
    SAGA_CHECK_ARG(do_y2x || do_x2y, "saga_p2p_signed_fwd: no direction requested");
    SAGA_CHECK_ARG(do_y2x ? yidx != nullptr : (yidx == nullptr && y2x_vec == nullptr),
                   "saga_p2p_signed_fwd: y2x_signed, yidx (and y2x_vec) are given together or all NULL");
    SAGA_CHECK_ARG(do_x2y ? xidx != nullptr : (xidx == nullptr && x2y_vec == nullptr),
                   "saga_p2p_signed_fwd: x2y_signed, xidx (and x2y_vec) are given together or all NULL");
    SAGA_CHECK_ARG(N <= 65535, "saga_p2p_signed_fwd: batch %d exceeds 65535", N);
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    if (ws && nnp_layout(N, P1, P2).total) {
        // (indices only: idx1 = x -> y, idx2 = y -> x; a NULL pair skips that direction)
        rc = saga_chamfer_fwd_ws(x, y, nullptr, nullptr, do_x2y ? xidx : nullptr, do_y2x ? yidx : nullptr, N, P1, P2, ws, ws_bytes, stream);
    } else {
        void* prof = saga_prof_begin(SAGA_PROF_CHAMFER, stream);
        if (do_x2y) rc = launch_scan(x, y, nullptr, xidx, N, P1, P2, st);
        if (!rc && do_y2x) rc = launch_scan(y, x, nullptr, yidx, N, P2, P1, st);
        saga_prof_end(prof);
    }
    if (rc) return rc;
    // x2y = x - y[xidx], signed by y_normals[xidx] (:76-81); y2x = y - x[yidx], signed by x_normals[yidx] (:68-74)
    if (do_x2y) {
        p2p_signed_kernel<<<(unsigned)(((size_t)N * P1 + 255) / 256), 256, 0, st>>>(x, y, y_normals, xidx, x2y_signed, x2y_vec, N, P1, P2);
        SAGA_LAUNCH_CHECK();
    }
    if (do_y2x) {
        p2p_signed_kernel<<<(unsigned)(((size_t)N * P2 + 255) / 256), 256, 0, st>>>(y, x, x_normals, yidx, y2x_signed, y2x_vec, N, P2, P1);
        SAGA_LAUNCH_CHECK();
    }
    return 0;
}

// Backward.  A point p of cloud a receives (i) its own term -- d|a_p - b_k| = sign^2-free: g * sign * unit(a_p - b_k), plus the
// gradient of its difference vector -- and (ii) minus the same quantities of every point q of b whose neighbour is p (the
// reference's gather backward is a scatter-add, :59-63).  Like saga_chamfer_bwd this is a GATHER in a fixed order (own term,
// then ascending q; the other cloud's neighbour indices stream through LDS): no atomics, bit-reproducible, batch-independent.
// sign comes from the forward's signed distance (its sign IS the sign factor; 0 -> no gradient, like torch.sign).
__global__ __launch_bounds__(NG_THREADS) void p2p_grad_gather_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                     const int32_t* __restrict__ idx_ab, const float* __restrict__ sd_a,
                                                                     const float* __restrict__ g_sd_a, const float* __restrict__ g_vec_a,
                                                                     const int32_t* __restrict__ idx_ba, const float* __restrict__ sd_b,
                                                                     const float* __restrict__ g_sd_b, const float* __restrict__ g_vec_b,
                                                                     float* __restrict__ g_a, int Pa, int Pb) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) int32_t sidx[NG_TILE];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int p = blockIdx.x * NG_THREADS + tid;
    const bool live = p < Pa;
    const float* an = a + (size_t)n * Pa * 3;
    const float* bn = b + (size_t)n * Pb * 3;
    auto sgn = [](float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); };
    float ax = 0.f, ay = 0.f, az = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
    if (live) {
        const size_t i = (size_t)n * Pa + p;
        ax = an[p * 3]; ay = an[p * 3 + 1]; az = an[p * 3 + 2];
        if (g_sd_a) {
            const int k = idx_ab[i];
            const float dx = ax - bn[k * 3], dy = ay - bn[k * 3 + 1], dz = az - bn[k * 3 + 2];
            const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz);
            const float c = nrm > 0.f ? g_sd_a[i] * sgn(sd_a[i]) / nrm : 0.f;
            sx = c * dx; sy = c * dy; sz = c * dz;
        }
        if (g_vec_a) { sx += g_vec_a[i * 3]; sy += g_vec_a[i * 3 + 1]; sz += g_vec_a[i * 3 + 2]; }
    }
    if (g_sd_b || g_vec_b) {
        const int32_t* ib = idx_ba + (size_t)n * Pb;
        const int me = live ? p : -2;
        for (int t0 = 0; t0 < Pb; t0 += NG_TILE) {
            const int cnt = min(NG_TILE, Pb - t0);
            if (t0) __syncthreads();
            for (int i = tid; i < NG_TILE; i += NG_THREADS) sidx[i] = i < cnt ? ib[t0 + i] : -1;
            __syncthreads();
            const int cnt4 = (cnt + 3) & ~3;
            for (int j = 0; j < cnt4; j += 4) {
                const int4 v = *reinterpret_cast<const int4*>(sidx + j);
                if (v.x == me || v.y == me || v.z == me || v.w == me) {
                    const int vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4)
                        if (vv[c4] == me) {
                            const int q = t0 + j + c4;
                            const size_t qi = (size_t)n * Pb + q;
                            if (g_sd_b) {
                                const float dx = bn[q * 3] - ax, dy = bn[q * 3 + 1] - ay, dz = bn[q * 3 + 2] - az;
                                const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz);
                                const float c = nrm > 0.f ? g_sd_b[qi] * sgn(sd_b[qi]) / nrm : 0.f;
                                sx -= c * dx; sy -= c * dy; sz -= c * dz;
                            }
                            if (g_vec_b) { sx -= g_vec_b[qi * 3]; sy -= g_vec_b[qi * 3 + 1]; sz -= g_vec_b[qi * 3 + 2]; }
                        }
                }
            }
        }
    }
    if (live) {
        float* o = g_a + ((size_t)n * Pa + p) * 3;
        o[0] = sx; o[1] = sy; o[2] = sz;
    }
}

// The same backward for small target clouds (Pa <= P2P_FX_MAX: every call SAGA makes -- 143 markers / 778 hand vertices against
// 2,048 object points), O(Pa + Pb) per batch element instead of the gather's O(Pa Pb) compares: one block per batch element
// scatter-adds the other cloud's terms into 64-bit FIXED-POINT accumulators in LDS.  Integer adds are associative, so the result
// does not depend on the order the LDS atomics land in (bit-reproducible, batch-independent).  The grid is 2^-k with k chosen per
// batch element from the largest term M so that M Pb 2^k < 2^62: every term is quantised to within 2^-(62 - lp) M (lp = ceil
// log2 Pb; 2^-51 M for 2,048 points), so a target's sum carries an ABSOLUTE error of at most (its number of terms) x that --
// far below float32 rounding of M, but a target whose incoming terms are all much smaller than the batch element's largest term
// M loses relative precision accordingly (below 2^-27 M nothing is left of float32's 24 bits).  A NaN / Inf term sets a flag for
// its target, whose row is written as NaN -- what torch's scatter-add gives.  (Float atomics, the CUDA reference's way, make the
// sum order-dependent.)
#define P2P_FX_MAX 2048
#define P2P_FX_THREADS 256
__global__ __launch_bounds__(P2P_FX_THREADS) void p2p_grad_fixed_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                        const int32_t* __restrict__ idx_ab, const float* __restrict__ sd_a,
                                                                        const float* __restrict__ g_sd_a, const float* __restrict__ g_vec_a,
                                                                        const int32_t* __restrict__ idx_ba, const float* __restrict__ sd_b,
                                                                        const float* __restrict__ g_sd_b, const float* __restrict__ g_vec_b,
                                                                        float* __restrict__ g_a, int Pa, int Pb) {
#pragma clang fp contract(off)
    __shared__ unsigned long long acc[P2P_FX_MAX * 3];
    __shared__ float smax[P2P_FX_THREADS / 64];
    __shared__ unsigned int sbad[P2P_FX_MAX / 32];              // bit p: a non-finite term landed on target p
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* an = a + (size_t)n * Pa * 3;
    const float* bn = b + (size_t)n * Pb * 3;
    auto sgn = [](float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); };
    for (int i = tid; i < Pa * 3; i += P2P_FX_THREADS) acc[i] = 0ull;
    for (int i = tid; i < P2P_FX_MAX / 32; i += P2P_FX_THREADS) sbad[i] = 0u;
    // the term point q of b sends to its neighbour in a (evaluated twice: once for the scale, once to add -- cheaper than storing it)
    auto term = [&](int q, float& tx, float& ty, float& tz, int& p) {
        const size_t qi = (size_t)n * Pb + q;
        p = idx_ba[qi];
        tx = 0.f; ty = 0.f; tz = 0.f;
        if (g_sd_b) {
            const float dx = bn[q * 3] - an[p * 3], dy = bn[q * 3 + 1] - an[p * 3 + 1], dz = bn[q * 3 + 2] - an[p * 3 + 2];
            const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz);
            const float c = nrm > 0.f ? g_sd_b[qi] * sgn(sd_b[qi]) / nrm : 0.f;
            tx = c * dx; ty = c * dy; tz = c * dz;
        }
        if (g_vec_b) { tx += g_vec_b[qi * 3]; ty += g_vec_b[qi * 3 + 1]; tz += g_vec_b[qi * 3 + 2]; }
    };
    const bool scatter = g_sd_b || g_vec_b;
    double inv_scale = 1.0;
    if (scatter) {
        float m = 0.f;
        for (int q = tid; q < Pb; q += P2P_FX_THREADS) {
            float tx, ty, tz; int p;
            term(q, tx, ty, tz, p);
            const float t = fmaxf(fabsf(tx), fmaxf(fabsf(ty), fabsf(tz)));
            m = (t <= 3.0e38f) ? fmaxf(m, t) : m;              // NaN / Inf terms do not set the scale (they poison their target below)
        }
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((tid & 63) == 0) smax[tid >> 6] = m;
        __syncthreads();                                        // (also: acc is zero)
        m = smax[0];
        for (int w = 1; w < P2P_FX_THREADS / 64; ++w) m = fmaxf(m, smax[w]);
        int e = 0;
        if (m > 0.f) (void)frexpf(m, &e);                      // m < 2^e
        int lp = 1;
        while ((1 << lp) < Pb) ++lp;                            // Pb <= 2^lp
        const int k = max(min(61 - e - lp, 1000), -1000);
        const double scale = ldexp(1.0, k);
        inv_scale = ldexp(1.0, -k);
        for (int q = tid; q < Pb; q += P2P_FX_THREADS) {
            float tx, ty, tz; int p;
            term(q, tx, ty, tz, p);
            // a NaN / Inf term adds nothing and flags its target (the row is written as NaN below): never undefined behaviour,
            // never a finite wrong number
            if (!(fabsf(tx) <= 3.0e38f && fabsf(ty) <= 3.0e38f && fabsf(tz) <= 3.0e38f)) {
                atomicOr(&sbad[p >> 5], 1u << (p & 31));
                continue;
            }
            auto fx = [&](float v) { return __double2ll_rn((double)v * scale); };       // |v| <= M: inside the grid
            atomicAdd(&acc[p * 3], (unsigned long long)fx(tx));
            atomicAdd(&acc[p * 3 + 1], (unsigned long long)fx(ty));
            atomicAdd(&acc[p * 3 + 2], (unsigned long long)fx(tz));
        }
    }
    __syncthreads();
    for (int p = tid; p < Pa; p += P2P_FX_THREADS) {
        const size_t i = (size_t)n * Pa + p;
        double sx = 0.0, sy = 0.0, sz = 0.0;
        if (g_sd_a) {
            const int k = idx_ab[i];
            const float dx = an[p * 3] - bn[k * 3], dy = an[p * 3 + 1] - bn[k * 3 + 1], dz = an[p * 3 + 2] - bn[k * 3 + 2];
            const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz);
            const float c = nrm > 0.f ? g_sd_a[i] * sgn(sd_a[i]) / nrm : 0.f;
            sx = c * dx; sy = c * dy; sz = c * dz;
        }
        if (g_vec_a) { sx += (double)g_vec_a[i * 3]; sy += (double)g_vec_a[i * 3 + 1]; sz += (double)g_vec_a[i * 3 + 2]; }
        if (scatter) {
            sx -= (double)(long long)acc[p * 3] * inv_scale;
            sy -= (double)(long long)acc[p * 3 + 1] * inv_scale;
            sz -= (double)(long long)acc[p * 3 + 2] * inv_scale;
        }
        if (scatter && ((sbad[p >> 5] >> (p & 31)) & 1u)) { sx = NAN; sy = NAN; sz = NAN; }
        float* o = g_a + i * 3;
        o[0] = (float)sx; o[1] = (float)sy; o[2] = (float)sz;
    }
}

static int p2p_grad_launch(const float* a, const float* b, const int32_t* idx_ab, const float* sd_a, const float* g_sd_a,
                           const float* g_vec_a, const int32_t* idx_ba, const float* sd_b, const float* g_sd_b, const float* g_vec_b,
                           float* g_a, int N, int Pa, int Pb, hipStream_t st) {
    const bool force_gather = getenv("SAGA_P2P_GATHER") != nullptr;   // cross-check switch (read per call): the O(Pa Pb) gather
    if (Pa <= P2P_FX_MAX && Pb <= (1 << 24) && !force_gather)
        p2p_grad_fixed_kernel<<<N, P2P_FX_THREADS, 0, st>>>(a, b, idx_ab, sd_a, g_sd_a, g_vec_a, idx_ba, sd_b, g_sd_b, g_vec_b, g_a, Pa, Pb);
    else
        p2p_grad_gather_kernel<<<dim3((Pa + NG_THREADS - 1) / NG_THREADS, N), NG_THREADS, 0, st>>>(a, b, idx_ab, sd_a, g_sd_a, g_vec_a,
                                                                                                  idx_ba, sd_b, g_sd_b, g_vec_b, g_a, Pa, Pb);
    SAGA_LAUNCH_CHECK();
    return 0;
}

extern "C" int saga_p2p_signed_bwd(const float* x, const float* y, const int32_t* yidx, const int32_t* xidx, const float* y2x_signed,
                                   const float* x2y_signed, const float* g_y2x_signed, const float* g_x2y_signed,
                                   const float* g_y2x_vec, const float* g_x2y_vec, float* g_x, float* g_y, int N, int P1, int P2,
                                   void* stream) {
    SAGA_CHECK_ARG(N >= 0 && P1 >= 0 && P2 >= 0, "saga_p2p_signed_bwd: negative size");
    if (N == 0 || P1 == 0 || P2 == 0) return 0;
    SAGA_CHECK_ARG(x && y, "saga_p2p_signed_bwd: null argument");
    SAGA_CHECK_ARG((xidx && x2y_signed) || (!g_x2y_signed && !g_x2y_vec),
                   "saga_p2p_signed_bwd: gradients of the x -> y outputs need xidx and x2y_signed");
    SAGA_CHECK_ARG((yidx && y2x_signed) || (!g_y2x_signed && !g_y2x_vec),
                   "saga_p2p_signed_bwd: gradients of the y -> x outputs need yidx and y2x_signed");
    SAGA_CHECK_ARG(g_x || g_y, "saga_p2p_signed_bwd: no gradient requested");
    SAGA_CHECK_ARG(N <= 65535, "saga_p2p_signed_bwd: batch %d exceeds 65535", N);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (g_x && (rc = p2p_grad_launch(x, y, xidx, x2y_signed, g_x2y_signed, g_x2y_vec, yidx, y2x_signed, g_y2x_signed, g_y2x_vec, g_x, N,
                                     P1, P2, st)))
        return rc;
    if (g_y && (rc = p2p_grad_launch(y, x, yidx, y2x_signed, g_y2x_signed, g_y2x_vec, xidx, x2y_signed, g_x2y_signed, g_x2y_vec, g_y, N,
                                     P2, P1, st)))
        return rc;
    return 0;
}
