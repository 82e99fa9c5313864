// FP32-MFMA "NT" GEMM for the blend-shape contraction of SMPL-X (the dominant kernel of the path):
//   forward : v_posed[B, Npad]  = v_template + feat[B,512] . Bt[Npad,512]^T      (shape + pose blend shapes, K = 20+486 -> 512)
//   backward: g_feat[B, 512]    = g_v_posed[B,Npad] . Bk[512,Npad]^T              (split-K slabs, summed by pose_bwd_kernel)
// Both operands are stored with the contraction index contiguous.  Staging is LDS-DMA (global_load_lds_dwordx4: no
// staging VGPRs, no ds_write): one wave-instruction moves 8 rows x 128 B into a lane-linear [128 rows][8 x 16 B] image;
// the bank-conflict swizzle (16-B slot ^= (row>>1)&7) is applied on the per-lane SOURCE address and again on the
// ds_read_b128 address.  Two LDS stages, tile t+1 in flight while tile t is multiplied, one barrier per k-tile.
// v_mfma_f32_32x32x2_f32 is an exact f32 FMA chain (no xf32 on gfx950): lanes 0-31 supply k = 8g+t, lanes 32-63
// k = 8g+4+t for MFMA t of k-group g -- any k permutation is valid as long as A and B agree.
// Replaces smplx.lbs.blend_shapes + the posedirs matmul.
#include <stdlib.h>

#include <algorithm>

#include "saga_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 128
// tuned on MI355X with scripts/ab/gemm_ab.hip (one process, interleaved rounds, random operands): 16-deep tiles with
// three tiles in flight beside the one being multiplied (4 stages = 64 KB, two blocks per CU) run the forward shape at
// 125 TFLOP/s and the split-K backward at 131 (5 stages: 134); 32-deep/2-stage gives 106/107.
#define GEMM_BKT 16         // k-depth of an LDS tile
#ifndef GEMM_NST_FWD
#define GEMM_NST_FWD 4      // LDS stages, forward
#endif
#define GEMM_NST_BWD 5      // LDS stages, backward (split-K)
#define GEMM_SMALL_TILES 256 // launches of at most this many 128 x 128 tiles (less than one per CU) run 64 x 64 tiles instead

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int N>
__device__ __forceinline__ void gemm_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else static_assert(N == 0, "unsupported vmcnt");
}

// EPI 0: C[row][col] = acc + bias[col]           (C row stride ldc; bias may be NULL)
// EPI 1: C[z][row][col] = acc                    (slab z = blockIdx.z, slab stride M*ldc)
// EPI 2: C[row][col] = leaky_relu(acc + bias[col], 0.2)                    (VPoser decoder layers, fit_step.hip)
// EPI 3: C[row][col] = acc * (aux[row][col] > 0 ? 1 : 0.2)                 (their backward; aux = the layer's output)
// BKT  : k-depth of one LDS tile (16 or 32); NST: LDS stages (NST-1 tiles in flight beside the one being multiplied)
// WT   : 32x32 MFMA tiles per wave and dimension: 2 -> 128x128 block tile (the blend contraction), 1 -> 64x64 (the small
//        decoder layers, which need 4x the blocks to fill 256 CUs)
// TRS  : accumulate the TRANSPOSED 32x32 tiles (MFMA operands swapped; same products, same order, same bits): a lane then
//        holds 4 consecutive output columns per register quad and the epilogue is 16 dwordx4 stores per wave instead of
//        64 dword stores -- the store issue of a 128x128 tile was ~8 % of its time with K = 512
template <int EPI, int BKT, int NST, int WT = 2, bool TRS = true, bool LEPI = false>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, int lda,
                                                          const float* __restrict__ Bt, int ldb, float* __restrict__ C,
                                                          int ldc, const float* __restrict__ bias,
                                                          const float* __restrict__ aux, int M, int N, int K,
                                                          int kchunk, int tiles_m, int xcd_map = 0, int n_tiles_x = 0) {
    constexpr int TBM = 64 * WT, TBN = 64 * WT;      // block tile
    constexpr int TILE_FLOATS = TBM * BKT;           // one operand tile
    constexpr int PPW = TILE_FLOATS * 4 / 1024 / 4;  // 1-KiB LDS-DMA pieces per wave per operand tile: 4 (BKT 32) / 2 (16)
    constexpr int RPP = 1024 / (BKT * 4);            // tile rows per piece: 8 / 16
    constexpr int CPR = BKT / 4;                     // 16-B chunks per row: 8 / 4
    // ONE shared array (a second LDS object beside an LDS-DMA target makes hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(1024))) float smem[NST * 2 * TILE_FLOATS];   // [stage][A|B][128][BKT]

    // n_tiles_x > gridDim.x: a PERSISTENT launch -- a block walks the x tiles bx = blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x a
    // multiple of 8, so a block's tiles stay on its XCD's share of the ownership map below); 0: one tile per block
    if (n_tiles_x == 0) n_tiles_x = gridDim.x;
    for (int bx = blockIdx.x; bx < n_tiles_x; bx += gridDim.x) {
    if (bx != (int)blockIdx.x) __syncthreads();      // every wave has left the previous tile's last LDS stage
    // consecutive blocks walk M first so the (large, batch-shared) Bt tile stays hot in L2 while A is re-read
    int tile_m = bx % tiles_m;
    int tile_n = bx / tiles_m;
    int slab = blockIdx.z;
    if (xcd_map == 2) {
        // split-K launch (gridDim.z a multiple of 8): XCD x owns the slabs z = x (mod 8) entirely, so every A and B tile of the
        // launch is fetched by exactly one L2 (linear workgroup id = z * gridDim.x + x, dealt to the XCDs round-robin)
        const int per = n_tiles_x;
        const int L = blockIdx.z * per + bx;
        const int x = L & 7, i = L >> 3;
        slab = x + 8 * (i / per);
        const int t = i % per, tiles_n = per / tiles_m;
        tile_n = t % tiles_n;          // the N tiles of one M tile run together: its (large, streamed) A rows are fetched once
        tile_m = t / tiles_n;
    } else if (xcd_map) {
        // workgroups go to the 8 XCDs round-robin: XCD x owns a quarter of the M tiles (its A rows stay in its L2) and half of the
        // N tiles, so a Bt tile is fetched by four L2s instead of eight (host checks tiles_m % 4 == 0 and tiles_n % 2 == 0)
        const int x = bx & 7, i = bx >> 3;
        const int mq = tiles_m >> 2, nh = (n_tiles_x / tiles_m) >> 1;
        tile_m = (x & 3) * mq + i % mq;
        tile_n = (x >> 2) * nh + i / mq;
    }
    const int m0 = tile_m * TBM, n0 = tile_n * TBN;
    const int k_begin = slab * kchunk;
    const int k_end = min(K, k_begin + kchunk);
    const int T = (k_end - k_begin) / BKT;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    // bank-conflict swizzle of the 16-B slot within a row: BKT 32 (128-B rows, two per bank row): (row>>1)&7;
    // BKT 16 (64-B rows, four per bank row): (row>>2)&3 -- the 16 lanes of every ds_read_b128 group hit 16 distinct slots
    auto swz = [](int row) { return BKT == 32 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

    // LDS-DMA source addresses: piece p = wave*PPW + j covers tile rows RPP*p ..; lane -> (row, slot) and fetches the
    // 16-B k-chunk slot ^ swz(row) of that row.
    const float* a_src[PPW];
    const float* b_src[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = (wave * PPW + j) * RPP + lane / CPR;
        const int chunk = (lane % CPR) ^ swz(row);
        a_src[j] = A + (size_t)min(m0 + row, M - 1) * lda + k_begin + chunk * 4;
        b_src[j] = Bt + (size_t)min(n0 + row, N - 1) * ldb + k_begin + chunk * 4;
    }
    auto issue = [&](int stage, int koff) {
        float* sa = smem + stage * (2 * TILE_FLOATS) + wave * (PPW * 256);
        float* sb = sa + TILE_FLOATS;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[j] + koff), (lds_ptr_t)(sa + j * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(b_src[j] + koff), (lds_ptr_t)(sb + j * 256), 16, 0, 0);
        }
    };

    f32x16 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (floats): row * BKT + ((2g + half) ^ swz(row)) * 4; swz(row) depends on l31 only
    const int fsw = swz(l31);
    const int a_row = (wm * 32 * WT + l31) * BKT, b_row = (wn * 32 * WT + l31) * BKT;

#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < T) issue(s, s * BKT);
    for (int t = 0; t < T; ++t) {
        // tile t has landed once at most the NST-2 younger tiles are outstanding (fewer near the tail: drain)
        if (NST > 2 && t + NST - 2 < T) gemm_wait_vmcnt<(NST - 2) * 2 * PPW>();
        else gemm_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();       // everyone's pieces have; everyone is done reading stage (t-1) % NST
        const float* As = smem + (t % NST) * (2 * TILE_FLOATS);
        const float* Bs = As + TILE_FLOATS;
        // all fragment reads of this k-tile first, THEN the LDS-DMA of a later tile: hipcc waits vmcnt(0) before any
        // ds_read that follows an outstanding LDS-DMA in program order, which would serialise the prefetch with the MFMAs
        float4 af[BKT / 8][WT], bf[BKT / 8][WT];
#pragma unroll
        for (int g = 0; g < BKT / 8; ++g) {
            const int koff = ((2 * g + half) ^ fsw) * 4;
#pragma unroll
            for (int u = 0; u < WT; ++u) {
                af[g][u] = *reinterpret_cast<const float4*>(As + a_row + u * 32 * BKT + koff);
                bf[g][u] = *reinterpret_cast<const float4*>(Bs + b_row + u * 32 * BKT + koff);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + NST - 1 < T) issue((t + NST - 1) % NST, (t + NST - 1) * BKT);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < BKT / 8; ++g) {
#pragma unroll
            for (int mt = 0; mt < WT; ++mt)
#pragma unroll
                for (int nt = 0; nt < WT; ++nt) {
                    if (TRS) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].x, af[g][mt].x, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].y, af[g][mt].y, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].z, af[g][mt].z, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].w, af[g][mt].w, acc[mt][nt], 0, 0, 0);
                    } else {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][mt].x, bf[g][nt].x, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][mt].y, bf[g][nt].y, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][mt].z, bf[g][nt].z, acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][mt].w, bf[g][nt].w, acc[mt][nt], 0, 0, 0);
                    }
                }
        }
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cz = C;
    if (EPI == 1) Cz += (size_t)slab * M * ldc;
    const bool full = (m0 + TBM <= M);       // N is always a multiple of BN here (Npad / 512)
    if (TRS && LEPI && EPI == 0 && WT == 2 && full && NST * 2 * TILE_FLOATS >= 4 * 32 * 68) {
        // whole-line epilogue: the wave's 64 x 64 tile goes through the (now idle) LDS, 32 rows at a time, and leaves as 256-B row
        // segments (16 lanes x 16 B), non-temporal -- the output then does not displace the operands in L2
        __syncthreads();
        float* sw = smem + wave * (32 * 68);
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int mt = 0; mt < WT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < WT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(sw + l31 * 68 + nt * 32 + 8 * q + 4 * half) =
                        make_float4(acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]);
            __builtin_amdgcn_wave_barrier();
            const int c4 = (lane & 15) * 4;
            const int col = n0 + wn * 64 + c4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) bv = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = it * 4 + (lane >> 4);
                const float4 v = *reinterpret_cast<const float4*>(sw + r * 68 + c4);
                const f32x4_t o = {v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w};
                __builtin_nontemporal_store(o, reinterpret_cast<f32x4_t*>(Cz + (size_t)(m0 + wm * 64 + mt * 32 + r) * ldc + col));
            }
            __builtin_amdgcn_wave_barrier();
        }
        continue;
    }
    if (TRS) {
        // transposed tiles: row = lane&31, col = (r&3) + 8*(r>>2) + 4*(lane>>5): one 16-B store per register quad
#pragma unroll
        for (int mt = 0; mt < WT; ++mt) {
            const int row = m0 + wm * 32 * WT + mt * 32 + l31;
            if (!full && row >= M) continue;
#pragma unroll
            for (int nt = 0; nt < WT; ++nt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = n0 + wn * 32 * WT + nt * 32 + 8 * q + 4 * half;
                    if (col >= N) continue;
                    const size_t off = (size_t)row * ldc + col;
                    float4 v = make_float4(acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]);
                    if ((EPI == 0 || EPI == 2) && bias) {
                        const float4 bv = *reinterpret_cast<const float4*>(bias + col);
                        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    }
                    if (EPI == 2) {
                        v.x = v.x > 0.f ? v.x : 0.2f * v.x; v.y = v.y > 0.f ? v.y : 0.2f * v.y;
                        v.z = v.z > 0.f ? v.z : 0.2f * v.z; v.w = v.w > 0.f ? v.w : 0.2f * v.w;
                    }
                    if (EPI == 3) {
                        const float4 x = *reinterpret_cast<const float4*>(aux + off);
                        v.x = x.x > 0.f ? v.x : 0.2f * v.x; v.y = x.y > 0.f ? v.y : 0.2f * v.y;
                        v.z = x.z > 0.f ? v.z : 0.2f * v.z; v.w = x.w > 0.f ? v.w : 0.2f * v.w;
                    }
#ifdef GEMM_EXP_NO_STORE        // timing experiment: what the epilogue's stores cost (wrong results)
                    if (v.x == 123.456f)
#endif
                    *reinterpret_cast<float4*>(Cz + off) = v;
                }
            }
        }
        continue;
    }
#pragma unroll
    for (int nt = 0; nt < WT; ++nt) {
        const int col = n0 + wn * 32 * WT + nt * 32 + l31;
        if (col >= N) continue;
        const float bv = ((EPI == 0 || EPI == 2) && bias) ? bias[col] : 0.f;
        auto epi = [&](float v, size_t off) -> float {
            if (EPI == 2) return v > 0.f ? v : 0.2f * v;
            if (EPI == 3) return aux[off] > 0.f ? v : 0.2f * v;
            return v;
        };
#pragma unroll
        for (int mt = 0; mt < WT; ++mt) {
            const size_t cbase = (size_t)(m0 + wm * 32 * WT + mt * 32 + 4 * half) * ldc + col;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const size_t off = cbase + (size_t)((r & 3) + 8 * (r >> 2)) * ldc;
                    Cz[off] = epi(acc[mt][nt][r] + bv, off);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 32 * WT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < M) Cz[(size_t)row * ldc + col] = epi(acc[mt][nt][r] + bv, (size_t)row * ldc + col);
                }
            }
        }
    }
    }       // tiles of this block
}

// v_posed[B,Npad] = v_template + feat . Bt^T
int saga_launch_blend_fwd(const saga_model* m, const float* feat, float* v_posed, int B, hipStream_t st) {
    const int tiles_m = (B + BM - 1) / BM, tiles_n = m->Npad / BN;
    dim3 grid(tiles_m * tiles_n, 1, 1);
    static const int want_xcd = getenv("SAGA_GEMM_XCD") ? atoi(getenv("SAGA_GEMM_XCD")) : 1;     // SAGA_GEMM_XCD=0: plain tile order (A/B)
    int xcd_map = want_xcd && tiles_m % 4 == 0 && tiles_n % 2 == 0;
    int n_tiles = tiles_m * tiles_n;
    // whole-line non-temporal epilogue (LEPI, SAGA_GEMM_LEPI=1): the 516 MB of output no longer displaces the operands in L2 -- HBM-side
    // traffic of the launch 1.14 -> 0.81 GB, same bits -- but the launch is 0.8 % slower and the skinning forward, which then finds
    // nothing of v_posed in the Infinity Cache, 0.013 ms slower: 4.324 against 4.300 ms per iteration.  Off by default: time wins.
    static const int want_lepi = getenv("SAGA_GEMM_LEPI") ? atoi(getenv("SAGA_GEMM_LEPI")) : 0;
    // Few tiles (small batches, vertex-subset models: the 256 x 79-vertex launches of GraspMotion-Opt stage 1 are 8 tiles): the
    // launch is bound by the LATENCY of one tile's k-loop (32 k-steps x 2,048 MFMA cycles = 27 us for 128 x 128), so 64 x 64 tiles
    // (a quarter of the cycles per k-step, four times the blocks) finish sooner: 39 -> ? us.  Same products in the same order:
    // same bits (SAGA_GEMM_SMALL=0: always 128 x 128).
    static const int want_small = getenv("SAGA_GEMM_SMALL") ? atoi(getenv("SAGA_GEMM_SMALL")) : 1;
    if (want_small && !want_lepi && n_tiles <= GEMM_SMALL_TILES) {
        const int tm = (B + 63) / 64, tn = m->Npad / 64;
        gemm_nt_f32_kernel<0, GEMM_BKT, GEMM_NST_FWD, 1><<<dim3(tm * tn), 256, 0, st>>>(feat, FEAT_K, m->Bt, FEAT_K, v_posed, m->Npad, m->v_template,
                                                                                      nullptr, B, m->Npad, FEAT_K, FEAT_K, tm);
        SAGA_LAUNCH_CHECK();
        return 0;
    }
    if (want_lepi) {
        grid.x = n_tiles;
        gemm_nt_f32_kernel<0, GEMM_BKT, GEMM_NST_FWD, 2, true, true><<<grid, 256, 0, st>>>(feat, FEAT_K, m->Bt, FEAT_K, v_posed, m->Npad, m->v_template,
                                                                                            nullptr, B, m->Npad, FEAT_K, FEAT_K, tiles_m, xcd_map, n_tiles);
    } else {
        // persistent launch: two blocks per CU (what the 64 KB of LDS stages allow) walk the tiles -- no block turnover between
        // tiles: 1.068 -> 1.059 ms at B = 4096 (SAGA_GEMM_PERSIST=0: one block per tile; =n: n blocks).  What the per-tile
        // overheads are worth in all (measured with -DGEMM_EXP_NO_STORE, wrong results): no epilogue stores at all 1.027 ms.
        static const int persist_env = getenv("SAGA_GEMM_PERSIST") ? atoi(getenv("SAGA_GEMM_PERSIST")) : -1;
        int persist = persist_env;
        if (persist < 0) {
            static int n_cu = 0;
            if (n_cu == 0) {
                int dev = 0;
                hipDeviceProp_t prop;
                if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
                else n_cu = 256;
            }
            persist = 2 * n_cu;
        }
        persist &= ~7;          // a multiple of 8: a block's tiles stay on one XCD's share of the ownership map
        grid.x = n_tiles;
        if (persist > 0 && n_tiles > persist) grid.x = persist;
        gemm_nt_f32_kernel<0, GEMM_BKT, GEMM_NST_FWD><<<grid, 256, 0, st>>>(feat, FEAT_K, m->Bt, FEAT_K, v_posed, m->Npad, m->v_template, nullptr, B, m->Npad,
                                                    FEAT_K, FEAT_K, tiles_m, xcd_map, n_tiles);
    }
    SAGA_LAUNCH_CHECK();
    return 0;
}

// Split-K slabs of the backward contraction: 16 for the whole mesh; a vertex subset (K = 3 x a few hundred to a few thousand) gets
// one slab per 768 columns so that a block still runs >= 48 k-tiles.  A function of the model only: a candidate's sums must
// not depend on the batch size.
int saga_blend_bwd_splitk(const saga_model* m) { return std::min(SAGA_BWD_SPLITK, std::max(1, m->Npad / 768)); }

// slabs[z][B][512] = g_v_posed[:, kz] . Bk[:, kz]^T
int saga_launch_blend_bwd(const saga_model* m, const float* g_vposed, float* slabs, int B, hipStream_t st) {
    const int tiles_m = (B + BM - 1) / BM, tiles_n = FEAT_K / BN;
    const int splitk = saga_blend_bwd_splitk(m);
    const int kchunk = saga_roundup((m->Npad + splitk - 1) / splitk, 32);
    dim3 grid(tiles_m * tiles_n, 1, splitk);
    static const int want_xcd = getenv("SAGA_GEMM_XCD") ? atoi(getenv("SAGA_GEMM_XCD")) : 1;     // SAGA_GEMM_XCD=0: plain tile order (A/B)
    const int xcd_map = (want_xcd && splitk % 8 == 0) ? 2 : 0;
    static const int want_small = getenv("SAGA_GEMM_SMALL") ? atoi(getenv("SAGA_GEMM_SMALL")) : 1;
    if (want_small && tiles_m * tiles_n * splitk <= GEMM_SMALL_TILES) {        // latency-bound launch: 64 x 64 tiles (see the forward)
        const int tm = (B + 63) / 64, tn = FEAT_K / 64;
        gemm_nt_f32_kernel<1, GEMM_BKT, GEMM_NST_BWD, 1><<<dim3(tm * tn, 1, splitk), 256, 0, st>>>(g_vposed, m->Npad, m->Bk, m->Npad, slabs, FEAT_K,
                                                                                                 nullptr, nullptr, B, FEAT_K, m->Npad, kchunk, tm);
        SAGA_LAUNCH_CHECK();
        return 0;
    }
    gemm_nt_f32_kernel<1, GEMM_BKT, GEMM_NST_BWD><<<grid, 256, 0, st>>>(g_vposed, m->Npad, m->Bk, m->Npad, slabs, FEAT_K, nullptr, nullptr, B, FEAT_K,
                                                m->Npad, kchunk, tiles_m, xcd_map);
    SAGA_LAUNCH_CHECK();
    return 0;
}

// Small dense layers of the parameter decoder (fit_step.hip): C[M,N] = epi(A[M,K] . Bt[N,K]^T), 64x64 block tiles.  N a multiple of 64,
// K a multiple of 16, rows of A and Bt 16-B aligned.
int saga_launch_gemm_nt(int epi, const float* A, int lda, const float* Bt, int ldb, float* C, int ldc, const float* bias,
                        const float* aux, int M, int N, int K, hipStream_t st) {
    SAGA_CHECK_ARG(N % 64 == 0 && K % GEMM_BKT == 0 && lda % 4 == 0 && ldb % 4 == 0 && M > 0, "gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
    const int tiles_m = (M + 63) / 64, tiles_n = N / 64;
    dim3 grid(tiles_m * tiles_n, 1, 1);
    if (epi == 0)
        gemm_nt_f32_kernel<0, GEMM_BKT, GEMM_NST_FWD, 1><<<grid, 256, 0, st>>>(A, lda, Bt, ldb, C, ldc, bias, aux, M, N, K, K, tiles_m);
    else if (epi == 2)
        gemm_nt_f32_kernel<2, GEMM_BKT, GEMM_NST_FWD, 1><<<grid, 256, 0, st>>>(A, lda, Bt, ldb, C, ldc, bias, aux, M, N, K, K, tiles_m);
    else if (epi == 3)
        gemm_nt_f32_kernel<3, GEMM_BKT, GEMM_NST_FWD, 1><<<grid, 256, 0, st>>>(A, lda, Bt, ldb, C, ldc, bias, aux, M, N, K, K, tiles_m);
    else
        SAGA_CHECK_ARG(false, "gemm_nt: unknown epilogue %d", epi);
    SAGA_LAUNCH_CHECK();
    return 0;
}
