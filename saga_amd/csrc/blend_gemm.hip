// FP32-MFMA "NT" GEMM for the blend-shape contraction of SMPL-X (the dominant kernel of the path):
//   forward : v_posed[B, Npad]  = v_template + feat[B,512] . Bt[Npad,512]^T      (shape + pose blend shapes, K = 20+486 -> 512)
//   backward: g_feat[B, 512]    = g_v_posed[B,Npad] . Bk[512,Npad]^T              (split-K slabs, summed by pose_bwd_kernel)
// Both operands are stored with the contraction index contiguous.  Staging is LDS-DMA (global_load_lds_dwordx4: no
// staging VGPRs, no ds_write): one wave-instruction moves 16 rows x 64 B into a lane-linear [128 rows][4 x 16 B] image;
// the bank-conflict swizzle (16-B slot ^= (row>>2)&3) is applied on the per-lane SOURCE offset and again on the
// ds_read_b128 address.  4 (forward) / 5 (split-K backward) LDS stages, one counted vmcnt wait and one barrier per k-tile.
// v_mfma_f32_32x32x2_f32 is an exact f32 FMA chain (no xf32 on gfx950): lanes 0-31 supply k = 8g+t, lanes 32-63
// k = 8g+4+t for MFMA t of k-group g -- any k permutation is valid as long as A and B agree.
//
// What shapes the k-loop (measured: scripts/ab/mfma_share_ab.hip, scripts/ab/store_pattern_ab.hip, scripts/exp/gemm_trace.py;
// DESIGN.md section 4): two blocks share a CU, i.e. two waves a SIMD.  Their MFMAs share the matrix pipe perfectly (64 cycles
// each, 128 per wave and MFMA when both issue) -- but whatever ELSE a wave issues while its partner is inside a burst of MFMAs
// waits for a gap in that burst: an epilogue of ~400 instructions took 25-30 k cycles, 16 stores alone 55 k in the microbenchmark
// (700 cycles with the CU to itself).  So the k-loop contains no vector instruction but the MFMAs: scalar LDS-DMA targets (uniform
// wave id), scalar-base LDS-DMA addresses (inline asm: the builtin takes per-lane 64-bit addresses), fragment reads whose stage and
// row tile are immediate offsets (k-loop unrolled by the stage count), and the epilogue is one address per 32-row tile + the bias
// adds + 16 stores.  Forward 1.06 -> 1.03 ms, backward 0.97 -> 0.90 ms at B = 4096, bit-identical.
// Replaces smplx.lbs.blend_shapes + the posedirs matmul.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

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
#ifndef GEMM_NST_BWD
#define GEMM_NST_BWD 5      // LDS stages, backward (split-K)
#endif
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
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else static_assert(N == 0, "unsupported vmcnt");
}

template <int I, int N, class F>
__device__ __forceinline__ void gemm_static_for(F&& f) {       // f(integral_constant<int, I>) for I..N-1: every index a constant
    if constexpr (I < N) {                                      // (a plain unrolled loop left the accumulator array in scratch)
        f(std::integral_constant<int, I>());
        gemm_static_for<I + 1, N>(f);
    }
}
// One LDS-DMA wave-instruction (64 lanes x 16 B -> 1 KiB of LDS at lds_addr, lane-linear) in the scalar-base form: global address =
// 64-bit scalar base + the lane's 32-bit byte offset.  hipcc's builtin always takes a 64-bit per-lane address, i.e. one 64-bit
// vector add per transfer -- vector instructions are what this kernel cannot afford inside its k-loop (see below).  M0 (the LDS
// target) is set here; the kernel has no other user of it.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void gemm_lds_dma16(uint32_t voff, const char* sbase, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop
// Fragment reads as inline asm.  hipcc treats every ds_read after an outstanding LDS-DMA as a possible alias of it and drains vmcnt(0)
// first -- in a loop that is EVERY k-step (the DMA of the step before is still in flight), so the deep stage ring degenerates to
// "wait for the youngest tile": the load latency of one tile had to fit into one k-step.  The asm reads are invisible to that rule;
// the counted vmcnt wait + barrier at the top of the step is what orders them after the DMA that filled their stage, and the
// lgkmcnt waits below (tied to the destination registers, so the MFMAs cannot move above them) order the MFMAs after the reads.
typedef float gemm_f4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ gemm_f4 gemm_lds_read128(uint32_t addr) {
    gemm_f4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void gemm_wait_lgkm(gemm_f4& a, gemm_f4& b, gemm_f4& c, gemm_f4& d) {
    if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
    else static_assert(N == 0, "unsupported lgkmcnt");
}
template <int N>        // (every register once: a repeated "+v" operand is copied BEFORE the wait, i.e. before its read has landed)
__device__ __forceinline__ void gemm_wait_lgkm(gemm_f4& a, gemm_f4& b) {
    if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b)::"memory");
    else static_assert(N == 0, "unsupported lgkmcnt");
}

#if defined(SAGA_EXPERIMENT_BUILD) && defined(GEMM_EXP_TRACE)      // investigation build only: per-block time stamps of the persistent forward launch (scripts/exp/gemm_trace.py)
#define GEMM_TRACE_STRIDE 96
__device__ unsigned long long gemm_trace_buf[1024 * GEMM_TRACE_STRIDE];
extern "C" int saga_debug_gemm_trace(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gemm_trace_buf), n * sizeof(unsigned long long));
}
#define GEMM_STAMP(slot)                                                                                                  \
    if (EPI == 0 && WT == 2 && threadIdx.x == 0 && n_tiles_x > (int)gridDim.x && (slot) < GEMM_TRACE_STRIDE)              \
        gemm_trace_buf[blockIdx.x * GEMM_TRACE_STRIDE + (slot)] = clock64();
#else
#define GEMM_STAMP(slot)
#endif

// EPI 0: C[row][col] = acc + bias[col]           (C row stride ldc; bias may be NULL)
// EPI 1: C[z][row][col] = acc                    (slab z = blockIdx.z, slab stride M*ldc)
// EPI 2: C[row][col] = leaky_relu(acc + bias[col], 0.2)                    (VPoser decoder layers, fit_step.hip)
// EPI 3: C[row][col] = acc * (aux[row][col] > 0 ? 1 : 0.2)                 (their backward; aux = the layer's output)
// BKT  : k-depth of one LDS tile (16); NST: LDS stages (NST-1 tiles in flight beside the one being multiplied)
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
#if defined(SAGA_EXPERIMENT_BUILD) && defined(GEMM_EXP_STAGGER)
    // The two blocks of a CU start together and every tile takes the same time, so their prologues / epilogues (no MFMAs) coincide
    // for the whole launch: the second half of the blocks (the second block of every CU) starts GEMM_EXP_STAGGER x 1,024 cycles late
    if (n_tiles_x > (int)gridDim.x && blockIdx.x >= gridDim.x / 2)
        for (int i = 0; i < GEMM_EXP_STAGGER; ++i) __builtin_amdgcn_s_sleep(16);
#endif
#if defined(SAGA_EXPERIMENT_BUILD) && defined(GEMM_EXP_TRACE)
    if (EPI == 0 && WT == 2 && threadIdx.x == 0 && n_tiles_x > (int)gridDim.x)
        gemm_trace_buf[blockIdx.x * GEMM_TRACE_STRIDE] =
            (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32);
    int trace_i = 0;
#endif
    for (int bx = blockIdx.x; bx < n_tiles_x; bx += gridDim.x) {
    GEMM_STAMP(1 + 6 * trace_i)
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
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: the LDS-DMA targets become scalar arithmetic)
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    // bank-conflict swizzle of the 16-B slot within a row: BKT 32 (128-B rows, two per bank row): (row>>1)&7;
    // BKT 16 (64-B rows, four per bank row): (row>>2)&3 -- the 16 lanes of every ds_read_b128 group hit 16 distinct slots
    auto swz = [](int row) { return BKT == 32 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

    // LDS-DMA source addresses: piece p = wave*PPW + j covers tile rows RPP*p ..; lane -> (row, slot) and fetches the
    // 16-B k-chunk slot ^ swz(row) of that row.  A uniform 64-bit base (the tile's first row at k_begin: scalar registers, moved on by
    // scalar adds) plus a per-lane 32-bit byte offset that never changes: the transfer's address costs no vector instruction
    // (an immediate offset would not do: the instruction adds it to the LDS address as well).
    const char* a_base = reinterpret_cast<const char*>(A + (size_t)m0 * lda + k_begin);
    const char* b_base = reinterpret_cast<const char*>(Bt + (size_t)n0 * ldb + k_begin);
    uint32_t a_off[PPW], b_off[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = (wave * PPW + j) * RPP + lane / CPR;
        const int chunk = (lane % CPR) ^ swz(row);
        a_off[j] = ((uint32_t)min(row, M - 1 - m0) * (uint32_t)lda + chunk * 4) * 4u;
        b_off[j] = ((uint32_t)min(row, N - 1 - n0) * (uint32_t)ldb + chunk * 4) * 4u;
    }
    // LDS-DMA of the k-tile at byte offset koff of the bases, into stage STG (LDS target: scalar arithmetic, wave is uniform)
    auto issue = [&](auto stg, int koff) __attribute__((always_inline)) {
        constexpr int STG = decltype(stg)::value;
        float* sa = smem + STG * (2 * TILE_FLOATS) + wave * (PPW * 256);
        float* sb = sa + TILE_FLOATS;
        const char* ak = a_base + koff;
        const char* bk = b_base + koff;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            gemm_lds_dma16(a_off[j], ak, (uint32_t)(uintptr_t)(lds_ptr_t)(sa + j * 256));
            gemm_lds_dma16(b_off[j], bk, (uint32_t)(uintptr_t)(lds_ptr_t)(sb + j * 256));
        }
    };

    f32x16 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read addresses (LDS bytes) in stage 0, per k-group: row * BKT + ((2g + half) ^ swz(row)) * 4; swz(row) depends on l31
    // only.  The stage and the second 32-row tile are immediate offsets of the reads (16 bits: a second base from stage HI_STG on).
    static_assert(BKT == 16, "the fragment reads are written for two k-groups per tile");
    constexpr int STAGE_BYTES = 2 * TILE_FLOATS * 4;
    constexpr int HI_STG = (65535 - TILE_FLOATS * 4 - 32 * BKT * 4) / STAGE_BYTES + 1;      // first stage the 16-bit offset cannot reach
    const int fsw = swz(l31);
    uint32_t fa[2][2], fb[2][2];        // [base: stage 0 / stage HI_STG][k-group]
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int koff = ((2 * g + half) ^ fsw) * 4;
        fa[0][g] = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + (wm * 32 * WT + l31) * BKT + koff);
        fb[0][g] = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + TILE_FLOATS + (wn * 32 * WT + l31) * BKT + koff);
        fa[1][g] = fa[0][g] + HI_STG * STAGE_BYTES;
        fb[1][g] = fb[0][g] + HI_STG * STAGE_BYTES;
    }

    // EPI 0 adds the bias as ONE MORE K-STEP on the matrix cores: operands (bias[col], 1) in the k = 0 lanes, (0, 0) in the k = 1
    // lanes, so acc = fma(1, bias, acc) = acc + bias -- the same rounding as the vector add it replaces (bit-identical), and the
    // epilogue shrinks to its stores: every vector instruction there is issued under the co-resident block's MFMA burst (~100
    // cycles each: 64 bias adds + 8 quad loads per wave and tile were most of the epilogue).  The two operand words per lane are
    // requested here, before the k-loop's first LDS-DMA: the oldest entries of vmcnt, back long before they are used.
    constexpr bool MFMA_BIAS = EPI == 0 && TRS && !LEPI;
    float bias_op[WT];
    if (MFMA_BIAS && bias) {
#pragma unroll
        for (int nt = 0; nt < WT; ++nt) bias_op[nt] = bias[n0 + wn * 32 * WT + nt * 32 + l31];
    }
    gemm_static_for<0, NST - 1>([&](auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        if (S < T) issue(sc, S * BKT * 4);
    });
    // The k-loop, NST steps per trip so that a step's stage is a constant.  Per step and wave: one counted wait, one barrier, 4 WT
    // fragment reads, 2 PPW LDS-DMA instructions and 32 (WT 2) MFMAs -- and no vector arithmetic: the co-resident block's wave on
    // this SIMD is inside an MFMA burst most of the time, and whatever vector instruction this wave issues meanwhile waits for a slot
    // between two of its MFMAs (64 cycles each; measured: scripts/ab/store_pattern_ab.hip)
    for (int t0 = 0; t0 < T; t0 += NST) {
        gemm_static_for<0, NST>([&](auto uc) __attribute__((always_inline)) {
            constexpr int U = decltype(uc)::value;
            const int t = t0 + U;
            if (t < T) {
                // tile t has landed once at most the NST-2 younger tiles are outstanding (fewer near the tail: drain)
                if (NST > 2 && t + NST - 2 < T) gemm_wait_vmcnt<(NST - 2) * 2 * PPW>();
                else gemm_wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();       // everyone's pieces have; everyone is done reading stage (t-1) % NST
#if defined(SAGA_EXPERIMENT_BUILD) && defined(GEMM_EXP_TRACE)
                if (t == 0) { GEMM_STAMP(2 + 6 * trace_i) }
#endif
                // fragment reads as asm (see gemm_lds_read128), all of them before the LDS-DMA of a later tile
                constexpr int HB = U >= HI_STG ? 1 : 0, SOFF = (U - HB * HI_STG) * STAGE_BYTES;
                gemm_f4 af[2][WT], bf[2][WT];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    af[g][0] = gemm_lds_read128<SOFF>(fa[HB][g]);
                    if constexpr (WT == 2) af[g][1] = gemm_lds_read128<SOFF + 32 * BKT * 4>(fa[HB][g]);
                    bf[g][0] = gemm_lds_read128<SOFF>(fb[HB][g]);
                    if constexpr (WT == 2) bf[g][1] = gemm_lds_read128<SOFF + 32 * BKT * 4>(fb[HB][g]);
                }
                if (t + NST - 1 < T) issue(std::integral_constant<int, (U + NST - 1) % NST>(), (t + NST - 1) * BKT * 4);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    // LDS returns in order: group g's 2 WT reads are back once at most the later groups' are outstanding
                    if (g == 1) __builtin_amdgcn_sched_barrier(0);      // (keeps group 1's wait behind group 0's MFMAs)
                    if constexpr (WT == 2) {
                        if (g == 0) gemm_wait_lgkm<4>(af[0][0], af[0][1], bf[0][0], bf[0][1]);
                        else gemm_wait_lgkm<0>(af[1][0], af[1][1], bf[1][0], bf[1][1]);
                    } else {
                        if (g == 0) gemm_wait_lgkm<2>(af[0][0], bf[0][0]);
                        else gemm_wait_lgkm<0>(af[1][0], bf[1][0]);
                    }
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
        });
    }

    if (MFMA_BIAS && bias) {
        const float one = half ? 0.f : 1.f;
#pragma unroll
        for (int nt = 0; nt < WT; ++nt) {
            const float bv = half ? 0.f : bias_op[nt];
#pragma unroll
            for (int mt = 0; mt < WT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, one, acc[mt][nt], 0, 0, 0);
        }
    }
    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    GEMM_STAMP(3 + 6 * trace_i)
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
        // transposed tiles: row = lane&31, col = (r&3) + 8*(r>>2) + 4*(lane>>5): one 16-B store per register quad.  One address per
        // 32-row tile, the quads at immediate offsets -- the epilogue is vector instructions the co-resident block's MFMA burst
        // starves, so it is kept to the 16 WT adds of the bias and the stores (N is a multiple of the tile width for every caller).
        // Everything the epilogue READS (bias quads: shared by the row tiles; EPI 3: the aux quads) is fetched before the first
        // store: stores count in vmcnt, so a load issued between them would make its wait drain every store before it.
        const int colw = n0 + wn * 32 * WT + 4 * half;
        float4 bq[WT][4];
        if ((EPI == 0 || EPI == 2) && bias && !MFMA_BIAS) {
            const float* bp = bias + colw;
#pragma unroll
            for (int nt = 0; nt < WT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) bq[nt][q] = *reinterpret_cast<const float4*>(bp + nt * 32 + 8 * q);
        }
        float4 xq[EPI == 3 ? WT : 1][EPI == 3 ? WT : 1][4];
        if (EPI == 3) {
#pragma unroll
            for (int mt = 0; mt < WT; ++mt) {
                const float* xp = aux + (size_t)min(m0 + wm * 32 * WT + mt * 32 + l31, M - 1) * ldc + colw;
#pragma unroll
                for (int nt = 0; nt < WT; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) xq[EPI == 3 ? mt : 0][EPI == 3 ? nt : 0][q] = *reinterpret_cast<const float4*>(xp + nt * 32 + 8 * q);
            }
        }
#if defined(SAGA_EXPERIMENT_BUILD) && defined(GEMM_EXP_TRACE)
        GEMM_STAMP(4 + 6 * trace_i)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GEMM_STAMP(5 + 6 * trace_i)
#endif
#pragma unroll
        for (int mt = 0; mt < WT; ++mt) {
            const int row = m0 + wm * 32 * WT + mt * 32 + l31;
            if (!full && row >= M) continue;
            float* cp = Cz + (size_t)row * ldc + colw;
#pragma unroll
            for (int nt = 0; nt < WT; ++nt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4(acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]);
                    if ((EPI == 0 || EPI == 2) && bias && !MFMA_BIAS) {
                        const float4 bv = bq[nt][q];
                        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    }
                    if (EPI == 2) {
                        v.x = v.x > 0.f ? v.x : 0.2f * v.x; v.y = v.y > 0.f ? v.y : 0.2f * v.y;
                        v.z = v.z > 0.f ? v.z : 0.2f * v.z; v.w = v.w > 0.f ? v.w : 0.2f * v.w;
                    }
                    if (EPI == 3) {
                        const float4 x = xq[EPI == 3 ? mt : 0][EPI == 3 ? nt : 0][q];
                        v.x = x.x > 0.f ? v.x : 0.2f * v.x; v.y = x.y > 0.f ? v.y : 0.2f * v.y;
                        v.z = x.z > 0.f ? v.z : 0.2f * v.z; v.w = x.w > 0.f ? v.w : 0.2f * v.w;
                    }
#if defined(SAGA_EXPERIMENT_BUILD) && defined(GEMM_EXP_NO_STORE)        // timing experiment: what the epilogue's stores cost (wrong results)
                    if (v.x == 123.456f)
#endif
                    *reinterpret_cast<float4*>(cp + nt * 32 + 8 * q) = v;
                }
            }
        }
#if defined(SAGA_EXPERIMENT_BUILD) && defined(GEMM_EXP_TRACE)
        GEMM_STAMP(6 + 6 * trace_i)
        ++trace_i;
#endif
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


// The forward contraction at fitting batch sizes: a PERSISTENT kernel whose tiles overlap (blend_fwd_xt_kernel).  With one tile per block
// (gemm_nt_f32_kernel) ~9 % of a tile's time is its prologue (three k-tiles requested, nothing to multiply yet) and its epilogue (16
// stores per wave, each waiting for a gap in the co-resident block's MFMA burst).  Here
//   * the stores of tile i leave during the k-loop of tile i+1, the four quads of a 32 x 32 tile every eighth k-step, from a second
//     accumulator set (2 x 64 registers; 190 VGPRs, still two blocks per CU) -- still no vector arithmetic in a k-step;
//   * the first three k-tiles of tile i+1 are requested during the last three k-steps of tile i: the stage ring never drains;
//   * tiles come from a queue per XCD (the two blocks of a CU progress very unequally, equal shares lose: see the launch below), drawn
//     ONE TILE AHEAD: the ticket is requested at the start of a tile, published to the block through LDS at k-step 19 and read at
//     k-step 21, behind barriers the k-loop has anyway;
//   * everything the compiler would wait for with a count that ignores the LDS-DMA transfers (i.e. drain them) -- the ticket, the
//     bias words -- is inline asm, retired by the counted waits of the k-steps between its issue and its use.
// Same products in the same order as the one-tile kernel: same bits (tests/test_gpu_smplx.py).  0.985 -> 0.939 ms at B = 4096.
// Full 128 x 128 tiles, K = 512, lda = ldb = 512 only.  The queue is 8 counters the CALLER provides, zero at the start of the launch:
// saga_smplx_fwd lends the first 32 bytes of its `vertices` output, which the pose kernel before this launch zeroes and the skinning
// kernel after it overwrites -- no state in the library, nothing shared between calls.
__device__ __forceinline__ void gemm_lds_write32(uint32_t addr, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ int gemm_lds_read32_wait(uint32_t addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// vector memory operations the compiler must not see: it would wait for them with counts that ignore the LDS-DMA transfers in flight
// (i.e. drain them).  Their results are used more than three k-steps later: the counted waits of the steps between have retired them.
__device__ __forceinline__ int gemm_atomic_inc_async(int* p) {
    int v = 1;
    asm volatile("global_atomic_add %0, %1, %0, off sc0" : "+v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float gemm_load_async(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__global__ __launch_bounds__(256) void blend_fwd_xt_kernel(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C, int ldc,
                                                           const float* __restrict__ bias, int tiles_m, int xcd_map, int n_tiles, int* __restrict__ queues) {
    constexpr int BKT = 16, NST = 4, LD = 512, T = 32, TILE_FLOATS = 128 * BKT, STAGE_BYTES = 2 * TILE_FLOATS * 4;
    __shared__ __attribute__((aligned(1024))) float smem[NST * 2 * TILE_FLOATS + 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    auto swz = [](int row) { return (row >> 2) & 3; };
    uint32_t off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + lane / 4;
        off[j] = ((uint32_t)row * LD + (((lane & 3) ^ swz(row)) * 4)) * 4u;
    }
    const int fsw = swz(l31);
    uint32_t fa[2], fb[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int koff = ((2 * g + half) ^ fsw) * 4;
        fa[g] = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + (wm * 64 + l31) * BKT + koff);
        fb[g] = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + TILE_FLOATS + (wn * 64 + l31) * BKT + koff);
    }
    const uint32_t ticket_addr = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + NST * 2 * TILE_FLOATS);
    const int qx = blockIdx.x & 7, gx = gridDim.x >> 3, nx = n_tiles >> 3;        // this XCD's queue: gx blocks, nx tiles
    int* const queue = queues + qx;

    auto tile_origin = [&](int bx, int& m0, int& n0) {
        int tile_m = bx % tiles_m, tile_n = bx / tiles_m;
        if (xcd_map) {
            const int x = bx & 7, i = bx >> 3;
            const int mq = tiles_m >> 2, nh = (n_tiles / tiles_m) >> 1;
            tile_m = (x & 3) * mq + i % mq;
            tile_n = (x >> 2) * nh + i / mq;
        }
        m0 = tile_m * 128;
        n0 = tile_n * 128;
    };
    auto issue = [&](auto stg, const char* ak, const char* bk) __attribute__((always_inline)) {
        constexpr int STG = decltype(stg)::value;
        float* sa = smem + STG * (2 * TILE_FLOATS) + wave * 512;
        float* sb = sa + TILE_FLOATS;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gemm_lds_dma16(off[j], ak, (uint32_t)(uintptr_t)(lds_ptr_t)(sa + j * 256));
            gemm_lds_dma16(off[j], bk, (uint32_t)(uintptr_t)(lds_ptr_t)(sb + j * 256));
        }
    };

    int m0, n0;
    tile_origin(blockIdx.x, m0, n0);
    const char* a_base = reinterpret_cast<const char*>(A + (size_t)m0 * LD);
    const char* b_base = reinterpret_cast<const char*>(Bt + (size_t)n0 * LD);
    const char* a_next = a_base;
    const char* b_next = b_base;
    float* cp_cur[2];
    float* cp_prev[2] = {nullptr, nullptr};
    float* cp_next[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) cp_next[mt] = cp_cur[mt] = C + (size_t)(m0 + wm * 64 + mt * 32 + l31) * ldc + n0 + wn * 64 + 4 * half;
    float bias_cur[2], bias_next[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bias_cur[nt] = gemm_load_async(bias + n0 + wn * 64 + nt * 32 + l31);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias_cur[0]), "+v"(bias_cur[1])::"memory");
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bias_next[nt] = bias_cur[nt];
    gemm_static_for<0, NST - 1>([&](auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        issue(sc, a_base + S * BKT * 4, b_base + S * BKT * 4);
    });
    bool has_next = false;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // one tile: cur accumulates, prev (the tile before, if has_prev) is stored one quad per even k-step
    auto trip = [&](f32x16 (&cur)[2][2], f32x16 (&prev)[2][2], auto hp) __attribute__((always_inline)) {
        constexpr bool has_prev = decltype(hp)::value;
        int ticket = 0;
        if (tid == 0) ticket = gemm_atomic_inc_async(queue);
        gemm_static_for<0, T>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            constexpr int U = t % NST;
            // k-tile t has landed once only what was issued after it is outstanding: two k-tiles (8 transfers) and, with a tile being
            // stored, the four stores of k-step 0, 8, 16 or 24 if that is one of the three steps before this one; at the tail of a
            // block's last tile nothing follows.  (Whatever else is in flight -- the ticket, the bias words -- only makes a count
            // stricter than it need be.)
            if (t >= T - 2 && !has_next) {
                if (t == T - 2) gemm_wait_vmcnt<4>();
                else gemm_wait_vmcnt<0>();
            } else if (has_prev && t % 8 >= 1 && t % 8 <= 3) gemm_wait_vmcnt<12>();
            else gemm_wait_vmcnt<8>();
            __builtin_amdgcn_s_barrier();
            if (t == 19 && tid == 0) {
                const int i = gx + ticket;
                gemm_lds_write32(ticket_addr, i < nx ? qx + 8 * i : -1);
            }
            if (t == 21) {
                const int nb = __builtin_amdgcn_readfirstlane(gemm_lds_read32_wait(ticket_addr));
                has_next = nb >= 0;
                int m1, n1;
                tile_origin(has_next ? nb : (int)blockIdx.x, m1, n1);
                a_next = reinterpret_cast<const char*>(A + (size_t)m1 * LD);
                b_next = reinterpret_cast<const char*>(Bt + (size_t)n1 * LD);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) cp_next[mt] = C + (size_t)(m1 + wm * 64 + mt * 32 + l31) * ldc + n1 + wn * 64 + 4 * half;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bias_next[nt] = gemm_load_async(bias + n1 + wn * 64 + nt * 32 + l31);
            }
            constexpr int SOFF = U * STAGE_BYTES;
            gemm_f4 af[2][2], bf[2][2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                af[g][0] = gemm_lds_read128<SOFF>(fa[g]);
                af[g][1] = gemm_lds_read128<SOFF + 32 * BKT * 4>(fa[g]);
                bf[g][0] = gemm_lds_read128<SOFF>(fb[g]);
                bf[g][1] = gemm_lds_read128<SOFF + 32 * BKT * 4>(fb[g]);
            }
            if (t + NST - 1 < T) issue(std::integral_constant<int, (U + NST - 1) % NST>(), a_base + (t + NST - 1) * BKT * 4, b_base + (t + NST - 1) * BKT * 4);
            else if (has_next) issue(std::integral_constant<int, (U + NST - 1) % NST>(), a_next + (t + NST - 1 - T) * BKT * 4, b_next + (t + NST - 1 - T) * BKT * 4);
            if (t % 8 == 0 && has_prev) {
                // the four quads of one 32 x 32 tile together: with the lane pair's 32 B per row and store, they complete the
                // rows' 128-B lines while those are still in L2 (one quad per even k-step: 0.92 GB written instead of 0.52)
                constexpr int mt = t >> 4, nt = (t >> 3) & 1;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(cp_prev[mt] + nt * 32 + 8 * q) =
                        make_float4(prev[mt][nt][4 * q], prev[mt][nt][4 * q + 1], prev[mt][nt][4 * q + 2], prev[mt][nt][4 * q + 3]);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (g == 1) __builtin_amdgcn_sched_barrier(0);
                if (g == 0) gemm_wait_lgkm<4>(af[0][0], af[0][1], bf[0][0], bf[0][1]);
                else gemm_wait_lgkm<0>(af[1][0], af[1][1], bf[1][0], bf[1][1]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if (t == 0 && g == 0) cur[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].x, af[g][mt].x, zero16, 0, 0, 0);
                        else cur[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].x, af[g][mt].x, cur[mt][nt], 0, 0, 0);
                        cur[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].y, af[g][mt].y, cur[mt][nt], 0, 0, 0);
                        cur[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].z, af[g][mt].z, cur[mt][nt], 0, 0, 0);
                        cur[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].w, af[g][mt].w, cur[mt][nt], 0, 0, 0);
                    }
            }
        });
        const float one = half ? 0.f : 1.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float bv = half ? 0.f : bias_cur[nt];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) cur[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, one, cur[mt][nt], 0, 0, 0);
        }
    };
    auto flush = [&](f32x16 (&cur)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(cp_cur[mt] + nt * 32 + 8 * q) =
                        make_float4(cur[mt][nt][4 * q], cur[mt][nt][4 * q + 1], cur[mt][nt][4 * q + 2], cur[mt][nt][4 * q + 3]);
    };
    auto advance = [&]() __attribute__((always_inline)) {
        a_base = a_next;
        b_base = b_next;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            cp_prev[i] = cp_cur[i];
            cp_cur[i] = cp_next[i];
            bias_cur[i] = bias_next[i];
        }
    };
    f32x16 accA[2][2], accB[2][2];
    trip(accA, accB, std::false_type());
    if (!has_next) {
        flush(accA);
        return;
    }
    advance();
    for (;;) {
        trip(accB, accA, std::true_type());
        if (!has_next) { flush(accB); break; }
        advance();
        trip(accA, accB, std::true_type());
        if (!has_next) { flush(accA); break; }
        advance();
    }
}

// The split-K backward contraction the same way (slabs[z] = g_v_posed[:, kz] . Bk[:, kz]^T; EPI 1 of gemm_nt_f32_kernel): its tiles are
// 124 (last slab: 108) k-steps long, so the k-loop is a 28-step prefix (fully unrolled: the stores of the tile before, the ticket),
// a counted loop of four-step groups, and a last group that requests the next tile's first k-tiles.  XCD x owns the slabs z = x (mod 8)
// as in the one-tile launch (xcd_map 2), and its queue hands out their tiles with the four N tiles of an M tile together.  Whole
// 128-row tiles, N = 512, every slab a multiple of 64 columns and at least 512 long; 4 LDS stages (the one-tile launch: 5).
__global__ __launch_bounds__(256) void blend_bwd_xt_kernel(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C, int ld,
                                                           int M, int K, int kchunk, int tiles_m, int n_slab, int* __restrict__ queues) {
    constexpr int BKT = 16, NST = 4, TILE_FLOATS = 128 * BKT, STAGE_BYTES = 2 * TILE_FLOATS * 4, LDC = FEAT_K, TILES_N = FEAT_K / 128, PRE = 28;
    __shared__ __attribute__((aligned(1024))) float smem[NST * 2 * TILE_FLOATS + 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    auto swz = [](int row) { return (row >> 2) & 3; };
    uint32_t off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + lane / 4;
        off[j] = ((uint32_t)row * (uint32_t)ld + (((lane & 3) ^ swz(row)) * 4)) * 4u;
    }
    const int fsw = swz(l31);
    uint32_t fa[2], fb[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int koff = ((2 * g + half) ^ fsw) * 4;
        fa[g] = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + (wm * 64 + l31) * BKT + koff);
        fb[g] = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + TILE_FLOATS + (wn * 64 + l31) * BKT + koff);
    }
    const uint32_t ticket_addr = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + NST * 2 * TILE_FLOATS);
    const int per_slab = tiles_m * TILES_N;
    const int qx = blockIdx.x & 7, gx = gridDim.x >> 3, nx = per_slab * (n_slab >> 3);      // this XCD's queue: gx blocks, nx tiles
    int* const queue = queues + qx;
    uint32_t lane_c[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) lane_c[mt] = (uint32_t)((wm * 64 + mt * 32 + l31) * LDC + wn * 64 + 4 * half);

    // tile i of this XCD's queue -> operand bases (bytes), output tile origin (floats), k-steps
    auto tile_of = [&](int i, const char*& ab, const char*& bb, size_t& c0, int& T) {
        const int slab = qx + 8 * (i / per_slab), t = i % per_slab;
        const int m0 = (t / TILES_N) * 128, n0 = (t % TILES_N) * 128;
        const int k_begin = slab * kchunk;
        T = (min(K, k_begin + kchunk) - k_begin) / BKT;
        ab = reinterpret_cast<const char*>(A + (size_t)m0 * ld + k_begin);
        bb = reinterpret_cast<const char*>(Bt + (size_t)n0 * ld + k_begin);
        c0 = ((size_t)slab * M + m0) * LDC + n0;
    };
    auto issue = [&](auto stg, const char* ak, const char* bk) __attribute__((always_inline)) {
        constexpr int STG = decltype(stg)::value;
        float* sa = smem + STG * (2 * TILE_FLOATS) + wave * 512;
        float* sb = sa + TILE_FLOATS;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gemm_lds_dma16(off[j], ak, (uint32_t)(uintptr_t)(lds_ptr_t)(sa + j * 256));
            gemm_lds_dma16(off[j], bk, (uint32_t)(uintptr_t)(lds_ptr_t)(sb + j * 256));
        }
    };

    const char *a_base, *b_base, *a_next, *b_next;
    size_t c_cur, c_next;
    int T, T_next;
    tile_of(blockIdx.x >> 3, a_base, b_base, c_cur, T);
    a_next = a_base; b_next = b_base; c_next = c_cur; T_next = T;
    float* cp_cur[2];
    float* cp_prev[2] = {nullptr, nullptr};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) cp_cur[mt] = C + c_cur + lane_c[mt];
    gemm_static_for<0, NST - 1>([&](auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        issue(sc, a_base + S * BKT * 4, b_base + S * BKT * 4);
    });
    bool has_next = false;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto trip = [&](f32x16 (&cur)[2][2], f32x16 (&prev)[2][2], auto hp) __attribute__((always_inline)) {
        constexpr bool has_prev = decltype(hp)::value;
        int ticket = 0;
        if (tid == 0) ticket = gemm_atomic_inc_async(queue);
        // one k-step.  PH 0: step TC of the prefix (a constant); PH 1: a step of the counted loop, t = tr; PH 2: step T - 4 + TC
        auto kstep = [&](auto phc, auto tcc, const int tr) __attribute__((always_inline)) {
            constexpr int PH = decltype(phc)::value, TC = decltype(tcc)::value, U = TC % NST;
            // what is outstanding behind k-tile t: two k-tiles (8 transfers) and, behind the prefix steps 0, 8, 16, 24 of a tile whose
            // predecessor is being stored, that step's four stores; at the tail of a block's last tile nothing follows
            if (PH == 2 && U >= 2 && !has_next) {
                if (U == 2) gemm_wait_vmcnt<4>();
                else gemm_wait_vmcnt<0>();
            } else if (PH == 0 && has_prev && TC % 8 >= 1 && TC % 8 <= 3) gemm_wait_vmcnt<12>();
            else gemm_wait_vmcnt<8>();
            __builtin_amdgcn_s_barrier();
            if (PH == 0 && TC == 19 && tid == 0) {
                const int i = gx + ticket;
                gemm_lds_write32(ticket_addr, i < nx ? i : -1);
            }
            if (PH == 0 && TC == 21) {
                const int ni = __builtin_amdgcn_readfirstlane(gemm_lds_read32_wait(ticket_addr));
                has_next = ni >= 0;
                tile_of(has_next ? ni : 0, a_next, b_next, c_next, T_next);
            }
            constexpr int SOFF = U * STAGE_BYTES;
            gemm_f4 af[2][2], bf[2][2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                af[g][0] = gemm_lds_read128<SOFF>(fa[g]);
                af[g][1] = gemm_lds_read128<SOFF + 32 * BKT * 4>(fa[g]);
                bf[g][0] = gemm_lds_read128<SOFF>(fb[g]);
                bf[g][1] = gemm_lds_read128<SOFF + 32 * BKT * 4>(fb[g]);
            }
            constexpr auto stg = std::integral_constant<int, (U + NST - 1) % NST>();
            if (PH == 0) issue(stg, a_base + (TC + NST - 1) * BKT * 4, b_base + (TC + NST - 1) * BKT * 4);
            else if (PH == 1) issue(stg, a_base + (tr + NST - 1) * BKT * 4, b_base + (tr + NST - 1) * BKT * 4);
            else if (U == 0) issue(stg, a_base + (T - 1) * BKT * 4, b_base + (T - 1) * BKT * 4);
            else if (has_next) issue(stg, a_next + (U - 1) * BKT * 4, b_next + (U - 1) * BKT * 4);
            if (PH == 0 && TC % 8 == 0 && has_prev) {
                constexpr int mt = (TC >> 4) & 1, nt = (TC >> 3) & 1;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(cp_prev[mt] + nt * 32 + 8 * q) =
                        make_float4(prev[mt][nt][4 * q], prev[mt][nt][4 * q + 1], prev[mt][nt][4 * q + 2], prev[mt][nt][4 * q + 3]);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (g == 1) __builtin_amdgcn_sched_barrier(0);
                if (g == 0) gemm_wait_lgkm<4>(af[0][0], af[0][1], bf[0][0], bf[0][1]);
                else gemm_wait_lgkm<0>(af[1][0], af[1][1], bf[1][0], bf[1][1]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if (PH == 0 && TC == 0 && g == 0) cur[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].x, af[g][mt].x, zero16, 0, 0, 0);
                        else cur[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].x, af[g][mt].x, cur[mt][nt], 0, 0, 0);
                        cur[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].y, af[g][mt].y, cur[mt][nt], 0, 0, 0);
                        cur[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].z, af[g][mt].z, cur[mt][nt], 0, 0, 0);
                        cur[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[g][nt].w, af[g][mt].w, cur[mt][nt], 0, 0, 0);
                    }
            }
        };
        gemm_static_for<0, PRE>([&](auto tc) __attribute__((always_inline)) { kstep(std::integral_constant<int, 0>(), tc, 0); });
        for (int tt = PRE; tt < T - NST; tt += NST)
            gemm_static_for<0, NST>([&](auto uc) __attribute__((always_inline)) { kstep(std::integral_constant<int, 1>(), uc, tt + decltype(uc)::value); });
        gemm_static_for<0, NST>([&](auto uc) __attribute__((always_inline)) { kstep(std::integral_constant<int, 2>(), uc, 0); });
    };
    auto flush = [&](f32x16 (&cur)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(cp_cur[mt] + nt * 32 + 8 * q) =
                        make_float4(cur[mt][nt][4 * q], cur[mt][nt][4 * q + 1], cur[mt][nt][4 * q + 2], cur[mt][nt][4 * q + 3]);
    };
    auto advance = [&]() __attribute__((always_inline)) {
        a_base = a_next;
        b_base = b_next;
        T = T_next;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            cp_prev[i] = cp_cur[i];
            cp_cur[i] = C + c_next + lane_c[i];
        }
    };
    f32x16 accA[2][2], accB[2][2];
    trip(accA, accB, std::false_type());
    if (!has_next) {
        flush(accA);
        return;
    }
    advance();
    for (;;) {
        trip(accB, accA, std::true_type());
        if (!has_next) { flush(accB); break; }
        advance();
        trip(accA, accB, std::true_type());
        if (!has_next) { flush(accA); break; }
        advance();
    }
}

static int gemm_cu_count() {       // of the current device
    static int n_cu[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n_cu[dev] == 0) {
        int v = 0;
        n_cu[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return n_cu[dev];
}
// Whether the forward of this batch runs the persistent kernel (and needs its 8 queue counters zeroed before the launch): whole
// 128-row tiles and at least one tile per block of two blocks per CU (B >= 512 on the whole mesh; measured from one tile per block
// on: B = 512 0.137 -> 0.134 ms, B = 1,024 0.261 -> 0.249 ms).
// SAGA_GEMM_XT=0: never (one tile per block); =n: from n tiles per block on.
bool saga_blend_fwd_uses_queue(const saga_model* m, int B) {
    static const int want_xt = getenv("SAGA_GEMM_XT") ? atoi(getenv("SAGA_GEMM_XT")) : 1;
    static const int want_lepi = getenv("SAGA_GEMM_LEPI") ? atoi(getenv("SAGA_GEMM_LEPI")) : 0;
    if (want_xt <= 0 || want_lepi || B % BM != 0 || !m->v_template) return false;
    const int n_tiles = (B / BM) * (m->Npad / BN);
    return n_tiles % 8 == 0 && n_tiles >= want_xt * ((2 * gemm_cu_count()) & ~7);
}

// v_posed[B,Npad] = v_template + feat . Bt^T          (queues: 8 zeroed ints for saga_blend_fwd_uses_queue(m, B) batches, else unused)
int saga_launch_blend_fwd(const saga_model* m, const float* feat, float* v_posed, int B, hipStream_t st, int* queues) {
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
    // Fitting batch sizes: the persistent kernel with overlapping tiles
    if (saga_blend_fwd_uses_queue(m, B) && queues) {
        const int blocks = (2 * gemm_cu_count()) & ~7;
        blend_fwd_xt_kernel<<<dim3(blocks), 256, 0, st>>>(feat, m->Bt, v_posed, m->Npad, m->v_template, tiles_m, xcd_map, n_tiles, queues);
        SAGA_LAUNCH_CHECK();
        return 0;
    }
    if (want_lepi) {
        grid.x = n_tiles;
        gemm_nt_f32_kernel<0, GEMM_BKT, GEMM_NST_FWD, 2, true, true><<<grid, 256, 0, st>>>(feat, FEAT_K, m->Bt, FEAT_K, v_posed, m->Npad, m->v_template,
                                                                                            nullptr, B, m->Npad, FEAT_K, FEAT_K, tiles_m, xcd_map, n_tiles);
    } else {
        // One block per tile, handed out by the hardware.  (A persistent launch -- SAGA_GEMM_PERSIST=-1: two blocks per CU, =n: n blocks,
        // each walking an equal share of the tiles -- was the default while the k-loop still carried vector arithmetic: 1.068 -> 1.059
        // ms.  With the lean loop the time stamps show the two blocks of a CU running very unequally -- the wave that came first wins
        // the SIMD's arbitration and takes a tile in ~100 k cycles, its partner fills the gaps and needs ~260 k -- so equal shares
        // leave the second block alone on its CU for the last third of the launch: 1.027 ms against 0.993 for this form.)
        static const int persist_env = getenv("SAGA_GEMM_PERSIST") ? atoi(getenv("SAGA_GEMM_PERSIST")) : 0;
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

// Whether the backward of this batch runs the persistent kernel (and needs its 8 queue counters zeroed before the launch): whole
// 128-row tiles, slabs dealt to the XCDs whole (a multiple of 8 of them), every slab a whole number of four-step groups and at
// least 32 k-steps long, and at least one tile per block of two blocks per CU (B >= 1,024 on the whole mesh; B = 1,024, one tile per
// block: 0.237 -> 0.228 ms; B = 2,048: 0.460 -> 0.449; B = 4,096: 0.91 -> 0.885).
// SAGA_GEMM_XT=0 or SAGA_GEMM_XT_BWD=0: never.
bool saga_blend_bwd_uses_queue(const saga_model* m, int B) {
    static const int want_xt = getenv("SAGA_GEMM_XT") ? atoi(getenv("SAGA_GEMM_XT")) : 1;
    static const int want_bwd = getenv("SAGA_GEMM_XT_BWD") ? atoi(getenv("SAGA_GEMM_XT_BWD")) : 1;
    static const int want_xcd = getenv("SAGA_GEMM_XCD") ? atoi(getenv("SAGA_GEMM_XCD")) : 1;
    if (want_xt <= 0 || !want_bwd || !want_xcd || B <= 0 || B % BM != 0) return false;
    const int splitk = saga_blend_bwd_splitk(m);
    if (splitk % 8 != 0) return false;
    const int kchunk = saga_roundup((m->Npad + splitk - 1) / splitk, 32);
    const int last = m->Npad - (splitk - 1) * kchunk;
    if (kchunk % (4 * GEMM_BKT) != 0 || last % (4 * GEMM_BKT) != 0 || last < 32 * GEMM_BKT) return false;
    const int n_tiles = (B / BM) * (FEAT_K / BN) * splitk;
    return n_tiles >= want_xt * ((2 * gemm_cu_count()) & ~7);
}

// slabs[z][B][512] = g_v_posed[:, kz] . Bk[:, kz]^T          (queues: 8 zeroed ints for saga_blend_bwd_uses_queue(m, B) batches, else unused)
int saga_launch_blend_bwd(const saga_model* m, const float* g_vposed, float* slabs, int B, hipStream_t st, int* queues) {
    const int tiles_m = (B + BM - 1) / BM, tiles_n = FEAT_K / BN;
    const int splitk = saga_blend_bwd_splitk(m);
    const int kchunk = saga_roundup((m->Npad + splitk - 1) / splitk, 32);
    if (queues && saga_blend_bwd_uses_queue(m, B)) {
        const int blocks = (2 * gemm_cu_count()) & ~7;
        blend_bwd_xt_kernel<<<dim3(blocks), 256, 0, st>>>(g_vposed, m->Bk, slabs, m->Npad, B, m->Npad, kchunk, tiles_m, splitk, queues);
        SAGA_LAUNCH_CHECK();
        return 0;
    }
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
