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
#include "saga_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 128
#define BK 32
#define TILE_FLOATS (BM * BK)   // one operand tile: 128 rows x 32 floats = 16 KiB, 16 LDS-DMA pieces of 1 KiB

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// EPI 0: C[row][col] = acc + bias[col]           (C row stride ldc)
// EPI 1: C[z][row][col] = acc                    (slab z = blockIdx.z, slab stride M*ldc)
template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, int lda,
                                                          const float* __restrict__ Bt, int ldb, float* __restrict__ C,
                                                          int ldc, const float* __restrict__ bias, int M, int N, int K,
                                                          int kchunk, int tiles_m) {
    // ONE shared array (a second LDS object beside an LDS-DMA target makes hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(1024))) float smem[2 * 2 * TILE_FLOATS];   // [stage][A|B][128][32]

    // consecutive blocks walk M first so the (large, batch-shared) Bt tile stays hot in L2 while A is re-read
    const int tile_m = blockIdx.x % tiles_m;
    const int tile_n = blockIdx.x / tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k_begin = blockIdx.z * kchunk;
    const int k_end = min(K, k_begin + kchunk);
    const int T = (k_end - k_begin) / BK;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    // LDS-DMA source addresses: piece p = wave*4 + j covers tile rows 8p .. 8p+7; lane -> (row 8p + lane>>3, slot lane&7)
    // and fetches the 16-B k-chunk slot ^ ((row>>1)&7) of that row.
    const float* a_src[4];
    const float* b_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_src[j] = A + (size_t)min(m0 + row, M - 1) * lda + k_begin + chunk * 4;
        b_src[j] = Bt + (size_t)min(n0 + row, N - 1) * ldb + k_begin + chunk * 4;
    }
    auto issue = [&](int stage, int koff) {
        float* sa = smem + stage * (2 * TILE_FLOATS) + wave * 1024;
        float* sb = sa + TILE_FLOATS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[j] + koff), (lds_ptr_t)(sa + j * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(b_src[j] + koff), (lds_ptr_t)(sb + j * 256), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (floats): row * 32 + ((2g + half) ^ fsw) * 4, fsw = (row>>1)&7 = (l31>>1)&7
    const int fsw = (l31 >> 1) & 7;
    const int a_row = (wm * 64 + l31) * BK, b_row = (wn * 64 + l31) * BK;

    if (T > 0) issue(0, 0);
    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t have landed
        __syncthreads();                                    // everyone's have; everyone is done reading stage (t+1)&1
        const float* As = smem + (t & 1) * (2 * TILE_FLOATS);
        const float* Bs = As + TILE_FLOATS;
        // all 16 fragment reads of this k-tile first, THEN the LDS-DMA of the next tile: hipcc waits vmcnt(0) before any
        // ds_read that follows an outstanding LDS-DMA in program order, which would serialise the prefetch with the MFMAs
        float4 af[BK / 8][2], bf[BK / 8][2];
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            const int koff = ((2 * g + half) ^ fsw) * 4;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                af[g][u] = *reinterpret_cast<const float4*>(As + a_row + u * 32 * BK + koff);
                bf[g][u] = *reinterpret_cast<const float4*>(Bs + b_row + u * 32 * BK + koff);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < T) issue((t + 1) & 1, (t + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][mt].x, bf[g][nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][mt].y, bf[g][nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][mt].z, bf[g][nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][mt].w, bf[g][nt].w, acc[mt][nt], 0, 0, 0);
                }
        }
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cz = C;
    if (EPI == 1) Cz += (size_t)blockIdx.z * M * ldc;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = n0 + wn * 64 + nt * 32 + l31;
        if (col >= N) continue;
        const float bv = (EPI == 0) ? bias[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < M) Cz[(size_t)row * ldc + col] = acc[mt][nt][r] + bv;
            }
        }
    }
}

// v_posed[B,Npad] = v_template + feat . Bt^T
int saga_launch_blend_fwd(const saga_model* m, const float* feat, float* v_posed, int B, hipStream_t st) {
    const int tiles_m = (B + BM - 1) / BM, tiles_n = m->Npad / BN;
    dim3 grid(tiles_m * tiles_n, 1, 1);
    gemm_nt_f32_kernel<0><<<grid, 256, 0, st>>>(feat, FEAT_K, m->Bt, FEAT_K, v_posed, m->Npad, m->v_template, B, m->Npad,
                                                FEAT_K, FEAT_K, tiles_m);
    SAGA_LAUNCH_CHECK();
    return 0;
}

// slabs[z][B][512] = g_v_posed[:, kz] . Bk[:, kz]^T
int saga_launch_blend_bwd(const saga_model* m, const float* g_vposed, float* slabs, int B, hipStream_t st) {
    const int tiles_m = (B + BM - 1) / BM, tiles_n = FEAT_K / BN;
    const int kchunk = saga_roundup((m->Npad + SAGA_BWD_SPLITK - 1) / SAGA_BWD_SPLITK, BK);
    dim3 grid(tiles_m * tiles_n, 1, SAGA_BWD_SPLITK);
    gemm_nt_f32_kernel<1><<<grid, 256, 0, st>>>(g_vposed, m->Npad, m->Bk, m->Npad, slabs, FEAT_K, nullptr, B, FEAT_K,
                                                m->Npad, kchunk, tiles_m);
    SAGA_LAUNCH_CHECK();
    return 0;
}
