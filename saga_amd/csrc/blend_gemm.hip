// FP32-MFMA "NT" GEMM for the blend-shape contraction of SMPL-X (the dominant kernel of the path):
//   forward : v_posed[B, Npad]  = v_template + feat[B,512] . Bt[Npad,512]^T      (shape + pose blend shapes, K = 20+486 -> 512)
//   backward: g_feat[B, 512]    = g_v_posed[B,Npad] . Bk[512,Npad]^T              (split-K slabs, summed by pose_bwd_kernel)
// Both operands are stored with the contraction index contiguous, so global loads are 16 B/lane and the
// LDS image is [row][k] read back with ds_read_b128.  v_mfma_f32_32x32x2_f32 is an exact f32 FMA chain
// (no xf32 on gfx950): lanes 0-31 supply k = 8g+t, lanes 32-63 k = 8g+4+t for MFMA t of k-group g -- any
// k permutation is valid as long as A and B agree.
// Replaces smplx.lbs.blend_shapes + the posedirs matmul.
#include "saga_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 128
#define BK 32
#define LDS_LD (BK + 4)   // 36 floats: conflict-free ds_read_b128 over the 16-lane groups

// EPI 0: C[row][col] = acc + bias[col]           (C row stride ldc)
// EPI 1: C[z][row][col] = acc                    (slab z = blockIdx.z, slab stride M*ldc)
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_f32_kernel(const float* __restrict__ A, int lda,
                                                             const float* __restrict__ Bt, int ldb,
                                                             float* __restrict__ C, int ldc,
                                                             const float* __restrict__ bias, int M, int N, int K,
                                                             int kchunk, int tiles_m) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];

    // consecutive blocks walk M first so the (large, batch-shared) Bt tile stays hot in L2 while A is re-read
    const int tile_m = blockIdx.x % tiles_m;
    const int tile_n = blockIdx.x / tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k_begin = blockIdx.z * kchunk;
    const int k_end = min(K, k_begin + kchunk);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    // global -> register staging: thread covers rows (tid>>3) + 32*i, float4 column (tid&7)
    const int ld_row = tid >> 3, ld_c4 = (tid & 7) * 4;
    const float* a_ptr[4];
    const float* b_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = min(m0 + ld_row + 32 * i, M - 1);
        const int rb = min(n0 + ld_row + 32 * i, N - 1);
        a_ptr[i] = A + (size_t)ra * lda + ld_c4;
        b_ptr[i] = Bt + (size_t)rb * ldb + ld_c4;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[4];
    if (k_begin < k_end) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const float4*>(a_ptr[i] + k_begin);
            rb[i] = *reinterpret_cast<const float4*>(b_ptr[i] + k_begin);
        }
    }
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&As[(ld_row + 32 * i) * LDS_LD + ld_c4]) = ra[i];
            *reinterpret_cast<float4*>(&Bs[(ld_row + 32 * i) * LDS_LD + ld_c4]) = rb[i];
        }
        __syncthreads();
        if (k0 + BK < k_end) {   // prefetch the next k-tile while this one is multiplied
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *reinterpret_cast<const float4*>(a_ptr[i] + k0 + BK);
                rb[i] = *reinterpret_cast<const float4*>(b_ptr[i] + k0 + BK);
            }
        }
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            const int koff = g * 8 + half * 4;
            float4 af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = *reinterpret_cast<const float4*>(&As[(wm * 64 + t * 32 + l31) * LDS_LD + koff]);
                bf[t] = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + t * 32 + l31) * LDS_LD + koff]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].x, bf[nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].y, bf[nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].z, bf[nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].w, bf[nt].w, acc[mt][nt], 0, 0, 0);
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
