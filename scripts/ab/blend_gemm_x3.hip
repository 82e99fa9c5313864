// FP32-accurate blend contraction on the BF16 matrix cores ("bf16x3"): every FP32 operand is split exactly into three
// bf16 pieces x = h + m + l (8 + 8 + 8 significant bits), and the product is rebuilt from the six largest cross terms
//     a.b ~= al.bh + ah.bl + am.bm + am.bh + ah.bm + ah.bh        (dropped: am.bl, al.bm, al.bl  <= 2^-23 |a||b|)
// each a v_mfma_f32_32x32x16_bf16 with FP32 accumulation.  gfx950 runs bf16 MFMA at 16x the FP32-MFMA rate, so six
// bf16 MFMAs cost 192 cycles per 32x32x16 block against 512 for eight v_mfma_f32_32x32x2_f32: 2.7x the throughput at
// FP32-level accuracy (bf16 x bf16 products are exact in FP32; only the accumulation rounds, as in the FP32 kernel).
// EXPERIMENT (scripts/ab/gemm_x3_ab.hip), not in the library: measured on MI355X at M=4096, N=31488, K=512 the accuracy equals
// the FP32-MFMA kernel (max error vs double 1.6e-7 / 1.7e-7 forward) but at 128x128 tiles the kernel is bound by LDS-DMA
// issue (12 one-KiB pieces per wave per 48 MFMAs): 1.03 ms forward, 0.81 ms backward against 1.06 / 0.98 for FP32 MFMA.
// A 256x256 tile (half the pieces per MFMA) is the next step before this can replace the FP32 kernel.
//
// Same tiling machinery as gemm_nt_f32_kernel: 128x128 block tile, 64-byte tile rows (32 bf16 of K), LDS-DMA staging with
// the source-side XOR swizzle, NST stages; a stage holds six tiles (A h/m/l, B h/m/l) = 48 KB.
#include "saga_common.h"   // -I saga_amd/csrc

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* x3_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* x3_gbl_ptr_t;

#define X3_BM 128
#define X3_BN 128
#define X3_BK 32                 // bf16 per tile row (64 B)
#define X3_TILE_BYTES (X3_BM * X3_BK * 2)
#define X3_STAGE_BYTES (6 * X3_TILE_BYTES)

template <int N>
__device__ __forceinline__ void x3_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else static_assert(N == 0, "unsupported vmcnt");
}

// x -> (h, m, l) bf16, round-to-nearest-even at every step; x - h and (x - h) - m are exact in FP32
__device__ __forceinline__ void x3_split(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r = x - (float)h;
    m = (__bf16)r;
    l = (__bf16)(r - (float)m);
}

// src [rows, ld_src] fp32 (cols used) -> dst pieces [3][rows][ld_dst] bf16 (columns >= cols zero-filled up to ld_dst)
__global__ void x3_split_kernel(const float* __restrict__ src, int ld_src, int rows, int cols, __bf16* __restrict__ dst,
                                size_t piece_stride, int ld_dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)rows * ld_dst;
    if (i >= total) return;
    const int r = (int)(i / ld_dst), c = (int)(i % ld_dst);
    const float x = c < cols ? src[(size_t)r * ld_src + c] : 0.f;
    __bf16 h, m, l;
    x3_split(x, h, m, l);
    dst[i] = h; dst[piece_stride + i] = m; dst[2 * piece_stride + i] = l;
}

// EPI 0: C[row][col] = acc + bias[col];  EPI 1: C[z][row][col] = acc (split-K slabs, z = blockIdx.z)
template <int EPI, int NST>
__global__ __launch_bounds__(256) void gemm_nt_bf16x3_kernel(const __bf16* __restrict__ A3, size_t a_piece, int lda,
                                                             const __bf16* __restrict__ B3, size_t b_piece, int ldb,
                                                             float* __restrict__ C, int ldc, const float* __restrict__ bias, int M,
                                                             int N, int K, int kchunk, int tiles_m) {
    // ONE shared array (see gemm_nt_f32_kernel); [stage][A h,m,l | B h,m,l][128 rows][64 B]
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * X3_STAGE_BYTES];
    const int tile_m = blockIdx.x % tiles_m, tile_n = blockIdx.x / tiles_m;
    const int m0 = tile_m * X3_BM, n0 = tile_n * X3_BN;
    const int k_begin = blockIdx.z * kchunk;
    const int k_end = min(K, k_begin + kchunk);
    const int T = (k_end - k_begin) / X3_BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    auto swz = [](int row) { return (row >> 2) & 3; };     // 64-B rows: four rows per 256-B bank row

    // LDS-DMA: 1-KiB piece p = wave*2 + j of a tile covers rows 16p..16p+15; lane -> (row, 16-B slot ^ swz(row))
    const __bf16* a_src[2];
    const __bf16* b_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ swz(row);
        a_src[j] = A3 + (size_t)min(m0 + row, M - 1) * lda + k_begin + chunk * 8;
        b_src[j] = B3 + (size_t)min(n0 + row, N - 1) * ldb + k_begin + chunk * 8;
    }
    auto issue = [&](int stage, int koff) {
        unsigned char* s = smem + stage * X3_STAGE_BYTES + wave * 2048;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                __builtin_amdgcn_global_load_lds((x3_gbl_ptr_t)(a_src[j] + p * a_piece + koff),
                                                 (x3_lds_ptr_t)(s + p * X3_TILE_BYTES + j * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((x3_gbl_ptr_t)(b_src[j] + p * b_piece + koff),
                                                 (x3_lds_ptr_t)(s + (3 + p) * X3_TILE_BYTES + j * 1024), 16, 0, 0);
            }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fsw = swz(l31);
    const int a_row = (wm * 64 + l31) * 64, b_row = (wn * 64 + l31) * 64;    // byte offsets inside a tile

#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < T) issue(s, s * X3_BK);
    for (int t = 0; t < T; ++t) {
        if (NST > 2 && t + NST - 2 < T) x3_wait_vmcnt<(NST - 2) * 12>();
        else x3_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const unsigned char* S = smem + (t % NST) * X3_STAGE_BYTES;
        // every fragment of this k-tile first, then the LDS-DMA of a later tile (see gemm_nt_f32_kernel)
        bf16x8 af[2][3][2], bf[2][3][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int koff = ((2 * kk + half) ^ fsw) * 16;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    af[kk][p][u] = *reinterpret_cast<const bf16x8*>(S + p * X3_TILE_BYTES + a_row + u * 32 * 64 + koff);
                    bf[kk][p][u] = *reinterpret_cast<const bf16x8*>(S + (3 + p) * X3_TILE_BYTES + b_row + u * 32 * 64 + koff);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + NST - 1 < T) issue((t + NST - 1) % NST, (t + NST - 1) * X3_BK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x16 c = acc[mt][nt];
                    // smallest cross terms first: l.h, h.l, m.m, m.h, h.m, h.h
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][2][mt], bf[kk][0][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][0][mt], bf[kk][2][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][1][mt], bf[kk][1][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][1][mt], bf[kk][0][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][0][mt], bf[kk][1][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][0][mt], bf[kk][0][nt], c, 0, 0, 0);
                    acc[mt][nt] = c;
                }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cz = C;
    if (EPI == 1) Cz += (size_t)blockIdx.z * M * ldc;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = n0 + wn * 64 + nt * 32 + l31;
        if (col >= N) continue;
        const float bv = (EPI == 0 && bias) ? bias[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < M) Cz[(size_t)row * ldc + col] = acc[mt][nt][r] + bv;
            }
    }
}
