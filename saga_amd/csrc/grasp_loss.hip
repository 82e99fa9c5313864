// Fused contact / penetration loss of GraspPose-Opt: everything WholeGraspPose/models/fittingop.py:152-212 does
// between the SMPL-X forward and the weighted sum -- marker / foot / hand gathers, right-hand vertex normals
// (Open3D on the host in the reference, :158-164), the two point2point_signed calls (utils/train_helper.py:17-86:
// bidirectional nearest neighbours, gathered difference vectors, norms, normal-dot signs) and the per-candidate
// means -- in ONE kernel per direction, one workgroup per candidate.
//
// Forward: object cloud tiles, the 778 hand vertices + 143 markers and the hand normals live in LDS; lanes own
// query points and scan the other cloud with broadcast ds_read_b128 (same exact-rounding distance and strict '<'
// as saga_chamfer_fwd, so the indices are identical to it).  FP32-VALU bound.
// Backward: analytic gradient of the eight loss terms w.r.t. the gathered vertices, accumulated per candidate in
// LDS (object points scatter-add into their nearest hand vertex -> LDS float atomics) and written as the dense
// grad_vertices [B,V,3] the SMPL-X backward consumes.
#include <math.h>

#include "saga_common.h"
#include "nn_scan.h"
#include "nn_fast.h"
#include "vertex_normal.h"

void* saga_prof_begin(int id, void* stream);
void saga_prof_end(void* scope);

#define GL_THREADS 256          // forward: 4 waves; lanes own 4 queries (x->y) / 8 object points (y->x)
#define GL_BWD_THREADS 512
#define GL_TILE 2048            // object points staged per LDS tile
#define GL_MAX_HAND 1024
#define GL_MAX_MARK 256
#define GL_MAX_UNIQUE 2048
#define GL_NQ 4                 // queries per lane, hand/marker -> object  (4 * 256 = 1024 queries)
#define GL_NO 8                 // object points per lane, object -> hand   (8 * 256 = 2048 points per pass)
#define GL_NSUM 12
#define NT SAGA_GRASP_NTERMS

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// Sum over the wave, valid in lane 0: the additions of the __shfl_down tree (offsets 32, 16, .., 1) in the same order, so lane 0 holds
// the same bits -- but with the lane exchanges as v_permlane32_swap / ds_swizzle, whose patterns are immediates: __shfl_down keeps
// six per-lane ds_bpermute addresses live from the first reduction of a kernel to its last (they were spilled to scratch).
__device__ __forceinline__ float gl_wave_sum_lane0(float v) {
    {
        const unsigned u = (unsigned)__float_as_int(v);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // [1]: lanes 0..31 hold lanes 32..63
        v += __int_as_float((int)sw[1]);
    }
#define GL_SWZ_XOR(o) __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), ((o) << 10) | 0x1f))
    v += GL_SWZ_XOR(16);
    v += GL_SWZ_XOR(8);
    v += GL_SWZ_XOR(4);
    v += GL_SWZ_XOR(2);
    v += GL_SWZ_XOR(1);
#undef GL_SWZ_XOR
    return v;
}

// prior-contact weight of hand vertex i (fittingop.py:195,199): rhand_weight where rhand_weight^(1/2.5) > 0.7, else 0.  The mask
// depends on the table only: the host precomputes it (tables' hand_prior_w) when no entry sits within rounding of the threshold;
// tables without it evaluate the power here.
__device__ __forceinline__ float gl_prior_weight(const saga_grasp_tables_t& tb, int i) {
    if (tb.hand_prior_w) return tb.hand_prior_w[i];
    const float vw = tb.hand_weights[i];
    return powf(vw, 1.0f / 2.5f) > 0.7f ? vw : 0.f;
}

__global__ __launch_bounds__(GL_THREADS) void grasp_loss_fwd_kernel(
    saga_grasp_tables_t tb, const float* __restrict__ verts, const int32_t* __restrict__ vf_ptr,
    const int32_t* __restrict__ vf_tri, const float* __restrict__ obj,
    const float* __restrict__ obj_n, const float* __restrict__ mk_target, const float* __restrict__ c_mk,
    const float* __restrict__ c_obj, float* __restrict__ terms, float* __restrict__ markers_rec,
    float* __restrict__ hand_rec, float* __restrict__ hand_normals, float* __restrict__ h2o_marker,
    float* __restrict__ h2o_signed, float* __restrict__ o2h_signed, int32_t* __restrict__ idx_m2o,
    int32_t* __restrict__ idx_h2o, int32_t* __restrict__ idx_o2h, int V, int O) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float sObj[NN_TILE_FLOATS(GL_TILE)];        // paired layout (nn_scan.h)
    __shared__ __attribute__((aligned(16))) float sHand[NN_TILE_FLOATS(GL_MAX_HAND)];   // hand vertices, paired + padded
    __shared__ float sNrm[GL_MAX_HAND * 3];
    __shared__ float red[GL_THREADS / 64][GL_NSUM];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int H = tb.n_hand, M = tb.n_markers, NQ = H + M;
    const int Hpad = (H + NN_CH - 1) / NN_CH * NN_CH;
    const float* vb = verts + (size_t)b * V * 3;
    const float* ob = obj + (size_t)b * O * 3;
    const float* onb = obj_n + (size_t)b * O * 3;

    // ---- x -> y queries: lane owns hand vertices / markers q = tid + i*GL_THREADS, i < GL_NQ ------------------
    float qx[GL_NQ], qy[GL_NQ], qz[GL_NQ];
#pragma unroll
    for (int i = 0; i < GL_NQ; ++i) {
        const int q = tid + i * GL_THREADS;
        const int qc = min(q, NQ - 1);
        const int v = qc < H ? tb.hand_ids[qc] : tb.marker_ids[qc - H];
        qx[i] = vb[v * 3]; qy[i] = vb[v * 3 + 1]; qz[i] = vb[v * 3 + 2];
        if (q < NQ) {
            float* dst = q < H ? hand_rec + ((size_t)b * H + q) * 3 : markers_rec + ((size_t)b * M + (q - H)) * 3;
            dst[0] = qx[i]; dst[1] = qy[i]; dst[2] = qz[i];
            if (q < H) {
                float* d = sHand + (q >> 1) * 6 + (q & 1);
                d[0] = qx[i]; d[2] = qy[i]; d[4] = qz[i];
            }
        }
    }
    for (int q = H + tid; q < Hpad; q += GL_THREADS) {
        float* d = sHand + (q >> 1) * 6 + (q & 1);
        d[0] = NN_FAR; d[2] = NN_FAR; d[4] = NN_FAR;
    }
    // ---- hand normals (area-weighted, ascending face order) --------------------------------------------------
    for (int i = tid; i < H; i += GL_THREADS) {
        float nx, ny, nz;
        saga_vertex_normal(vb, vf_ptr, vf_tri, tb.hand_ids[i], nx, ny, nz);
        sNrm[i * 3] = nx; sNrm[i * 3 + 1] = ny; sNrm[i * 3 + 2] = nz;
        float* dst = hand_normals + ((size_t)b * H + i) * 3;
        dst[0] = nx; dst[1] = ny; dst[2] = nz;
    }

    float sums[GL_NSUM];
#pragma unroll
    for (int i = 0; i < GL_NSUM; ++i) sums[i] = 0.f;
    // sums: 0 rec 1 body_rec 2 foot 3 marker contact 4 object contact 5 prior 6 hand colli 7 obj colli
    //       8 n hand colli 9 n obj colli 10 n contact 11 n marker colli

    // ---- x -> y : every hand vertex / marker finds its nearest object point -------------------------------
    {
        float best[GL_NQ];
        int bc[GL_NQ], bi[GL_NQ];
#pragma unroll
        for (int i = 0; i < GL_NQ; ++i) { best[i] = INFINITY; bc[i] = -1; bi[i] = 0; }
        for (int t0 = 0; t0 < O; t0 += GL_TILE) {
            const int cnt = min(GL_TILE, O - t0);
            if (t0) __syncthreads();
            const int cnt_pad = nn_stage_tile(sObj, ob + (size_t)t0 * 3, cnt, tid, GL_THREADS);
            __syncthreads();                 // also publishes sHand / sNrm on the first pass
            nn_scan_tile<GL_NQ>(sObj, cnt_pad, t0, qx, qy, qz, best, bc);
            nn_resolve_tile<GL_NQ>(sObj, t0, qx, qy, qz, best, bc, bi);
        }
        // per-query terms
#pragma unroll
        for (int i = 0; i < GL_NQ; ++i) {
            const int q = tid + i * GL_THREADS;
            if (q >= NQ) continue;
            const float x = qx[i], y = qy[i], z = qz[i];
            const int k = bi[i];
            const float dx = x - ob[k * 3], dy = y - ob[k * 3 + 1], dz = z - ob[k * 3 + 2];
            const float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
            if (q < H) {
                // h2o_signed: sign from the OBJECT normal at the neighbour (train_helper.py:76-79)
                const float dot = (onb[k * 3] * dx + onb[k * 3 + 1] * dy) + onb[k * 3 + 2] * dz;
                const float d = dist * sgn(dot);
                h2o_signed[(size_t)b * H + q] = d;
                idx_h2o[(size_t)b * H + q] = k;
                const float pw = gl_prior_weight(tb, q);                                        // vw where vw^(1/2.5) > 0.7, else 0
                const bool near = (d < 0.02f) && (d > 0.f);                                      // fittingop.py:195
                sums[5] += fabsf(d) * (near ? pw : 0.f);                                          // :199,210
                sums[6] += fabsf(d) * (d < 0.f ? 10.f : 0.f);                                     // :200,211
                sums[8] += d < 0.f ? 1.f : 0.f;
                sums[10] += (d < 0.001f && d > -0.001f) ? 1.f : 0.f;
            } else {
                const int p = q - H;
                // marker call (:183): y_normals is None -> unsigned distance
                h2o_marker[(size_t)b * M + p] = dist;
                idx_m2o[(size_t)b * M + p] = k;
                sums[3] += dist * c_mk[(size_t)b * M + p];                                        // :190
                const float* tg = mk_target + ((size_t)b * M + p) * 3;
                const float e = (fabsf(x - tg[0]) + fabsf(y - tg[1])) + fabsf(z - tg[2]);
                sums[0] += e;                                                                     // :168
                if (p < tb.n_body_markers) sums[1] += e;                                          // :169
                sums[11] += dist < 0.f ? 1.f : 0.f;
            }
        }
    }
    // foot term: 0.1 * mean |z| (:173)
    for (int i = tid; i < tb.n_foot; i += GL_THREADS) sums[2] += fabsf(vb[tb.foot_ids[i] * 3 + 2]);

    // ---- y -> x : every object point finds its nearest HAND vertex (the marker direction is never used) ------
    // the hand cloud sHand[0 .. Hpad) is the (single) database tile; lane owns object points k0 + tid + i*GL_THREADS
    for (int k0 = 0; k0 < O; k0 += GL_THREADS * GL_NO) {
        float ox[GL_NO], oy[GL_NO], oz[GL_NO], best[GL_NO];
        int bc[GL_NO], bj[GL_NO];
#pragma unroll
        for (int i = 0; i < GL_NO; ++i) {
            const int k = min(k0 + tid + i * GL_THREADS, O - 1);
            ox[i] = ob[k * 3]; oy[i] = ob[k * 3 + 1]; oz[i] = ob[k * 3 + 2];
            best[i] = INFINITY; bc[i] = -1; bj[i] = 0;
        }
        nn_scan_tile<GL_NO>(sHand, Hpad, 0, ox, oy, oz, best, bc);
        nn_resolve_tile<GL_NO>(sHand, 0, ox, oy, oz, best, bc, bj);
#pragma unroll
        for (int i = 0; i < GL_NO; ++i) {
            const int k = k0 + tid + i * GL_THREADS;
            if (k >= O) continue;
            const int j = bj[i];
            float hx, hy, hz;
            nn_tile_point(sHand, j, hx, hy, hz);
            const float dx = ox[i] - hx, dy = oy[i] - hy, dz = oz[i] - hz;
            const float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
            const float dot = (sNrm[j * 3] * dx + sNrm[j * 3 + 1] * dy) + sNrm[j * 3 + 2] * dz;   // HAND normal (:68-71)
            const float d = dist * sgn(dot);
            o2h_signed[(size_t)b * O + k] = d;
            idx_o2h[(size_t)b * O + k] = j;
            sums[4] += fabsf(d) * c_obj[(size_t)b * O + k];                                  // :191
            sums[7] += fabsf(d) * (d < 0.f ? 20.f : 0.f);                                    // :206,212
            sums[9] += d < 0.f ? 1.f : 0.f;
        }
    }

    // ---- per-candidate means: wave shuffles, then the four wave partials in wave order ---------------------
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int i = 0; i < GL_NSUM; ++i) {
        float v = sums[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float sm[GL_NSUM];
#pragma unroll
        for (int i = 0; i < GL_NSUM; ++i) sm[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
        float* t = terms + (size_t)b * NT;
        t[0] = sm[0] / (float)(M * 3);
        t[1] = sm[1] / (float)(tb.n_body_markers * 3);
        t[2] = 0.1f * (sm[2] / (float)tb.n_foot);
        t[3] = sm[3] / (float)M;
        t[4] = sm[4] / (float)O;
        t[5] = sm[5] / (float)H;
        t[6] = sm[6] / (float)H;
        t[7] = sm[7] / (float)O;
        t[8] = sm[8]; t[9] = sm[9]; t[10] = sm[10]; t[11] = sm[11];
        t[12] = 0.f; t[13] = 0.f; t[14] = 0.f; t[15] = 0.f;
    }
}

#if defined(SAGA_EXPERIMENT_BUILD) && defined(GL_PHASE_TIMING)
__device__ unsigned long long gl_phase[8];
extern "C" int saga_debug_read_phases(unsigned long long* out) {
    unsigned long long z[8] = {0};
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(gl_phase), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(gl_phase), z, sizeof(z));
    return (int)e;
}
#define GL_T(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); atomicAdd(&gl_phase[i], t_ - t_prev); t_prev = t_; } } while (0)
#define GL_T0() unsigned long long t_prev = clock64()
#else
#define GL_T(i)
#define GL_T0()
#endif

// ---- forward, object clouds that fit one LDS tile (O <= GL_TILE: every call SAGA makes) -------------------------------
// Same outputs as grasp_loss_fwd_kernel, bit for bit, with the two-level scan of nn_fast.h: the object tile and the hand
// tile live in LDS in their own centred frames as (-2x,-2y,-2z,|p|^2); the approximate pass costs 3 fmas per pair, the
// exact pass re-evaluates the two candidate chunks of every query with the arithmetic of nn_scan.h.
#define GLF_NO 4                // object points per lane and pass, object -> hand
#ifndef GL_PAIR_XY
#define GL_PAIR_XY true         // hand / marker -> object scan (64 object tiles): one sorted insert per TWO matrix tiles (nn_fast.h, PAIR)
#endif
#define GL_OPL (GL_TILE / GL_THREADS)   // object points per lane of the fast kernel (8)
#ifndef GL_NRM_FPS
#define GL_NRM_FPS 2             // faces per vertex and gather step of the hand normals (3: 1.22 ms against 1.10, registers)
#endif
// PRE (saga_fit_step): the vertices the loss reads come as compact rows uniq_rows [B, n_unique, 3] (markers / hand / feet through
// the tables' *_slot arrays) and hand_normals is an INPUT -- both written by the whole-candidate skinning kernel (skin.hip,
// skin_fwd_cand_kernel); this kernel then touches neither the mesh nor its faces.
__host__ __device__ constexpr int glf_hand_off(int H) {       // float offset of the framed hand tile inside the arena
    return (NNF_TILE_FLOATS(GL_TILE) + NNF_TILE_FLOATS(GL_MAX_HAND) - NNF_TILE_FLOATS((H + 31) & ~31)) & ~3;
}
__host__ __device__ constexpr bool glf_hand_fits(int H) {     // A-operand image of the hand tiles + normals in front of the hand tile
    return (((H + 31) & ~31) >> 5) * 256 + ((H * 3 + 3) & ~3) <= glf_hand_off(H);
}
static_assert(glf_hand_fits(778) && glf_hand_fits(800), "SMPL-X's right hand must take the fast kernel");
template <bool PRE>
__global__ __launch_bounds__(GL_THREADS, 3) void grasp_loss_fwd_fast_kernel(
    saga_grasp_tables_t tb, const float* __restrict__ verts, const int32_t* __restrict__ vf_ptr,
    const int32_t* __restrict__ vf_tri, const float* __restrict__ obj,
    const float* __restrict__ obj_n, const float* __restrict__ mk_target, const float* __restrict__ c_mk,
    const float* __restrict__ c_obj, float* __restrict__ terms, float* __restrict__ markers_rec,
    float* __restrict__ hand_rec, float* __restrict__ hand_normals, float* __restrict__ h2o_marker,
    float* __restrict__ h2o_signed, float* __restrict__ o2h_signed, int32_t* __restrict__ idx_m2o,
    int32_t* __restrict__ idx_h2o, int32_t* __restrict__ idx_o2h, int V, int O, int touch_verts, const float* __restrict__ uniq_rows) {
#pragma clang fp contract(off)
    // One LDS arena (52 KB: three blocks per CU), used in two phases.
    //   hand/markers -> object:  [ framed object tile (f32) | A operands of 16 object tiles at a time ]
    //   object -> hand:          [ A operands of every hand tile | hand normals | ... | framed hand tile (f32), at the arena's end ]
    constexpr int SM_OBJ = NNF_TILE_FLOATS(GL_TILE), SM_TOTAL = SM_OBJ + NNF_TILE_FLOATS(GL_MAX_HAND);
    __shared__ __attribute__((aligned(16))) float smem[SM_TOTAL];
    __shared__ float red[GL_THREADS / 64][GL_NSUM];
    __shared__ float sbox[18 * (GL_THREADS / 64)];
    float* const sObj = smem;
    float* const sImgO = smem + SM_OBJ;
    // candidates in reverse launch order: the producer (skinning) wrote them in ascending order, so the most recently
    // written vertices -- the ones still in the Infinity Cache -- are read first
    const int b = gridDim.x - 1 - blockIdx.x, tid = threadIdx.x;
    const int H = tb.n_hand, M = tb.n_markers, NQ = H + M;
    const int Hpad = (H + 31) & ~31;           // both clouds are padded with far points to whole 32-point matrix tiles
    // phase 2 layout: the hand tile sits at the arena's end; its A-operand image (1 KB per 32 vertices) and the normals in front of
    // it (they fit for H <= GLF_MAX_HAND, checked by the host: larger hand tables take grasp_loss_fwd_kernel)
    const int h_tiles = Hpad >> 5;
    const int hand_off = glf_hand_off(H);
    float* const sHand = smem + hand_off;
    float* const sAimg = smem;
    float* const sNrm = smem + h_tiles * 256;
    const float* vb = verts + (size_t)b * V * 3;
    const float* ob = obj + (size_t)b * O * 3;
    const float* onb = obj_n + (size_t)b * O * 3;
    GL_T0();

    // the lane's object points (O <= GL_TILE = GL_OPL * GL_THREADS): requested first, together with the query gathers below, and
    // kept in registers for the bounding box AND the tile -- one trip to memory instead of two
    float3 op[GL_OPL];
#pragma unroll
    for (int i = 0; i < GL_OPL; ++i) {
        typedef float f32x3_t __attribute__((ext_vector_type(3)));
        const f32x3_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x3_t*>(ob + (size_t)min(tid + i * GL_THREADS, O - 1) * 3));
        op[i] = make_float3(t.x, t.y, t.z);
    }
    // ---- x -> y queries: lane owns hand vertices / markers q = tid + i*GL_THREADS, i < GL_NQ ------------------
    float qx[GL_NQ], qy[GL_NQ], qz[GL_NQ];
    float hlo[3] = {INFINITY, INFINITY, INFINITY}, hhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float mlo[3] = {INFINITY, INFINITY, INFINITY}, mhi[3] = {-INFINITY, -INFINITY, -INFINITY};      // the markers' box
#pragma unroll
    for (int i = 0; i < GL_NQ; ++i) {
        const int q = tid + i * GL_THREADS;
        const int qc = min(q, NQ - 1);
        float* rec = qc < H ? hand_rec + ((size_t)b * H + qc) * 3 : markers_rec + ((size_t)b * M + (qc - H)) * 3;
        if (PRE) {
            const float* u = uniq_rows + ((size_t)b * tb.n_unique + (qc < H ? tb.hand_slot[qc] : tb.marker_slot[qc - H])) * 3;
            qx[i] = u[0]; qy[i] = u[1]; qz[i] = u[2];
        } else {
            const int v = qc < H ? tb.hand_ids[qc] : tb.marker_ids[qc - H];
            qx[i] = vb[v * 3]; qy[i] = vb[v * 3 + 1]; qz[i] = vb[v * 3 + 2];
        }
        if (q < NQ) {
            rec[0] = qx[i]; rec[1] = qy[i]; rec[2] = qz[i];
            if (q < H) {
                hlo[0] = fminf(hlo[0], qx[i]); hhi[0] = fmaxf(hhi[0], qx[i]);
                hlo[1] = fminf(hlo[1], qy[i]); hhi[1] = fmaxf(hhi[1], qy[i]);
                hlo[2] = fminf(hlo[2], qz[i]); hhi[2] = fmaxf(hhi[2], qz[i]);
            } else {
                mlo[0] = fminf(mlo[0], qx[i]); mhi[0] = fmaxf(mhi[0], qx[i]);
                mlo[1] = fminf(mlo[1], qy[i]); mhi[1] = fmaxf(mhi[1], qy[i]);
                mlo[2] = fminf(mlo[2], qz[i]); mhi[2] = fmaxf(mhi[2], qz[i]);
            }
        }
    }
    GL_T(0);
    // frames of the two clouds (block-wide bounding boxes) and the two tiles
    float olo[3] = {INFINITY, INFINITY, INFINITY}, ohi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int i = 0; i < GL_OPL; ++i) {      // (a clamped duplicate of point O - 1 does not change the box)
        olo[0] = fminf(olo[0], op[i].x); ohi[0] = fmaxf(ohi[0], op[i].x);
        olo[1] = fminf(olo[1], op[i].y); ohi[1] = fmaxf(ohi[1], op[i].y);
        olo[2] = fminf(olo[2], op[i].z); ohi[2] = fmaxf(ohi[2], op[i].z);
    }
    // the object's frame, the hand's frame, and the box of ALL queries (hand + markers): one scale of the matrix-core pass for the
    // whole block (nnf_scales' rq: any upper bound of the framed queries' largest |coordinate|), so its A operands can be shared.
    // The three boxes are reduced over the block together (one pair of barriers).
    NnfFrame fr[3];
    {
        float blo[3][3], bhi[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            blo[0][a] = olo[a]; bhi[0][a] = ohi[a];
            blo[1][a] = hlo[a]; bhi[1][a] = hhi[a];
            blo[2][a] = fminf(hlo[a], mlo[a]); bhi[2][a] = fmaxf(hhi[a], mhi[a]);
        }
        nnf_frames_from_boxes<3>(blo, bhi, sbox, tid, GL_THREADS, fr);
    }
    const NnfFrame fo = fr[0], fh = fr[1], fq = fr[2];      // fq.rp >= half diagonal of the queries' box about (fq.cx, fq.cy, fq.cz)
    const int Opad = (O + 31) & ~31;
#pragma unroll
    for (int i = 0; i < GL_OPL; ++i) {
        const int k = tid + i * GL_THREADS;
        if (k < O) nnf_store_point(sObj, k, op[i].x, op[i].y, op[i].z, fo);
        else if (k < Opad) nnf_store_point(sObj, k, NN_FAR, NN_FAR, NN_FAR, fo);
    }
    __syncthreads();
    GL_T(1);

    // Register budget (168 per lane at three blocks per CU): the matrix-core pass holds 2 x 16 accumulators, the queries' B operands
    // (32) and the sorted triples (40); what it does not read is kept out of its live range -- the framed query coordinates are
    // formed again after it (same subtraction, same bits), the per-lane sums start after it, the sums of the first direction are
    // folded to their wave partials before the second direction starts (the order of the additions is unchanged), frames and scales
    // are scalar (nnf_uniform), and the wave reductions use immediate lane patterns (gl_wave_sum_lane0).
    float sums[GL_NSUM];

    // ---- x -> y : every hand vertex / marker finds its nearest object point -------------------------------
    {
        float cx[GL_NQ], cy[GL_NQ], cz[GL_NQ], m1[GL_NQ], m2[GL_NQ], m3[GL_NQ], best[GL_NQ];
        int c1[GL_NQ], c2[GL_NQ], bi[GL_NQ];
#pragma unroll
        for (int i = 0; i < GL_NQ; ++i) { cx[i] = qx[i] - fo.cx; cy[i] = qy[i] - fo.cy; cz[i] = qz[i] - fo.cz; }
        float ef_o;         // error floor of the object scan's scales (nn_fast.h)
        // approximate pass on the matrix cores (f16-split operands, nn_fast.h), exact pass on the two candidate chunks.  The four
        // waves scan the same 64 object tiles: their A operands are built ONCE per block, 16 tiles (16 KB) at a time, into the LDS
        // the hand tile will take afterwards (a wave rebuilding all of them itself spent a quarter of this scan's instructions there)
        {
            const float rq = fq.rp + fmaxf(fabsf(fq.cx - fo.cx), fmaxf(fabsf(fq.cy - fo.cy), fabsf(fq.cz - fo.cz)));
            const NnfScales ko = nnf_scales(fo.rp, rq);
            ef_o = nnf_mfma_error_floor(fo, ko);
            // query tile (i, hh) of this wave starts at query wave*64 + hh*32 + i*256: all-padding tiles are skipped
            const int wave = nnf_again(tid) >> 6, lane = nnf_again(tid) & 63;
            unsigned act = 0;
#pragma unroll
            for (int i = 0; i < GL_NQ; ++i)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
                    if (wave * 64 + hh * 32 + i * GL_THREADS < NQ) act |= 1u << (2 * i + hh);
            NnfMfmaState<GL_NQ> st;
            nnf_mfma_begin<GL_NQ>(st, ko, cx, cy, cz);
            constexpr int IMG_TILES = 16;
            static_assert(IMG_TILES * 256 <= NNF_TILE_FLOATS(GL_MAX_HAND), "the A-operand image lives behind the object tile");
            const int o_tiles = Opad >> 5;
#if defined(SAGA_EXPERIMENT_BUILD) && defined(GL_EXP_XY_PRIVATE)        // timing experiment (scripts/exp): every wave builds its own A operands, no rounds, no barriers
            nnf_mfma_tiles<GL_NQ, false, true>(st, sObj, Opad / NNF_CH, 0, o_tiles, 0, ko, act);
#else
            for (int r = 0; r < o_tiles; r += IMG_TILES) {
                const int r1 = min(r + IMG_TILES, o_tiles);
                if (r) __syncthreads();
                for (int t = r + wave; t < r1; t += GL_THREADS / 64)
                    reinterpret_cast<nn_h8*>(sImgO)[(t - r) * 64 + lane] = nnf_build_A(sObj, Opad, t, lane, ko);
                __syncthreads();
                nnf_mfma_tiles<GL_NQ, true, true, GL_PAIR_XY>(st, sImgO, Opad / NNF_CH, r, r1, r, ko, act);
            }
#endif
            nnf_mfma_end<GL_NQ>(st, ko, m1, m2, m3, c1, c2);
        }
        GL_T(2);
#pragma unroll
        for (int i = 0; i < GL_NQ; ++i) {
            asm volatile("" : "+v"(qx[i]), "+v"(qy[i]), "+v"(qz[i]));     // not the same value to the compiler: cx.. do not stay live above
            cx[i] = qx[i] - fo.cx; cy[i] = qy[i] - fo.cy; cz[i] = qz[i] - fo.cz;
            best[i] = INFINITY; bi[i] = 0;
        }
#pragma unroll
        for (int i = 0; i < GL_NSUM; ++i) sums[i] = 0.f;
        nnf_resolve<GL_NQ, true, false, GL_PAIR_XY>(sObj, Opad / NNF_CH, fo, 0, cx, cy, cz, qx, qy, qz, m1, m2, m3, c1, c2, best, bi, nullptr, ef_o);
        GL_T(3);
#pragma unroll
        for (int i = 0; i < GL_NQ; ++i) {
            const int q = tid + i * GL_THREADS;
            if (q >= NQ) continue;
            const float x = qx[i], y = qy[i], z = qz[i];
            const int k = bi[i];
            float px, py, pz;
            nnf_tile_point(sObj, fo, k, px, py, pz);           // bit-exact object point from the LDS tile
            const float dx = x - px, dy = y - py, dz = z - pz;
            const float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
            if (q < H) {
                const float dot = (onb[k * 3] * dx + onb[k * 3 + 1] * dy) + onb[k * 3 + 2] * dz;
                const float d = dist * sgn(dot);
                h2o_signed[(size_t)b * H + q] = d;
                idx_h2o[(size_t)b * H + q] = k;
                const float pw = gl_prior_weight(tb, q);                                        // vw where vw^(1/2.5) > 0.7, else 0
                const bool near = (d < 0.02f) && (d > 0.f);                                      // fittingop.py:195
                sums[5] += fabsf(d) * (near ? pw : 0.f);                                          // :199,210
                sums[6] += fabsf(d) * (d < 0.f ? 10.f : 0.f);                                     // :200,211
                sums[8] += d < 0.f ? 1.f : 0.f;
                sums[10] += (d < 0.001f && d > -0.001f) ? 1.f : 0.f;
            } else {
                const int p = q - H;
                h2o_marker[(size_t)b * M + p] = dist;                                             // :183 quirk: unsigned
                idx_m2o[(size_t)b * M + p] = k;
                sums[3] += dist * c_mk[(size_t)b * M + p];                                        // :190
                const float* tg = mk_target + ((size_t)b * M + p) * 3;
                const float e = (fabsf(x - tg[0]) + fabsf(y - tg[1])) + fabsf(z - tg[2]);
                sums[0] += e;                                                                     // :168
                if (p < tb.n_body_markers) sums[1] += e;                                          // :169
                sums[11] += dist < 0.f ? 1.f : 0.f;
            }
        }
    }
    // foot term: 0.1 * mean |z| (:173)
    for (int i = tid; i < tb.n_foot; i += GL_THREADS)
        sums[2] += fabsf(PRE ? uniq_rows[((size_t)b * tb.n_unique + tb.foot_slot[i]) * 3 + 2] : vb[tb.foot_ids[i] * 3 + 2]);

    // sums 0-3, 5, 6, 8, 10, 11 are complete: fold them to wave partials now (4, 7, 9 belong to the second direction)
#pragma unroll
    for (int i = 0; i < GL_NSUM; ++i) {
        if (i == 4 || i == 7 || i == 9) continue;
        const float v = gl_wave_sum_lane0(sums[i]);
        if ((nnf_again(tid) & 63) == 0) red[__builtin_amdgcn_readfirstlane(nnf_again(tid) >> 6)][i] = v;
    }
    sums[4] = 0.f; sums[7] = 0.f; sums[9] = 0.f;
    GL_T(4);
    // ---- hand normals (area-weighted, ascending face order) into the LDS space the object tile no longer needs ----
    // The normals gather three corners per incident face.  When the mesh's faces join vertices that are far apart in memory
    // (touch_verts: set by the host from the model's faces), those are ~14k scattered 12-B reads per candidate that miss L2
    // (4x the bytes the candidate has).  Touching every 128-B line of the candidate's vertex block once, coalesced, brings
    // the block (126 KB) into L2 first.
#if !(defined(SAGA_EXPERIMENT_BUILD) && defined(GL_EXP_NO_NORMALS))
    if (!PRE && touch_verts) {
        const int n_lines = (V * 3 + 31) / 32;
        for (int l = tid; l < n_lines; l += GL_THREADS) {
            const float x = vb[min(l * 32, V * 3 - 1)];
            asm volatile("" ::"v"(x));
        }
    }
#endif
    __syncthreads();                          // every wave is done with the object tile and its A-operand image: phase 2 of the arena
#pragma unroll
    for (int i = 0; i < GL_NQ; ++i) {
        const int q = nnf_again(tid) + i * GL_THREADS;
        if (q < H) nnf_store_point(sHand, q, qx[i], qy[i], qz[i], fh);
    }
    for (int q = H + nnf_again(tid); q < Hpad; q += GL_THREADS) nnf_store_point(sHand, q, NN_FAR, NN_FAR, NN_FAR, fh);
    if (PRE) {
        const float* hn = hand_normals + (size_t)b * H * 3;
        for (int i = tid; i < H * 3; i += GL_THREADS) sNrm[i] = hn[i];
    } else {
        // all of a lane's hand vertices (GL_MAX_HAND / GL_THREADS = 4) at once: their read chains advance together
        constexpr int NVL = GL_MAX_HAND / GL_THREADS;
        int hv[NVL];
        float nr[NVL][3];
#pragma unroll
        for (int r = 0; r < NVL; ++r) hv[r] = (tid + r * GL_THREADS < H) ? tb.hand_ids[tid + r * GL_THREADS] : -1;
#if defined(SAGA_EXPERIMENT_BUILD) && defined(GL_EXP_NO_NORMALS)            // timing experiment only (scripts/exp): what the kernel costs without the normals' gathers
#pragma unroll
        for (int r = 0; r < NVL; ++r) { nr[r][0] = 0.f; nr[r][1] = 0.f; nr[r][2] = 1.f; }
#else
        saga_vertex_normals_multi<NVL, GL_NRM_FPS>(vb, vf_ptr, vf_tri, hv, nr);
#endif
#pragma unroll
        for (int r = 0; r < NVL; ++r) {
            const int i = tid + r * GL_THREADS;
            if (i >= H) continue;
            sNrm[i * 3] = nr[r][0]; sNrm[i * 3 + 1] = nr[r][1]; sNrm[i * 3 + 2] = nr[r][2];
            float* dst = hand_normals + ((size_t)b * H + i) * 3;
            dst[0] = nr[r][0]; dst[1] = nr[r][1]; dst[2] = nr[r][2];
        }
    }
    // The hand tile is scanned once per pass below by every wave: its A operands (matrix-core pass, nn_fast.h) are built once
    // per block into the rest of the LDS space the object tile no longer needs (1 KB per 32 hand vertices, after the normals),
    // with one scale for the whole block: every object point is within fo.rp of the object frame's centre.
    const NnfScales kh = nnf_scales(fh.rp, fo.rp * 1.0001f + fmaxf(fabsf(fo.cx - fh.cx), fmaxf(fabsf(fo.cy - fh.cy), fabsf(fo.cz - fh.cz))));
    __syncthreads();                          // the hand tile is complete
    {
        const int t2 = nnf_again(tid);
        for (int t = t2 >> 6; t < h_tiles; t += GL_THREADS / 64)
            reinterpret_cast<nn_h8*>(sAimg)[t * 64 + (t2 & 63)] = nnf_build_A(sHand, Hpad, t, t2 & 63, kh);
    }
    __syncthreads();

    GL_T(5);
    // ---- y -> x : every object point finds its nearest HAND vertex (the marker direction is never used) ------
    for (int k0 = 0; k0 < O; k0 += GL_THREADS * GLF_NO) {
        float ox[GLF_NO], oy[GLF_NO], oz[GLF_NO], cx[GLF_NO], cy[GLF_NO], cz[GLF_NO], m1[GLF_NO], m2[GLF_NO], m3[GLF_NO], best[GLF_NO];
        int c1[GLF_NO], c2[GLF_NO], bj[GLF_NO];
#pragma unroll
        for (int i = 0; i < GLF_NO; ++i) {
            const int k = min(k0 + tid + i * GL_THREADS, O - 1);
            ox[i] = ob[k * 3]; oy[i] = ob[k * 3 + 1]; oz[i] = ob[k * 3 + 2];
            cx[i] = ox[i] - fh.cx; cy[i] = oy[i] - fh.cy; cz[i] = oz[i] - fh.cz;
        }
        {
            unsigned act = 0;
#pragma unroll
            for (int i = 0; i < GLF_NO; ++i)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
                    if (k0 + (nnf_again(tid) >> 6) * 64 + hh * 32 + i * GL_THREADS < O) act |= 1u << (2 * i + hh);
            NnfMfmaState<GLF_NO> st;
            nnf_mfma_begin<GLF_NO>(st, kh, cx, cy, cz);
            nnf_mfma_tiles<GLF_NO, true, true>(st, sAimg, Hpad / NNF_CH, 0, h_tiles, 0, kh, act);
            nnf_mfma_end<GLF_NO>(st, kh, m1, m2, m3, c1, c2);
        }
#pragma unroll
        for (int i = 0; i < GLF_NO; ++i) {
            asm volatile("" : "+v"(ox[i]), "+v"(oy[i]), "+v"(oz[i]));
            cx[i] = ox[i] - fh.cx; cy[i] = oy[i] - fh.cy; cz[i] = oz[i] - fh.cz;
            best[i] = INFINITY; bj[i] = 0;
        }
        nnf_resolve<GLF_NO, true>(sHand, Hpad / NNF_CH, fh, 0, cx, cy, cz, ox, oy, oz, m1, m2, m3, c1, c2, best, bj, nullptr,
                                  nnf_mfma_error_floor(fh, kh));
#pragma unroll
        for (int i = 0; i < GLF_NO; ++i) {
            const int k = k0 + nnf_again(tid) + i * GL_THREADS;
            if (k >= O) continue;
            const int j = bj[i];
            float hx, hy, hz;
            nnf_tile_point(sHand, fh, j, hx, hy, hz);
            const float dx = ox[i] - hx, dy = oy[i] - hy, dz = oz[i] - hz;
            const float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
            const float dot = (sNrm[j * 3] * dx + sNrm[j * 3 + 1] * dy) + sNrm[j * 3 + 2] * dz;   // HAND normal (:68-71)
            const float d = dist * sgn(dot);
            o2h_signed[(size_t)b * O + k] = d;
            idx_o2h[(size_t)b * O + k] = j;
            sums[4] += fabsf(d) * c_obj[(size_t)b * O + k];                                  // :191
            sums[7] += fabsf(d) * (d < 0.f ? 20.f : 0.f);                                    // :206,212
            sums[9] += d < 0.f ? 1.f : 0.f;
        }
    }

    GL_T(6);
    // ---- per-candidate means: wave sums, then the four wave partials in wave order ---------------------
#pragma unroll
    for (int i = 0; i < GL_NSUM; ++i) {
        if (!(i == 4 || i == 7 || i == 9)) continue;
        const float v = gl_wave_sum_lane0(sums[i]);
        if ((nnf_again(tid) & 63) == 0) red[__builtin_amdgcn_readfirstlane(nnf_again(tid) >> 6)][i] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float sm[GL_NSUM];
#pragma unroll
        for (int i = 0; i < GL_NSUM; ++i) sm[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
        float* t = terms + (size_t)b * NT;
        t[0] = sm[0] / (float)(M * 3);
        t[1] = sm[1] / (float)(tb.n_body_markers * 3);
        t[2] = 0.1f * (sm[2] / (float)tb.n_foot);
        t[3] = sm[3] / (float)M;
        t[4] = sm[4] / (float)O;
        t[5] = sm[5] / (float)H;
        t[6] = sm[6] / (float)H;
        t[7] = sm[7] / (float)O;
        t[8] = sm[8]; t[9] = sm[9]; t[10] = sm[10]; t[11] = sm[11];
        t[12] = 0.f; t[13] = 0.f; t[14] = 0.f; t[15] = 0.f;
    }
}

// grad_vertices [B,V,3] (zero-filled here, then the touched vertices) from the upstream gradients of the eight differentiable terms
__global__ __launch_bounds__(GL_BWD_THREADS) void grasp_loss_bwd_kernel(
    saga_grasp_tables_t tb, const float* __restrict__ g_terms, const float* __restrict__ verts,
    const float* __restrict__ obj, const float* __restrict__ mk_target, const float* __restrict__ c_mk,
    const float* __restrict__ c_obj, const float* __restrict__ markers_rec, const float* __restrict__ hand_rec,
    const float* __restrict__ h2o_signed, const float* __restrict__ o2h_signed, const int32_t* __restrict__ idx_m2o,
    const int32_t* __restrict__ idx_h2o, const int32_t* __restrict__ idx_o2h, float* __restrict__ g_verts, int V, int O,
    int32_t* __restrict__ zero_tag, int32_t tag, int compact) {
    // compact: g_verts is [B, n_unique, 3] -- row i of a candidate is the gradient of vertex unique_ids[i] (what the
    // structurally sparse backward of saga_fit_step_sparse consumes); nothing to zero-fill, every entry is written.
    // Accumulators are 64-bit fixed point (2^-40 units, range +-2^23): integer adds are associative, so the scatter-add of
    // many object points into one hand vertex gives the same bits whatever order the LDS atomics land in, and the sum is
    // the exactly rounded one (float atomics would make the whole iteration run-to-run irreproducible).
    __shared__ unsigned long long acc[GL_MAX_UNIQUE * 3];
    auto add_fx = [](unsigned long long* p, float v) {
        atomicAdd(p, (unsigned long long)__float2ll_rn(v * 1099511627776.0f));
    };
    const int b = blockIdx.x, tid = threadIdx.x;
    const int H = tb.n_hand, M = tb.n_markers, U = tb.n_unique;
    const float* gt = g_terms + (size_t)b * NT;
    const float g_rec = gt[0], g_brec = gt[1], g_foot = gt[2], g_mc = gt[3], g_oc = gt[4], g_prior = gt[5], g_hcol = gt[6],
                g_ocol = gt[7];
    const float* vb = verts + (size_t)b * V * 3;
    const float* ob = obj + (size_t)b * O * 3;
    for (int i = tid; i < U * 3; i += GL_BWD_THREADS) acc[i] = 0ull;
    // Zero-fill this candidate's row of the dense gradient (no separate memset pass): scalar head up to 16-B alignment,
    // float4 body, scalar tail; the scatter below comes after a barrier.  With a tag array (saga_fit_step's workspace,
    // which nobody else writes) the fill happens only the first time: every later call rewrites exactly the same
    // vertices (the tables' unique ids) and everything else is still zero.
    if (!compact && (!zero_tag || zero_tag[b] != tag)) {
        float* gb0 = g_verts + (size_t)b * V * 3;
        const int n = V * 3;
        const int head = min(n, (int)((16 - ((uintptr_t)gb0 & 15)) & 15) >> 2);
        const int body = (n - head) >> 2;
        if (tid < head) gb0[tid] = 0.f;
        float4* g4 = reinterpret_cast<float4*>(gb0 + head);
        for (int i = tid; i < body; i += GL_BWD_THREADS) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = head + body * 4 + tid; i < n; i += GL_BWD_THREADS) gb0[i] = 0.f;
    }
    __syncthreads();
    // markers: L1 reconstruction + contact-weighted distance to the nearest object point
    for (int p = tid; p < M; p += GL_BWD_THREADS) {
        const float* m = markers_rec + ((size_t)b * M + p) * 3;
        const float* tg = mk_target + ((size_t)b * M + p) * 3;
        const int k = idx_m2o[(size_t)b * M + p];
        const float dx = m[0] - ob[k * 3], dy = m[1] - ob[k * 3 + 1], dz = m[2] - ob[k * 3 + 2];
        const float n = sqrtf(dx * dx + dy * dy + dz * dz);
        const float w = n > 0.f ? g_mc * c_mk[(size_t)b * M + p] / ((float)M * n) : 0.f;
        const float wr = g_rec / (float)(M * 3) + (p < tb.n_body_markers ? g_brec / (float)(tb.n_body_markers * 3) : 0.f);
        unsigned long long* a = acc + tb.marker_slot[p] * 3;
        add_fx(a + 0, wr * sgn(m[0] - tg[0]) + w * dx);
        add_fx(a + 1, wr * sgn(m[1] - tg[1]) + w * dy);
        add_fx(a + 2, wr * sgn(m[2] - tg[2]) + w * dz);
    }
    // foot: 0.1 * mean |z|
    for (int f = tid; f < tb.n_foot; f += GL_BWD_THREADS) {
        const float z = vb[tb.foot_ids[f] * 3 + 2];
        add_fx(acc + tb.foot_slot[f] * 3 + 2, g_foot * 0.1f * sgn(z) / (float)tb.n_foot);
    }
    // hand -> object: |d_h| weighted by the prior-contact / collision masks (masks carry no gradient)
    for (int p = tid; p < H; p += GL_BWD_THREADS) {
        const float d = h2o_signed[(size_t)b * H + p];
        const float pw = gl_prior_weight(tb, p);
        const bool near = (d < 0.02f) && (d > 0.f);
        const float coef = (g_prior * (near ? pw : 0.f) + g_hcol * (d < 0.f ? 10.f : 0.f)) / (float)H;
        // d|d|/dh = sign(d)^2 * unit(h - y): zero where the sign is zero
        if (coef != 0.f && d != 0.f) {
            const float* h = hand_rec + ((size_t)b * H + p) * 3;
            const int k = idx_h2o[(size_t)b * H + p];
            const float dx = h[0] - ob[k * 3], dy = h[1] - ob[k * 3 + 1], dz = h[2] - ob[k * 3 + 2];
            const float w = coef / sqrtf(dx * dx + dy * dy + dz * dz);
            unsigned long long* a = acc + tb.hand_slot[p] * 3;
            add_fx(a + 0, w * dx); add_fx(a + 1, w * dy); add_fx(a + 2, w * dz);
        }
    }
    // object -> hand: gradient lands on the nearest hand vertex (scatter-add; many object points share one vertex)
    for (int k = tid; k < O; k += GL_BWD_THREADS) {
        const float d = o2h_signed[(size_t)b * O + k];
        const float coef = (g_oc * c_obj[(size_t)b * O + k] + g_ocol * (d < 0.f ? 20.f : 0.f)) / (float)O;
        if (coef != 0.f && d != 0.f) {
            const int j = idx_o2h[(size_t)b * O + k];
            const float* h = hand_rec + ((size_t)b * H + j) * 3;
            const float dx = ob[k * 3] - h[0], dy = ob[k * 3 + 1] - h[1], dz = ob[k * 3 + 2] - h[2];
            const float w = -coef / sqrtf(dx * dx + dy * dy + dz * dz);
            unsigned long long* a = acc + tb.hand_slot[j] * 3;
            add_fx(a + 0, w * dx); add_fx(a + 1, w * dy); add_fx(a + 2, w * dz);
        }
    }
    __syncthreads();
    if (compact) {
        float* gc = g_verts + (size_t)b * U * 3;
        for (int i = tid; i < U * 3; i += GL_BWD_THREADS) gc[i] = (float)((double)(long long)acc[i] * 9.094947017729282e-13);
        return;
    }
    float* gb = g_verts + (size_t)b * V * 3;
    for (int i = tid; i < U * 3; i += GL_BWD_THREADS) gb[tb.unique_ids[i / 3] * 3 + (i % 3)] = (float)((double)(long long)acc[i] * 9.094947017729282e-13);
    if (zero_tag && tid == 0) zero_tag[b] = tag;
}

// The terms a stage with every contact / collision weight gated off needs (fittingop.py:215-223 with stage < 2 or only_rec):
// marker reconstruction (:168-169) and the foot term (:173); the gated terms and the four counts are written as 0.  One block per
// candidate; lane partials in query order, wave partials in wave order (fixed summation order).
__global__ __launch_bounds__(GL_THREADS) void grasp_loss_fwd_rec_kernel(saga_grasp_tables_t tb, const float* __restrict__ verts,
                                                                        const float* __restrict__ mk_target, float* __restrict__ terms,
                                                                        float* __restrict__ markers_rec, int32_t* __restrict__ idx_m2o,
                                                                        int V) {
#pragma clang fp contract(off)
    __shared__ float red[GL_THREADS / 64][3];
    const int b = blockIdx.x, tid = threadIdx.x, M = tb.n_markers;
    const float* vb = verts + (size_t)b * V * 3;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int p = tid; p < M; p += GL_THREADS) {
        const int v = tb.marker_ids[p];
        const float x = vb[v * 3], y = vb[v * 3 + 1], z = vb[v * 3 + 2];
        float* dst = markers_rec + ((size_t)b * M + p) * 3;
        dst[0] = x; dst[1] = y; dst[2] = z;
        idx_m2o[(size_t)b * M + p] = 0;       // the backward forms (marker - obj[idx]) before it multiplies by the zero weight
        const float* tg = mk_target + ((size_t)b * M + p) * 3;
        const float e = (fabsf(x - tg[0]) + fabsf(y - tg[1])) + fabsf(z - tg[2]);
        s0 += e;
        if (p < tb.n_body_markers) s1 += e;
    }
    for (int i = tid; i < tb.n_foot; i += GL_THREADS) s2 += fabsf(vb[tb.foot_ids[i] * 3 + 2]);
    const int wave = tid >> 6, lane = tid & 63;
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_down(s0, o, 64); s1 += __shfl_down(s1, o, 64); s2 += __shfl_down(s2, o, 64);
    }
    if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; }
    __syncthreads();
    if (tid == 0) {
        float sm[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) sm[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
        float* t = terms + (size_t)b * NT;
        t[0] = sm[0] / (float)(M * 3);
        t[1] = sm[1] / (float)(tb.n_body_markers * 3);
        t[2] = 0.1f * (sm[2] / (float)tb.n_foot);
#pragma unroll
        for (int i = 3; i < NT; ++i) t[i] = 0.f;
    }
}

static int check_tables(const saga_grasp_tables_t* tb) {
    SAGA_CHECK_ARG(tb, "grasp loss: null tables");
    SAGA_CHECK_ARG(tb->n_hand > 0 && tb->n_hand <= GL_MAX_HAND && tb->n_markers > 0 && tb->n_markers <= GL_MAX_MARK &&
                       tb->n_hand + tb->n_markers <= GL_NQ * GL_THREADS,
                   "grasp loss: n_hand=%d (max %d), n_markers=%d (max %d)", tb->n_hand, GL_MAX_HAND, tb->n_markers, GL_MAX_MARK);
    SAGA_CHECK_ARG(tb->n_foot > 0 && tb->n_unique > 0 && tb->n_unique <= GL_MAX_UNIQUE && tb->n_body_markers > 0 &&
                       tb->n_body_markers <= tb->n_markers,
                   "grasp loss: bad table sizes (foot %d, unique %d, body markers %d)", tb->n_foot, tb->n_unique, tb->n_body_markers);
    SAGA_CHECK_ARG(tb->marker_ids && tb->hand_ids && tb->foot_ids && tb->hand_weights && tb->unique_ids && tb->marker_slot &&
                       tb->hand_slot && tb->foot_slot,
                   "grasp loss: null table pointer");
    return 0;
}

extern "C" int saga_grasp_loss_fwd(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* verts, const float* obj,
                                   const float* obj_normals, const float* markers_target, const float* c_markers,
                                   const float* c_object, float* terms, float* markers_rec, float* hand_rec,
                                   float* hand_normals, float* h2o_marker, float* h2o_signed, float* o2h_signed,
                                   int32_t* idx_m2o, int32_t* idx_h2o, int32_t* idx_o2h, int B, int O, void* stream) {
    SAGA_CHECK_ARG(m && m->F > 0, "saga_grasp_loss_fwd: model without faces");
    SAGA_CHECK_DEVICE(m, "saga_grasp_loss_fwd");
    int rc = check_tables(tb);
    if (rc) return rc;
    SAGA_CHECK_ARG(B >= 0 && O > 0, "saga_grasp_loss_fwd: bad sizes B=%d O=%d", B, O);
    if (B == 0) return 0;
    SAGA_CHECK_ARG(verts && obj && obj_normals && markers_target && c_markers && c_object && terms && markers_rec && hand_rec &&
                       hand_normals && h2o_marker && h2o_signed && o2h_signed && idx_m2o && idx_h2o && idx_o2h,
                   "saga_grasp_loss_fwd: null buffer");
    void* prof = saga_prof_begin(SAGA_PROF_GRASP_FWD, stream);
    if (O <= GL_TILE && glf_hand_fits(tb->n_hand))
        grasp_loss_fwd_fast_kernel<false><<<B, GL_THREADS, 0, (hipStream_t)stream>>>(*tb, verts, m->vf_ptr, m->vf_tri, obj,
                                                                              obj_normals, markers_target, c_markers, c_object, terms,
                                                                              markers_rec, hand_rec, hand_normals, h2o_marker,
                                                                              h2o_signed, o2h_signed, idx_m2o, idx_h2o, idx_o2h, m->V, O, m->scattered_faces, nullptr);
    else
        grasp_loss_fwd_kernel<<<B, GL_THREADS, 0, (hipStream_t)stream>>>(*tb, verts, m->vf_ptr, m->vf_tri, obj, obj_normals,
                                                                         markers_target, c_markers, c_object, terms, markers_rec,
                                                                         hand_rec, hand_normals, h2o_marker, h2o_signed, o2h_signed,
                                                                         idx_m2o, idx_h2o, idx_o2h, m->V, O);
    saga_prof_end(prof);
    SAGA_LAUNCH_CHECK();
    return 0;
}

// whether these tables and this object size take grasp_loss_fwd_fast_kernel (saga_fit_step asks before it runs the ring kernel)
int saga_grasp_loss_fast_ok(const saga_grasp_tables_t* tb, int O) { return tb && O > 0 && O <= GL_TILE && glf_hand_fits(tb->n_hand); }

// saga_fit_step after saga_launch_skin_fwd_cand: uniq_rows and hand_normals are inputs (see the kernel).
// O <= GL_TILE only (the caller falls back to saga_grasp_loss_fwd otherwise).
int saga_grasp_loss_fwd_pre(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* obj, const float* obj_normals,
                            const float* markers_target, const float* c_markers, const float* c_object, float* terms,
                            float* markers_rec, float* hand_rec, float* hand_normals, const float* uniq_rows, float* h2o_marker,
                            float* h2o_signed, float* o2h_signed, int32_t* idx_m2o, int32_t* idx_h2o, int32_t* idx_o2h, int B, int O,
                            void* stream) {
    SAGA_CHECK_ARG(m, "saga_grasp_loss_fwd_pre: null model");
    SAGA_CHECK_DEVICE(m, "saga_grasp_loss_fwd_pre");
    int rc = check_tables(tb);
    if (rc) return rc;
    SAGA_CHECK_ARG(B >= 0 && saga_grasp_loss_fast_ok(tb, O), "saga_grasp_loss_fwd_pre: bad sizes B=%d O=%d n_hand=%d", B, O, tb->n_hand);
    if (B == 0) return 0;
    SAGA_CHECK_ARG(obj && obj_normals && markers_target && c_markers && c_object && terms && markers_rec && hand_rec && hand_normals &&
                       uniq_rows && h2o_marker && h2o_signed && o2h_signed && idx_m2o && idx_h2o && idx_o2h,
                   "saga_grasp_loss_fwd_pre: null buffer");
    void* prof = saga_prof_begin(SAGA_PROF_GRASP_FWD, stream);
    grasp_loss_fwd_fast_kernel<true><<<B, GL_THREADS, 0, (hipStream_t)stream>>>(*tb, nullptr, m->vf_ptr, m->vf_tri, obj, obj_normals,
                                                                                markers_target, c_markers, c_object, terms, markers_rec,
                                                                                hand_rec, hand_normals, h2o_marker, h2o_signed, o2h_signed,
                                                                                idx_m2o, idx_h2o, idx_o2h, m->V, O, 0, uniq_rows);
    saga_prof_end(prof);
    SAGA_LAUNCH_CHECK();
    return 0;
}

// saga_fit_step with cfg->skip_gated_terms in a stage whose contact / collision weights are all zero
int saga_grasp_loss_fwd_rec(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* verts, const float* markers_target,
                            float* terms, float* markers_rec, int32_t* idx_m2o, int B, void* stream) {
    SAGA_CHECK_ARG(m && idx_m2o, "saga_grasp_loss_fwd_rec: null model");
    SAGA_CHECK_DEVICE(m, "saga_grasp_loss_fwd_rec");
    int rc = check_tables(tb);
    if (rc) return rc;
    SAGA_CHECK_ARG(B >= 0 && verts && markers_target && terms && markers_rec, "saga_grasp_loss_fwd_rec: bad arguments");
    if (B == 0) return 0;
    void* prof = saga_prof_begin(SAGA_PROF_GRASP_FWD, stream);
    grasp_loss_fwd_rec_kernel<<<B, GL_THREADS, 0, (hipStream_t)stream>>>(*tb, verts, markers_target, terms, markers_rec, idx_m2o, m->V);
    saga_prof_end(prof);
    SAGA_LAUNCH_CHECK();
    return 0;
}

// zero_tag (device int32 [B], may be NULL) / tag / compact: see the kernel; used by saga_fit_step[_sparse] only
int saga_grasp_loss_bwd_tagged(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* g_terms, const float* verts,
                               const float* obj, const float* markers_target, const float* c_markers, const float* c_object,
                               const float* markers_rec, const float* hand_rec, const float* h2o_signed,
                               const float* o2h_signed, const int32_t* idx_m2o, const int32_t* idx_h2o,
                               const int32_t* idx_o2h, float* g_verts, int B, int O, int32_t* zero_tag, int32_t tag, int compact, void* stream);

extern "C" int saga_grasp_loss_bwd(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* g_terms, const float* verts,
                                   const float* obj, const float* markers_target, const float* c_markers, const float* c_object,
                                   const float* markers_rec, const float* hand_rec, const float* h2o_signed,
                                   const float* o2h_signed, const int32_t* idx_m2o, const int32_t* idx_h2o,
                                   const int32_t* idx_o2h, float* g_verts, int B, int O, void* stream) {
    return saga_grasp_loss_bwd_tagged(m, tb, g_terms, verts, obj, markers_target, c_markers, c_object, markers_rec, hand_rec, h2o_signed,
                                      o2h_signed, idx_m2o, idx_h2o, idx_o2h, g_verts, B, O, nullptr, 0, 0, stream);
}

int saga_grasp_loss_bwd_tagged(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* g_terms, const float* verts,
                               const float* obj, const float* markers_target, const float* c_markers, const float* c_object,
                               const float* markers_rec, const float* hand_rec, const float* h2o_signed,
                               const float* o2h_signed, const int32_t* idx_m2o, const int32_t* idx_h2o,
                               const int32_t* idx_o2h, float* g_verts, int B, int O, int32_t* zero_tag, int32_t tag, int compact, void* stream) {
    SAGA_CHECK_ARG(m, "saga_grasp_loss_bwd: null model");
    SAGA_CHECK_DEVICE(m, "saga_grasp_loss_bwd");
    int rc = check_tables(tb);
    if (rc) return rc;
    SAGA_CHECK_ARG(B >= 0 && O > 0, "saga_grasp_loss_bwd: bad sizes B=%d O=%d", B, O);
    if (B == 0) return 0;
    SAGA_CHECK_ARG(g_terms && verts && obj && markers_target && c_markers && c_object && markers_rec && hand_rec && h2o_signed &&
                       o2h_signed && idx_m2o && idx_h2o && idx_o2h && g_verts,
                   "saga_grasp_loss_bwd: null buffer");
    hipStream_t st = (hipStream_t)stream;
    void* prof = saga_prof_begin(SAGA_PROF_GRASP_BWD, stream);
    hipError_t e = hipSuccess;
    grasp_loss_bwd_kernel<<<B, GL_BWD_THREADS, 0, st>>>(*tb, g_terms, verts, obj, markers_target, c_markers, c_object, markers_rec,
                                                        hand_rec, h2o_signed, o2h_signed, idx_m2o, idx_h2o, idx_o2h, g_verts, m->V, O,
                                                        zero_tag, tag, compact);
    saga_prof_end(prof);
    SAGA_HIP(e);
    SAGA_LAUNCH_CHECK();
    return 0;
}
