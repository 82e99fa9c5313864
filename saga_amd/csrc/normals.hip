// Area-weighted vertex normals.  Replaces Open3D TriangleMesh.compute_vertex_normals() as called every
// iteration at WholeGraspPose/models/fittingop.py:158-164 (a device->host->device round trip in the reference).
// Gather formulation: each output vertex walks its incident faces in ascending face order (the order Open3D's
// face loop adds them), sums the un-normalised cross products and normalises; zero rows stay zero, NaN -> (0,0,1).
// No atomics, deterministic.  HBM/L2-bound gather.
#include <math.h>
#include <stdlib.h>

#include "saga_common.h"
#include "vertex_normal.h"

void* saga_prof_begin(int id, void* stream);
void saga_prof_end(void* scope);

__global__ void vertex_normals_kernel(const float* __restrict__ verts, const int32_t* __restrict__ vf_ptr,
                                      const int32_t* __restrict__ vf_tri, const int32_t* __restrict__ sel, int n_out,
                                      float* __restrict__ normals, int V, int T, int t0, int n_a, float* __restrict__ normals_b) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (o >= n_out) return;
    const int v = sel ? sel[o] : o;
    float nx, ny, nz;
    const int row = (b / (T - t0)) * T + t0 + b % (T - t0);       // output b = frame t0 + b % (T - t0) of sequence b / (T - t0)
    saga_vertex_normal(verts + (size_t)row * V * 3, vf_ptr, vf_tri, v, nx, ny, nz);
    // outputs 0..n_a-1 -> normals [B,n_a,3], the rest -> normals_b [B,n_out-n_a,3] (n_a = n_out: one array)
    float* out = o < n_a ? normals + ((size_t)b * n_a + o) * 3 : normals_b + ((size_t)b * (n_out - n_a) + (o - n_a)) * 3;
    out[0] = nx; out[1] = ny; out[2] = nz;
}

// The same normals with the whole mesh of one batch element staged in LDS first (V * 12 B: 126 KB for the full SMPL-X mesh): one
// coalesced pass over the row instead of ~18 scattered 12-B corner reads per output vertex, each a dependent global round trip.
// For batches large enough to fill the chip with one block per element (the frames of GraspMotion-Opt: 7,936 per iteration).
// Same sums, face by face in ascending order: same bits.
#define VNL_THREADS 512
__global__ __launch_bounds__(VNL_THREADS) void vertex_normals_lds_kernel(const float* __restrict__ verts, const int32_t* __restrict__ vf_ptr,
                                                                         const int32_t* __restrict__ vf_tri, const int32_t* __restrict__ sel,
                                                                         int n_out, float* __restrict__ normals, int V, int T, int t0,
                                                                         int n_a, float* __restrict__ normals_b) {
    extern __shared__ __attribute__((aligned(16))) float vnl_smem[];     // [3 + V*3]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int row = (b / (T - t0)) * T + t0 + b % (T - t0);
    const float* vb = verts + (size_t)row * V * 3;
    // the first output vertex's incidence range is requested before the row (model constants: L2)
    const int v_first = tid < n_out ? (sel ? sel[tid] : tid) : 0;
    const int e0_first = vf_ptr[v_first], e1_first = vf_ptr[v_first + 1];
    // row -> LDS with 16-B loads: LDS element i sits at vnl_smem[mis + i], so aligned global quads are aligned LDS quads
    const int n = V * 3;
    const int mis = (int)((reinterpret_cast<uintptr_t>(vb) >> 2) & 3);
    float* sv = vnl_smem + mis;
    const int head = min((4 - mis) & 3, n);
    if (tid < head) sv[tid] = vb[tid];
    const int nq = (n - head) >> 2;
    const float4* src = reinterpret_cast<const float4*>(vb + head);
    float4* dst = reinterpret_cast<float4*>(sv + head);
    for (int i = tid; i < nq; i += VNL_THREADS) dst[i] = src[i];
    for (int i = head + nq * 4 + tid; i < n; i += VNL_THREADS) sv[i] = vb[i];
    __syncthreads();
    for (int o = tid; o < n_out; o += VNL_THREADS) {
        int e0 = e0_first, e1 = e1_first;
        if (o != tid) {
            const int v = sel ? sel[o] : o;
            e0 = vf_ptr[v]; e1 = vf_ptr[v + 1];
        }
        float nn[3] = {0.f, 0.f, 0.f};
        for (int e = e0; e < e1; e += 8) {
            int4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = reinterpret_cast<const int4*>(vf_tri)[min(e + u, e1 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e + u < e1) saga_face_normal_add(sv, t[u].x, t[u].y, t[u].z, nn);
        }
        saga_normal_finish(nn);
        float* out = o < n_a ? normals + ((size_t)b * n_a + o) * 3 : normals_b + ((size_t)b * (n_out - n_a) + (o - n_a)) * 3;
        out[0] = nn[0]; out[1] = nn[1]; out[2] = nn[2];
    }
}

// B outputs; T, t0: verts holds sequences of T meshes and output b is mesh t0 + b % (T - t0) of sequence b / (T - t0) (1, 0: mesh b)
static int launch(const saga_model_t* m, const float* verts, const int32_t* sel, int n_out, float* normals, int B, int T, int t0,
                  void* stream, int n_a = -1, float* normals_b = nullptr) {
    if (n_a < 0) n_a = n_out;
    SAGA_CHECK_ARG(m && m->F > 0, "saga_vertex_normals: model has no faces");
    SAGA_CHECK_DEVICE(m, "saga_vertex_normals");
    SAGA_CHECK_ARG(B >= 0 && B <= 65535 && n_out >= 0, "saga_vertex_normals: bad sizes B=%d n=%d", B, n_out);
    if (B == 0 || n_out == 0) return 0;
    SAGA_CHECK_ARG(verts && normals, "saga_vertex_normals: null argument");
    const size_t lds = ((size_t)m->V * 3 + 3) * sizeof(float);
    const bool staged = B >= 512 && lds <= 160 * 1024 && getenv("SAGA_NORMALS_GATHER") == nullptr;   // (cross-check switch, read per call)
    if (staged) {
        static bool attr_set[64] = {};
        if (!attr_set[m->device & 63]) {
            SAGA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&vertex_normals_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[m->device & 63] = true;
        }
    }
    dim3 grid((n_out + 255) / 256, B);
    void* prof = saga_prof_begin(SAGA_PROF_NORMALS, stream);
    if (staged)
        vertex_normals_lds_kernel<<<B, VNL_THREADS, lds, (hipStream_t)stream>>>(verts, m->vf_ptr, m->vf_tri, sel, n_out, normals, m->V, T, t0, n_a,
                                                                                normals_b);
    else
        vertex_normals_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(verts, m->vf_ptr, m->vf_tri, sel, n_out, normals, m->V, T, t0, n_a,
                                                                     normals_b);
    saga_prof_end(prof);
    SAGA_LAUNCH_CHECK();
    return 0;
}

extern "C" int saga_vertex_normals(const saga_model_t* m, const float* verts, float* normals, int B, void* stream) {
    return launch(m, verts, nullptr, m ? m->V : 0, normals, B, 1, 0, stream);
}

// `sel` (device, int32) must hold vertex ids in [0, V): checked by the host wrapper that uploads it.
extern "C" int saga_vertex_normals_subset(const saga_model_t* m, const float* verts, const int32_t* sel, int n_sel, float* normals,
                                          int B, void* stream) {
    SAGA_CHECK_ARG(sel || n_sel == 0, "saga_vertex_normals_subset: null selection");
    return launch(m, verts, sel, n_sel, normals, B, 1, 0, stream);
}

// verts [S,T,V,3]: the listed vertices on the frames t0.. of every sequence only -> normals [S,T-t0,n_sel,3]
extern "C" int saga_vertex_normals_subset_window(const saga_model_t* m, const float* verts, const int32_t* sel, int n_sel,
                                                 float* normals, int S, int T, int t0, void* stream) {
    SAGA_CHECK_ARG(sel || n_sel == 0, "saga_vertex_normals_subset_window: null selection");
    SAGA_CHECK_ARG(S >= 0 && T > 0 && t0 >= 0 && t0 < T, "saga_vertex_normals_subset_window: bad sizes S=%d T=%d t0=%d", S, T, t0);
    return launch(m, verts, sel, n_sel, normals, S * (T - t0), T, t0, stream);
}

// the same with the outputs split into two arrays: sel[0..n_a) -> normals_a [S,T-t0,n_a,3], sel[n_a..n_a+n_b) -> normals_b
// [S,T-t0,n_b,3] (internal: saga_motion_step hands the hand normals to the distance operator and the palm-marker normals to the
// term operator, each contiguous, from ONE staging of the mesh)
int saga_vertex_normals_window_split(const saga_model_t* m, const float* verts, const int32_t* sel, int n_a, int n_b, float* normals_a,
                                     float* normals_b, int S, int T, int t0, void* stream) {
    SAGA_CHECK_ARG(sel && n_a > 0 && n_b > 0 && normals_b, "saga_vertex_normals_window_split: bad arguments");
    SAGA_CHECK_ARG(S >= 0 && T > 0 && t0 >= 0 && t0 < T, "saga_vertex_normals_window_split: bad sizes S=%d T=%d t0=%d", S, T, t0);
    return launch(m, verts, sel, n_a + n_b, normals_a, S * (T - t0), T, t0, stream, n_a, normals_b);
}

// ---- the loss kernel's reads of the mesh, served once per candidate from LDS (saga_fit_step) --------------------------------
// One block per candidate stages the "ring" -- every corner of a face that touches a right-hand vertex (tables' ring_ids: 6,485
// vertices = 78 KB on the synthetic mesh with random faces, two blocks per CU; ~1 k on a real mesh) -- in LDS with one gather in
// ascending id order, computes the right-hand vertex normals from there (same sums as saga_vertex_normal, face by face in ascending
// order: same bits) and writes the vertices the loss reads (markers / hand / feet: tables' unique ids) as compact rows.  The loss
// kernel then touches neither the mesh nor its faces: its ~14k scattered 12-B corner reads per candidate were 2 GB of L2-missing
// traffic per launch and a fifth of its time (three blocks per CU each waiting on them).  The index chain of the normals (range
// of incident faces -> packed corner ids; model constants) is fetched BEFORE the gather, so the block waits for memory only twice.
#define RING_THREADS 512
#define RING_NF 12              // faces per hand vertex whose packed corner ids are fetched up front
#define RING_MAXH 2             // hand vertices per lane (n_hand <= 1,024)
__global__ __launch_bounds__(RING_THREADS) void hand_ring_kernel(saga_grasp_tables_t tb, const float* __restrict__ verts,
                                                                 float* __restrict__ uniq_rows, float* __restrict__ hand_normals, int V) {
    extern __shared__ __attribute__((aligned(16))) float ring_smem[];        // [n_ring*3]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* vb = verts + (size_t)b * V * 3;
    const int H = tb.n_hand;
    int e0[RING_MAXH], e1[RING_MAXH];
    int2 tri[RING_MAXH][RING_NF];
#pragma unroll
    for (int r = 0; r < RING_MAXH; ++r) {
        const int i = tid + r * RING_THREADS;
        e0[r] = i < H ? tb.hand_tri_ptr[i] : 0;
        e1[r] = i < H ? tb.hand_tri_ptr[i + 1] : 0;
    }
#pragma unroll
    for (int r = 0; r < RING_MAXH; ++r)
#pragma unroll
        for (int u = 0; u < RING_NF; ++u) tri[r][u] = reinterpret_cast<const int2*>(tb.hand_tri)[max(min(e0[r] + u, e1[r] - 1), 0)];
    typedef float f32x3_t __attribute__((ext_vector_type(3)));
    // ring gather, eight vertices per lane in flight
    for (int i0 = 0; i0 < tb.n_ring; i0 += RING_THREADS * 8) {
        f32x3_t p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u * RING_THREADS + tid, tb.n_ring - 1);
            p[u] = *reinterpret_cast<const f32x3_t*>(vb + (size_t)tb.ring_ids[i] * 3);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * RING_THREADS + tid;
            if (i < tb.n_ring) { ring_smem[i * 3] = p[u].x; ring_smem[i * 3 + 1] = p[u].y; ring_smem[i * 3 + 2] = p[u].z; }
        }
    }
    // compact rows of the loss vertices
    float* urow = uniq_rows + (size_t)b * tb.n_unique * 3;
    for (int i = tid; i < tb.n_unique; i += RING_THREADS) {
        const f32x3_t p = *reinterpret_cast<const f32x3_t*>(vb + (size_t)tb.unique_ids[i] * 3);
        urow[i * 3] = p.x; urow[i * 3 + 1] = p.y; urow[i * 3 + 2] = p.z;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RING_MAXH; ++r) {
        const int i = tid + r * RING_THREADS;
        if (i >= H) continue;
        float n[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < RING_NF; ++u)
            if (e0[r] + u < e1[r]) saga_face_normal_add(ring_smem, tri[r][u].x & 0xffff, (unsigned)tri[r][u].x >> 16, tri[r][u].y, n);
        for (int e = e0[r] + RING_NF; e < e1[r]; e += 8) {       // valence beyond RING_NF (rare): eight more per round trip
            int2 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = reinterpret_cast<const int2*>(tb.hand_tri)[min(e + u, e1[r] - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e + u < e1[r]) saga_face_normal_add(ring_smem, t[u].x & 0xffff, (unsigned)t[u].x >> 16, t[u].y, n);
        }
        saga_normal_finish(n);
        float* d = hand_normals + ((size_t)b * H + i) * 3;
        d[0] = n[0]; d[1] = n[1]; d[2] = n[2];
    }
}

// returns 1 (nothing launched) when the tables carry no ring or it does not fit the LDS of a CU
int saga_launch_hand_ring(const saga_model_t* m, const saga_grasp_tables_t* tb, const float* verts, float* uniq_rows,
                          float* hand_normals, int B, void* stream) {
    if (!tb->ring_ids || !tb->hand_tri_ptr || !tb->hand_tri || tb->n_ring <= 0 || tb->n_hand > RING_MAXH * RING_THREADS) return 1;
    const size_t lds = (size_t)tb->n_ring * 3 * sizeof(float);
    if (lds > 160 * 1024) return 1;
    static bool attr_set[64] = {};          // per device: the attribute belongs to the device's copy of the kernel
    if (!attr_set[m->device & 63]) {
        SAGA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hand_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
        attr_set[m->device & 63] = true;
    }
    void* prof = saga_prof_begin(SAGA_PROF_NORMALS, stream);
    hand_ring_kernel<<<B, RING_THREADS, lds, (hipStream_t)stream>>>(*tb, verts, uniq_rows, hand_normals, m->V);
    saga_prof_end(prof);
    SAGA_LAUNCH_CHECK();
    return 0;
}
