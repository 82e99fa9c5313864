// The geometric terms of the GraspMotion-Opt stage-2 objective (MotionFill/models/fittingop.py:132-280,356-409) for S sequences,
// as one forward launch and one backward launch, from the picked vertex sets (rows.hip) and the outputs of saga_p2p_signed_fwd:
//   0 rec_markers      mean |markers*mask - markers_rec*mask|                                   (:360-366)
//   1 end_markers_fit  mean |markers[T-1] - markers_end|                                        (:368)
//   2 skating          per heel / toe group: mean speed (30 fps) of the vertices moving faster than vel_thres on the frames
//                      labelled as contact; groups without such a vertex give 0                (:193-263)
//   3 hand_smooth      mean ||hand[t+1] - hand[t]|| over the frames start_t..                   (:386-387)
//   4 contact          mean |o2h|*contact_object*w + mean |h2o_marker|*contact_markers*w, w = 1 on the last five frames (:159-176)
//   5 collision        5 * mean |o2h| * (20 where o2h < -3 mm)                                  (:178-191)
//   6 hand_angle       mean [h2o_marker < 1 cm] (cos(palm normal, -marker->object vector) - 1)^2 (1 - ((t+1)/n)^2)   (:265-280)
// Every sum has a fixed order (lane-strided partial sums, wave tree, waves in order) and a sequence's block sees only its own
// data: bit-reproducible, and a sequence's terms do not depend on the batch around it.  The autograd composition of the same
// terms (saga_amd/motion_losses.py) is ~300 small launches per iteration over tensors of up to 65 MB; it stays as the checker.
#include <math.h>

#include "saga_common.h"

#define MT_THREADS 256
#define MT_NT 12            // floats per sequence in `terms`: 7 terms, 1 pad, 4 skating counts

struct MtDims {
    int S, T, n_tail, M, NF, fs[4], H, O, CM, NP, n_w;
    float vel_thres;
};

template <int K>
__device__ __forceinline__ void mt_block_sum(float (&v)[K], float* sm) {
#pragma unroll
    for (int k = 0; k < K; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) sm[(threadIdx.x >> 6) * K + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float s = sm[k];
        for (int w = 1; w < MT_THREADS / 64; ++w) s += sm[w * K + k];
        v[k] = s;
    }
}

__device__ __forceinline__ float mt_sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }
__device__ __forceinline__ float mt_frame_w(int t, int n_tail, int n_w) { return t >= n_tail - n_w ? 1.f : 0.f; }
__device__ __forceinline__ float mt_angle_w(int t, int n_tail) {
    const float r = (float)(t + 1) / (float)n_tail;
    return 1.f - r * r;
}

__global__ __launch_bounds__(MT_THREADS) void motion_terms_fwd_kernel(
    MtDims d, const float* __restrict__ markers, const float* __restrict__ markers_rec, const float* __restrict__ marker_mask,
    const float* __restrict__ markers_end, const float* __restrict__ feet, const float* __restrict__ lbl,
    const float* __restrict__ hand, const float* __restrict__ o2h, const float* __restrict__ c_obj, const float* __restrict__ h_mk,
    const float* __restrict__ c_mk, const float* __restrict__ vec, const float* __restrict__ palm_n, float* __restrict__ terms) {
#pragma clang fp contract(off)
    __shared__ float sm[(MT_THREADS / 64) * 8];
    const int s = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    float* out = terms + (size_t)s * MT_NT;
    if (part == 0) {
        float v[2] = {0.f, 0.f};
        const int n = d.T * d.M * 3;
        const float* mk = markers + (size_t)s * n;
        const float* rc = markers_rec + (size_t)s * n;
        for (int e = tid; e < n; e += MT_THREADS) {
            const float m = marker_mask[(e / 3) % d.M];
            v[0] += fabsf(mk[e] * m - rc[e] * m);
        }
        const float* last = mk + (size_t)(d.T - 1) * d.M * 3;
        const float* en = markers_end + (size_t)s * d.M * 3;
        for (int e = tid; e < d.M * 3; e += MT_THREADS) v[1] += fabsf(last[e] - en[e]);
        mt_block_sum<2>(v, sm);
        if (tid == 0) { out[0] = v[0] / (float)n; out[1] = v[1] / (float)(d.M * 3); }
    } else if (part == 1) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* f = feet + (size_t)s * d.T * d.NF * 3;
        const float* lb = lbl + (size_t)s * d.T * 4;
        int og = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ng = d.fs[g];
            for (int e = tid; e < (d.T - 1) * ng; e += MT_THREADS) {
                const int t = e / ng, r = og + (e - t * ng);
                const float* a = f + ((size_t)t * d.NF + r) * 3;
                const float* b = a + (size_t)d.NF * 3;
                const float x = (b[0] - a[0]) * 30.f, y = (b[1] - a[1]) * 30.f, z = (b[2] - a[2]) * 30.f;
                const float speed = sqrtf(x * x + y * y + z * z);
                const bool on = lb[t * 4 + g] == 1.f && speed > d.vel_thres;
                v[g] += on ? speed : 0.f;
                v[4 + g] += on ? 1.f : 0.f;
            }
            og += ng;
        }
        mt_block_sum<8>(v, sm);
        if (tid == 0) {
            float tot = 0.f;
            for (int g = 0; g < 4; ++g) { tot += v[4 + g] > 0.f ? v[g] / v[4 + g] : 0.f; out[8 + g] = v[4 + g]; }
            out[2] = tot;
        }
    } else if (part == 2) {
        float v[1] = {0.f};
        const float* h = hand + (size_t)s * d.n_tail * d.H * 3;
        for (int e = tid; e < (d.n_tail - 1) * d.H; e += MT_THREADS) {
            const float* a = h + (size_t)e * 3;
            const float* b = a + (size_t)d.H * 3;
            const float x = b[0] - a[0], y = b[1] - a[1], z = b[2] - a[2];
            v[0] += sqrtf(x * x + y * y + z * z);
        }
        mt_block_sum<1>(v, sm);
        if (tid == 0) out[3] = v[0] / (float)((d.n_tail - 1) * d.H);
    } else {
        float v[4] = {0.f, 0.f, 0.f, 0.f};       // contact (object), contact (markers), collision, angle
        const float* oh = o2h + (size_t)s * d.n_tail * d.O;
        const float* co = c_obj + (size_t)s * d.O;
        for (int e = tid; e < d.n_tail * d.O; e += MT_THREADS) {
            const int t = e / d.O, o = e - t * d.O;
            const float x = oh[e], a = fabsf(x);
            v[0] += a * co[o] * mt_frame_w(t, d.n_tail, d.n_w);
            v[2] += a * (x < -0.003f ? 20.f : 0.f);
        }
        const float* hm = h_mk + (size_t)s * d.n_tail * d.CM;
        const float* cm = c_mk + (size_t)s * d.CM;
        for (int e = tid; e < d.n_tail * d.CM; e += MT_THREADS) {
            const int t = e / d.CM, m = e - t * d.CM;
            v[1] += fabsf(hm[e]) * cm[m] * mt_frame_w(t, d.n_tail, d.n_w);
        }
        const float* vc = vec + (size_t)s * d.n_tail * d.CM * 3;
        const float* pn = palm_n + (size_t)s * d.n_tail * d.NP * 3;
        for (int e = tid; e < d.n_tail * d.NP; e += MT_THREADS) {
            const int t = e / d.NP, k = e - t * d.NP, m = d.CM - d.NP + k;
            const float* q = vc + ((size_t)t * d.CM + m) * 3;
            const float* nn = pn + (size_t)e * 3;
            const float vx = -q[0], vy = -q[1], vz = -q[2];
            const float dot = nn[0] * vx + nn[1] * vy + nn[2] * vz;
            const float ln = sqrtf(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]), lv = sqrtf(vx * vx + vy * vy + vz * vz);
            const float c = dot / (ln * lv) - 1.f;
            const float mask = hm[t * d.CM + m] < 0.01f ? 1.f : 0.f;
            v[3] += mask * (c * c) * mt_angle_w(t, d.n_tail);       // (0 * NaN stays NaN, as in the composition)
        }
        mt_block_sum<4>(v, sm);
        if (tid == 0) {
            out[4] = v[0] / (float)(d.n_tail * d.O) + v[1] / (float)(d.n_tail * d.CM);
            out[5] = 5.f * (v[2] / (float)(d.n_tail * d.O));
            out[6] = v[3] / (float)(d.n_tail * d.NP);
            out[7] = 0.f;
        }
    }
}

// Backward: every gradient element is written exactly once (no atomics).  blockIdx.y = sequence, blockIdx.z = which array.
__global__ __launch_bounds__(MT_THREADS) void motion_terms_bwd_kernel(
    MtDims d, const float* __restrict__ markers, const float* __restrict__ markers_rec, const float* __restrict__ marker_mask,
    const float* __restrict__ markers_end, const float* __restrict__ feet, const float* __restrict__ lbl,
    const float* __restrict__ hand, const float* __restrict__ o2h, const float* __restrict__ c_obj, const float* __restrict__ h_mk,
    const float* __restrict__ c_mk, const float* __restrict__ vec, const float* __restrict__ palm_n, const float* __restrict__ terms,
    const float* __restrict__ g_terms, float* __restrict__ g_markers, float* __restrict__ g_feet, float* __restrict__ g_hand,
    float* __restrict__ g_o2h, float* __restrict__ g_h_mk, float* __restrict__ g_vec) {
#pragma clang fp contract(off)
    const int s = blockIdx.y, which = blockIdx.z;
    const int stride = gridDim.x * MT_THREADS, first = blockIdx.x * MT_THREADS + threadIdx.x;
    const float* g = g_terms + (size_t)s * 7;
    if (which == 0) {
        const int n = d.T * d.M * 3;
        const float* mk = markers + (size_t)s * n;
        const float* rc = markers_rec + (size_t)s * n;
        const float* en = markers_end + (size_t)s * d.M * 3;
        const float k0 = g[0] / (float)n, k1 = g[1] / (float)(d.M * 3);
        for (int e = first; e < n; e += stride) {
            const int r = e / 3, t = r / d.M, m = r - t * d.M;
            const float mm = marker_mask[m];
            float a = k0 * mm * mt_sgn(mk[e] * mm - rc[e] * mm);
            if (t == d.T - 1) a += k1 * mt_sgn(mk[e] - en[(e - (d.T - 1) * d.M * 3)]);
            g_markers[(size_t)s * n + e] = a;
        }
    } else if (which == 1) {
        const float* f = feet + (size_t)s * d.T * d.NF * 3;
        const float* lb = lbl + (size_t)s * d.T * 4;
        const float* cnt = terms + (size_t)s * MT_NT + 8;
        const int n = d.T * d.NF;
        for (int e = first; e < n; e += stride) {
            const int t = e / d.NF, r = e - t * d.NF;
            int gi = 0, acc = d.fs[0];
            while (gi < 3 && r >= acc) acc += d.fs[++gi];
            const float sc = cnt[gi] > 0.f ? g[2] / cnt[gi] : 0.f;
            const float* p = f + (size_t)e * 3;
            float gx = 0.f, gy = 0.f, gz = 0.f;
            if (t > 0) {                    // interval (t-1, t): this row is its later end
                const float* a = p - (size_t)d.NF * 3;
                const float x = (p[0] - a[0]) * 30.f, y = (p[1] - a[1]) * 30.f, z = (p[2] - a[2]) * 30.f;
                const float speed = sqrtf(x * x + y * y + z * z);
                if (lb[(t - 1) * 4 + gi] == 1.f && speed > d.vel_thres) {
                    const float k = sc * 30.f / speed;
                    gx += k * x; gy += k * y; gz += k * z;
                }
            }
            if (t < d.T - 1) {              // interval (t, t+1): this row is its earlier end
                const float* b = p + (size_t)d.NF * 3;
                const float x = (b[0] - p[0]) * 30.f, y = (b[1] - p[1]) * 30.f, z = (b[2] - p[2]) * 30.f;
                const float speed = sqrtf(x * x + y * y + z * z);
                if (lb[t * 4 + gi] == 1.f && speed > d.vel_thres) {
                    const float k = sc * 30.f / speed;
                    gx -= k * x; gy -= k * y; gz -= k * z;
                }
            }
            float* o = g_feet + ((size_t)s * n + e) * 3;
            o[0] = gx; o[1] = gy; o[2] = gz;
        }
    } else if (which == 2) {
        const float* h = hand + (size_t)s * d.n_tail * d.H * 3;
        const int n = d.n_tail * d.H;
        const float k3 = g[3] / (float)((d.n_tail - 1) * d.H);
        for (int e = first; e < n; e += stride) {
            const int t = e / d.H;
            const float* p = h + (size_t)e * 3;
            float gx = 0.f, gy = 0.f, gz = 0.f;
            if (t > 0) {
                const float* a = p - (size_t)d.H * 3;
                const float x = p[0] - a[0], y = p[1] - a[1], z = p[2] - a[2];
                const float l = sqrtf(x * x + y * y + z * z);
                if (l > 0.f) { gx += k3 * x / l; gy += k3 * y / l; gz += k3 * z / l; }
            }
            if (t < d.n_tail - 1) {
                const float* b = p + (size_t)d.H * 3;
                const float x = b[0] - p[0], y = b[1] - p[1], z = b[2] - p[2];
                const float l = sqrtf(x * x + y * y + z * z);
                if (l > 0.f) { gx -= k3 * x / l; gy -= k3 * y / l; gz -= k3 * z / l; }
            }
            float* o = g_hand + ((size_t)s * n + e) * 3;
            o[0] = gx; o[1] = gy; o[2] = gz;
        }
    } else if (which == 3) {
        const float* oh = o2h + (size_t)s * d.n_tail * d.O;
        const float* co = c_obj + (size_t)s * d.O;
        const float k4 = g[4] / (float)(d.n_tail * d.O), k5 = g[5] * 5.f / (float)(d.n_tail * d.O);
        for (int e = first; e < d.n_tail * d.O; e += stride) {
            const int t = e / d.O, o = e - t * d.O;
            const float x = oh[e];
            g_o2h[(size_t)s * d.n_tail * d.O + e] =
                mt_sgn(x) * (k4 * co[o] * mt_frame_w(t, d.n_tail, d.n_w) + k5 * (x < -0.003f ? 20.f : 0.f));
        }
    } else {
        const float* hm = h_mk + (size_t)s * d.n_tail * d.CM;
        const float* cm = c_mk + (size_t)s * d.CM;
        const float* vc = vec + (size_t)s * d.n_tail * d.CM * 3;
        const float* pn = palm_n + (size_t)s * d.n_tail * d.NP * 3;
        const float k4 = g[4] / (float)(d.n_tail * d.CM), k6 = g[6] / (float)(d.n_tail * d.NP);
        for (int e = first; e < d.n_tail * d.CM; e += stride) {
            const int t = e / d.CM, m = e - t * d.CM;
            g_h_mk[(size_t)s * d.n_tail * d.CM + e] = k4 * mt_sgn(hm[e]) * cm[m] * mt_frame_w(t, d.n_tail, d.n_w);
            float gx = 0.f, gy = 0.f, gz = 0.f;
            const int k = m - (d.CM - d.NP);
            if (k >= 0) {
                const float* q = vc + (size_t)e * 3;
                const float* nn = pn + ((size_t)t * d.NP + k) * 3;
                const float vx = -q[0], vy = -q[1], vz = -q[2];
                const float dot = nn[0] * vx + nn[1] * vy + nn[2] * vz;
                const float ln = sqrtf(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]), lv = sqrtf(vx * vx + vy * vy + vz * vz);
                const float cs = dot / (ln * lv);
                const float mask = hm[e] < 0.01f ? 1.f : 0.f;
                const float c = k6 * 2.f * mask * mt_angle_w(t, d.n_tail) * (cs - 1.f);        // d term / d cos
                // d cos / d v = n / (|n||v|) - cos v / |v|^2 ;  vec = -v
                const float a = 1.f / (ln * lv), b = cs / (lv * lv);
                gx = -c * (nn[0] * a - vx * b); gy = -c * (nn[1] * a - vy * b); gz = -c * (nn[2] * a - vz * b);
            }
            float* o = g_vec + ((size_t)s * d.n_tail * d.CM + e) * 3;
            o[0] = gx; o[1] = gy; o[2] = gz;
        }
    }
}

static int mt_dims(const saga_motion_terms_cfg_t* c, MtDims& d, const char* what) {
    SAGA_CHECK_ARG(c, "%s: null cfg", what);
    SAGA_CHECK_ARG(c->n_seq >= 0 && c->n_seq <= 65535 && c->n_frames >= 2 && c->n_tail >= 1 && c->n_tail <= c->n_frames,
                   "%s: bad sizes S=%d T=%d tail=%d", what, c->n_seq, c->n_frames, c->n_tail);
    SAGA_CHECK_ARG(c->n_markers > 0 && c->n_hand > 0 && c->n_obj > 0 && c->n_contact_markers > 0 && c->n_palm > 0 &&
                       c->n_palm <= c->n_contact_markers && c->n_last_frames >= 0,
                   "%s: bad set sizes", what);
    d.S = c->n_seq; d.T = c->n_frames; d.n_tail = c->n_tail; d.M = c->n_markers; d.H = c->n_hand; d.O = c->n_obj;
    d.CM = c->n_contact_markers; d.NP = c->n_palm; d.n_w = c->n_last_frames; d.vel_thres = c->vel_thres;
    d.NF = 0;
    for (int g = 0; g < 4; ++g) {
        SAGA_CHECK_ARG(c->foot_group_size[g] >= 0, "%s: negative foot group size", what);
        d.fs[g] = c->foot_group_size[g];
        d.NF += d.fs[g];
    }
    SAGA_CHECK_ARG(d.NF > 0, "%s: no foot vertices", what);
    return 0;
}

extern "C" int saga_motion_terms_fwd(const saga_motion_terms_cfg_t* cfg, const float* markers, const float* markers_rec,
                                     const float* marker_mask, const float* markers_end, const float* feet, const float* contact_lbl,
                                     const float* hand_tail, const float* o2h_signed, const float* contact_object,
                                     const float* h2o_marker, const float* contact_markers, const float* h2o_marker_vec,
                                     const float* palm_normals, float* terms, void* stream) {
    MtDims d;
    if (int rc = mt_dims(cfg, d, "saga_motion_terms_fwd")) return rc;
    if (d.S == 0) return 0;
    SAGA_CHECK_ARG(markers && markers_rec && marker_mask && markers_end && feet && contact_lbl && hand_tail && o2h_signed &&
                       contact_object && h2o_marker && contact_markers && h2o_marker_vec && palm_normals && terms,
                   "saga_motion_terms_fwd: null argument");
    if (d.S == 0) return 0;
    motion_terms_fwd_kernel<<<dim3(d.S, 4), MT_THREADS, 0, (hipStream_t)stream>>>(d, markers, markers_rec, marker_mask, markers_end, feet,
                                                                                  contact_lbl, hand_tail, o2h_signed, contact_object,
                                                                                  h2o_marker, contact_markers, h2o_marker_vec,
                                                                                  palm_normals, terms);
    SAGA_LAUNCH_CHECK();
    return 0;
}

extern "C" int saga_motion_terms_bwd(const saga_motion_terms_cfg_t* cfg, const float* markers, const float* markers_rec,
                                     const float* marker_mask, const float* markers_end, const float* feet, const float* contact_lbl,
                                     const float* hand_tail, const float* o2h_signed, const float* contact_object,
                                     const float* h2o_marker, const float* contact_markers, const float* h2o_marker_vec,
                                     const float* palm_normals, const float* terms, const float* g_terms, float* g_markers,
                                     float* g_feet, float* g_hand_tail, float* g_o2h_signed, float* g_h2o_marker,
                                     float* g_h2o_marker_vec, void* stream) {
    MtDims d;
    if (int rc = mt_dims(cfg, d, "saga_motion_terms_bwd")) return rc;
    if (d.S == 0) return 0;
    SAGA_CHECK_ARG(markers && markers_rec && marker_mask && markers_end && feet && contact_lbl && hand_tail && o2h_signed &&
                       contact_object && h2o_marker && contact_markers && h2o_marker_vec && palm_normals && terms && g_terms &&
                       g_markers && g_feet && g_hand_tail && g_o2h_signed && g_h2o_marker && g_h2o_marker_vec,
                   "saga_motion_terms_bwd: null argument");
    if (d.S == 0) return 0;
    motion_terms_bwd_kernel<<<dim3(8, d.S, 5), MT_THREADS, 0, (hipStream_t)stream>>>(
        d, markers, markers_rec, marker_mask, markers_end, feet, contact_lbl, hand_tail, o2h_signed, contact_object, h2o_marker,
        contact_markers, h2o_marker_vec, palm_normals, terms, g_terms, g_markers, g_feet, g_hand_tail, g_o2h_signed, g_h2o_marker,
        g_h2o_marker_vec);
    SAGA_LAUNCH_CHECK();
    return 0;
}
