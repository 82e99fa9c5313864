// Row picks for objectives that read several (overlapping) vertex sets of one cloud, some of them on a window of the frame axis
// only: src [S,T,U,3] -> out_k [S,T-t0_k,n_k,3] with out_k[s][t-t0_k][i] = src[s][t][pos_k[i]], every set in ONE launch, and the
// backward as ONE gather over the inverse maps -- g_src[s][t][u] = sum over the sets k (in set order, t >= t0_k) of the picks i of
// u (ascending i) of g_out_k[s][t-t0_k][i]: no atomics, no zero-filled full-size temporaries, no pairwise adds of them.
// (The autograd form of MotionFill/models/fittingop.py:356-409 is one index_select and one frame slice per set: each backward
// zero-fills and index_adds a full-size gradient and autograd then adds them pairwise.)
#include <algorithm>

#include "saga_common.h"

#define PICK_THREADS 256

struct PickSets {
    saga_pick_set_t s[SAGA_PICK_MAX_SETS];
    int n;
};

__global__ __launch_bounds__(PICK_THREADS) void pick_sets_fwd_kernel(const float* __restrict__ src, PickSets ps, int T, int U) {
    const int frame = blockIdx.y, s = frame / T, t = frame - s * T;
    const float* sp = src + (size_t)frame * U * 3;
    for (int k = 0; k < ps.n; ++k) {
        const saga_pick_set_t& st = ps.s[k];
        if (t < st.t0) continue;
        float* o = st.out + ((size_t)s * (T - st.t0) + (t - st.t0)) * st.n * 3;
        typedef float f32x3_t __attribute__((ext_vector_type(3)));
        for (int i = blockIdx.x * PICK_THREADS + threadIdx.x; i < st.n; i += gridDim.x * PICK_THREADS)       // one 12-B row per lane
            *reinterpret_cast<f32x3_t*>(o + i * 3) = *reinterpret_cast<const f32x3_t*>(sp + st.pos[i] * 3);
    }
}

__global__ __launch_bounds__(PICK_THREADS) void pick_sets_bwd_kernel(PickSets ps, float* __restrict__ g_src, int T, int U) {
    const int frame = blockIdx.y, s = frame / T, t = frame - s * T;
    float* o = g_src + (size_t)frame * U * 3;
    typedef float f32x3_t __attribute__((ext_vector_type(3)));
    for (int u = blockIdx.x * PICK_THREADS + threadIdx.x; u < U; u += gridDim.x * PICK_THREADS) {         // one 12-B row per lane
        f32x3_t acc = {0.f, 0.f, 0.f};
        for (int k = 0; k < ps.n; ++k) {
            const saga_pick_set_t& st = ps.s[k];
            if (t < st.t0 || !st.g_out) continue;
            const float* g = st.g_out + ((size_t)s * (T - st.t0) + (t - st.t0)) * st.n * 3;
            for (int j = st.inv_ptr[u]; j < st.inv_ptr[u + 1]; ++j) acc += *reinterpret_cast<const f32x3_t*>(g + st.inv_idx[j] * 3);
        }
        *reinterpret_cast<f32x3_t*>(o + u * 3) = acc;
    }
}

static int pick_check(const saga_pick_set_t* sets, int n_sets, int S, int T, int U, bool fwd, PickSets& ps, const char* what) {
    SAGA_CHECK_ARG(sets && n_sets > 0 && n_sets <= SAGA_PICK_MAX_SETS, "%s: 1..%d sets", what, SAGA_PICK_MAX_SETS);
    SAGA_CHECK_ARG(S >= 0 && T > 0 && U > 0 && (long long)S * T <= 65535, "%s: bad sizes S=%d T=%d U=%d (S*T <= 65535)", what, S, T, U);
    ps.n = n_sets;
    if (S == 0) return 0;        // an empty batch: nothing to check the pointers of (empty tensors have none)
    for (int k = 0; k < n_sets; ++k) {
        const saga_pick_set_t& st = sets[k];
        SAGA_CHECK_ARG(st.n > 0 && st.t0 >= 0 && st.t0 < T, "%s: set %d has n=%d t0=%d (T=%d)", what, k, st.n, st.t0, T);
        if (fwd) SAGA_CHECK_ARG(st.pos && st.out, "%s: set %d: null pos / out", what, k);
        else SAGA_CHECK_ARG(st.inv_ptr && st.inv_idx, "%s: set %d: null inverse map", what, k);
        ps.s[k] = st;
    }
    return 0;
}

extern "C" int saga_pick_sets_fwd(const float* src, const saga_pick_set_t* sets, int n_sets, int S, int T, int U, void* stream) {
    PickSets ps;
    if (int rc = pick_check(sets, n_sets, S, T, U, true, ps, "saga_pick_sets_fwd")) return rc;
    if (S == 0) return 0;
    SAGA_CHECK_ARG(src, "saga_pick_sets_fwd: null src");
    int n_max = 0;
    for (int k = 0; k < n_sets; ++k) n_max = std::max(n_max, sets[k].n);
    dim3 grid(std::min(4, (n_max + PICK_THREADS - 1) / PICK_THREADS), S * T);
    pick_sets_fwd_kernel<<<grid, PICK_THREADS, 0, (hipStream_t)stream>>>(src, ps, T, U);
    SAGA_LAUNCH_CHECK();
    return 0;
}

extern "C" int saga_pick_sets_bwd(const saga_pick_set_t* sets, int n_sets, float* g_src, int S, int T, int U, void* stream) {
    PickSets ps;
    if (int rc = pick_check(sets, n_sets, S, T, U, false, ps, "saga_pick_sets_bwd")) return rc;
    if (S == 0) return 0;
    SAGA_CHECK_ARG(g_src, "saga_pick_sets_bwd: null g_src");
    dim3 grid(std::min(4, (U + PICK_THREADS - 1) / PICK_THREADS), S * T);
    pick_sets_bwd_kernel<<<grid, PICK_THREADS, 0, (hipStream_t)stream>>>(ps, g_src, T, U);
    SAGA_LAUNCH_CHECK();
    return 0;
}
