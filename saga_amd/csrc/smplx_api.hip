// extern "C" orchestration of the SMPL-X forward / backward (kernel sequence on one stream, no allocation, no sync).
#include <stdlib.h>

#include "saga_common.h"

int saga_launch_pose_fwd(const saga_model* m, const float* shape, const float* full_pose, float* feat, float* rot,
                         float* jrest, float* chain, float* A, int B, hipStream_t st, int* zero8);
int saga_launch_pose_bwd(const saga_model* m, const float* dA, const float* g_pj, const float* g_jsum, const float* slabs,
                         int n_slabs, const float* rot, const float* jrest, const float* chain, const float* full_pose,
                         float* g_shape, float* g_full_pose, float* g_transl, int B, int has_verts, hipStream_t st);
int saga_launch_blend_fwd(const saga_model* m, const float* feat, float* v_posed, int B, hipStream_t st, int* queues);
bool saga_blend_fwd_uses_queue(const saga_model* m, int B);
int saga_launch_blend_bwd(const saga_model* m, const float* g_vposed, float* slabs, int B, hipStream_t st, int* queues);
bool saga_blend_bwd_uses_queue(const saga_model* m, int B);
int saga_blend_bwd_splitk(const saga_model* m);
int saga_launch_skin_fwd(const saga_model* m, const float* A, const float* v_posed, const float* transl, float* verts, int B,
                         hipStream_t st);
int saga_launch_skin_fwd_sel(const saga_model* m, const saga_model* sub, const float* A, const float* v_posed, const float* transl,
                             float* verts, float* sel_rows, float* sel_verts, int B, hipStream_t st);
int saga_launch_joints_fwd(const saga_model* m, const float* chain, const float* verts, const float* transl, float* joints, int B,
                           hipStream_t st);
int saga_launch_joints_bwd(const saga_model* m, const float* g_joints, float* g_pj, float* g_jsum, float* gv, int B,
                           int small_stride, hipStream_t st);
int saga_launch_skin_bwd(const saga_model* m, const float* A, const float* g_verts, const float* v_posed, float* g_vposed,
                         float* dA, int dA_stride, float* scratch, int B, hipStream_t st, int* zero8, bool* zeroed);

// ---- optional per-kernel timing (HIP events on the launch stream) -------------------------------------------
// bench.py turns this on for its timed region: every kernel launch of the path is bracketed by an event pair on
// the stream it is launched on; saga_profile_read() resolves the pairs.  Off by default (zero overhead).
#include <vector>
struct ProfPair { hipEvent_t a, b; int id; };
static int g_prof_on = 0;       // 0 off, 1 every kernel, 2 only the blend GEMM launches (the roofline kernel)
static thread_local int g_prof_suspend = 0;  // set for the duration of saga_smplx_bwd_noalloc ON THE CALLING THREAD: event creation allocates
static std::vector<ProfPair> g_prof_pairs;
static std::vector<ProfPair> g_prof_pool;

struct ProfScope {
    hipStream_t st; hipEvent_t b = nullptr; bool on;
    ProfScope(int id, hipStream_t s)
        : st(s), on(!g_prof_suspend && (g_prof_on == 1 || (g_prof_on == 2 && (id == SAGA_PROF_BLEND_FWD || id == SAGA_PROF_BLEND_BWD)))) {
        if (!on) return;
        ProfPair p;
        if (!g_prof_pool.empty()) { p = g_prof_pool.back(); g_prof_pool.pop_back(); }
        else if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) { on = false; return; }
        p.id = id;
        (void)hipEventRecord(p.a, st);
        b = p.b;
        g_prof_pairs.push_back(p);
    }
    ~ProfScope() { if (on) (void)hipEventRecord(b, st); }
};

int saga_prof_is_on() { return g_prof_on != 0; }
void* saga_prof_begin(int id, void* stream) { return g_prof_on ? new ProfScope(id, (hipStream_t)stream) : nullptr; }
void saga_prof_end(void* scope) { delete (ProfScope*)scope; }

extern "C" int saga_profile_enable(int on) {
    g_prof_on = on < 0 ? 0 : (on > 2 ? 1 : on);
    return 0;
}

// Synchronises the recorded events; total_ms[id] += elapsed, count[id] += 1 for id in [0, SAGA_PROF_NKERNELS).
extern "C" int saga_profile_read(double* total_ms_host, int32_t* count_host) {
    SAGA_CHECK_ARG(total_ms_host && count_host, "saga_profile_read: null argument");
    for (int i = 0; i < SAGA_PROF_NKERNELS; ++i) { total_ms_host[i] = 0.0; count_host[i] = 0; }
    for (ProfPair& p : g_prof_pairs) {
        SAGA_HIP(hipEventSynchronize(p.b));
        float ms = 0.f;
        SAGA_HIP(hipEventElapsedTime(&ms, p.a, p.b));
        if (p.id >= 0 && p.id < SAGA_PROF_NKERNELS) { total_ms_host[p.id] += ms; count_host[p.id] += 1; }
        g_prof_pool.push_back(p);
    }
    g_prof_pairs.clear();
    return 0;
}

#define PROF(id) ProfScope prof_scope_##id(id, st)

extern "C" int saga_smplx_fwd_is_persistent(const saga_model_t* m, int B) { return m ? (saga_blend_fwd_uses_queue(m, B) ? 1 : 0) : -1; }
extern "C" int saga_smplx_bwd_is_persistent(const saga_model_t* m, int B) {
    static const int force_mfma = getenv("SAGA_DA_MFMA") != nullptr;       // (the dense skinning backward does not zero the queue)
    return m ? ((saga_blend_bwd_uses_queue(m, B) && m->ds_parts > 0 && !force_mfma) ? 1 : 0) : -1;
}

// sub / sel_rows (both NULL for the plain forward): see saga_launch_skin_fwd_sel; used by saga_fit_step_sparse
int saga_smplx_fwd_sel(const saga_model_t* m, const saga_model_t* sub, const float* shape, const float* full_pose, const float* transl,
                       float* vertices, float* joints, float* feat, float* rot, float* jrest, float* chain, float* A, float* v_posed,
                       float* sel_rows, int B, void* stream, float* sel_verts = nullptr);

extern "C" int saga_smplx_fwd(const saga_model_t* m, const float* shape, const float* full_pose, const float* transl,
                              float* vertices, float* joints, float* feat, float* rot, float* jrest, float* chain, float* A,
                              float* v_posed, int B, void* stream) {
    return saga_smplx_fwd_sel(m, nullptr, shape, full_pose, transl, vertices, joints, feat, rot, jrest, chain, A, v_posed, nullptr, B,
                              stream);
}

// The forward with the rows of a vertex subset written compact as well (for callers whose loss reads those vertices only and whose
// backward runs on the subset model: smplx forward(vertex_grad_ids=...)): v_posed_sel [B, npad(sub)] in sub's layout (what
// saga_smplx_bwd(sub, ...) takes as v_posed) and vertices_sel [B, V(sub), 3].  m may be a subset model too; sub's vertices must be
// vertices of m.
extern "C" int saga_smplx_fwd_selected(const saga_model_t* m, const saga_model_t* sub, const float* shape, const float* full_pose,
                                       const float* transl, float* vertices, float* joints, float* feat, float* rot, float* jrest,
                                       float* chain, float* A, float* v_posed, float* v_posed_sel, float* vertices_sel, int B,
                                       void* stream) {
    SAGA_CHECK_ARG(sub && v_posed_sel && vertices_sel, "saga_smplx_fwd_selected: null subset argument");
    SAGA_CHECK_DEVICE(sub, "saga_smplx_fwd_selected (subset model)");
    return saga_smplx_fwd_sel(m, sub, shape, full_pose, transl, vertices, joints, feat, rot, jrest, chain, A, v_posed, v_posed_sel, B,
                              stream, vertices_sel);
}

int saga_smplx_fwd_sel(const saga_model_t* m, const saga_model_t* sub, const float* shape, const float* full_pose, const float* transl,
                       float* vertices, float* joints, float* feat, float* rot, float* jrest, float* chain, float* A, float* v_posed,
                       float* sel_rows, int B, void* stream, float* sel_verts) {
    SAGA_CHECK_ARG(m, "saga_smplx_fwd: null model");
    SAGA_CHECK_DEVICE(m, "saga_smplx_fwd");
    SAGA_CHECK_ARG(B >= 0 && B <= 65535, "saga_smplx_fwd: batch %d outside [0, 65535]", B);
    if (B == 0) return 0;
    SAGA_CHECK_ARG(full_pose && vertices && feat && rot && jrest && chain && A && v_posed && (m->S == 0 || shape),
                   "saga_smplx_fwd: null buffer");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // the persistent blend kernel draws its tiles from 8 counters: the head of `vertices`, free until the skinning kernel writes it
    int* queues = saga_blend_fwd_uses_queue(m, B) ? reinterpret_cast<int*>(vertices) : nullptr;
    { PROF(SAGA_PROF_POSE_FWD); if ((rc = saga_launch_pose_fwd(m, shape, full_pose, feat, rot, jrest, chain, A, B, st, queues))) return rc; }
    { PROF(SAGA_PROF_BLEND_FWD); if ((rc = saga_launch_blend_fwd(m, feat, v_posed, B, st, queues))) return rc; }
    {
        PROF(SAGA_PROF_SKIN_FWD);
        if ((rc = sub ? saga_launch_skin_fwd_sel(m, sub, A, v_posed, transl, vertices, sel_rows, sel_verts, B, st)
                      : saga_launch_skin_fwd(m, A, v_posed, transl, vertices, B, st)))
            return rc;
    }
    if (joints) { PROF(SAGA_PROF_JOINTS); if ((rc = saga_launch_joints_fwd(m, chain, vertices, transl, joints, B, st))) return rc; }
    return 0;
}

extern "C" int saga_smplx_bwd(const saga_model_t* m, const float* g_vertices, const float* g_joints, const float* full_pose,
                              const float* rot, const float* jrest, const float* chain, const float* A, const float* v_posed,
                              float* g_shape, float* g_full_pose, float* g_transl, float* ws_gvp, float* ws_slab, float* ws_dA,
                              float* ws_gv, int B, void* stream) {
    SAGA_CHECK_ARG(m, "saga_smplx_bwd: null model");
    SAGA_CHECK_DEVICE(m, "saga_smplx_bwd");
    SAGA_CHECK_ARG(B >= 0 && B <= 65535, "saga_smplx_bwd: batch %d outside [0, 65535]", B);
    if (B == 0) return 0;
    SAGA_CHECK_ARG(g_vertices || g_joints, "saga_smplx_bwd: no incoming gradient");
    SAGA_CHECK_ARG(full_pose && rot && jrest && chain && A && v_posed && g_full_pose && g_transl && ws_dA && (m->S == 0 || g_shape),
                   "saga_smplx_bwd: null buffer");
    hipStream_t st = (hipStream_t)stream;
    const int V = m->V;
    const int stride = SAGA_DA_STRIDE;
    float* g_pj = nullptr;
    float* g_jsum = nullptr;
    const float* gv = g_vertices;
    int has_verts = g_vertices != nullptr;
    int rc;
    if (g_joints) {
        // extra joints and landmarks are functions of the vertices: fold their gradient into a private copy
        const bool touches_verts = (m->E + m->L) > 0;
        g_pj = ws_dA + DA_ROWS * 12;
        g_jsum = g_pj + NJ * 3;
        if (touches_verts) {
            SAGA_CHECK_ARG(ws_gv, "saga_smplx_bwd: ws_gv is required when g_joints is given");
            if (g_vertices)
                SAGA_HIP(hipMemcpyAsync(ws_gv, g_vertices, (size_t)B * V * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
            else
                SAGA_HIP(hipMemsetAsync(ws_gv, 0, (size_t)B * V * 3 * sizeof(float), st));
            gv = ws_gv;
            has_verts = 1;
        }
        { PROF(SAGA_PROF_JOINTS); if ((rc = saga_launch_joints_bwd(m, g_joints, g_pj, g_jsum, ws_gv, B, stride, st))) return rc; }
    }
    if (has_verts) {
        SAGA_CHECK_ARG(ws_gvp && ws_slab, "saga_smplx_bwd: workspace missing");
        // the persistent blend kernel draws its tiles from 8 counters: the head of g_full_pose, free until the pose kernel writes it
        int* queues = saga_blend_bwd_uses_queue(m, B) ? reinterpret_cast<int*>(g_full_pose) : nullptr;
        bool zeroed = false;
        { PROF(SAGA_PROF_SKIN_BWD); if ((rc = saga_launch_skin_bwd(m, A, gv, v_posed, ws_gvp, ws_dA, stride, ws_slab, B, st, queues, &zeroed))) return rc; }
        { PROF(SAGA_PROF_BLEND_BWD); if ((rc = saga_launch_blend_bwd(m, ws_gvp, ws_slab, B, st, zeroed ? queues : nullptr))) return rc; }
    }
    PROF(SAGA_PROF_POSE_BWD);
    return saga_launch_pose_bwd(m, ws_dA, g_pj, g_jsum, ws_slab, saga_blend_bwd_splitk(m), rot, jrest, chain, full_pose, g_shape, g_full_pose,
                                g_transl, B, has_verts, st);
}

// ---- the entry point that allocates nothing -----------------------------------------------------------------------
// saga_smplx_bwd with its four workspaces in a file-scope device array instead of caller-allocated buffers, for callers
// that cannot call hipMalloc (under HIP graph capture, tight serving loops, harnesses that forbid it).  Each GPU the library
// runs on holds its own copy of the array (18.4 MB) until the process ends.  In host code the array's NAME is a host-side
// stand-in, not a device address: its memory is reached only through the pointer hipGetSymbolAddress returns, looked up once
// per device with that device current (at saga_model_create -- saga_noalloc_init_device -- or on the first call) and kept per device.
#define NOALLOC_MAX_B    SAGA_NOALLOC_MAX_B
#define NOALLOC_MAX_NPAD SAGA_NOALLOC_MAX_NPAD
#define NOALLOC_MAX_DEVICES 64
static constexpr size_t SCR_GVP  = (size_t)NOALLOC_MAX_B * NOALLOC_MAX_NPAD;             // ws_gvp  [B, Npad]
static constexpr size_t SCR_SLAB = (size_t)SAGA_BWD_SPLITK * NOALLOC_MAX_B * FEAT_K;     // ws_slab [16, B, 512]
static constexpr size_t SCR_DA   = ((size_t)NOALLOC_MAX_B * SAGA_DA_STRIDE + 63) / 64 * 64;   // ws_dA [B, 840]
static constexpr size_t SCR_GV   = (size_t)NOALLOC_MAX_B * NOALLOC_MAX_NPAD;             // ws_gv   [B, V, 3]  (3V <= Npad)
static constexpr size_t SCR_TOTAL = SCR_GVP + SCR_SLAB + SCR_DA + SCR_GV;
static __device__ __align__(256) float scratch[SCR_TOTAL];
static float* g_scratch_ptr[NOALLOC_MAX_DEVICES];      // per device: the address of ITS copy of `scratch`

// looks up (once) the current device's copy of the scratch array; *out = its device address
static int noalloc_scratch(int device, float** out) {
    SAGA_CHECK_ARG(device >= 0 && device < NOALLOC_MAX_DEVICES, "saga_smplx_bwd_noalloc: device %d beyond %d", device, NOALLOC_MAX_DEVICES);
    if (!g_scratch_ptr[device]) {
        float* p = nullptr;
        SAGA_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));       // of the CURRENT device (checked by the callers)
        SAGA_CHECK_ARG(p, "saga_smplx_bwd_noalloc: hipGetSymbolAddress returned a null address");
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return 0;
}

// called by saga_model_create with the model's device current: the lookup happens at set-up, not inside a capture
int saga_noalloc_init_device(int device) {
    float* p;
    return noalloc_scratch(device, &p);
}

extern "C" int saga_smplx_bwd_noalloc(const saga_model_t* m, const float* g_vertices, const float* g_joints, const float* full_pose,
                                      const float* rot, const float* jrest, const float* chain, const float* A,
                                      const float* v_posed, float* g_shape, float* g_full_pose, float* g_transl, int B,
                                      void* stream) {
    SAGA_CHECK_ARG(m, "saga_smplx_bwd_noalloc: null model");
    SAGA_CHECK_DEVICE(m, "saga_smplx_bwd_noalloc");
    SAGA_CHECK_ARG(B >= 0 && B <= NOALLOC_MAX_B, "saga_smplx_bwd_noalloc: batch %d exceeds the static scratch (at most %d candidates "
                   "per call; use saga_smplx_bwd with caller workspaces beyond that)", B, NOALLOC_MAX_B);
    SAGA_CHECK_ARG(m->Npad <= NOALLOC_MAX_NPAD, "saga_smplx_bwd_noalloc: the model's padded row (%d floats) exceeds the static scratch (%d)",
                   m->Npad, NOALLOC_MAX_NPAD);
    if (B == 0) return 0;
    float* base = nullptr;
    int rc = noalloc_scratch(m->device, &base);
    if (rc) return rc;
    // nothing in the scratch has to start at zero: saga_smplx_bwd treats its workspaces as undefined on entry (ws_gv is
    // copied / cleared inside when g_joints is given)
    g_prof_suspend = 1;
    rc = saga_smplx_bwd(m, g_vertices, g_joints, full_pose, rot, jrest, chain, A, v_posed, g_shape, g_full_pose, g_transl,
                        base, base + SCR_GVP, base + SCR_GVP + SCR_SLAB, base + SCR_GVP + SCR_SLAB + SCR_DA, B, stream);
    g_prof_suspend = 0;
    return rc;
}
