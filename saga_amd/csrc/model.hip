// Model constants: host-side re-layout + upload.  Replaces the buffers smplx.create() registers
// (reference call site WholeGraspPose/models/fittingop.py:30-59).
#include <stdarg.h>

#include <algorithm>
#include <vector>

#include "saga_common.h"

static thread_local char g_err[512] = "";

void saga_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* saga_last_error(void) { return g_err; }

int saga_check_device(int device, const char* what) {
    int cur = -1;
    SAGA_HIP(hipGetDevice(&cur));
    SAGA_CHECK_ARG(cur == device, "%s: the handle's constants live on device %d but the current device is %d "
                   "(make that device current -- hipSetDevice / torch.cuda.device -- and pass one of its streams)", what, device, cur);
    return 0;
}
// Release builds report a positive version.  A library compiled with -DSAGA_EXPERIMENT_BUILD -- the only way the timing
// experiments in the sources (-DGEMM_EXP_*, -DDS_EXP_*, -DGL_EXP_*, -DGL_PHASE_TIMING: several give WRONG results by design)
// compile at all -- reports the negated version, and saga_amd._lib.load() refuses it outside scripts/exp.
#define SAGA_VERSION 300
extern "C" int saga_version(void) {
#ifdef SAGA_EXPERIMENT_BUILD
    return -SAGA_VERSION;
#else
    return SAGA_VERSION;
#endif
}

int saga_noalloc_init_device(int device);

template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    SAGA_HIP(hipMalloc((void**)dst, bytes));
    if (!src.empty()) SAGA_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

extern "C" void saga_model_destroy(saga_model_t* m) {
    if (!m) return;
    void* ptrs[] = {m->Bt, m->Bk, m->v_template, m->J_template, m->J_dirs, m->parents, m->depth, m->child_ptr,
                    m->child_idx, m->w_idx, m->w_val, m->Wd, m->ds_ent, m->ds_meta, m->ds_item_ptr, m->faces, m->vf_ptr, m->vf_idx, m->vf_tri, m->lmk_verts,
                    m->lmk_bary, m->extra_verts, m->sel_ids, m->full_slot};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete m;
}

extern "C" int saga_model_npad(const saga_model_t* m) { return m ? m->Npad : SAGA_EINVAL; }
extern "C" int saga_model_num_joints_out(const saga_model_t* m) { return m ? NJ + m->E + m->L : SAGA_EINVAL; }
extern "C" int saga_model_weight_nnz(const saga_model_t* m) { return m ? m->KW : SAGA_EINVAL; }

// sel (optional): ascending ids of the vertices the model keeps.  VF = vertices of the arrays in the descriptor, V = vertices of
// the model; the joint regression always runs over the full arrays.
static int model_create_impl(saga_model_t** out, const saga_model_desc_t* d, const int32_t* sel, int n_sel) {
    SAGA_CHECK_ARG(out && d, "saga_model_create: null argument");
    *out = nullptr;
    const int VF = d->n_verts, S = d->n_shape, L = d->n_landmarks, E = d->n_extra_joints;
    const int V = sel ? n_sel : VF;
    int F = d->n_faces;
    SAGA_CHECK_ARG(VF > 0 && V > 0 && F >= 0 && L >= 0 && E >= 0, "saga_model_create: bad sizes V=%d F=%d L=%d E=%d", VF, F, L, E);
    if (sel) {
        SAGA_CHECK_ARG(L == 0 && E == 0, "saga_model_create_subset: a vertex subset has no landmark / extra-joint outputs (pass L = E = 0)");
        for (int i = 0; i < n_sel; ++i)
            SAGA_CHECK_ARG(sel[i] >= 0 && sel[i] < VF && (i == 0 || sel[i] > sel[i - 1]), "saga_model_create_subset: ids must be ascending and < V");
    }
    auto sv = [&](int v) { return sel ? sel[v] : v; };
    SAGA_CHECK_ARG(S >= 0 && S + NPF <= FEAT_K, "saga_model_create: n_shape=%d does not fit K=%d", S, FEAT_K);
    SAGA_CHECK_ARG(d->v_template_host && d->posedirs_host && d->J_regressor_host && d->parents_host &&
                       d->lbs_weights_host && (S == 0 || d->shapedirs_host),
                   "saga_model_create: null array");
    SAGA_CHECK_ARG(F == 0 || d->faces_host, "saga_model_create: faces missing");
    SAGA_CHECK_ARG(L == 0 || (d->lmk_faces_idx_host && d->lmk_bary_host && F > 0), "saga_model_create: landmarks missing");
    SAGA_CHECK_ARG(E == 0 || d->extra_joint_verts_host, "saga_model_create: extra joints missing");
    SAGA_CHECK_ARG(d->parents_host[0] == -1, "saga_model_create: parents[0] must be -1");
    for (int j = 1; j < NJ; ++j)
        SAGA_CHECK_ARG(d->parents_host[j] >= 0 && d->parents_host[j] < j, "saga_model_create: parents[%d]=%d not topological", j,
                       d->parents_host[j]);
    for (int f = 0; f < 3 * F; ++f)
        SAGA_CHECK_ARG(d->faces_host[f] >= 0 && d->faces_host[f] < VF, "saga_model_create: face index out of range");
    for (int l = 0; l < L; ++l)
        SAGA_CHECK_ARG(d->lmk_faces_idx_host[l] >= 0 && d->lmk_faces_idx_host[l] < F, "saga_model_create: landmark face out of range");
    for (int e = 0; e < E; ++e)
        SAGA_CHECK_ARG(d->extra_joint_verts_host[e] >= 0 && d->extra_joint_verts_host[e] < VF,
                       "saga_model_create: extra joint vertex %d out of range", d->extra_joint_verts_host[e]);

    hipDeviceProp_t prop;
    int dev = 0;
    SAGA_HIP(hipGetDevice(&dev));
    SAGA_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        saga_set_error("libsaga_hip is built for gfx950 only; device %d is %s", dev, prop.gcnArchName);
        return SAGA_EDEVICE;
    }

    saga_model* m = new saga_model();
    memset(m, 0, sizeof(*m));
    m->V = V; m->S = S; m->L = L; m->E = E;       // m->F below (a subset keeps only the faces inside it)
    m->N = 3 * V;
    m->Npad = saga_roundup(3 * V, SAGA_N_ALIGN);
    m->Vpad4 = saga_roundup(V, 256);
    m->device = dev;
    m->VF = VF;
    const int N = m->N, Npad = m->Npad;
    int rc = 0;
    if (sel) {
        std::vector<int32_t> ids(sel, sel + n_sel), slot(VF, -1);
        for (int i = 0; i < n_sel; ++i) slot[sel[i]] = i;
        if ((rc = upload(&m->sel_ids, ids)) || (rc = upload(&m->full_slot, slot))) goto fail;
    }

    {   // blend operands, both orientations
        std::vector<float> Bt((size_t)Npad * FEAT_K, 0.f), Bk((size_t)FEAT_K * Npad, 0.f), vt(Npad, 0.f);
        for (int n = 0; n < N; ++n) {
            float* row = &Bt[(size_t)n * FEAT_K];
            const size_t ns = (size_t)sv(n / 3) * 3 + n % 3;         // coordinate index in the descriptor's arrays
            for (int s = 0; s < S; ++s) row[s] = d->shapedirs_host[ns * S + s];
            for (int p = 0; p < NPF; ++p) row[S + p] = d->posedirs_host[(size_t)p * (3 * (size_t)VF) + ns];
            vt[n] = d->v_template_host[ns];
        }
        for (int k = 0; k < FEAT_K; ++k)
            for (int n = 0; n < N; ++n) Bk[(size_t)k * Npad + n] = Bt[(size_t)n * FEAT_K + k];
        if ((rc = upload(&m->Bt, Bt)) || (rc = upload(&m->Bk, Bk)) || (rc = upload(&m->v_template, vt))) goto fail;
    }
    {   // folded joint regression (accumulated in double)
        std::vector<float> Jt(NJ * 3), Jd((size_t)std::max(S, 1) * NJ * 3, 0.f);
        for (int j = 0; j < NJ; ++j) {
            const float* reg = d->J_regressor_host + (size_t)j * VF;
            double acc[3] = {0, 0, 0};
            std::vector<double> accd((size_t)3 * S, 0.0);
            for (int v = 0; v < VF; ++v) {
                double w = reg[v];
                if (w == 0.0) continue;
                for (int c = 0; c < 3; ++c) {
                    acc[c] += w * d->v_template_host[v * 3 + c];
                    for (int s = 0; s < S; ++s) accd[c * S + s] += w * d->shapedirs_host[((size_t)v * 3 + c) * S + s];
                }
            }
            for (int c = 0; c < 3; ++c) {
                Jt[j * 3 + c] = (float)acc[c];
                for (int s = 0; s < S; ++s) Jd[(size_t)s * NJ * 3 + j * 3 + c] = (float)accd[c * S + s];
            }
        }
        if ((rc = upload(&m->J_template, Jt)) || (rc = upload(&m->J_dirs, Jd))) goto fail;
    }
    {   // kinematic tree
        std::vector<int32_t> par(d->parents_host, d->parents_host + NJ), depth(NJ, 0), cptr(NJ + 1, 0), cidx;
        for (int j = 1; j < NJ; ++j) depth[j] = depth[par[j]] + 1;
        m->max_depth = *std::max_element(depth.begin(), depth.end());
        for (int p = 0; p < NJ; ++p) {
            cptr[p] = (int32_t)cidx.size();
            for (int j = 1; j < NJ; ++j)
                if (par[j] == p) cidx.push_back(j);
        }
        cptr[NJ] = (int32_t)cidx.size();
        memcpy(m->h_parents, par.data(), sizeof(int32_t) * NJ);
        if ((rc = upload(&m->parents, par)) || (rc = upload(&m->depth, depth)) || (rc = upload(&m->child_ptr, cptr)) ||
            (rc = upload(&m->child_idx, cidx)))
            goto fail;
    }
    {   // skinning weights
        int KW = 1;
        for (int v = 0; v < V; ++v) {
            int nnz = 0;
            for (int j = 0; j < NJ; ++j) nnz += d->lbs_weights_host[(size_t)sv(v) * NJ + j] != 0.f;
            KW = std::max(KW, nnz);
        }
        KW = saga_roundup(KW, 2);
        m->KW = KW;
        std::vector<int32_t> wi((size_t)V * KW, 0);
        std::vector<float> wv((size_t)V * KW, 0.f), Wd((size_t)m->Vpad4 * WT_COLS, 0.f);
        for (int v = 0; v < V; ++v) {
            int k = 0;
            for (int j = 0; j < NJ; ++j) {
                float w = d->lbs_weights_host[(size_t)sv(v) * NJ + j];
                Wd[(size_t)v * WT_COLS + (j & 15) * 4 + (j >> 4)] = w;      // column j at slot (j%16)*4 + j/16
                if (w != 0.f) { wi[(size_t)v * KW + k] = j; wv[(size_t)v * KW + k] = w; ++k; }
            }
            Wd[(size_t)v * WT_COLS + (NJ & 15) * 4 + (NJ >> 4)] = 1.f;
        }
        if ((rc = upload(&m->w_idx, wi)) || (rc = upload(&m->w_val, wv)) || (rc = upload(&m->Wd, Wd))) goto fail;
    }
    {   // sparse dA reduction tables (see saga_common.h)
        const int P = (V + DS_VR - 1) / DS_VR;
        const int Vr = saga_roundup((V + P - 1) / P, 64);
        std::vector<int2> ent;
        std::vector<int32_t> meta, iptr;
        bool ok = P <= DS_MAX_PARTS && (P - 1) * Vr < V && Vr <= DS_VR;
        for (int r = 0; r < P && ok; ++r) {
            const int lo = r * Vr, hi = std::min(V, lo + Vr);
            std::vector<std::vector<int2>> lists(NJ);
            for (int v = lo; v < hi; ++v)
                for (int j = 0; j < NJ; ++j) {
                    const float w = d->lbs_weights_host[(size_t)sv(v) * NJ + j];
                    if (w != 0.f) { int2 e; e.x = v - lo; memcpy(&e.y, &w, 4); lists[j].push_back(e); }
                }
            int C = 4;
            auto n_items = [&](int c) { int n = 0; for (int j = 0; j < NJ; ++j) n += ((int)lists[j].size() + c - 1) / c; return n; };
            while (C <= DS_MAX_CHUNK && n_items(C) > DS_THREADS) C += 4;
            if (C > DS_MAX_CHUNK) { ok = false; break; }
            const size_t off = ent.size();
            meta.push_back((int32_t)off); meta.push_back(C);
            int2 zero; zero.x = 0; zero.y = 0;
            ent.resize(off + (size_t)C * DS_THREADS, zero);
            int item = 0;
            for (int j = 0; j < NJ; ++j) {
                iptr.push_back(item);
                // entry e of the joint goes to its item e % n, slot e / n: at every step the lanes of a joint read consecutive
                // list entries (consecutive vertices when a body part's vertices are numbered together -> no LDS bank conflicts)
                const size_t n = (lists[j].size() + C - 1) / C;
                for (size_t e = 0; e < lists[j].size(); ++e) ent[off + (e / n) * DS_THREADS + item + e % n] = lists[j][e];
                item += (int)n;
            }
            iptr.push_back(item);
        }
        if (ok) {
            m->ds_parts = P; m->ds_vr = Vr;
            if ((rc = upload(&m->ds_ent, ent)) || (rc = upload(&m->ds_meta, meta)) || (rc = upload(&m->ds_item_ptr, iptr))) goto fail;
        }
    }
    {   // faces, vertex->face incidence (ascending face order), landmarks, extra joints
        std::vector<int32_t> faces(d->faces_host, d->faces_host + (size_t)3 * F);
        if (sel) {      // keep the faces whose three corners are kept, in their order, with the corners renumbered
            std::vector<int32_t> pos(VF, -1), kept;
            for (int i = 0; i < n_sel; ++i) pos[sel[i]] = i;
            for (int f = 0; f < F; ++f) {
                const int a = pos[faces[f * 3]], b = pos[faces[f * 3 + 1]], c = pos[faces[f * 3 + 2]];
                if (a >= 0 && b >= 0 && c >= 0) { kept.push_back(a); kept.push_back(b); kept.push_back(c); }
            }
            faces.swap(kept);
            F = (int)(faces.size() / 3);
        }
        m->F = F;
        {
            size_t far = 0;
            for (int f = 0; f < F; ++f) {
                const int a = faces[f * 3], b = faces[f * 3 + 1], c = faces[f * 3 + 2];
                const int lo = std::min(a, std::min(b, c)), hi = std::max(a, std::max(b, c));
                far += (hi - lo) > 1024;
            }
            m->scattered_faces = F > 0 && far * 2 > (size_t)F;
        }
        std::vector<int32_t> vptr(V + 1, 0), vidx((size_t)3 * F);
        for (int f = 0; f < 3 * F; ++f) vptr[faces[f] + 1]++;
        for (int v = 0; v < V; ++v) vptr[v + 1] += vptr[v];
        std::vector<int32_t> fill(vptr.begin(), vptr.end() - 1);
        for (int f = 0; f < F; ++f)
            for (int c = 0; c < 3; ++c) vidx[fill[faces[f * 3 + c]]++] = f;
        std::vector<int32_t> vtri((size_t)12 * F, 0);
        for (size_t e = 0; e < (size_t)3 * F; ++e)
            for (int c = 0; c < 3; ++c) vtri[e * 4 + c] = faces[(size_t)vidx[e] * 3 + c];
        if ((rc = upload(&m->vf_tri, vtri))) goto fail;
        std::vector<int32_t> lv((size_t)3 * L);
        std::vector<float> lb(d->lmk_bary_host, d->lmk_bary_host + (size_t)3 * L);
        for (int l = 0; l < L; ++l)
            for (int c = 0; c < 3; ++c) lv[l * 3 + c] = faces[d->lmk_faces_idx_host[l] * 3 + c];
        std::vector<int32_t> ev(d->extra_joint_verts_host, d->extra_joint_verts_host + E);
        if ((rc = upload(&m->faces, faces)) || (rc = upload(&m->vf_ptr, vptr)) || (rc = upload(&m->vf_idx, vidx)) ||
            (rc = upload(&m->lmk_verts, lv)) || (rc = upload(&m->lmk_bary, lb)) || (rc = upload(&m->extra_verts, ev)))
            goto fail;
    }
    if ((rc = saga_noalloc_init_device(dev))) goto fail;       // this device's copy of the static scratch (smplx_api.hip)
    *out = m;
    return 0;
fail:
    saga_model_destroy(m);
    return rc;
}

extern "C" int saga_model_create(saga_model_t** out, const saga_model_desc_t* d) { return model_create_impl(out, d, nullptr, 0); }

extern "C" int saga_model_create_subset(saga_model_t** out, const saga_model_desc_t* d, const int32_t* vertex_ids_host, int n_ids) {
    SAGA_CHECK_ARG(vertex_ids_host && n_ids > 0, "saga_model_create_subset: empty vertex list");
    return model_create_impl(out, d, vertex_ids_host, n_ids);
}

extern "C" int saga_model_num_faces(const saga_model_t* m) { return m ? m->F : SAGA_EINVAL; }
