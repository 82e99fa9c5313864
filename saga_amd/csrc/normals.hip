// Area-weighted vertex normals.  Replaces Open3D TriangleMesh.compute_vertex_normals() as called every
// iteration at WholeGraspPose/models/fittingop.py:158-164 (a device->host->device round trip in the reference).
// Gather formulation: each output vertex walks its incident faces in ascending face order (the order Open3D's
// face loop adds them), sums the un-normalised cross products and normalises; zero rows stay zero, NaN -> (0,0,1).
// No atomics, deterministic.  HBM/L2-bound gather.
#include <math.h>

#include "saga_common.h"
#include "vertex_normal.h"

void* saga_prof_begin(int id, void* stream);
void saga_prof_end(void* scope);

__global__ void vertex_normals_kernel(const float* __restrict__ verts, const int32_t* __restrict__ vf_ptr,
                                      const int32_t* __restrict__ vf_tri, const int32_t* __restrict__ sel, int n_out,
                                      float* __restrict__ normals, int V) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (o >= n_out) return;
    const int v = sel ? sel[o] : o;
    float nx, ny, nz;
    saga_vertex_normal(verts + (size_t)b * V * 3, vf_ptr, vf_tri, v, nx, ny, nz);
    float* out = normals + ((size_t)b * n_out + o) * 3;
    out[0] = nx; out[1] = ny; out[2] = nz;
}

static int launch(const saga_model_t* m, const float* verts, const int32_t* sel, int n_out, float* normals, int B, void* stream) {
    SAGA_CHECK_ARG(m && m->F > 0, "saga_vertex_normals: model has no faces");
    SAGA_CHECK_DEVICE(m, "saga_vertex_normals");
    SAGA_CHECK_ARG(B >= 0 && B <= 65535 && n_out >= 0, "saga_vertex_normals: bad sizes B=%d n=%d", B, n_out);
    if (B == 0 || n_out == 0) return 0;
    SAGA_CHECK_ARG(verts && normals, "saga_vertex_normals: null argument");
    dim3 grid((n_out + 255) / 256, B);
    void* prof = saga_prof_begin(SAGA_PROF_NORMALS, stream);
    vertex_normals_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(verts, m->vf_ptr, m->vf_tri, sel, n_out, normals, m->V);
    saga_prof_end(prof);
    SAGA_LAUNCH_CHECK();
    return 0;
}

extern "C" int saga_vertex_normals(const saga_model_t* m, const float* verts, float* normals, int B, void* stream) {
    return launch(m, verts, nullptr, m ? m->V : 0, normals, B, stream);
}

// `sel` (device, int32) must hold vertex ids in [0, V): checked by the host wrapper that uploads it.
extern "C" int saga_vertex_normals_subset(const saga_model_t* m, const float* verts, const int32_t* sel, int n_sel, float* normals,
                                          int B, void* stream) {
    SAGA_CHECK_ARG(sel || n_sel == 0, "saga_vertex_normals_subset: null selection");
    return launch(m, verts, sel, n_sel, normals, B, stream);
}
