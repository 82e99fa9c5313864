// Internal definitions shared by the HIP translation units of libsaga_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/saga_hip.h"

#define NJ SAGA_NUM_JOINTS      // 55
#define NPF 486                 // pose-feature length (54 joints x 9)
#define FEAT_K SAGA_FEAT_K      // 512
#define DA_ROWS 56              // 55 joints + 1 all-ones row (sum of gradients -> d transl)
#define DS_THREADS 256          // sparse dA reduction: lanes (= items) per block
#ifndef DS_VR
#define DS_VR 1024              // ... and the most vertices a block stages in LDS (6 floats each: 42 KB)
#endif
#ifndef DS_MAX_PARTS
#define DS_MAX_PARTS 16         // partial sums live in the split-K slab space: parts * SAGA_DA_STRIDE <= 16 * 512 floats per candidate
#endif
#define DS_MAX_CHUNK 64         // longer chunks mean the weights are dense: the MFMA reduction takes over
#define WT_COLS 64              // dense weight table padded to 64 columns for the MFMA dA reduction

void saga_set_error(const char* fmt, ...);
// 0 when `device` (where a model / decoder handle's constants live) is the calling thread's current HIP device, else
// SAGA_EINVAL: the kernels would otherwise launch on another GPU with this one's pointers
int saga_check_device(int device, const char* what);
#define SAGA_CHECK_DEVICE(handle, what)                                   \
    do {                                                                  \
        int rc_dev_ = saga_check_device((handle)->device, what);          \
        if (rc_dev_) return rc_dev_;                                      \
    } while (0)

#define SAGA_CHECK_ARG(cond, ...)                      \
    do {                                               \
        if (!(cond)) {                                 \
            saga_set_error(__VA_ARGS__);               \
            return SAGA_EINVAL;                        \
        }                                              \
    } while (0)

#define SAGA_HIP(call)                                                              \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            saga_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                         \
        }                                                                           \
    } while (0)

#define SAGA_LAUNCH_CHECK()                                                         \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            saga_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                         \
        }                                                                           \
    } while (0)

// Device-resident model constants, laid out for the kernels (see DESIGN.md 'Data layout in HBM').
struct saga_model {
    int V, S, F, L, E;
    int N;        // 3V
    int Npad;     // roundup(3V, 128)
    int KW;       // entries per vertex in the compressed weight table
    int max_depth;
    int device;
    // blend contraction operands
    float* Bt;          // [Npad, 512]  Bt[n][k]: k<S shapedirs, S<=k<S+486 posedirs, else 0 (forward B operand)
    float* Bk;          // [512, Npad]  the transpose (backward B operand)
    float* v_template;  // [Npad] (pad = 0)
    // folded joint regression: J = J_template + shape . J_dirs
    float* J_template;  // [55*3]
    float* J_dirs;      // [S, 165]
    // kinematic tree
    int32_t* parents;   // [55]
    int32_t* depth;     // [55]
    int32_t* child_ptr; // [56] CSR of children
    int32_t* child_idx; // [54]
    int32_t h_parents[NJ];
    // skinning weights: compressed rows (joint, weight) and dense padded table for the dA reduction
    int32_t* w_idx;     // [V, KW]
    float*   w_val;     // [V, KW] (pad entries: idx 0, weight 0)
    float*   Wd;        // [Vpad4, 64]  column j of W (j<55; j=55: all ones; j>55: 0) sits at slot (j%16)*4 + j/16, so the
                        // four N-tile values of an MFMA lane (joints i16, i16+16, i16+32, i16+48) are one 16-B load; rows >= V are 0
    int Vpad4;          // roundup(V, 256)
    // sparse dA reduction (skin.hip, skin_bwd_sparse_kernel): the vertices are cut into ds_parts ranges of ds_vr; per range the
    // non-zero weights are listed joint by joint (ascending vertex), each joint's list cut into chunks of at most C entries
    // so that a range has at most DS_THREADS chunks ("items", one per lane).  ds_parts = 0: table not built (weights too dense).
    int ds_parts, ds_vr;
    int2*    ds_ent;      // per range: [C][DS_THREADS] (vertex - range start, weight bits); pad entries have weight 0
    int32_t* ds_meta;     // [ds_parts][2]: offset of the range's entries in ds_ent, C
    int32_t* ds_item_ptr; // [ds_parts][NJ+1]: first item of each joint
    // faces / landmarks / extra joints
    int scattered_faces; // 1 when most faces join vertices more than 1,024 ids apart (kernels that gather face corners then warm L2 first)
    int32_t* faces;     // [F,3]
    int32_t* vf_ptr;    // [V+1] CSR vertex -> incident (face, corner) in ascending face order
    int32_t* vf_idx;    // [3F]  face id
    int32_t* vf_tri;    // [3F,4] the three vertex ids of that face (+ pad): one 16-B load per incidence entry
    int32_t* lmk_verts; // [L,3] vertex ids of each landmark's triangle
    float*   lmk_bary;  // [L,3]
    int32_t* extra_verts; // [E]
    // vertex subset (saga_model_create_subset): VF = vertices of the full mesh, sel_ids [V] = full-mesh id of every kept vertex
    // (ascending); a full model has VF = V and sel_ids = NULL
    int VF;
    int32_t* sel_ids;
    int32_t* full_slot;   // [VF]: row of full-mesh vertex v in this subset, or -1 (NULL for a full model)
};

static inline int saga_roundup(int x, int m) { return (x + m - 1) / m * m; }
