/* CPU restatement of the chamfer_distance extension's nearest-neighbour kernel (NmDistanceKernel of
 * github otaheri/chamfer_distance; no commit pinned by the reference, README.md:59; source absent from
 * /root/reference, so PARITY UNPINNED).  TEST INFRASTRUCTURE ONLY -- built by __graft_entry__.build() into
 * oracle/lib/liboracle_nn.so and loaded by oracle/chamfer_ref.py; the product never links it.
 *
 * Contract (call site utils/train_helper.py:57-63): for every point of `q` the squared euclidean distance to, and the
 * int32 index of, its nearest neighbour in `db`; candidates are scanned in index order with a strict '<', so the
 * lowest index wins ties.  The squared distance is the extension's expression
 *     (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)
 * under the multiply-add contraction a CUDA compiler applies by default: fma(dz,dz, fma(dy,dy, dx*dx)).
 * Compile WITHOUT -ffast-math and with -ffp-contract=off so that only the explicit fmaf() calls fuse.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

void oracle_nn_scan_f32(const float* q, const float* db, float* dist, int32_t* idx, int N, int Pq, int Pd) {
    for (int n = 0; n < N; ++n) {
        const float* qn = q + (size_t)n * Pq * 3;
        const float* dn = db + (size_t)n * Pd * 3;
        for (int p = 0; p < Pq; ++p) {
            const float x = qn[p * 3], y = qn[p * 3 + 1], z = qn[p * 3 + 2];
            float best = INFINITY;
            int32_t bi = 0;
            for (int k = 0; k < Pd; ++k) {
                const float dx = x - dn[k * 3], dy = y - dn[k * 3 + 1], dz = z - dn[k * 3 + 2];
                const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                if (d < best) {
                    best = d;
                    bi = k;
                }
            }
            if (dist) dist[(size_t)n * Pq + p] = best;
            if (idx) idx[(size_t)n * Pq + p] = bi;
        }
    }
}

/* same scan in double precision (used when the oracle runs in float64: fusion is immaterial there, but keep the chain) */
void oracle_nn_scan_f64(const double* q, const double* db, double* dist, int32_t* idx, int N, int Pq, int Pd) {
    for (int n = 0; n < N; ++n) {
        const double* qn = q + (size_t)n * Pq * 3;
        const double* dn = db + (size_t)n * Pd * 3;
        for (int p = 0; p < Pq; ++p) {
            const double x = qn[p * 3], y = qn[p * 3 + 1], z = qn[p * 3 + 2];
            double best = INFINITY;
            int32_t bi = 0;
            for (int k = 0; k < Pd; ++k) {
                const double dx = x - dn[k * 3], dy = y - dn[k * 3 + 1], dz = z - dn[k * 3 + 2];
                const double d = fma(dz, dz, fma(dy, dy, dx * dx));
                if (d < best) {
                    best = d;
                    bi = k;
                }
            }
            if (dist) dist[(size_t)n * Pq + p] = best;
            if (idx) idx[(size_t)n * Pq + p] = bi;
        }
    }
}
