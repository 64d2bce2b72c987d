/* CPU restatement of the reference's Chamfer kernels.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows /root/reference/extensions/chamfer_distance/chamfer3D.cu:
 *   - forward  NmDistanceKernel      (chamfer3D.cu:12-134): for every point j of xyz1 the
 *     squared distance to, and the index of, its nearest point of xyz2; direct
 *     (x2-x1)^2 form in fp32, strict '<' compares so the FIRST minimum wins (:36,:46,:126).
 *   - backward NmDistanceGradKernel  (chamfer3D.cu:155-174): g = 2*grad_dist[j];
 *     grad_xyz1[j] += g*(x1-x2[idx]);  grad_xyz2[idx] -= g*(x1-x2[idx]).
 * The product d = x*x + y*y + z*z is written as fmaf(z,z,fmaf(y,y,x*x)), the contraction
 * nvcc (-fmad=true) and hipcc (-ffp-contract=fast) both apply to that expression, so the
 * HIP kernel and this file agree bit for bit.
 *
 * Parity status: pinned by tests/golden/chamfer_*.npz (outputs of the reference's own
 * pure-PyTorch equivalent, extensions/ChamferDistancePytorch/chamfer_python.py:18-39).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* one direction: for each of the n points of a, nearest of the m points of b */
void chamfer_oracle_nn(int batch, int n, const float *a, int m, const float *b,
                       float *dist, int32_t *idx)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < batch; ++i) {
        for (int j = 0; j < n; ++j) {
            const float x1 = a[((size_t)i * n + j) * 3 + 0];
            const float y1 = a[((size_t)i * n + j) * 3 + 1];
            const float z1 = a[((size_t)i * n + j) * 3 + 2];
            const float *q = b + (size_t)i * m * 3;
            float best = 0.f;
            int best_i = 0;
            for (int k = 0; k < m; ++k) {
                const float x2 = q[k * 3 + 0] - x1;
                const float y2 = q[k * 3 + 1] - y1;
                const float z2 = q[k * 3 + 2] - z1;
                const float d = fmaf(z2, z2, fmaf(y2, y2, x2 * x2));
                if (k == 0 || d < best) { best = d; best_i = k; }
            }
            dist[(size_t)i * n + j] = best;
            idx[(size_t)i * n + j] = best_i;
        }
    }
}

void chamfer_oracle_forward(int batch, int n, const float *xyz1, int m, const float *xyz2,
                            float *dist1, int32_t *idx1, float *dist2, int32_t *idx2)
{
    chamfer_oracle_nn(batch, n, xyz1, m, xyz2, dist1, idx1);
    chamfer_oracle_nn(batch, m, xyz2, n, xyz1, dist2, idx2);
}

static void grad_one(int batch, int n, const float *xyz1, int m, const float *xyz2,
                     const float *grad_dist1, const int32_t *idx1, float *grad_xyz1, float *grad_xyz2)
{
    for (int i = 0; i < batch; ++i) {
        for (int j = 0; j < n; ++j) {
            const size_t p = ((size_t)i * n + j) * 3;
            const int j2 = idx1[(size_t)i * n + j];
            const size_t q = ((size_t)i * m + j2) * 3;
            const float g = grad_dist1[(size_t)i * n + j] * 2.f;
            for (int t = 0; t < 3; ++t) {
                const float v = g * (xyz1[p + t] - xyz2[q + t]);
                grad_xyz1[p + t] += v;
                grad_xyz2[q + t] -= v;
            }
        }
    }
}

/* grad_xyz1 / grad_xyz2 are zeroed here (the reference zeroes them in Python,
 * chamfer_distance.py:63-64) */
void chamfer_oracle_backward(int batch, int n, const float *xyz1, int m, const float *xyz2,
                             const float *grad_dist1, const int32_t *idx1,
                             const float *grad_dist2, const int32_t *idx2,
                             float *grad_xyz1, float *grad_xyz2)
{
    memset(grad_xyz1, 0, sizeof(float) * (size_t)batch * n * 3);
    memset(grad_xyz2, 0, sizeof(float) * (size_t)batch * m * 3);
    grad_one(batch, n, xyz1, m, xyz2, grad_dist1, idx1, grad_xyz1, grad_xyz2);
    grad_one(batch, m, xyz2, n, xyz1, grad_dist2, idx2, grad_xyz2, grad_xyz1);
}
