/* CPU restatement of the reference's approximate Earth Mover's Distance.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows /root/reference/extensions/earth_movers_distance/emd_kernel.cu line for line per sample:
 *   approxmatch    :25-158  (levels -4^j, j = 7..-1, then 0; integer n/m multipliers :29-35; 1e-9 guards :64,115;
 *                            fminf/fmaxf clamps :115-117,152; match laid out [m][n] :145)
 *   matchcost      :200-243, matchcostgrad2 :286-327, matchcostgrad1 :333-355
 * with expf() where the CUDA code uses __expf (fast-math, ~2 ulp): compare at ~1e-4 relative.
 *
 * PARITY UNPINNED: the reference has no test, golden vector or CPU equivalent for EMD, and emd_cuda
 * cannot be built here (needs nvcc); this file is a restatement only.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float d2(const float *a, const float *b)
{
    return (b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1]) + (b[2] - a[2]) * (b[2] - a[2]);
}

void emd_oracle_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match)
{
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < b; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * 3, *p2 = xyz2 + (size_t)i * m * 3;
        float *mt = match + (size_t)i * n * m;
        float *remainL = (float *)malloc(sizeof(float) * (n + m) * 2);
        float *remainR = remainL + n, *ratioL = remainL + n + m, *ratioR = remainL + n + m + n;
        const float multiL = n >= m ? 1.f : (float)(m / n), multiR = n >= m ? (float)(n / m) : 1.f;
        memset(mt, 0, sizeof(float) * (size_t)n * m);
        for (int k = 0; k < n; ++k) remainL[k] = multiL;
        for (int l = 0; l < m; ++l) remainR[l] = multiR;
        for (int j = 7; j >= -2; --j) {
            const float level = j == -2 ? 0.f : -powf(4.0f, (float)j);
            for (int k = 0; k < n; ++k) {
                float suml = 1e-9f;
                for (int l = 0; l < m; ++l) suml += expf(level * d2(p1 + k * 3, p2 + l * 3)) * remainR[l];
                ratioL[k] = remainL[k] / suml;
            }
            for (int l = 0; l < m; ++l) {
                float sumr = 0;
                for (int k = 0; k < n; ++k) sumr += expf(level * d2(p1 + k * 3, p2 + l * 3)) * ratioL[k];
                sumr *= remainR[l];
                const float consumption = fminf(remainR[l] / (sumr + 1e-9f), 1.0f);
                ratioR[l] = consumption * remainR[l];
                remainR[l] = fmaxf(0.0f, remainR[l] - sumr);
            }
            for (int k = 0; k < n; ++k) {
                float suml = 0;
                for (int l = 0; l < m; ++l) {
                    const float w = expf(level * d2(p1 + k * 3, p2 + l * 3)) * ratioL[k] * ratioR[l];
                    mt[(size_t)l * n + k] += w;
                    suml += w;
                }
                remainL[k] = fmaxf(0.0f, remainL[k] - suml);
            }
        }
        free(remainL);
    }
}

void emd_oracle_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *cost)
{
    for (int i = 0; i < b; ++i) {
        double s = 0;
        for (int k = 0; k < n; ++k)
            for (int l = 0; l < m; ++l)
                s += (double)d2(xyz1 + ((size_t)i * n + k) * 3, xyz2 + ((size_t)i * m + l) * 3) * match[(size_t)i * n * m + (size_t)l * n + k];
        cost[i] = (float)s;
    }
}

void emd_oracle_matchcost_backward(int b, int n, int m, const float *grad_cost, const float *xyz1, const float *xyz2,
                                   const float *match, float *grad1, float *grad2)
{
    for (int i = 0; i < b; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * 3, *p2 = xyz2 + (size_t)i * m * 3, *mt = match + (size_t)i * n * m;
        for (int k = 0; k < n; ++k) {
            double g[3] = {0, 0, 0};
            for (int l = 0; l < m; ++l)
                for (int t = 0; t < 3; ++t) g[t] += (double)(p1[k * 3 + t] - p2[l * 3 + t]) * mt[(size_t)l * n + k] * 2;
            for (int t = 0; t < 3; ++t) grad1[((size_t)i * n + k) * 3 + t] = (float)(g[t] * grad_cost[i]);
        }
        for (int l = 0; l < m; ++l) {
            double g[3] = {0, 0, 0};
            for (int k = 0; k < n; ++k)
                for (int t = 0; t < 3; ++t) g[t] += (double)(p2[l * 3 + t] - p1[k * 3 + t]) * mt[(size_t)l * n + k] * 2;
            for (int t = 0; t < 3; ++t) grad2[((size_t)i * m + l) * 3 + t] = (float)(g[t] * grad_cost[i]);
        }
    }
}
