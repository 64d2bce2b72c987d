// Glue of the two non-equivariant members of the reference's plugin surface (dec_type "foldingnet", enc_type "dgcnn_fps":
// /root/reference/models/pcn.py:274-317, models/dgcnn.py:19-161): nn.BatchNorm1d over the point-major rows [P, C] of a [B, C, N] tensor with
// the activation that follows it (ReLU / LeakyReLU / none) in one pass per direction.  Statistics are Welford partials per chunk of
// rows merged in double by vnp_vn_bn_finalize_chunked (which also updates the running statistics with torch's momentum / unbiased
// variance convention); the backward is two passes (per-channel sums, then the gradient), the pre-activation recomputed from x.
// Outside the measured hot path; same layout conventions as vn_ops.hip (64 channels x 4 row lanes per workgroup).
#include "vnp_common.h"

#define VNP_B1_CHUNK 1024          // rows per chunk of the statistics / sums passes

// Chan's pairwise merge of (count, mean, M2), as in vn_ops.hip
__device__ __forceinline__ void vnp_b1_chan_merge(double& na, double& ma, double& Ma, double nb, double mb, double Mb)
{
    if (nb <= 0.0) return;
    const double n = na + nb, delta = mb - ma;
    ma += delta * nb / n;
    Ma += Mb + delta * delta * na * nb / n;
    na = n;
}

__device__ __forceinline__ float vnp_b1_act(float y, int act, float slope) { return act == 1 ? fmaxf(y, 0.f) : act == 2 ? (y > 0.f ? y : y * slope) : y; }
__device__ __forceinline__ float vnp_b1_dact(float y, int act, float slope) { return act == 1 ? (y > 0.f ? 1.f : 0.f) : act == 2 ? (y > 0.f ? 1.f : slope) : 1.f; }

// partial[chunk][c] = (mean, M2) of x[rows of the chunk][c]
__global__ void __launch_bounds__(256) vnp_bn1d_stats_kernel(const float* __restrict__ x, int ldx, long long P, int C, float2* __restrict__ partial)
{
    __shared__ float rm[4][64], rM[4][64];
    __shared__ int rn[4];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const long long r0 = (long long)blockIdx.y * VNP_B1_CHUNK, r1 = min(P, r0 + VNP_B1_CHUNK);
    float mean = 0.f, m2 = 0.f; int n = 0;
    if (c < C) {
#pragma unroll 4
        for (long long r = r0 + py; r < r1; r += 4) {
            const float v = x[(size_t)r * ldx + c];
            ++n;
            const float d = v - mean;
            mean += d / (float)n;
            m2 += d * (v - mean);
        }
    }
    rm[py][cx] = mean; rM[py][cx] = m2;
    if (cx == 0) rn[py] = (int)((r1 - r0 - py + 3) / 4 > 0 ? (r1 - r0 - py + 3) / 4 : 0);
    __syncthreads();
    if (py == 0 && c < C) {
        double na = rn[0], ma = rm[0][cx], Ma = rM[0][cx];
#pragma unroll
        for (int q = 1; q < 4; ++q) if (rn[q] > 0) vnp_b1_chan_merge(na, ma, Ma, (double)rn[q], (double)rm[q][cx], (double)rM[q][cx]);
        partial[(size_t)blockIdx.y * C + c] = make_float2((float)ma, (float)Ma);
    }
}

// y = act((x - mean) rstd gamma + beta)
__global__ void __launch_bounds__(256) vnp_bn1d_act_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long long P, int C,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int act, float slope)
{
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    if (c >= C) return;
    const float mu = mean[c], sc = rstd[c] * gamma[c], be = beta[c];
    const long long r0 = (long long)blockIdx.y * VNP_B1_CHUNK, r1 = min(P, r0 + VNP_B1_CHUNK);
#pragma unroll 4
    for (long long r = r0 + py; r < r1; r += 4) {
        const float v = __builtin_nontemporal_load(x + (size_t)r * ldx + c);
        __builtin_nontemporal_store(vnp_b1_act((v - mu) * sc + be, act, slope), y + (size_t)r * ldy + c);
    }
}

// MODE 1: partial[chunk][c] = (sum g', sum g' xhat) with g' = gy act'(pre);  MODE 2: gx = gamma rstd (g' - m1 - xhat m2)  (m1 = m2 = 0 in
// eval mode: sums == nullptr)
template <int MODE>
__global__ void __launch_bounds__(256) vnp_bn1d_act_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gy, int ldgy,
                                                              float* __restrict__ gx, int ldgx, long long P, int C,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int act, float slope,
                                                              float2* __restrict__ partial, const float* __restrict__ sums, float inv_count)
{
    __shared__ double2 red[4][64];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const bool live = c < C;
    const int cc = live ? c : 0;
    const float mu = mean[cc], rs = rstd[cc], ga = gamma[cc], be = beta[cc];
    const float m1 = (MODE == 2 && sums) ? sums[2 * cc] * inv_count : 0.f, m2 = (MODE == 2 && sums) ? sums[2 * cc + 1] * inv_count : 0.f;
    const long long r0 = (long long)blockIdx.y * VNP_B1_CHUNK, r1 = min(P, r0 + VNP_B1_CHUNK);
    double s1 = 0.0, s2 = 0.0;
    if (live) {
#pragma unroll 4
        for (long long r = r0 + py; r < r1; r += 4) {
            const float v = __builtin_nontemporal_load(x + (size_t)r * ldx + c);
            const float g = __builtin_nontemporal_load(gy + (size_t)r * ldgy + c);
            const float xh = (v - mu) * rs;
            const float gp = g * vnp_b1_dact(xh * ga + be, act, slope);
            if (MODE == 1) { s1 += (double)gp; s2 += (double)(gp * xh); }
            else __builtin_nontemporal_store(ga * rs * (gp - m1 - xh * m2), gx + (size_t)r * ldgx + c);
        }
    }
    if (MODE == 1) {
        red[py][cx] = make_double2(s1, s2);
        __syncthreads();
        if (py == 0 && live) {
            const double2 a = red[0][cx], b = red[1][cx], e = red[2][cx], f = red[3][cx];
            partial[(size_t)blockIdx.y * C + c] = make_float2((float)((a.x + b.x) + (e.x + f.x)), (float)((a.y + b.y) + (e.y + f.y)));
        }
    }
}

extern "C" int vnp_bn1d_chunk_rows(void) { return VNP_B1_CHUNK; }

// mode 0: statistics partials [ceil(P / vnp_bn1d_chunk_rows())][C][2] (-> vnp_vn_bn_finalize_chunked);  1: y = act(bn(x));
// 2: backward sums partials (same shape; -> vnp_reduce_pairs);  3: gx (sums [C][2] from mode 2, or NULL in eval mode; count = P).
// act: 0 none, 1 ReLU, 2 LeakyReLU(slope).
extern "C" int vnp_bn1d_act(hipStream_t st, int mode, const float* x, int ldx, long long P, int C, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, int act, float slope, float* y, int ldy, const float* gy, int ldgy,
                            float* partial, const float* sums, long long count)
{
    if (!x || P <= 0 || C <= 0 || mode < 0 || mode > 3 || act < 0 || act > 2 || ldx < C) return VNP_EINVAL;
    if (mode >= 1 && (!mean || !rstd || !gamma || !beta)) return VNP_EINVAL;
    if ((mode == 0 || mode == 2) && !partial) return VNP_EINVAL;
    if ((mode == 1 || mode == 3) && (!y || ldy < C)) return VNP_EINVAL;
    if (mode >= 2 && (!gy || ldgy < C)) return VNP_EINVAL;
    const long long nchunk = (P + VNP_B1_CHUNK - 1) / VNP_B1_CHUNK;
    if (nchunk > 65535) return VNP_EINVAL;
    dim3 grid(vnp_cdiv(C, 64), (unsigned)nchunk), block(256);
    if (mode == 0) hipLaunchKernelGGL(vnp_bn1d_stats_kernel, grid, block, 0, st, x, ldx, P, C, (float2*)partial);
    else if (mode == 1) hipLaunchKernelGGL(vnp_bn1d_act_fwd_kernel, grid, block, 0, st, x, ldx, y, ldy, P, C, mean, rstd, gamma, beta, act, slope);
    else if (mode == 2) hipLaunchKernelGGL((vnp_bn1d_act_bwd_kernel<1>), grid, block, 0, st, x, ldx, gy, ldgy, (float*)nullptr, 0, P, C, mean, rstd, gamma, beta,
                                           act, slope, (float2*)partial, (const float*)nullptr, 0.f);
    else hipLaunchKernelGGL((vnp_bn1d_act_bwd_kernel<2>), grid, block, 0, st, x, ldx, gy, ldgy, y, ldy, P, C, mean, rstd, gamma, beta,
                            act, slope, (float2*)nullptr, sums, count > 0 ? 1.0f / (float)count : 0.f);
    return vnp_last_error();
}
