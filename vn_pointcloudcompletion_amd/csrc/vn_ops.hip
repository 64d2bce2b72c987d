// Vector-Neuron pointwise / reduction kernels on the rows layout (vnp_common.h):
//   norm statistics + BatchNorm-over-norms + leaky projection (models/vn_layers.py:25-43, 46-74, 107-127),
//   arg-max pooling (models/vn_layers.py:153-167), the FoldingNet rank-2 expansion (models/pcn.py:375-385)
//   and the small column / row reductions their backward passes need.
// All of them are HBM-bound: one thread owns one (point, channel) pair, a wave reads 64 consecutive
// channels of a row (256 contiguous bytes) for each of the 3 Cartesian components.
#include "vnp_common.h"

// point-loop unrolling of the streaming BatchNorm / leaky passes.  Measured with 2 (loads of two points in flight before the
// first store): decoder-sized maps bwd1 1588 -> 1627 us, bwd2 878 -> 882, fwd 501 -> 469 — neutral, so 1.  These passes already
// sit at the chip's mixed read/write ceiling (scripts/hbm_yardstick.hip: 5.5-5.8 TB/s for 2R1W / 3R2W streams, 6.4-6.9 read-only).
#ifndef VNP_PT_UNROLL_N
#define VNP_PT_UNROLL_N 1
#endif
#define VNP_STR2(x) #x
#define VNP_STR(x) VNP_STR2(x)
#define VNP_PT_UNROLL _Pragma(VNP_STR(unroll VNP_PT_UNROLL_N))
#define VNP_CHUNK 1024         // points (rows) per block in the chunked kernels (also exported via vnp_chunk_points)

extern "C" int vnp_chunk_points() { return VNP_CHUNK; }

// Points per workgroup of the BatchNorm / leaky passes for a [P points, C channels] map: VNP_CHUNK, halved (down to 64) until the
// grid has at least 1024 workgroups — a 128-channel layer over 65536 points is only 128 chunks of 1024 and left half the chip idle
// (measured: 1.0-1.5 TB/s on first_conv[0] against 5+ TB/s on the wide layers).  Statistics / gradient partials are per chunk of
// this size: callers size them with this function and pass the value on (vnp_vn_norm_stats, vnp_vn_bn_leaky_bwd1,
// vnp_vn_bn_finalize_chunked).
extern "C" int vnp_chunk_points_pc(long long P, int C)
{
    int chunk = VNP_CHUNK;
    const long long cols = (C + 63) / 64;
    while (chunk > 64 && ((P + chunk - 1) / chunk) * cols < 1024) chunk >>= 1;
    return chunk;
}

// ------------------------------------------------------------------------------------------------
// block = 256 threads = 64 channels x 4 point lanes; grid = (ceil(C/64), nchunk)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float vnp_sq3(float a, float b, float c) { return a * a + b * b + c * c; }

// partial[chunk][c] = (mean, M2) of n = |p| + EPS over the chunk's points (vn_layers.py:121).
// Welford per thread + Chan's pairwise merge: the batch variance of the norms can be 1e4-1e5 times smaller than
// mean^2 (FoldingNet layer 0: every point of a sample shares the dominant global term), where sum / sum-of-squares
// in fp32 loses the variance to cancellation.
__device__ __forceinline__ void vnp_chan_merge(double& na, double& ma, double& Ma, double nb, double mb, double Mb)
{
    if (nb <= 0.0) return;
    const double n = na + nb, delta = mb - ma;
    ma += delta * nb / n;
    Ma += Mb + delta * delta * na * nb / n;
    na = n;
}

__global__ void __launch_bounds__(256) vnp_norm_stats_kernel(const float* __restrict__ p, int ld, long long P, int C,
                                                            float2* __restrict__ partial, int chunk)
{
    __shared__ float2 red[4][64];
    __shared__ int cnt[4];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const long long p0 = (long long)blockIdx.y * chunk;
    const long long p1 = min(P, p0 + chunk);
    float mean = 0.f, m2 = 0.f;
    int k = 0;
    if (c < C) {
        long long pt = p0 + py;
        // four points in flight per thread (12 independent loads), then the four dependent Welford updates in the same order
        for (; pt + 12 < p1; pt += 16) {
            float nn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* r = p + (size_t)(pt + 4 * u) * 3 * ld + c;
                nn[u] = sqrtf(vnp_sq3(__builtin_nontemporal_load(r), __builtin_nontemporal_load(r + ld), __builtin_nontemporal_load(r + 2 * ld))) + VNP_EPS;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ++k;
                const float delta = nn[u] - mean;
                mean += delta / (float)k;
                m2 += delta * (nn[u] - mean);
            }
        }
        for (; pt < p1; pt += 4) {
            const float* r = p + (size_t)pt * 3 * ld + c;
            const float n = sqrtf(vnp_sq3(__builtin_nontemporal_load(r), __builtin_nontemporal_load(r + ld), __builtin_nontemporal_load(r + 2 * ld))) + VNP_EPS;
            ++k;
            const float delta = n - mean;
            mean += delta / (float)k;
            m2 += delta * (n - mean);
        }
    }
    red[py][cx] = make_float2(mean, m2);
    if (cx == 0) cnt[py] = (int)((p1 - p0 - py + 3) / 4 > 0 ? (p1 - p0 - py + 3) / 4 : 0);
    __syncthreads();
    if (py == 0 && c < C) {
        double na = cnt[0], ma = red[0][cx].x, Ma = red[0][cx].y;
#pragma unroll
        for (int j = 1; j < 4; ++j) vnp_chan_merge(na, ma, Ma, (double)cnt[j], (double)red[j][cx].x, (double)red[j][cx].y);
        partial[(size_t)blockIdx.y * C + c] = make_float2((float)ma, (float)Ma);
    }
}

// mean / rstd per channel from the chunk partials (Chan merge in double), running-stat update as torch does
// (momentum, unbiased variance for running_var).
// A workgroup = 64 channels x 16 chunk lanes: the chain of dependent double-precision merges per thread is nchunk / 16 long (C / 64
// workgroups is all the parallelism the channels offer: with 4 chunk lanes the 512-chunk decoder layer took 42 us).
#define VNP_FIN_LANES 16
__global__ void __launch_bounds__(64 * VNP_FIN_LANES) vnp_bn_finalize_kernel(const float2* __restrict__ partial, int nchunk, int chunk_points, int C,
                                                             long long count, float* __restrict__ mean, float* __restrict__ rstd,
                                                             float* running_mean, float* running_var, float momentum, float eps)
{
    __shared__ double rn[VNP_FIN_LANES][64], rm[VNP_FIN_LANES][64], rM[VNP_FIN_LANES][64];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double na = 0.0, ma = 0.0, Ma = 0.0;
    if (c < C) {
        for (int k = py; k < nchunk; k += VNP_FIN_LANES) {
            const float2 v = partial[(size_t)k * C + c];
            const long long left = count - (long long)k * chunk_points;
            const double nb = (double)(left < chunk_points ? left : chunk_points);
            if (na == 0.0) { na = nb; ma = v.x; Ma = v.y; }
            else vnp_chan_merge(na, ma, Ma, nb, (double)v.x, (double)v.y);
        }
    }
    rn[py][cx] = na; rm[py][cx] = ma; rM[py][cx] = Ma;
    __syncthreads();
    if (py != 0 || c >= C) return;
    for (int j = 1; j < VNP_FIN_LANES; ++j) {
        if (rn[j][cx] == 0.0) continue;
        if (na == 0.0) { na = rn[j][cx]; ma = rm[j][cx]; Ma = rM[j][cx]; }
        else vnp_chan_merge(na, ma, Ma, rn[j][cx], rm[j][cx], rM[j][cx]);
    }
    const double m = ma;
    double var = Ma / (double)count;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
}

__global__ void vnp_rstd_from_var_kernel(const float* __restrict__ var, float* __restrict__ rstd, int C, float eps)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) rstd[c] = 1.0f / sqrtf(var[c] + eps);
}

// out = leaky(bn(p), d).  d == nullptr: BatchNorm only.  mean == nullptr: no BatchNorm (plain VNLeakyReLU).
__global__ void __launch_bounds__(256) vnp_bn_leaky_fwd_kernel(const float* __restrict__ p, int ldp, const float* __restrict__ d, int ldd,
                                                              float* __restrict__ out, int ldo, long long P, int C,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float ns, int chunk)
{
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    if (c >= C) return;
    const long long p0 = (long long)blockIdx.y * chunk;
    const long long p1 = min(P, p0 + chunk);
    float mu = 0.f, rs = 0.f, ga = 0.f, be = 0.f;
    if (mean) { mu = mean[c]; rs = rstd[c]; ga = gamma[c]; be = beta[c]; }
    VNP_PT_UNROLL
    for (long long pt = p0 + py; pt < p1; pt += 4) {
        const float* r = p + (size_t)pt * 3 * ldp + c;
        float q0 = r[0], q1 = r[ldp], q2 = r[2 * ldp];
        if (mean) {
            const float n = sqrtf(vnp_sq3(q0, q1, q2)) + VNP_EPS;
            const float s = ((n - mu) * rs * ga + be) / n;
            q0 *= s; q1 *= s; q2 *= s;
        }
        if (d) {
            const float* rd = d + (size_t)pt * 3 * ldd + c;
            const float d0 = __builtin_nontemporal_load(rd), d1 = __builtin_nontemporal_load(rd + ldd), d2 = __builtin_nontemporal_load(rd + 2 * ldd);
            const float dot = q0 * d0 + q1 * d1 + q2 * d2;
            if (dot < 0.f) {
                const float a = (1.f - ns) * dot / (vnp_sq3(d0, d1, d2) + VNP_EPS);
                q0 -= a * d0; q1 -= a * d1; q2 -= a * d2;
            }
        }
        float* o = out + (size_t)pt * 3 * ldo + c;
        __builtin_nontemporal_store(q0, o); __builtin_nontemporal_store(q1, o + ldo); __builtin_nontemporal_store(q2, o + 2 * ldo);
    }
}

// (Measured: giving every lane 4 consecutive channels (dwordx4) does not speed these HBM-bound passes up: 1.71 -> 1.79 ms
// for bwd2.  Non-temporal loads and stores do: these tensors are streamed once per kernel and do not fit any cache —
// bwd1 3.05 -> 2.73 ms, bwd2 1.71 -> 1.50 ms, forward 0.81 -> 0.74 ms per step.)
// Backward, phase 1.  Upstream gradient go is either a matrix [P*3, ldgo] or (go == nullptr) the rank-1 form
// go[r][c] = go_row[r] * go_col[c] (gradient of the final VNLinear(C->1), models/pcn.py:346).
// Writes gq = dL/dq (q = bn(p)) and gd = dL/dd, and the per-chunk partial sums S1 = sum g_nhat, S2 = sum g_nhat*xhat
// (= dL/dbeta, dL/dgamma).  Without BatchNorm (mean == nullptr) gq is already dL/dp.
// MODE 0: as described above (gq = dL/dq, finished by vnp_bn_bwd2_kernel).  The two-pass form reads the inputs twice and never
// writes dL/dq:  MODE 1 = sums only (partial, gw_partial; nothing else is written);  MODE 2 = final gradients: with the reduced
// sums S ([C][2]; nullptr in eval mode) it applies phase 2 on the fly and writes gq = dL/dp and gd.  Against bwd1 + bwd2 this
// trades a written-then-re-read map for a second read of the inputs: 6 maps instead of 7 in the rank-1 form (2R + 2R2W against
// 2R2W + 2R1W), 8 = 8 in the matrix form but with 6 reads / 2 writes instead of 5 / 3 (reads stream faster: hbm_yardstick).
// gradients only: v_rcp_f32 / v_sqrt_f32 (1 ulp) instead of the ~12-instruction IEEE sequences (the rank-1 form was issue-bound)
#define BDIV(a, b) ((a) * __builtin_amdgcn_rcpf(b))
#define BSQRT(a) __builtin_amdgcn_sqrtf(a)
#define VNP_GRAD_HW 1
template <int MODE>
__global__ void __launch_bounds__(256) vnp_bn_leaky_bwd1_kernel(const float* __restrict__ go, int ldgo,
                                                               const float* __restrict__ go_row, const float* __restrict__ go_col,
                                                               const float* __restrict__ p, int ldp, const float* __restrict__ d, int ldd,
                                                               float* __restrict__ gq, int ldgq, float* __restrict__ gd, int ldgd,
                                                               long long P, int C,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, float ns,
                                                               float2* __restrict__ partial, float* __restrict__ gw_partial, int chunk,
                                                               const float* __restrict__ S, float inv_count, float* __restrict__ colsum)
{
    __shared__ double2 red[4][64];             // S1 / S2 are carried in double (as in vnp_fold_fused_kernel; free in an HBM-bound pass)
    __shared__ float red3[4][64];
    __shared__ float red6[4][6][64];
    float cs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // MODE 2 + colsum: sums over the chunk's points of (dL/dp, dL/dd) per component
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const long long p0 = (long long)blockIdx.y * chunk;
    const long long p1 = min(P, p0 + chunk);
    double s1 = 0.0, s2 = 0.0;
    float s3 = 0.f;
    if (c < C) {
        float mu = 0.f, rs = 0.f, ga = 0.f, be = 0.f, gc = 0.f;
        if (mean) { mu = mean[c]; rs = rstd[c]; ga = gamma[c]; be = beta[c]; }
        if (!go) gc = go_col[c];
        const float m1 = (MODE == 2 && S) ? S[2 * c] * inv_count : 0.f, m2 = (MODE == 2 && S) ? S[2 * c + 1] * inv_count : 0.f;
        VNP_PT_UNROLL
    for (long long pt = p0 + py; pt < p1; pt += 4) {
            const float* r = p + (size_t)pt * 3 * ldp + c;
            const float x0 = __builtin_nontemporal_load(r), x1 = __builtin_nontemporal_load(r + ldp), x2 = __builtin_nontemporal_load(r + 2 * ldp);
            float g0, g1, g2;
            if (go) {
                const float* rg = go + (size_t)pt * 3 * ldgo + c;
                g0 = __builtin_nontemporal_load(rg); g1 = __builtin_nontemporal_load(rg + ldgo); g2 = __builtin_nontemporal_load(rg + 2 * ldgo);
            } else {
                g0 = go_row[pt * 3] * gc; g1 = go_row[pt * 3 + 1] * gc; g2 = go_row[pt * 3 + 2] * gc;
            }
            float n = 1.f, s = 1.f, xh = 0.f;
            if (mean) {
                n = BSQRT(vnp_sq3(x0, x1, x2)) + VNP_EPS;
                xh = (n - mu) * rs;
                s = BDIV(xh * ga + be, n);
            }
            const float q0 = x0 * s, q1 = x1 * s, q2 = x2 * s;
            if (d) {
                const float* rd = d + (size_t)pt * 3 * ldd + c;
                const float d0 = __builtin_nontemporal_load(rd), d1 = __builtin_nontemporal_load(rd + ldd), d2 = __builtin_nontemporal_load(rd + 2 * ldd);
                const float dot = q0 * d0 + q1 * d1 + q2 * d2;
                float e0 = 0.f, e1 = 0.f, e2 = 0.f;
                if (MODE != 2 && gw_partial) {
                    // layer output (q or its projection) against the upstream gradient: d(loss)/d(w_out) of the fused
                    // final VNLinear(C -> 1)
                    float o0 = q0, o1 = q1, o2 = q2;
                    if (dot < 0.f) {
                        const float al = BDIV((1.f - ns) * dot, vnp_sq3(d0, d1, d2) + VNP_EPS);
                        o0 -= al * d0; o1 -= al * d1; o2 -= al * d2;
                    }
                    s3 += go_row[pt * 3] * o0 + go_row[pt * 3 + 1] * o1 + go_row[pt * 3 + 2] * o2;
                }
                if (dot < 0.f) {
                    const float cc = 1.f - ns;
                    const float inv = BDIV(1.f, vnp_sq3(d0, d1, d2) + VNP_EPS);
                    const float a = dot * inv;
                    const float gdd = g0 * d0 + g1 * d1 + g2 * d2;
                    const float k1 = cc * gdd * inv;
                    // dL/dd = -c [ a*go + gdd*(q*inv - 2 a inv d) ]
                    e0 = -cc * (a * g0 + gdd * (q0 * inv - 2.f * a * inv * d0));
                    e1 = -cc * (a * g1 + gdd * (q1 * inv - 2.f * a * inv * d1));
                    e2 = -cc * (a * g2 + gdd * (q2 * inv - 2.f * a * inv * d2));
                    g0 -= k1 * d0; g1 -= k1 * d1; g2 -= k1 * d2;
                }
                if (MODE != 1) {
                    float* wd = gd + (size_t)pt * 3 * ldgd + c;
                    __builtin_nontemporal_store(e0, wd); __builtin_nontemporal_store(e1, wd + ldgd); __builtin_nontemporal_store(e2, wd + 2 * ldgd);
                }
                if (MODE == 2 && colsum) { cs[3] += e0; cs[4] += e1; cs[5] += e2; }
            }
            if (MODE == 2 && mean) {
                // phase 2 on the fly (the arithmetic of vnp_bn_bwd2_kernel, term for term): g = dL/dq -> dL/dp
                const float pn = n - VNP_EPS;
                const float nh = xh * ga + be;
                const float gs = g0 * x0 + g1 * x1 + g2 * x2;
                const float gnh = BDIV(gs, n);
                const float gn = ga * rs * (gnh - m1 - xh * m2) - BDIV(gs * nh, n * n);
                const float k = pn > 0.f ? BDIV(gn, pn) : 0.f;
                g0 = s * g0 + k * x0; g1 = s * g1 + k * x1; g2 = s * g2 + k * x2;
            }
            if (MODE != 1) {
                float* wq = gq + (size_t)pt * 3 * ldgq + c;
                __builtin_nontemporal_store(g0, wq); __builtin_nontemporal_store(g1, wq + ldgq); __builtin_nontemporal_store(g2, wq + 2 * ldgq);
            }
            if (MODE == 2 && colsum) { cs[0] += g0; cs[1] += g1; cs[2] += g2; }
            if (MODE != 2 && mean) {
                const float gnh = BDIV(g0 * x0 + g1 * x1 + g2 * x2, n);     // dL/dnhat = (gq . p) / n
                s1 += (double)gnh; s2 += (double)(gnh * xh);
            }
        }
    }
    if (MODE == 2 && colsum) {
        // colsum[chunk][t][2C]: the per-sample bias gradient of a layer with de-duplicated broadcast channels is the sum of these
        // rows over the sample's chunks (no second pass over the 2C-wide gradient)
#pragma unroll
        for (int q = 0; q < 6; ++q) red6[py][q][cx] = cs[q];
        __syncthreads();
        if (py == 0 && c < C) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float* dst = colsum + ((size_t)blockIdx.y * 3 + t) * 2 * C;
                dst[c] = (red6[0][t][cx] + red6[1][t][cx]) + (red6[2][t][cx] + red6[3][t][cx]);
                dst[C + c] = (red6[0][3 + t][cx] + red6[1][3 + t][cx]) + (red6[2][3 + t][cx] + red6[3][3 + t][cx]);
            }
        }
    }
    if (MODE != 2 && (partial || gw_partial)) {
        red[py][cx] = make_double2(s1, s2);
        red3[py][cx] = s3;
        __syncthreads();
        if (py == 0 && c < C) {
            const double2 a = red[0][cx], b = red[1][cx], e = red[2][cx], f = red[3][cx];
            if (partial) partial[(size_t)blockIdx.y * C + c] = make_float2((float)((a.x + b.x) + (e.x + f.x)), (float)((a.y + b.y) + (e.y + f.y)));
            if (gw_partial) gw_partial[(size_t)blockIdx.y * C + c] = (red3[0][cx] + red3[1][cx]) + (red3[2][cx] + red3[3][cx]);
        }
    }
}
#undef BDIV
#undef BSQRT

// fine[pt*3+t] = sum_c leaky(bn(p), d)[pt,t,c] * w[c] + add[((pt / add_group))*3 + t]: the last VNLinearLeakyReLU of the
// decoder fused with the final VNLinear(C -> 1) and the residual point_feat (models/pcn.py:341-346,387): the [P*3, C]
// layer output is never written.  One wave per point (4 points per block), lanes stride over the channels.
__global__ void __launch_bounds__(256) vnp_bn_leaky_rowdot_kernel(const float* __restrict__ p, int ldp, const float* __restrict__ d, int ldd,
                                                                 long long P, int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float ns,
                                                                 const float* __restrict__ w, const float* __restrict__ add, int add_group,
                                                                 float* __restrict__ out)
{
    const long long pt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pt >= P) return;
    const float* rp = p + (size_t)pt * 3 * ldp;
    const float* rd = d + (size_t)pt * 3 * ldd;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        float q0 = __builtin_nontemporal_load(rp + c), q1 = __builtin_nontemporal_load(rp + ldp + c), q2 = __builtin_nontemporal_load(rp + 2 * ldp + c);
        const float n = sqrtf(vnp_sq3(q0, q1, q2)) + VNP_EPS;
        const float s = ((n - mean[c]) * rstd[c] * gamma[c] + beta[c]) / n;
        q0 *= s; q1 *= s; q2 *= s;
        const float d0 = __builtin_nontemporal_load(rd + c), d1 = __builtin_nontemporal_load(rd + ldd + c), d2 = __builtin_nontemporal_load(rd + 2 * ldd + c);
        const float dot = q0 * d0 + q1 * d1 + q2 * d2;
        if (dot < 0.f) {
            const float al = (1.f - ns) * dot / (vnp_sq3(d0, d1, d2) + VNP_EPS);
            q0 -= al * d0; q1 -= al * d1; q2 -= al * d2;
        }
        const float wc = w[c];
        a0 += q0 * wc; a1 += q1 * wc; a2 += q2 * wc;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
    if (lane == 0) {
        const float* ad = add ? add + (pt / add_group) * 3 : nullptr;
        out[pt * 3] = a0 + (ad ? ad[0] : 0.f); out[pt * 3 + 1] = a1 + (ad ? ad[1] : 0.f); out[pt * 3 + 2] = a2 + (ad ? ad[2] : 0.f);
    }
}

// Backward, phase 2 (BatchNorm only): g (in: dL/dq, out: dL/dp).
//   train: g_n = gamma*rstd*(g_nhat - S1/count - xhat*S2/count) - g_s*nhat/n^2     eval (S == nullptr): without the batch terms
//   g_p = s*g_q + g_n * p/|p|
__global__ void __launch_bounds__(256) vnp_bn_bwd2_kernel(float* __restrict__ g, int ldg, const float* __restrict__ p, int ldp,
                                                         long long P, int C,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ S, float inv_count, int chunk)
{
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    if (c >= C) return;
    const long long p0 = (long long)blockIdx.y * chunk;
    const long long p1 = min(P, p0 + chunk);
    const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    const float m1 = S ? S[2 * c] * inv_count : 0.f, m2 = S ? S[2 * c + 1] * inv_count : 0.f;
    VNP_PT_UNROLL
    for (long long pt = p0 + py; pt < p1; pt += 4) {
        const float* r = p + (size_t)pt * 3 * ldp + c;
        const float x0 = __builtin_nontemporal_load(r), x1 = __builtin_nontemporal_load(r + ldp), x2 = __builtin_nontemporal_load(r + 2 * ldp);
        float* rg = g + (size_t)pt * 3 * ldg + c;
        const float g0 = __builtin_nontemporal_load(rg), g1 = __builtin_nontemporal_load(rg + ldg), g2 = __builtin_nontemporal_load(rg + 2 * ldg);
        const float pn = sqrtf(vnp_sq3(x0, x1, x2));
        const float n = pn + VNP_EPS;
        const float xh = (n - mu) * rs;
        const float nh = xh * ga + be;
        const float s = nh / n;
        const float gs = g0 * x0 + g1 * x1 + g2 * x2;
        const float gnh = gs / n;
        const float gn = ga * rs * (gnh - m1 - xh * m2) - gs * nh / (n * n);
        const float k = pn > 0.f ? gn / pn : 0.f;
        __builtin_nontemporal_store(s * g0 + k * x0, rg); __builtin_nontemporal_store(s * g1 + k * x1, rg + ldg); __builtin_nontemporal_store(s * g2 + k * x2, rg + 2 * ldg);
    }
}

// out[b][j] = sum_{i<n} in[b*batch_stride + i*stride + j]   (double accumulation; small finalisation reductions)
// block = 64 columns x 16 row lanes (two loads in flight per lane), grid (ceil(W/64), nb): these are latency-bound, a lane's chain of
// dependent loads is n / 32 long
#define VNP_RR_LANES 16
__global__ void __launch_bounds__(64 * VNP_RR_LANES) vnp_reduce_rows_kernel(const float* __restrict__ in, long long batch_stride, int n, long long stride, int W,
                                                             float* __restrict__ out, long long out_stride, float* __restrict__ planar = nullptr)
{
    __shared__ double red[VNP_RR_LANES][64];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cx;
    double s = 0.0, s2 = 0.0;
    if (j < W) {
        const float* src = in + (size_t)blockIdx.y * batch_stride + j;
        int i = py;
        for (; i + VNP_RR_LANES < n; i += 2 * VNP_RR_LANES) { s += src[(size_t)i * stride]; s2 += src[(size_t)(i + VNP_RR_LANES) * stride]; }
        if (i < n) s += src[(size_t)i * stride];
    }
    red[py][cx] = s + s2;
    __syncthreads();
    if (py == 0 && j < W) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < VNP_RR_LANES; ++q) t += red[q][cx];
        out[(size_t)blockIdx.y * out_stride + j] = (float)t;
        if (planar) planar[(size_t)(j & 1) * (W >> 1) + (j >> 1)] = (float)t;        // vnp_reduce_pairs: the same sums as [2][W / 2]
    }
}

// ------------------------------------------------------------------------------------------------ pooling
// VNMaxPool (vn_layers.py:153-167): per group of gs consecutive points and channel c, argmax_n <x, d>, first max wins.
// Stage 1: grid (ceil(C/64), nchunk, G): best (value, local index) of each chunk of the group.
__global__ void __launch_bounds__(256) vnp_pool_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ d, int ldd,
                                                              int gs, int C, int nchunk, float* __restrict__ pval, int* __restrict__ pidx)
{
    __shared__ float rv[4][64];
    __shared__ int ri[4][64];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const long long g = blockIdx.z;
    const int n0 = blockIdx.y * VNP_CHUNK, n1 = min(gs, n0 + VNP_CHUNK);
    float best = -INFINITY; int bi = 0x7fffffff;
    if (c < C) {
        for (int n = n0 + py; n < n1; n += 4) {
            const size_t row = ((size_t)g * gs + n) * 3;
            const float* rx = x + row * ldx + c;
            const float* rd = d + row * ldd + c;
            const float dot = fmaf(rx[2 * ldx], rd[2 * ldd], fmaf(rx[ldx], rd[ldd], rx[0] * rd[0]));   // same form as the fused epilogue
            if (dot > best || bi == 0x7fffffff) { best = dot; bi = n; }
        }
    }
    rv[py][cx] = best; ri[py][cx] = bi;
    __syncthreads();
    if (py == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float v = rv[k][cx]; const int i = ri[k][cx];
            if (i != 0x7fffffff && (bi == 0x7fffffff || v > best || (v == best && i < bi))) { best = v; bi = i; }
        }
        pval[((size_t)g * nchunk + blockIdx.y) * C + c] = best;
        pidx[((size_t)g * nchunk + blockIdx.y) * C + c] = bi;
    }
}

// Stage 2: grid (ceil(C/256), G): reduce the chunk partials (ascending chunk order keeps the first maximum),
// write idx and gather the selected vectors.
__global__ void vnp_pool_final_kernel(const float* __restrict__ x, int ldx, int gs, int C, int nchunk,
                                      const float* __restrict__ pval, const int* __restrict__ pidx,
                                      int* __restrict__ idx, float* __restrict__ out, int ldo)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const long long g = blockIdx.y;
    float best = pval[(size_t)g * nchunk * C + c];
    int bi = pidx[(size_t)g * nchunk * C + c];
    for (int k = 1; k < nchunk; ++k) {
        const float v = pval[((size_t)g * nchunk + k) * C + c];
        if (v > best) { best = v; bi = pidx[((size_t)g * nchunk + k) * C + c]; }
    }
    idx[(size_t)g * C + c] = bi;
    const size_t row = ((size_t)g * gs + bi) * 3;
#pragma unroll
    for (int t = 0; t < 3; ++t) out[((size_t)g * 3 + t) * ldo + c] = x[(row + t) * ldx + c];
}

// gx (zero-filled beforehand) [(g*gs + idx[g][c])*3 + t][c] = gout[g*3+t][c]
__global__ void vnp_pool_bwd_kernel(const float* __restrict__ gout, int ldgo, const int* __restrict__ idx, int gs, int C,
                                    float* __restrict__ gx, int ldgx)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const long long g = blockIdx.y;
    const size_t row = ((size_t)g * gs + idx[(size_t)g * C + c]) * 3;
#pragma unroll
    for (int t = 0; t < 3; ++t) gx[(row + t) * ldgx + c] = gout[((size_t)g * 3 + t) * ldgo + c];
}

// gx [(g*gs + idx[g][c])*3 + t][c] += gout[g*3+t][c]: the pool's gradient added into an existing gradient of its input (each
// (row, channel) target has exactly one writer: no atomics)
__global__ void vnp_pool_bwd_acc_kernel(const float* __restrict__ gout, int ldgo, const int* __restrict__ idx, int gs, int C,
                                        float* __restrict__ gx, int ldgx)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const long long g = blockIdx.y;
    const size_t row = ((size_t)g * gs + idx[(size_t)g * C + c]) * 3;
#pragma unroll
    for (int t = 0; t < 3; ++t) gx[(row + t) * ldgx + c] += gout[((size_t)g * 3 + t) * ldgo + c];
}

// mean over groups of gs consecutive points (vn_layers.py:170-171 on the neighbour axis); grid (ceil(C/256), G*3)
__global__ void vnp_mean_pool_fwd_kernel(const float* __restrict__ x, int ldx, int gs, int C, float* __restrict__ out, int ldo)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const long long g = blockIdx.y / 3; const int t = blockIdx.y % 3;
    float s = 0.f;
    for (int n = 0; n < gs; ++n) s += x[(((size_t)g * gs + n) * 3 + t) * ldx + c];
    out[(size_t)blockIdx.y * ldo + c] = s / (float)gs;
}

__global__ void vnp_mean_pool_bwd_kernel(const float* __restrict__ gout, int ldgo, int gs, int C, float* __restrict__ gx, int ldgx)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const long long g = blockIdx.y / 3; const int t = blockIdx.y % 3;
    const float v = gout[(size_t)blockIdx.y * ldgo + c] / (float)gs;
    for (int n = 0; n < gs; ++n) gx[(((size_t)g * gs + n) * 3 + t) * ldgx + c] = v;
}

// ------------------------------------------------------------------------------------------------ folding
// Layer 1 of VN_FoldingNet with the broadcast input de-duplicated (models/pcn.py:375-387): of the latent+2 input
// channels, `latent` are the per-sample global feature, so W[:, :latent] g is a per-sample bias; what is left per
// point is the rank-2 term  w_seed * seed[s] + w_coarse * coarse[b, mc].
//   out[((b*Mc + mc)*S + s)*3 + t][j] = bias[b*3+t][j] + wseed[j]*seed[(b)][s][t] + wcoarse[j]*coarse[b][mc][t]
// grid (ceil(W/256), B*Mc): one block per coarse point, threads over the W (= 2*C: feat | dir) columns.
__global__ void __launch_bounds__(256) vnp_fold_fwd_kernel(const float* __restrict__ bias, int ldb, const float* __restrict__ wseed,
                                                          const float* __restrict__ wcoarse, const float* __restrict__ seed, int seed_batched,
                                                          const float* __restrict__ coarse, int Mc, int S, int W,
                                                          float* __restrict__ out, int ldo)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= W) return;
    const long long pc = blockIdx.y;             // b*Mc + mc
    const int b = (int)(pc / Mc);
    const float ws = wseed[j], wc = wcoarse[j];
    float base[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) base[t] = bias[((size_t)b * 3 + t) * ldb + j] + wc * coarse[pc * 3 + t];
    const float* sd = seed + (seed_batched ? (size_t)b * S * 3 : 0);
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
            out[(((size_t)pc * S + s) * 3 + t) * ldo + j] = base[t] + ws * sd[s * 3 + t];
    }
}

// Backward of the expansion: per coarse point partial sums over its S folded points, for column j:
//   part[pc][0..2][j] = sum_s g[(pc,s,t)][j]                (-> bias gradient after summing over mc)
//   part[pc][3][j]    = sum_{s,t} g * seed[s][t]            (-> wseed gradient)
//   part[pc][4][j]    = sum_{s,t} g * coarse[pc][t]         (-> wcoarse gradient)
// The caller reduces `part` over mc (per sample) / over everything with vnp_reduce_rows.
__global__ void __launch_bounds__(256) vnp_fold_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ seed, int seed_batched,
                                                          const float* __restrict__ coarse, int Mc, int S, int W, float* __restrict__ part)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= W) return;
    const long long pc = blockIdx.y;
    const int b = (int)(pc / Mc);
    const float* sd = seed + (seed_batched ? (size_t)b * S * 3 : 0);
    float a[3] = {0.f, 0.f, 0.f}, as = 0.f;
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float v = g[(((size_t)pc * S + s) * 3 + t) * ldg + j];
            a[t] += v; as += v * sd[s * 3 + t];
        }
    }
    float* o = part + (size_t)pc * 5 * W + j;
    o[0] = a[0]; o[W] = a[1]; o[2 * W] = a[2]; o[3 * W] = as;
    o[4 * W] = a[0] * coarse[pc * 3] + a[1] * coarse[pc * 3 + 1] + a[2] * coarse[pc * 3 + 2];
}

// ------------------------------------------------------------------------------------------------ row / column reductions
// out[r] = sum_c X[r][c] * w[c] (+ add[((r/3)/add_group)*3 + r%3]); one wave per row, 4 rows per block.
__global__ void __launch_bounds__(256) vnp_rowdot_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ w, long long R, int C,
                                                        const float* __restrict__ add, int add_group, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* row = X + (size_t)r * ldx;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += row[c] * w[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        if (add) s += add[((r / 3) / add_group) * 3 + r % 3];
        out[r] = s;
    }
}

// out[pc*3+t] (+)= sum_{s<S} in[((pc*S+s)*3+t)]
__global__ void vnp_group_sum3_kernel(const float* __restrict__ in, long long Pc, int S, float* __restrict__ out, int accumulate)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Pc * 3) return;
    const long long pc = i / 3; const int t = (int)(i % 3);
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += in[((size_t)pc * S + k) * 3 + t];
    out[i] = accumulate ? out[i] + s : s;
}

// partial[b][chunk][c] = sum over the chunk's rows r of X[b*batch_stride + r*row_stride + c] * (w ? w[b*R + r] : 1)
// grid (ceil(C/64), nchunk, nb); chunk = VNP_CHUNK rows.
__global__ void __launch_bounds__(256) vnp_colsum_kernel(const float* __restrict__ X, long long batch_stride, long long row_stride,
                                                        long long R, int C, const float* __restrict__ w, int nchunk, float* __restrict__ partial)
{
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const long long r0 = (long long)blockIdx.y * VNP_CHUNK, r1 = min(R, r0 + VNP_CHUNK);
    const float* base = X + (size_t)blockIdx.z * batch_stride;
    float s = 0.f;
    if (c < C) {
        long long r = r0 + py;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;                    // four rows in flight per thread
        for (; r + 12 < r1; r += 16) {
            const float x0 = __builtin_nontemporal_load(&base[(size_t)r * row_stride + c]), x1 = __builtin_nontemporal_load(&base[(size_t)(r + 4) * row_stride + c]);
            const float x2 = __builtin_nontemporal_load(&base[(size_t)(r + 8) * row_stride + c]), x3 = __builtin_nontemporal_load(&base[(size_t)(r + 12) * row_stride + c]);
            const float* ww = w ? w + (size_t)blockIdx.z * R + r : nullptr;
            t0 += x0 * (ww ? ww[0] : 1.f); t1 += x1 * (ww ? ww[4] : 1.f); t2 += x2 * (ww ? ww[8] : 1.f); t3 += x3 * (ww ? ww[12] : 1.f);
        }
        for (; r < r1; r += 4) t0 += __builtin_nontemporal_load(&base[(size_t)r * row_stride + c]) * (w ? w[(size_t)blockIdx.z * R + r] : 1.f);
        s = (t0 + t1) + (t2 + t3);
    }
    red[py][cx] = s;
    __syncthreads();
    if (py == 0 && c < C)
        partial[((size_t)blockIdx.z * nchunk + blockIdx.y) * C + c] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}

// out[i] = a[i] + b[i]  /  a[i] += b[i] style helpers are left to the GEMM's accumulate flag; this one converts layouts:
// rows layout [P*3, C] (ld) <-> reference layout [B, C, 3, N]: out_ref[((b*C + c)*3 + t)*N + n] = rows[((b*N+n)*3+t)*ld + c]
__global__ void vnp_rows_to_ref_kernel(const float* __restrict__ rows, int ld, int B, int C, int N, float* __restrict__ ref)
{
    const long long total = (long long)B * C * 3 * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N); long long k = i / N;
        const int t = (int)(k % 3); k /= 3;
        const int c = (int)(k % C); const int b = (int)(k / C);
        ref[i] = rows[(((size_t)b * N + n) * 3 + t) * ld + c];
    }
}

__global__ void vnp_ref_to_rows_kernel(const float* __restrict__ ref, int B, int C, int N, float* __restrict__ rows, int ld)
{
    const long long total = (long long)B * N * 3 * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); long long k = i / C;
        const int t = (int)(k % 3); k /= 3;
        const int n = (int)(k % N); const int b = (int)(k / N);
        rows[(((size_t)b * N + n) * 3 + t) * ld + c] = ref[(((size_t)b * C + c) * 3 + t) * N + n];
    }
}


// out[r][c] = u[r] * w[c]   (gradient of the final VNLinear(C->1) w.r.t. its input)
__global__ void vnp_outer_kernel(const float* __restrict__ u, const float* __restrict__ w, long long R, int C, float* __restrict__ out, int ldo)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    if (c >= C) return;
    const float wc = w[c];
    const long long r0 = (long long)blockIdx.y * 64;
    for (long long r = r0 + (threadIdx.x >> 6); r < min(R, r0 + 64); r += 4) out[(size_t)r * ldo + c] = u[r] * wc;
}

extern "C" int vnp_outer(hipStream_t st, const float* u, const float* w, long long R, int C, float* out, int ldo)
{
    if (!u || !w || !out || R <= 0 || C <= 0) return VNP_EINVAL;
    if ((R + 63) / 64 > 65535LL * 64) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_outer_kernel, dim3(vnp_cdiv(C, 64), vnp_cdiv(R, 64)), dim3(256), 0, st, u, w, R, C, out, ldo);
    return vnp_last_error();
}


// ------------------------------------------------------------------------------------------------ sparse pool backward
// VNLinear followed by VNMaxPool (models/pcn.py:169-170, 173-174): y = x W^T, out[g,t,o] = y[(g, idx[g,o]), t, o].
// The gradient of y has ONE non-zero point per (group, channel), so instead of materialising it and running two dense
// GEMMs (25.8 GFLOP per sample each for 1024 -> 2048), both gradients are gathered directly:
//   dW[o][c]        = sum_{g,t} gout[g*3+t][o] * x[(g*gs + idx[g][o])*3 + t][c]
//   dx[(g,n)*3+t][c] = sum_{o : idx[g][o] == n} gout[g*3+t][o] * W[o][c]
// dW: grid (Co), threads over c.  Deterministic (fixed g,t order).
__global__ void __launch_bounds__(256) vnp_pool_lin_bwd_w_kernel(const float* __restrict__ gout, int ldgo, const int* __restrict__ idx,
                                                                const float* __restrict__ x, int ldx, int G, int gs, int Co, int Ci,
                                                                float* __restrict__ gW, int ldgw)
{
    const int o = blockIdx.x;
    for (int c0 = threadIdx.x * 4; c0 < Ci; c0 += 1024) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const bool vec = (c0 + 3 < Ci) && ((ldx & 3) == 0);
        for (int g = 0; g < G; ++g) {
            const size_t row = ((size_t)g * gs + idx[(size_t)g * Co + o]) * 3;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float w = gout[((size_t)g * 3 + t) * ldgo + o];
                const float* r = x + (row + t) * ldx + c0;
                if (vec) { const vnp_f4 v = *(const vnp_f4*)r; a0 += w * v[0]; a1 += w * v[1]; a2 += w * v[2]; a3 += w * v[3]; }
                else {
                    a0 += w * r[0];
                    if (c0 + 1 < Ci) a1 += w * r[1];
                    if (c0 + 2 < Ci) a2 += w * r[2];
                    if (c0 + 3 < Ci) a3 += w * r[3];
                }
            }
        }
        float* dst = gW + (size_t)o * ldgw + c0;
        dst[0] = a0;
        if (c0 + 1 < Ci) dst[1] = a1;
        if (c0 + 2 < Ci) dst[2] = a2;
        if (c0 + 3 < Ci) dst[3] = a3;
    }
}

// dx: `order`[g][j] lists the channels of group g sorted by their arg-max point, `pts`[g][j] that point (ascending).  Block (n, g) owns
// point n of group g: it finds n's run in `pts` by bisection and writes the point's 3 rows — the sum over the run, or zeros for a point no
// channel selected (rounds 1-3: a zero fill of the whole 805 MB gradient beforehand, then one block per run).  No atomics, every row
// written exactly once.
#define VNP_POOL_BWD_PTS 8          // consecutive points per block: one bisection, then a walk along the sorted list
__global__ void __launch_bounds__(256) vnp_pool_lin_bwd_x_kernel(const float* __restrict__ gout, int ldgo, const int* __restrict__ order,
                                                                const int* __restrict__ pts, const float* __restrict__ W, int ldw,
                                                                int gs, int Co, int Ci, float* __restrict__ gx, int ldgx, int vec4)
{
    const int n0 = blockIdx.x * VNP_POOL_BWD_PTS, g = blockIdx.y;
    const int n1 = n0 + VNP_POOL_BWD_PTS < gs ? n0 + VNP_POOL_BWD_PTS : gs;
    const int* pg = pts + (size_t)g * Co;
    int lo = 0, hi = Co;                              // first j with pg[j] >= n0
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (pg[mid] < n0) lo = mid + 1; else hi = mid; }
    const int* og = order + (size_t)g * Co;
    int j1 = lo;
    for (int n = n0; n < n1; ++n) {
        const int j0 = j1;
        while (j1 < Co && pg[j1] == n) ++j1;          // an empty run (j1 == j0) stores zeros below
        const size_t row = ((size_t)g * gs + n) * 3;
        if (vec4) {
            // four consecutive input channels per thread: one pass over Ci = 1024 instead of four dependent ones (the kernel is a chain
            // of dependent gathers: index -> weight row -> store).  (One point per WAVE instead of per block — four chains in flight —
            // measured slower: 411 against 243 us.)
            for (int c0 = threadIdx.x * 4; c0 < Ci; c0 += 1024) {
                vnp_f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
                for (int j = j0; j < j1; ++j) {
                    const int o = og[j];
                    const vnp_f4 w = *(const vnp_f4*)(W + (size_t)o * ldw + c0);
                    const float g0 = gout[((size_t)g * 3 + 0) * ldgo + o], g1 = gout[((size_t)g * 3 + 1) * ldgo + o], g2 = gout[((size_t)g * 3 + 2) * ldgo + o];
                    a0 += g0 * w; a1 += g1 * w; a2 += g2 * w;
                }
                *(vnp_f4*)(gx + (row + 0) * ldgx + c0) = a0;
                *(vnp_f4*)(gx + (row + 1) * ldgx + c0) = a1;
                *(vnp_f4*)(gx + (row + 2) * ldgx + c0) = a2;
            }
            continue;
        }
        for (int c0 = threadIdx.x; c0 < Ci; c0 += 256) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int j = j0; j < j1; ++j) {
                const int o = og[j];
                const float w = W[(size_t)o * ldw + c0];
                a0 += gout[((size_t)g * 3 + 0) * ldgo + o] * w;
                a1 += gout[((size_t)g * 3 + 1) * ldgo + o] * w;
                a2 += gout[((size_t)g * 3 + 2) * ldgo + o] * w;
            }
            gx[(row + 0) * ldgx + c0] = a0;
            gx[(row + 1) * ldgx + c0] = a1;
            gx[(row + 2) * ldgx + c0] = a2;
        }
    }
}

extern "C" int vnp_pool_lin_bwd(hipStream_t st, const float* gout, int ldgo, const int* idx, const int* order, const int* pts,
                                const float* x, int ldx, const float* W, int ldw, int G, int gs, int Co, int Ci,
                                float* gW, int ldgw, float* gx, int ldgx)
{
    if (!gout || !idx || G <= 0 || gs <= 0 || Co <= 0 || Ci <= 0 || G > 65535) return VNP_EINVAL;
    if (gW) {
        if (!x) return VNP_EINVAL;
        hipLaunchKernelGGL(vnp_pool_lin_bwd_w_kernel, dim3(Co), dim3(256), 0, st, gout, ldgo, idx, x, ldx, G, gs, Co, Ci, gW, ldgw);
    }
    if (gx) {
        if (!order || !pts || !W) return VNP_EINVAL;
        const int vec4 = (Ci % 4 == 0 && ldw % 4 == 0 && ldgx % 4 == 0 && (uintptr_t)W % 16 == 0 && (uintptr_t)gx % 16 == 0) ? 1 : 0;
        hipLaunchKernelGGL(vnp_pool_lin_bwd_x_kernel, dim3(vnp_cdiv(gs, VNP_POOL_BWD_PTS), G), dim3(256), 0, st, gout, ldgo, order, pts, W, ldw, gs, Co, Ci, gx, ldgx, vec4);
    }
    return vnp_last_error();
}


// ------------------------------------------------------------------------------------------------ fused folding layer
// FoldingNet layer 0 end to end WITHOUT its [P*3, 2C] pre-activation tensor (3.2 GB at B = 32): p and d are recomputed
// from the per-sample bias and the rank-2 per-point term wherever they are needed
//   p[t][j] = bias[b*3+t][j]   + wseed[j]  *seed[(b)][s][t] + wcoarse[j]  *coarse[b][mc][t]        (j <  C)
//   d[t][j] = bias[b*3+t][C+j] + wseed[C+j]*seed[(b)][s][t] + wcoarse[C+j]*coarse[b][mc][t]
// MODE 0: Welford statistics of |p|+eps per chunk (-> vnp_vn_bn_finalize)          reads nothing big, writes nothing big
// MODE 1: out = leaky(bn(p), d)                                                     writes out
// MODE 2: backward sums S1 = sum g_nhat, S2 = sum g_nhat*xhat per chunk             reads go
// MODE 3: backward: gp, gd on the fly -> partial sums for dbias / dwseed / dwcoarse per chunk and dcoarse (atomics)
// block = 64 columns x 4 coarse-point lanes; a chunk = VNP_CHUNK points = VNP_CHUNK/S coarse points.
struct VnpFoldArgs {
    const float* bias; int ldb;
    const float* wseed; const float* wcoarse; const float* seed; int seed_batched;
    const float* coarse; int Mc, S, C; long long ncoarse;      // ncoarse = B*Mc
    int cpc;                              // coarse points per chunk (divides Mc: a chunk never straddles two samples)
    const float* mean; const float* rstd; const float* gamma; const float* beta; float ns;
    float* out; int ldo;                  // MODE 1
    const float* go; int ldgo;            // MODE 2, 3
    float2* partial;                      // MODE 0: (mean, M2); MODE 2: (S1, S2)   [nchunk][C]
    const float* sums; float inv_count;   // MODE 3: reduced sums [C][2] (nullptr in eval mode)
    float* part5;                         // MODE 3: [nchunk][5][2C]
    float* gcoarse;                       // MODE 3: [ceil(C / 64)][B*Mc][3]: one partial per 64-column block, summed by the caller (no atomics)
    // GRP (vnp_group_fold_fused): bias rows per COARSE point (group) instead of per sample, seed_batched = 2: seed points per coarse
    // point; coarse == nullptr: no coarse term
    float* gbias_cp;                      // MODE 3, GRP: dL/dbias per coarse point [B*Mc*3][2C]
    float* gx_part;                       // MODE 3, GRP: dL/dseed per point, one partial per column block [C/64][B*Mc][S*3] (or nullptr)
};

// two-lane helpers of the packed backward body (PK): one thread evaluates two seed points per step on v_pk_*_f32
__device__ __forceinline__ vnp_f2 vnp_sq3(vnp_f2 a, vnp_f2 b, vnp_f2 c) { return a * a + b * b + c * c; }
__device__ __forceinline__ vnp_f2 vnp_rcp2(vnp_f2 v) { return VNP_GRAD_HW ? vnp_f2{__builtin_amdgcn_rcpf(v[0]), __builtin_amdgcn_rcpf(v[1])} : vnp_f2{1.f / v[0], 1.f / v[1]}; }
__device__ __forceinline__ vnp_f2 vnp_sqrt2(vnp_f2 v) { return VNP_GRAD_HW ? vnp_f2{__builtin_amdgcn_sqrtf(v[0]), __builtin_amdgcn_sqrtf(v[1])} : vnp_f2{sqrtf(v[0]), sqrtf(v[1])}; }

// PK (backward modes, even S): the loop over a coarse point's S seed points takes two per step with packed fp32 arithmetic — the same
// formulas lane by lane; only the order in which the per-thread sums collect their terms changes (even and odd seed points separately)
#define VNP_GRP_MAXS 16          // GRP, MODE 3 with gx_part: seed points per coarse point (LDS transpose of S * 3 values x 64 columns per wave)
template <int MODE, bool PK = false, bool GRP = false>
__global__ void __launch_bounds__(256) vnp_fold_fused_kernel(const VnpFoldArgs a)
{
    static_assert(!PK || MODE >= 2, "the packed body exists for the backward modes");
    static_assert(!GRP || MODE < 2 || PK, "the per-group form of the backward modes exists in the packed body only (even S)");
    // GRP, MODE 3: dL/dseed[point][t] = sum over ALL columns of (dL/dp w_f + dL/dd w_d): every lane parks its column's S * 3 terms in LDS
    // (wave-private rows), then lane i < S * 3 sums row i over the 64 columns: a transpose-reduce for ~130 instructions per coarse
    // point instead of 6 shuffle steps per value (~600)
    __shared__ float xpose[(GRP && MODE == 3) ? 4 : 1][(GRP && MODE == 3) ? VNP_GRP_MAXS * 3 : 1][(GRP && MODE == 3) ? 65 : 1];
    // the backward modes (gradients) use v_rcp_f32 / v_sqrt_f32 (1 ulp each) where the forward modes keep IEEE division and sqrtf so that statistics and
    // outputs stay bit-identical to the unfused path
#define FDIV(a, b) ((MODE >= 2 && VNP_GRAD_HW) ? (a) * __builtin_amdgcn_rcpf(b) : (a) / (b))
#define FSQRT(a) ((MODE >= 2 && VNP_GRAD_HW) ? __builtin_amdgcn_sqrtf(a) : sqrtf(a))
    __shared__ float red[4][2][64];                               // MODE 0
    // backward sums are carried in double from one coarse point to the next (the 16 seed points of a coarse point are summed in fp32):
    // S1 / S2 enter every row's gradient through m1, m2 times gamma * rstd — rstd is 10^2..10^3 here, the norms of a sample share their
    // dominant term — as an error common to all rows; ten conversions and adds per coarse point cost nothing next to ~2400 VALU ops
    __shared__ double redd[MODE >= 2 ? 4 : 1][5][2][64];
    __shared__ int cnt[4];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cx;
    const bool live = j < a.C;
    const int C = a.C, S = a.S;
    const int cpc = a.cpc;                                         // coarse points per chunk
    const long long cp0 = (long long)blockIdx.y * cpc;
    const long long cp1 = min(a.ncoarse, cp0 + cpc);
    float wsf = 0, wcf = 0, wsd = 0, wcd = 0, mu = 0, rs = 0, ga = 0, be = 0;
    if (live) {
        wsf = a.wseed[j]; wsd = a.wseed[C + j];
        if (!GRP || a.wcoarse) { wcf = a.wcoarse[j]; wcd = a.wcoarse[C + j]; }
        if (a.mean) { mu = a.mean[j]; rs = a.rstd[j]; ga = a.gamma[j]; be = a.beta[j]; }
    }
    const float m1 = (MODE == 3 && a.sums) ? a.sums[2 * (live ? j : 0)] * a.inv_count : 0.f;
    const float m2 = (MODE == 3 && a.sums) ? a.sums[2 * (live ? j : 0) + 1] * a.inv_count : 0.f;
    float wmean = 0.f, wm2 = 0.f; int wk = 0;                      // MODE 0
    double s1 = 0.0, s2 = 0.0;                                     // MODE 2
    double acc[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};         // MODE 3: [feat|dir][bias t0..2, seed, coarse]
    for (long long cp = cp0 + py; cp < cp1; cp += 4) {
        const int b = (int)(cp / a.Mc);
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (!GRP || a.coarse) { c0 = a.coarse[cp * 3]; c1 = a.coarse[cp * 3 + 1]; c2 = a.coarse[cp * 3 + 2]; }
        float pf[3] = {0, 0, 0}, pdv[3] = {0, 0, 0};
        if (live) {
            const float* bb = a.bias + (size_t)(GRP ? cp : (long long)b) * 3 * a.ldb;
            pf[0] = bb[j] + wcf * c0; pf[1] = bb[a.ldb + j] + wcf * c1; pf[2] = bb[2 * a.ldb + j] + wcf * c2;
            pdv[0] = bb[C + j] + wcd * c0; pdv[1] = bb[a.ldb + C + j] + wcd * c1; pdv[2] = bb[2 * a.ldb + C + j] + wcd * c2;
        }
        const float* sd = a.seed + (a.seed_batched == 2 ? (size_t)cp * S * 3 : a.seed_batched ? (size_t)b * S * 3 : 0);
        float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f;                     // MODE 3: dcoarse contributions of this column
        // (measured: `#pragma unroll 4` here — more gradient loads in flight — costs registers instead: mode 3 0.99 -> 1.16 ms,
        // mode 2 0.52 -> 0.69 ms)
        if (PK) {
            vnp_f2 vs1 = {0.f, 0.f}, vs2 = {0.f, 0.f}, vgc0 = {0.f, 0.f}, vgc1 = {0.f, 0.f}, vgc2 = {0.f, 0.f};
            vnp_f2 vacc[2][5];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 5; ++r) vacc[h][r] = vnp_f2{0.f, 0.f};
            for (int s = 0; s < S; s += 2) {
                const vnp_f2 e0 = {sd[s * 3], sd[s * 3 + 3]}, e1 = {sd[s * 3 + 1], sd[s * 3 + 4]}, e2 = {sd[s * 3 + 2], sd[s * 3 + 5]};
                const vnp_f2 x0 = pf[0] + wsf * e0, x1 = pf[1] + wsf * e1, x2 = pf[2] + wsf * e2;
                const vnp_f2 pn = vnp_sqrt2(vnp_sq3(x0, x1, x2)), n = pn + VNP_EPS, rn = vnp_rcp2(n);
                const vnp_f2 xh = (n - mu) * rs, nh = xh * ga + be, sc = nh * rn;
                const vnp_f2 q0 = x0 * sc, q1 = x1 * sc, q2 = x2 * sc;
                const vnp_f2 d0 = pdv[0] + wsd * e0, d1 = pdv[1] + wsd * e1, d2 = pdv[2] + wsd * e2;
                const vnp_f2 dot = q0 * d0 + q1 * d1 + q2 * d2;
                const size_t row = ((size_t)cp * S + s) * 3;
                vnp_f2 g0 = {0.f, 0.f}, g1 = {0.f, 0.f}, g2 = {0.f, 0.f};
                if (live) {
                    const float* gp = a.go + row * a.ldgo + j;
                    g0 = vnp_f2{__builtin_nontemporal_load(gp), __builtin_nontemporal_load(gp + 3 * (size_t)a.ldgo)};
                    g1 = vnp_f2{__builtin_nontemporal_load(gp + a.ldgo), __builtin_nontemporal_load(gp + 4 * (size_t)a.ldgo)};
                    g2 = vnp_f2{__builtin_nontemporal_load(gp + 2 * (size_t)a.ldgo), __builtin_nontemporal_load(gp + 5 * (size_t)a.ldgo)};
                }
                // the leaky branch, evaluated for both lanes and selected lane by lane
                const float cc = 1.f - a.ns;
                const vnp_f2 inv = vnp_rcp2(vnp_sq3(d0, d1, d2) + VNP_EPS);
                const vnp_f2 al = dot * inv;
                const vnp_f2 gdd = g0 * d0 + g1 * d1 + g2 * d2;
                const vnp_f2 k1 = cc * gdd * inv;
                vnp_f2 h0 = -cc * (al * g0 + gdd * (q0 * inv - 2.f * al * inv * d0));
                vnp_f2 h1 = -cc * (al * g1 + gdd * (q1 * inv - 2.f * al * inv * d1));
                vnp_f2 h2 = -cc * (al * g2 + gdd * (q2 * inv - 2.f * al * inv * d2));
                const vnp_f2 t0 = g0 - k1 * d0, t1 = g1 - k1 * d1, t2 = g2 - k1 * d2;
#pragma unroll
                for (int l = 0; l < 2; ++l) {
                    const bool neg = dot[l] < 0.f;
                    h0[l] = neg ? h0[l] : 0.f; h1[l] = neg ? h1[l] : 0.f; h2[l] = neg ? h2[l] : 0.f;
                    g0[l] = neg ? t0[l] : g0[l]; g1[l] = neg ? t1[l] : g1[l]; g2[l] = neg ? t2[l] : g2[l];      // now dL/dq
                }
                const vnp_f2 gs = g0 * x0 + g1 * x1 + g2 * x2;
                const vnp_f2 gnh = gs * rn;
                if (MODE == 2) { vs1 += gnh; vs2 += gnh * xh; continue; }
                const vnp_f2 gn = ga * rs * (gnh - m1 - xh * m2) - gs * nh * vnp_rcp2(n * n);
                vnp_f2 kk = gn * vnp_rcp2(pn);
                kk[0] = pn[0] > 0.f ? kk[0] : 0.f; kk[1] = pn[1] > 0.f ? kk[1] : 0.f;
                const vnp_f2 r0 = sc * g0 + kk * x0, r1 = sc * g1 + kk * x1, r2 = sc * g2 + kk * x2;
                vacc[0][0] += r0; vacc[0][1] += r1; vacc[0][2] += r2;
                vacc[0][3] += r0 * e0 + r1 * e1 + r2 * e2;
                vacc[0][4] += r0 * c0 + r1 * c1 + r2 * c2;
                vacc[1][0] += h0; vacc[1][1] += h1; vacc[1][2] += h2;
                vacc[1][3] += h0 * e0 + h1 * e1 + h2 * e2;
                vacc[1][4] += h0 * c0 + h1 * c1 + h2 * c2;
                vgc0 += r0 * wcf + h0 * wcd; vgc1 += r1 * wcf + h1 * wcd; vgc2 += r2 * wcf + h2 * wcd;
                if (GRP && MODE == 3 && a.gx_part) {
                    const vnp_f2 y0 = r0 * wsf + h0 * wsd, y1 = r1 * wsf + h1 * wsd, y2 = r2 * wsf + h2 * wsd;
                    xpose[py][s * 3 + 0][cx] = y0[0]; xpose[py][s * 3 + 1][cx] = y1[0]; xpose[py][s * 3 + 2][cx] = y2[0];
                    xpose[py][s * 3 + 3][cx] = y0[1]; xpose[py][s * 3 + 4][cx] = y1[1]; xpose[py][s * 3 + 5][cx] = y2[1];
                }
            }
            s1 += (double)vs1[0] + (double)vs1[1]; s2 += (double)vs2[0] + (double)vs2[1];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 5; ++r) acc[h][r] += (double)vacc[h][r][0] + (double)vacc[h][r][1];
            gc0 = vgc0[0] + vgc0[1]; gc1 = vgc1[0] + vgc1[1]; gc2 = vgc2[0] + vgc2[1];
            if (GRP && MODE == 3) {
                if (live) {                                        // dL/dbias of THIS coarse point (the bias is per group here)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            a.gbias_cp[((size_t)cp * 3 + t) * 2 * C + h * C + j] = vacc[h][t][0] + vacc[h][t][1];
                }
                if (a.gx_part) {
                    __builtin_amdgcn_wave_barrier();               // wave-private LDS rows: the wave's own writes above, in program order
                    if (cx < S * 3) {
                        float v = 0.f;
#pragma unroll 8
                        for (int q = 0; q < 64; ++q) v += xpose[py][cx][q];
                        a.gx_part[((size_t)blockIdx.x * a.ncoarse + cp) * (S * 3) + cx] = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        } else
        for (int s = 0; s < S; ++s) {
            const float e0 = sd[s * 3], e1 = sd[s * 3 + 1], e2 = sd[s * 3 + 2];
            const float x0 = pf[0] + wsf * e0, x1 = pf[1] + wsf * e1, x2 = pf[2] + wsf * e2;
            const float pn = FSQRT(vnp_sq3(x0, x1, x2)), n = pn + VNP_EPS;
            if (MODE == 0) {
                if (live) { ++wk; const float dl = n - wmean; wmean += dl / (float)wk; wm2 += dl * (n - wmean); }
                continue;
            }
            const float xh = (n - mu) * rs, nh = xh * ga + be, sc = FDIV(nh, n);
            const float q0 = x0 * sc, q1 = x1 * sc, q2 = x2 * sc;
            const float d0 = pdv[0] + wsd * e0, d1 = pdv[1] + wsd * e1, d2 = pdv[2] + wsd * e2;
            const float dot = q0 * d0 + q1 * d1 + q2 * d2;
            const size_t row = ((size_t)cp * S + s) * 3;
            if (MODE == 1) {
                float o0 = q0, o1 = q1, o2 = q2;
                if (dot < 0.f) {
                    const float al = (1.f - a.ns) * dot / (vnp_sq3(d0, d1, d2) + VNP_EPS);
                    o0 -= al * d0; o1 -= al * d1; o2 -= al * d2;
                }
                if (live) { __builtin_nontemporal_store(o0, &a.out[row * a.ldo + j]); __builtin_nontemporal_store(o1, &a.out[(row + 1) * a.ldo + j]); __builtin_nontemporal_store(o2, &a.out[(row + 2) * a.ldo + j]); }
                continue;
            }
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (live) { g0 = __builtin_nontemporal_load(&a.go[row * a.ldgo + j]); g1 = __builtin_nontemporal_load(&a.go[(row + 1) * a.ldgo + j]); g2 = __builtin_nontemporal_load(&a.go[(row + 2) * a.ldgo + j]); }
            float h0 = 0.f, h1 = 0.f, h2 = 0.f;                    // dL/dd
            if (dot < 0.f) {
                const float cc = 1.f - a.ns;
                const float inv = FDIV(1.f, vnp_sq3(d0, d1, d2) + VNP_EPS);
                const float al = dot * inv;
                const float gdd = g0 * d0 + g1 * d1 + g2 * d2;
                const float k1 = cc * gdd * inv;
                h0 = -cc * (al * g0 + gdd * (q0 * inv - 2.f * al * inv * d0));
                h1 = -cc * (al * g1 + gdd * (q1 * inv - 2.f * al * inv * d1));
                h2 = -cc * (al * g2 + gdd * (q2 * inv - 2.f * al * inv * d2));
                g0 -= k1 * d0; g1 -= k1 * d1; g2 -= k1 * d2;       // now dL/dq
            }
            const float gs = g0 * x0 + g1 * x1 + g2 * x2;
            const float gnh = FDIV(gs, n);
            if (MODE == 2) { s1 += gnh; s2 += gnh * xh; continue; }
            // MODE 3: dL/dp = sc*gq + gn * p/|p|
            const float gn = ga * rs * (gnh - m1 - xh * m2) - FDIV(gs * nh, n * n);
            const float kk = pn > 0.f ? FDIV(gn, pn) : 0.f;
            const float r0 = sc * g0 + kk * x0, r1 = sc * g1 + kk * x1, r2 = sc * g2 + kk * x2;
            acc[0][0] += r0; acc[0][1] += r1; acc[0][2] += r2;
            acc[0][3] += r0 * e0 + r1 * e1 + r2 * e2;
            acc[0][4] += r0 * c0 + r1 * c1 + r2 * c2;
            acc[1][0] += h0; acc[1][1] += h1; acc[1][2] += h2;
            acc[1][3] += h0 * e0 + h1 * e1 + h2 * e2;
            acc[1][4] += h0 * c0 + h1 * c1 + h2 * c2;
            gc0 += r0 * wcf + h0 * wcd; gc1 += r1 * wcf + h1 * wcd; gc2 += r2 * wcf + h2 * wcd;
        }
        if (MODE == 3 && !GRP) {
            // dcoarse[cp][t] = sum over columns: this wave's 64 here (a fixed butterfly), the column blocks' partials by the caller
            // (vnp_reduce_rows) — plain stores, the same bits on every run (rounds 1-3: float atomics across the column blocks)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { gc0 += __shfl_xor(gc0, o); gc1 += __shfl_xor(gc1, o); gc2 += __shfl_xor(gc2, o); }
            if (cx == 0) {
                float* w = a.gcoarse + ((size_t)blockIdx.x * a.ncoarse + cp) * 3;
                w[0] = gc0; w[1] = gc1; w[2] = gc2;
            }
        }
    }
    if (MODE == 0) {
        red[py][0][cx] = wmean; red[py][1][cx] = wm2;
        if (cx == 0) cnt[py] = (int)((cp1 - cp0 - py + 3) / 4 > 0 ? ((cp1 - cp0 - py + 3) / 4) * S : 0);
        __syncthreads();
        if (py == 0 && live) {
            double na = cnt[0], ma = red[0][0][cx], Ma = red[0][1][cx];
#pragma unroll
            for (int q = 1; q < 4; ++q) vnp_chan_merge(na, ma, Ma, (double)cnt[q], (double)red[q][0][cx], (double)red[q][1][cx]);
            a.partial[(size_t)blockIdx.y * C + j] = make_float2((float)ma, (float)Ma);
        }
    } else if (MODE == 2) {
        redd[py][0][0][cx] = s1; redd[py][1][0][cx] = s2;
        __syncthreads();
        if (py == 0 && live)
            a.partial[(size_t)blockIdx.y * C + j] = make_float2((float)((redd[0][0][0][cx] + redd[1][0][0][cx]) + (redd[2][0][0][cx] + redd[3][0][0][cx])),
                                                                (float)((redd[0][1][0][cx] + redd[1][1][0][cx]) + (redd[2][1][0][cx] + redd[3][1][0][cx])));
    } else if (MODE == 3) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 5; ++r) redd[py][r][h][cx] = acc[h][r];
        __syncthreads();
        if (py == 0 && live) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 5; ++r)
                    a.part5[((size_t)blockIdx.y * 5 + r) * 2 * C + h * C + j] =
                        (float)((redd[0][r][h][cx] + redd[1][r][h][cx]) + (redd[2][r][h][cx] + redd[3][r][h][cx]));
        }
    }
}
#undef FSQRT
#undef FDIV

// mode 0: statistics partials; 1: forward; 2: backward sums; 3: backward gradients (see ops.FoldBnLeakyFn).
// partial: [nchunk][C][2]; part5: [nchunk][5][2C] with nchunk = B*Mc / coarse_per_chunk; gcoarse [ceil(C / 64)][B*Mc][3]: per-column-block partials of dL/dcoarse (mode 3), to be summed over the first axis.
extern "C" int vnp_fold_fused(hipStream_t st, int mode, const float* bias, int ldb, const float* wseed, const float* wcoarse,
                              const float* seed, int seed_batched, const float* coarse, int B, int Mc, int S, int C,
                              const float* mean, const float* rstd, const float* gamma, const float* beta, float negative_slope,
                              float* out, int ldo, const float* go, int ldgo, float* partial, const float* sums, long long count,
                              float* part5, float* gcoarse, int coarse_per_chunk)
{
    if (!bias || !wseed || !wcoarse || !seed || !coarse || B <= 0 || Mc <= 0 || S <= 0 || C <= 0) return VNP_EINVAL;
    // chunk = coarse_per_chunk coarse points (x S dense points): must divide Mc and hold at most vnp_chunk_points() dense points;
    // 0 = vnp_chunk_points() / S (the historical layout)
    const int cpc = coarse_per_chunk > 0 ? coarse_per_chunk : (VNP_CHUNK % S == 0 ? VNP_CHUNK / S : 0);
    if (cpc <= 0 || Mc % cpc || (long long)cpc * S > VNP_CHUNK) return VNP_EINVAL;
    if (mode < 0 || mode > 3) return VNP_EINVAL;
    if (mode > 0 && (!mean || !rstd || !gamma || !beta)) return VNP_EINVAL;
    if ((mode == 0 || mode == 2) && !partial) return VNP_EINVAL;
    if (mode == 1 && !out) return VNP_EINVAL;
    if (mode >= 2 && !go) return VNP_EINVAL;
    if (mode == 3 && (!part5 || !gcoarse)) return VNP_EINVAL;
    VnpFoldArgs a;
    a.bias = bias; a.ldb = ldb; a.wseed = wseed; a.wcoarse = wcoarse; a.seed = seed; a.seed_batched = seed_batched;
    a.coarse = coarse; a.Mc = Mc; a.S = S; a.C = C; a.ncoarse = (long long)B * Mc;
    a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.ns = negative_slope;
    a.out = out; a.ldo = ldo; a.go = go; a.ldgo = ldgo; a.partial = (float2*)partial;
    a.sums = sums; a.inv_count = count > 0 ? 1.0f / (float)count : 0.f; a.part5 = part5; a.gcoarse = gcoarse;
    a.gbias_cp = nullptr; a.gx_part = nullptr;
    a.cpc = cpc;
    const int nchunk = vnp_cdiv((long long)B * Mc, cpc);
    if (nchunk > 65535) return VNP_EINVAL;
    dim3 grid(vnp_cdiv(C, 64), nchunk), block(256);
    if (mode == 0) hipLaunchKernelGGL((vnp_fold_fused_kernel<0>), grid, block, 0, st, a);
    else if (mode == 1) hipLaunchKernelGGL((vnp_fold_fused_kernel<1>), grid, block, 0, st, a);
    else if (mode == 2) {
        if (S % 2 == 0) hipLaunchKernelGGL((vnp_fold_fused_kernel<2, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((vnp_fold_fused_kernel<2>), grid, block, 0, st, a);
    } else {
        if (S % 2 == 0) hipLaunchKernelGGL((vnp_fold_fused_kernel<3, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((vnp_fold_fused_kernel<3>), grid, block, 0, st, a);
    }
    return vnp_last_error();
}

// The same four passes for a layer whose per-point input is ONE VN channel x [G][S][3] beside per-GROUP features (Attention_VN_FoldingNet's
// folding stages, models/pcn.py:486-488: cat([point, per-centre features]) -> VNLinearLeakyReLU(385, 256)):
//   p | d [point (g, s)][t][j] = bias[g*3 + t][j | C + j] + w[j | C + j] * x[g][s][t]
// with bias [G*3][2C] = the group features times their weight columns (a 49152-row GEMM) and w [2C] the weight column of the point
// channel.  Nothing of the [G*S*3, 2C] pre-activation tensor is written or read (it was an outer-product launch, a statistics pass, the
// BatchNorm / leaky forward and two backward passes over 1.6 GB each, a column-sum pass for the bias gradient and a matrix-vector
// pass for the input gradient).  mode 3 writes dL/dbias per group (gbias [G*3][2C]), the per-chunk partials of dL/dw (part5 row 3)
// and, when gx_part is given, dL/dx as one partial per 64-column block (gx_part [ceil(C/64)][G][S*3]; S <= 16): deterministic.
// S even; chunk = coarse_per_chunk groups (must divide G; 0 = vnp_chunk_points() / S).
extern "C" int vnp_group_fold_fused(hipStream_t st, int mode, const float* bias, int ldb, const float* w, const float* x, long long G, int S, int C,
                                    const float* mean, const float* rstd, const float* gamma, const float* beta, float negative_slope,
                                    float* out, int ldo, const float* go, int ldgo, float* partial, const float* sums, long long count,
                                    float* part5, float* gbias, float* gx_part, int coarse_per_chunk)
{
    if (!bias || !w || !x || G <= 0 || S <= 0 || C <= 0 || G > 2147483647LL) return VNP_EINVAL;
    const int cpc = coarse_per_chunk > 0 ? coarse_per_chunk : (VNP_CHUNK % S == 0 ? VNP_CHUNK / S : 0);
    if (cpc <= 0 || G % cpc || (long long)cpc * S > VNP_CHUNK) return VNP_EINVAL;
    if (mode < 0 || mode > 3) return VNP_EINVAL;
    if (mode > 0 && (!mean || !rstd || !gamma || !beta)) return VNP_EINVAL;
    if ((mode == 0 || mode == 2) && !partial) return VNP_EINVAL;
    if (mode == 1 && !out) return VNP_EINVAL;
    if (mode >= 2 && (!go || S % 2)) return VNP_EINVAL;
    if (mode == 3 && (!part5 || !gbias || (gx_part && S > VNP_GRP_MAXS))) return VNP_EINVAL;
    VnpFoldArgs a;
    a.bias = bias; a.ldb = ldb; a.wseed = w; a.wcoarse = nullptr; a.seed = x; a.seed_batched = 2;
    a.coarse = nullptr; a.Mc = (int)G; a.S = S; a.C = C; a.ncoarse = G;
    a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.ns = negative_slope;
    a.out = out; a.ldo = ldo; a.go = go; a.ldgo = ldgo; a.partial = (float2*)partial;
    a.sums = sums; a.inv_count = count > 0 ? 1.0f / (float)count : 0.f; a.part5 = part5; a.gcoarse = nullptr;
    a.gbias_cp = gbias; a.gx_part = gx_part;
    a.cpc = cpc;
    const long long nchunk = G / cpc;
    if (nchunk > 65535) return VNP_EINVAL;
    dim3 grid(vnp_cdiv(C, 64), (unsigned)nchunk), block(256);
    if (mode == 0) hipLaunchKernelGGL((vnp_fold_fused_kernel<0, false, true>), grid, block, 0, st, a);
    else if (mode == 1) hipLaunchKernelGGL((vnp_fold_fused_kernel<1, false, true>), grid, block, 0, st, a);
    else if (mode == 2) hipLaunchKernelGGL((vnp_fold_fused_kernel<2, true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((vnp_fold_fused_kernel<3, true, true>), grid, block, 0, st, a);
    return vnp_last_error();
}

// ------------------------------------------------------------------------------------------------ fused EdgeConv
// VNLinearLeakyReLU on the kNN edge tensor + mean over the k neighbours (models/dgcnn.py:250-278 + conv1/4/5/6, the
// VN-PoinTr grouper) WITHOUT the edge tensor.  After the layer's linear maps have been applied per point (u = x W1^T,
// v = x (W2 - W1)^T, columns [feat | dir]) an edge's pre-activations are  (p | d)(q, j) = u[nbr[q][j]] + v[q];  they are
// recomputed from the (L2-resident, k times smaller) per-point matrices in each of the four passes instead of being
// written once and re-read eight times:
//   MODE 0  norm statistics of p over all edges (Welford partial per block of 64 queries)
//   MODE 1  out[q] = 1/k sum_j leaky(bn(p), d)(q, j)
//   MODE 2  backward sums  S1 = sum g_nhat, S2 = sum g_nhat * xhat  (= dL/dbeta, dL/dgamma), upstream gradient go[q] / k per edge
//   MODE 3  backward:  gv[q] = sum_j (dL/dp | dL/dd)(q, j)   and, when gu is given,  gu[nbr[q][j]] += (dL/dp | dL/dd)(q, j)  (float atomics:
//           6 per (edge, channel), 2.8 ms of config 3's step at ~1.3 TB/s of added bytes)
//   gather  (vnp_edge_gather_u_kernel, the default for gu)  gu[s] = sum over the edges INTO s, listed by the reverse adjacency
//           (vnp_edge_reverse): the edge's gradient is recomputed from u[s] + v[q] and go[q] — the same arithmetic term for term — and
//           summed per source point in double (exact for addends within 2^29 of each other, so the result does not depend on the order
//           of a source's list), stored once: no atomics, no zero-fill, deterministic
// The per-edge arithmetic is that of vnp_bn_leaky_fwd / bwd1 / bwd2 above, term for term.
struct VnpEdgeArgs {
    const float* u; int ldu; const float* v; int ldv; const int* nbr;
    long long Pq; int k, C;
    const float* mean; const float* rstd; const float* gamma; const float* beta; float ns;
    float2* partial;                           // MODE 0 / 2 output: [nblock][C]
    float* out; int ldo;                       // MODE 1
    const float* go; int ldgo;                 // MODE 2 / 3
    const float* sums; float inv_count;        // MODE 3 ([C][2]; nullptr in eval mode)
    float* gu; int ldgu; float* gv; int ldgv;  // MODE 3
    // R1 kernels (a layer whose input is ONE VN channel — the point itself, conv1 of every encoder, models/dgcnn.py:291): u and v are
    // the points' coordinates [P*3] (ldu = ldv = 1) and the per-point matrices are rebuilt as x[t] * wu[c], x[t] * wv[c] (wu, wv [2C])
    const float* wu; const float* wv;
};
#define VNP_EDGE_QB 64                         // queries per block

// Workgroup w of a launch runs on XCD w % 8, each with its own 4 MB L2.  The EdgeConv passes gather rows of u / v / go that belong to the
// block's own cloud (a few hundred KB to a few MB per cloud and channel block), so the blocks of one cloud should share an L2: logical
// block L = (blocks of XCD 0, then of XCD 1, ...) — XCD x takes a contiguous run of T/8 logical blocks, i.e. of whole clouds — instead of
// the dispatch order that deals a cloud's blocks out to all eight.  Bijective for any T.
__device__ __forceinline__ void vnp_xcd_block(int& bx, int& by)
{
    const unsigned gx = gridDim.x, T = gx * gridDim.y, w = blockIdx.y * gx + blockIdx.x;
    const unsigned q8 = T >> 3, r8 = T & 7u, xcd = w & 7u, i = w >> 3;
    const unsigned L = xcd * q8 + (xcd < r8 ? xcd : r8) + i;
    bx = (int)(L % gx); by = (int)(L / gx);
}

// one rounding of the product and no contraction with the addition behind it: the bits of the stored u = x W^T (a K = 1 product)
__device__ __forceinline__ float vnp_mul_alone(float a, float b) { float p = a * b; asm volatile("" : "+v"(p)); return p; }

template <int MODE, bool R1 = false>
__global__ void __launch_bounds__(256) vnp_edge_fused_kernel(const VnpEdgeArgs a)
{
    // the backward modes use v_rcp_f32 / v_sqrt_f32 (1 ulp each) as the other gradient kernels do; the forward modes keep IEEE division
    // and sqrtf, so statistics and outputs stay bit-identical to the materialising path
#define EDIV(x, y) ((MODE >= 2 && VNP_GRAD_HW) ? (x) * __builtin_amdgcn_rcpf(y) : (x) / (y))
#define ESQRT(x) ((MODE >= 2 && VNP_GRAD_HW) ? __builtin_amdgcn_sqrtf(x) : sqrtf(x))
    // 256 threads = CW channel lanes x NQ query lanes: 64 x 4, or 32 x 8 for a layer of at most 32 channels (conv1 of the encoders), whose
    // second half of the channel lanes would otherwise idle through every edge
    __shared__ float2 red[256];
    __shared__ int cnt[8];
    const int C = a.C, k = a.k;
    const int CW = C <= 32 ? 32 : 64, NQ = 256 / CW;
    const int cx = threadIdx.x & (CW - 1), py = threadIdx.x / CW;
    int bx, by;
    vnp_xcd_block(bx, by);
    const int c = bx * CW + cx;
    const bool live = c < C;
    const int cc = live ? c : 0;
    const long long q0 = (long long)by * VNP_EDGE_QB;
    const long long q1 = min(a.Pq, q0 + VNP_EDGE_QB);
    float mu = 0.f, rs = 0.f, ga = 0.f, be = 0.f, m1 = 0.f, m2 = 0.f;
    if (MODE >= 1) { mu = a.mean[cc]; rs = a.rstd[cc]; ga = a.gamma[cc]; be = a.beta[cc]; }
    if (MODE == 3 && a.sums) { m1 = a.sums[2 * cc] * a.inv_count; m2 = a.sums[2 * cc + 1] * a.inv_count; }
    const float kf = (float)k, kinv = 1.f / kf;
    float wmean = 0.f, wm2 = 0.f, s1 = 0.f, s2 = 0.f;
    int wn = 0;
    float wup = 0.f, wud = 0.f, wvp = 0.f, wvd = 0.f;
    if (R1) { wup = a.wu[cc]; wud = a.wu[C + cc]; wvp = a.wv[cc]; wvd = a.wv[C + cc]; }
    for (long long q = q0 + py; q < q1; q += NQ) {
        float vp0, vp1, vp2, vd0, vd1, vd2;
        if (R1) {
            const float* xq = a.v + (size_t)q * 3;
            const float c0 = xq[0], c1 = xq[1], c2 = xq[2];
            vp0 = vnp_mul_alone(c0, wvp); vp1 = vnp_mul_alone(c1, wvp); vp2 = vnp_mul_alone(c2, wvp);
            vd0 = vnp_mul_alone(c0, wvd); vd1 = vnp_mul_alone(c1, wvd); vd2 = vnp_mul_alone(c2, wvd);
        } else {
            const float* rv = a.v + (size_t)q * 3 * a.ldv + cc;
            vp0 = rv[0]; vp1 = rv[a.ldv]; vp2 = rv[2 * a.ldv];
            vd0 = rv[C]; vd1 = rv[a.ldv + C]; vd2 = rv[2 * a.ldv + C];
        }
        float ge0 = 0.f, ge1 = 0.f, ge2 = 0.f;
        if (MODE >= 2) {
            const float* rg = a.go + (size_t)q * 3 * a.ldgo + cc;
            ge0 = rg[0] * kinv; ge1 = rg[a.ldgo] * kinv; ge2 = rg[2 * a.ldgo] * kinv;     // vnp_mean_pool_bwd: g / gs (gradients: 1 ulp, as EDIV)
        }
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, h0 = 0.f, h1 = 0.f, h2 = 0.f;                 // MODE 1: sum of outputs; MODE 3: gv (p | d)
        const int* nb = a.nbr + (size_t)q * k;
#pragma unroll 4
        for (int j = 0; j < k; ++j) {
            const size_t urow = (size_t)nb[j] * 3;
            const float* ru = R1 ? a.u + urow : a.u + urow * a.ldu + cc;
            float x0, x1, x2, t0 = 0.f, t1 = 0.f, t2 = 0.f;
            if (R1) {
                t0 = ru[0]; t1 = ru[1]; t2 = ru[2];                                          // the neighbour's coordinates: 12 bytes, not 24 C
                x0 = vnp_mul_alone(t0, wup) + vp0; x1 = vnp_mul_alone(t1, wup) + vp1; x2 = vnp_mul_alone(t2, wup) + vp2;
            } else { x0 = ru[0] + vp0; x1 = ru[a.ldu] + vp1; x2 = ru[2 * a.ldu] + vp2; }
            if (MODE == 0) {
                // statistics pass: hardware sqrt and reciprocal (1 ulp each) — the pass was bound by the IEEE sqrt / division sequences
                // (~22 of its ~30 instructions per edge and channel); the running statistics move by ~1e-7 relative
                const float n0 = __builtin_amdgcn_sqrtf(vnp_sq3(x0, x1, x2)) + VNP_EPS;
                ++wn;
                const float delta = n0 - wmean;
                wmean += delta * __builtin_amdgcn_rcpf((float)wn);
                wm2 += delta * (n0 - wmean);
                continue;
            }
            const float pn = ESQRT(vnp_sq3(x0, x1, x2));
            const float n = pn + VNP_EPS;
            float d0, d1, d2;
            if (R1) { d0 = vnp_mul_alone(t0, wud) + vd0; d1 = vnp_mul_alone(t1, wud) + vd1; d2 = vnp_mul_alone(t2, wud) + vd2; }
            else { d0 = ru[C] + vd0; d1 = ru[a.ldu + C] + vd1; d2 = ru[2 * a.ldu + C] + vd2; }
            const float xh = (n - mu) * rs;
            const float nh = xh * ga + be;
            const float s = EDIV(nh, n);
            const float y0 = x0 * s, y1 = x1 * s, y2 = x2 * s;                             // q = bn(p)
            const float dot = y0 * d0 + y1 * d1 + y2 * d2;
            if (MODE == 1) {
                float r0 = y0, r1 = y1, r2 = y2;
                if (dot < 0.f) {
                    const float al = (1.f - a.ns) * dot / (vnp_sq3(d0, d1, d2) + VNP_EPS);
                    r0 -= al * d0; r1 -= al * d1; r2 -= al * d2;
                }
                o0 += r0; o1 += r1; o2 += r2;
                continue;
            }
            float g0 = ge0, g1 = ge1, g2 = ge2, e0 = 0.f, e1 = 0.f, e2 = 0.f;
            if (dot < 0.f) {
                const float cf = 1.f - a.ns;
                const float inv = EDIV(1.f, vnp_sq3(d0, d1, d2) + VNP_EPS);
                const float al = dot * inv;
                const float gdd = g0 * d0 + g1 * d1 + g2 * d2;
                const float k1 = cf * gdd * inv;
                e0 = -cf * (al * g0 + gdd * (y0 * inv - 2.f * al * inv * d0));
                e1 = -cf * (al * g1 + gdd * (y1 * inv - 2.f * al * inv * d1));
                e2 = -cf * (al * g2 + gdd * (y2 * inv - 2.f * al * inv * d2));
                g0 -= k1 * d0; g1 -= k1 * d1; g2 -= k1 * d2;
            }
            const float gs = g0 * x0 + g1 * x1 + g2 * x2;
            const float gnh = EDIV(gs, n);
            if (MODE == 2) { s1 += gnh; s2 += gnh * xh; continue; }
            const float gn = ga * rs * (gnh - m1 - xh * m2) - EDIV(gs * nh, n * n);
            const float kk = pn > 0.f ? EDIV(gn, pn) : 0.f;
            const float f0 = s * g0 + kk * x0, f1 = s * g1 + kk * x1, f2 = s * g2 + kk * x2;   // dL/dp of this edge
            o0 += f0; o1 += f1; o2 += f2; h0 += e0; h1 += e1; h2 += e2;
        }
        if (MODE == 1 && live) {
            float* w = a.out + (size_t)q * 3 * a.ldo + c;
            w[0] = o0 / kf; w[a.ldo] = o1 / kf; w[2 * a.ldo] = o2 / kf;                    // vnp_mean_pool_fwd: s / gs
        }
        if (MODE == 3 && live) {
            float* w = a.gv + (size_t)q * 3 * a.ldgv + c;
            w[0] = o0; w[a.ldgv] = o1; w[2 * a.ldgv] = o2;
            w[C] = h0; w[a.ldgv + C] = h1; w[2 * a.ldgv + C] = h2;
        }
    }
    if (MODE == 0) {
        red[py * CW + cx] = make_float2(wmean, wm2);
        if (cx == 0) cnt[py] = wn;
        __syncthreads();
        if (py == 0 && live) {
            double na = cnt[0], ma = red[cx].x, Ma = red[cx].y;
            for (int j = 1; j < NQ; ++j) {
                const float2 rj = red[j * CW + cx];
                if (na == 0.0) { na = cnt[j]; ma = rj.x; Ma = rj.y; }
                else vnp_chan_merge(na, ma, Ma, (double)cnt[j], (double)rj.x, (double)rj.y);
            }
            a.partial[(size_t)by * C + c] = make_float2((float)ma, (float)Ma);
        }
    }
    if (MODE == 2) {
        red[py * CW + cx] = make_float2(s1, s2);
        __syncthreads();
        if (py == 0 && live) {
            float2 t[2];
            for (int h = 0; h < NQ / 4; ++h) {                       // quads in lane order, then the two quads of the 32 x 8 form
                const float2 r0 = red[(4 * h) * CW + cx], r1 = red[(4 * h + 1) * CW + cx], r2 = red[(4 * h + 2) * CW + cx], r3 = red[(4 * h + 3) * CW + cx];
                t[h] = make_float2((r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y));
            }
            a.partial[(size_t)by * C + c] = NQ == 4 ? t[0] : make_float2(t[0].x + t[1].x, t[0].y + t[1].y);
        }
    }
}

// gu[s] = sum over the edges (q -> s) of (dL/dp | dL/dd)(q, s): MODE 3's arithmetic per edge, gathered per SOURCE point.
// A workgroup = 64 channels x 4 edge lanes walks VNP_EDGE_SB consecutive sources; the four lanes share a source's list (edge e, e + 4, ...:
// a hub of a feature-space graph with hundreds of incoming edges does not serialise on one thread) and meet in LDS in a fixed order.
#define VNP_EDGE_SB 16
template <bool R1 = false>
__global__ void __launch_bounds__(256) vnp_edge_gather_u_kernel(const VnpEdgeArgs a, const int* __restrict__ rstart, const int* __restrict__ rev,
                                                               long long Pu)
{
    constexpr int MODE = 3;
    __shared__ double red[7 * 6 * 32];               // [NQ - 1][6][CW]: 64 channel lanes x 4 edge lanes, or 32 x 8 for C <= 32
    const int C = a.C;
    const int CW = C <= 32 ? 32 : 64, NQ = 256 / CW;
    const int cx = threadIdx.x & (CW - 1), py = threadIdx.x / CW;
    int bx, by;
    vnp_xcd_block(bx, by);
    const int c = bx * CW + cx;
    const bool live = c < C;
    const int cc = live ? c : 0;
    const long long s0 = (long long)by * VNP_EDGE_SB;
    const long long s1 = min(Pu, s0 + VNP_EDGE_SB);
    const float mu = a.mean[cc], rs = a.rstd[cc], ga = a.gamma[cc], be = a.beta[cc];
    float m1 = 0.f, m2 = 0.f;
    if (a.sums) { m1 = a.sums[2 * cc] * a.inv_count; m2 = a.sums[2 * cc + 1] * a.inv_count; }
    const float kinv = 1.f / (float)a.k;
    float wup = 0.f, wud = 0.f, wvp = 0.f, wvd = 0.f;
    if (R1) { wup = a.wu[cc]; wud = a.wu[C + cc]; wvp = a.wv[cc]; wvd = a.wv[C + cc]; }
    for (long long sp = s0; sp < s1; ++sp) {
        float up0, up1, up2, ud0, ud1, ud2;
        if (R1) {
            const float* xs = a.u + (size_t)sp * 3;
            const float c0 = xs[0], c1 = xs[1], c2 = xs[2];
            up0 = vnp_mul_alone(c0, wup); up1 = vnp_mul_alone(c1, wup); up2 = vnp_mul_alone(c2, wup);
            ud0 = vnp_mul_alone(c0, wud); ud1 = vnp_mul_alone(c1, wud); ud2 = vnp_mul_alone(c2, wud);
        } else {
            const float* ru = a.u + (size_t)sp * 3 * a.ldu + cc;
            up0 = ru[0]; up1 = ru[a.ldu]; up2 = ru[2 * a.ldu];
            ud0 = ru[C]; ud1 = ru[a.ldu + C]; ud2 = ru[2 * a.ldu + C];
        }
        double F0 = 0.0, F1 = 0.0, F2 = 0.0, E0 = 0.0, E1 = 0.0, E2 = 0.0;
        const int e1 = rstart[sp + 1];
#pragma unroll 2
        for (int e = rstart[sp] + py; e < e1; e += NQ) {
            const size_t q = (size_t)rev[e];
            const float* rv = R1 ? a.v + q * 3 : a.v + q * 3 * a.ldv + cc;
            const float* rg = a.go + q * 3 * a.ldgo + cc;
            float x0, x1, x2, d0, d1, d2;
            if (R1) {
                const float c0 = rv[0], c1 = rv[1], c2 = rv[2];
                x0 = up0 + vnp_mul_alone(c0, wvp); x1 = up1 + vnp_mul_alone(c1, wvp); x2 = up2 + vnp_mul_alone(c2, wvp);
                d0 = ud0 + vnp_mul_alone(c0, wvd); d1 = ud1 + vnp_mul_alone(c1, wvd); d2 = ud2 + vnp_mul_alone(c2, wvd);
            } else {
                x0 = up0 + rv[0]; x1 = up1 + rv[a.ldv]; x2 = up2 + rv[2 * a.ldv];
                d0 = ud0 + rv[C]; d1 = ud1 + rv[a.ldv + C]; d2 = ud2 + rv[2 * a.ldv + C];
            }
            const float ge0 = rg[0] * kinv, ge1 = rg[a.ldgo] * kinv, ge2 = rg[2 * a.ldgo] * kinv;
            const float pn = ESQRT(vnp_sq3(x0, x1, x2));
            const float n = pn + VNP_EPS;
            const float xh = (n - mu) * rs;
            const float nh = xh * ga + be;
            const float s = EDIV(nh, n);
            const float y0 = x0 * s, y1 = x1 * s, y2 = x2 * s;
            const float dot = y0 * d0 + y1 * d1 + y2 * d2;
            float g0 = ge0, g1 = ge1, g2 = ge2, ee0 = 0.f, ee1 = 0.f, ee2 = 0.f;
            if (dot < 0.f) {
                const float cf = 1.f - a.ns;
                const float inv = EDIV(1.f, vnp_sq3(d0, d1, d2) + VNP_EPS);
                const float al = dot * inv;
                const float gdd = g0 * d0 + g1 * d1 + g2 * d2;
                const float k1 = cf * gdd * inv;
                ee0 = -cf * (al * g0 + gdd * (y0 * inv - 2.f * al * inv * d0));
                ee1 = -cf * (al * g1 + gdd * (y1 * inv - 2.f * al * inv * d1));
                ee2 = -cf * (al * g2 + gdd * (y2 * inv - 2.f * al * inv * d2));
                g0 -= k1 * d0; g1 -= k1 * d1; g2 -= k1 * d2;
            }
            const float gs = g0 * x0 + g1 * x1 + g2 * x2;
            const float gnh = EDIV(gs, n);
            const float gn = ga * rs * (gnh - m1 - xh * m2) - EDIV(gs * nh, n * n);
            const float kk = pn > 0.f ? EDIV(gn, pn) : 0.f;
            F0 += (double)(s * g0 + kk * x0); F1 += (double)(s * g1 + kk * x1); F2 += (double)(s * g2 + kk * x2);
            E0 += (double)ee0; E1 += (double)ee1; E2 += (double)ee2;
        }
        if (py > 0) {
            double* r = red + (size_t)(py - 1) * 6 * CW + cx;
            r[0] = F0; r[CW] = F1; r[2 * CW] = F2; r[3 * CW] = E0; r[4 * CW] = E1; r[5 * CW] = E2;
        }
        __syncthreads();
        if (py == 0 && live) {
            for (int j = 0; j < NQ - 1; ++j) {
                const double* r = red + (size_t)j * 6 * CW + cx;
                F0 += r[0]; F1 += r[CW]; F2 += r[2 * CW];
                E0 += r[3 * CW]; E1 += r[4 * CW]; E2 += r[5 * CW];
            }
            float* w = a.gu + (size_t)sp * 3 * a.ldgu + c;
            w[0] = (float)F0; w[a.ldgu] = (float)F1; w[2 * a.ldgu] = (float)F2;
            w[C] = (float)E0; w[a.ldgu + C] = (float)E1; w[2 * a.ldgu + C] = (float)E2;
        }
        __syncthreads();
    }
}
#undef EDIV
#undef ESQRT

// ---- reverse adjacency of a neighbour table: count -> exclusive scan -> fill (integer atomics: arrival order) -> sort of every
// source's list, so that the lists — and every sum taken over them in list order — are the same on every run
__global__ void vnp_edge_rev_hist_kernel(const int* __restrict__ nbr, long long total, int* __restrict__ deg)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) atomicAdd(&deg[nbr[i]], 1);
}

// start[i] = sum of deg[0 .. i), i = 0 .. n; deg is zeroed (it becomes the fill pass's cursor).  One workgroup of 1024 threads walks tiles
// of 4096 counts: a thread takes four consecutive ones (one 16-byte load; the first version gave every thread its own contiguous 1/1024 of
// the array — uncoalesced, 60 us for 65536 counts), a wave scans its 64 partials with shuffles, the 16 wave totals meet in LDS.
__global__ void __launch_bounds__(1024) vnp_edge_rev_scan_kernel(int* __restrict__ deg, long long n, int* __restrict__ start)
{
    __shared__ int wtot[16];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int carry = 0;
    for (long long base = 0; base < n; base += 4096) {
        const long long i0 = base + 4 * t;
        int v[4] = {0, 0, 0, 0};
        if (i0 + 3 < n && (((uintptr_t)deg) & 15) == 0) { const int4 q = *(const int4*)(deg + i0); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else
#pragma unroll
            for (int e = 0; e < 4; ++e) if (i0 + e < n) v[e] = deg[i0 + e];
        const int mine = v[0] + v[1] + v[2] + v[3];
        int incl = mine;                                        // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        int woff = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) if (w < wv) woff += wtot[w];
        int run = carry + woff + incl - mine;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (i0 + e < n) { start[i0 + e] = run; run += v[e]; deg[i0 + e] = 0; }
        if (t == 1023) carry_s = carry + woff + incl;
        __syncthreads();
        carry = carry_s;
    }
    if (t == 0) start[n] = carry;
}

__global__ void vnp_edge_rev_fill_kernel(const int* __restrict__ nbr, long long total, int k, const int* __restrict__ start, int* __restrict__ cursor,
                                         int* __restrict__ rev)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int sidx = nbr[i];
        rev[start[sidx] + atomicAdd(&cursor[sidx], 1)] = (int)(i / k);
    }
}

// Ascending sort of data[0 .. L) by one workgroup: the bitonic network in its all-ascending form (first stage of every merge mirrors the
// block, i ^ (2 j - 1); the others i ^ j; the smaller value always goes to the smaller index), so positions >= L behave as +infinity
// without being stored.  `data` is LDS or global memory (a workgroup's own global accesses are ordered by __syncthreads).
__device__ void vnp_block_sort_asc(int* data, int L)
{
    int n2 = 1;
    while (n2 < L) n2 <<= 1;
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool first = j == (k >> 1);
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                const int p = first ? (i ^ (k - 1)) : (i ^ j);
                if (p > i && p < L) {                             // (i < p < L: both real; p >= L is +infinity and stays where it is)
                    const int a = data[i], c = data[p];
                    if (c < a) { data[i] = c; data[p] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// Every source point's list `in[start[s] .. start[s + 1])` (query ids, in arrival order) -> ascending in `out`.  A workgroup takes four
// sources.  Lists of up to 64 entries (a kNN graph's in-degree is ~k) are sorted by one wave each, by counting: an entry's position is the
// number of entries that are smaller, or equal and earlier (equal ids — a query that lists the same neighbour twice — are
// interchangeable): one load, <= 64 shuffle-compare steps, one store per lane.  Longer lists (hubs: feature-space graphs) are sorted by
// the whole workgroup, in LDS up to VNP_REV_LDS entries, in place in the unsorted buffer beyond.
#define VNP_REV_LDS 4096
__global__ void __launch_bounds__(256) vnp_edge_rev_sort_kernel(const int* __restrict__ start, int* __restrict__ in, int* __restrict__ out,
                                                                long long Pu)
{
    __shared__ int buf[VNP_REV_LDS];
    const int lane = threadIdx.x & 63;
    {
        const long long s = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
        if (s < Pu) {
            const int b0 = start[s], L = start[s + 1] - b0;
            if (L > 0 && L <= 64) {
                const int v = lane < L ? in[b0 + lane] : 0x7fffffff;
                int rank = 0;
                for (int j = 0; j < L; ++j) {
                    const int u = __shfl(v, j);
                    rank += (u < v || (u == v && j < lane)) ? 1 : 0;
                }
                if (lane < L) out[b0 + rank] = v;
            }
        }
    }
    for (int q = 0; q < 4; ++q) {                               // workgroup-uniform: the long lists among the four
        const long long s = (long long)blockIdx.x * 4 + q;
        if (s >= Pu) break;
        const int b0 = start[s], L = start[s + 1] - b0;
        if (L <= 64) continue;
        if (L <= VNP_REV_LDS) {
            __syncthreads();
            for (int i = threadIdx.x; i < L; i += 256) buf[i] = in[b0 + i];
            __syncthreads();
            vnp_block_sort_asc(buf, L);
            for (int i = threadIdx.x; i < L; i += 256) out[b0 + i] = buf[i];
        } else {
            __syncthreads();
            vnp_block_sort_asc(in + b0, L);
            for (int i = threadIdx.x; i < L; i += 256) out[b0 + i] = in[b0 + i];
        }
    }
}

// ================================================================================================ C-ABI
static inline dim3 vnp_grid_pc(long long P, int C, int chunk) { return dim3(vnp_cdiv(C, 64), vnp_cdiv(P, chunk)); }
static inline bool vnp_chunk_ok(int chunk) { return chunk >= 4 && chunk <= VNP_CHUNK; }

// partial [ceil(P / chunk_points)][C][2]; chunk_points = 0: vnp_chunk_points()
extern "C" int vnp_vn_norm_stats(hipStream_t st, const float* p, int ldp, long long P, int C, float* partial, int chunk_points)
{
    if (!p || !partial || P <= 0 || C <= 0) return VNP_EINVAL;
    const int chunk = chunk_points > 0 ? chunk_points : VNP_CHUNK;
    if (!vnp_chunk_ok(chunk) || vnp_cdiv(P, chunk) > 65535) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_norm_stats_kernel, vnp_grid_pc(P, C, chunk), dim3(256), 0, st, p, ldp, P, C, (float2*)partial, chunk);
    return vnp_last_error();
}

extern "C" int vnp_vn_bn_finalize(hipStream_t st, const float* partial, int nchunk, int C, long long count,
                                  float* mean, float* rstd, float* running_mean, float* running_var, float momentum, float eps)
{
    if (!partial || !mean || !rstd || C <= 0 || count <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_bn_finalize_kernel, dim3(vnp_cdiv(C, 64)), dim3(64 * VNP_FIN_LANES), 0, st, (const float2*)partial, nchunk, VNP_CHUNK, C, count,
                       mean, rstd, running_mean, running_var, momentum, eps);
    return vnp_last_error();
}

extern "C" int vnp_rstd_from_var(hipStream_t st, const float* var, float* rstd, int C, float eps)
{
    if (!var || !rstd || C <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_rstd_from_var_kernel, dim3(vnp_cdiv(C, 256)), dim3(256), 0, st, var, rstd, C, eps);
    return vnp_last_error();
}

extern "C" int vnp_vn_bn_leaky_fwd(hipStream_t st, const float* p, int ldp, const float* d, int ldd, float* out, int ldo,
                                   long long P, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                   float negative_slope)
{
    if (!p || !out || P <= 0 || C <= 0) return VNP_EINVAL;
    if (mean && (!rstd || !gamma || !beta)) return VNP_EINVAL;
    const int chunk = vnp_chunk_points_pc(P, C);
    hipLaunchKernelGGL(vnp_bn_leaky_fwd_kernel, vnp_grid_pc(P, C, chunk), dim3(256), 0, st, p, ldp, d, ldd, out, ldo, P, C,
                       mean, rstd, gamma, beta, negative_slope, chunk);
    return vnp_last_error();
}

extern "C" int vnp_vn_bn_leaky_bwd1(hipStream_t st, const float* go, int ldgo, const float* go_row, const float* go_col,
                                    const float* p, int ldp, const float* d, int ldd,
                                    float* gq, int ldgq, float* gd, int ldgd, long long P, int C,
                                    const float* mean, const float* rstd, const float* gamma, const float* beta,
                                    float negative_slope, float* partial, float* gw_partial, int chunk_points)
{
    if (!p || !gq || P <= 0 || C <= 0) return VNP_EINVAL;
    const int chunk = chunk_points > 0 ? chunk_points : VNP_CHUNK;     // partial / gw_partial: one row per chunk of this many points
    if (!vnp_chunk_ok(chunk) || vnp_cdiv(P, chunk) > 65535) return VNP_EINVAL;
    if (gw_partial && (go || !d)) return VNP_EINVAL;
    if (!go && (!go_row || !go_col)) return VNP_EINVAL;
    if (d && !gd) return VNP_EINVAL;
    if (mean && (!rstd || !gamma || !beta || !partial)) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_bn_leaky_bwd1_kernel<0>, vnp_grid_pc(P, C, chunk), dim3(256), 0, st, go, ldgo, go_row, go_col, p, ldp, d, ldd,
                       gq, ldgq, gd, ldgd, P, C, mean, rstd, gamma, beta, negative_slope, (float2*)partial, gw_partial, chunk,
                       (const float*)nullptr, 0.f, (float*)nullptr);
    return vnp_last_error();
}

// Two-pass backward of BatchNorm-over-norms + leaky projection (see vnp_bn_leaky_bwd1_kernel).
//   pass 1: partial [nchunk][C][2] (and gw_partial) only — gq / gd are not touched and may be NULL;
//   pass 2: gq = dL/dp (phase 2 applied with the reduced sums S [C][2], NULL = eval mode: no batch terms) and gd; no partials;
//           colsum (nullable, needs d) [nchunk][3][2C]: per-chunk sums over points of (dL/dp | dL/dd) per component.
// mean must be non-NULL (without BatchNorm there is no phase 2: use vnp_vn_bn_leaky_bwd1).
extern "C" int vnp_vn_bn_leaky_bwd_pass(hipStream_t st, int pass, const float* go, int ldgo, const float* go_row, const float* go_col,
                                        const float* p, int ldp, const float* d, int ldd,
                                        float* gq, int ldgq, float* gd, int ldgd, long long P, int C,
                                        const float* mean, const float* rstd, const float* gamma, const float* beta,
                                        float negative_slope, float* partial, float* gw_partial, int chunk_points,
                                        const float* S, long long count, float* colsum)
{
    if (!p || P <= 0 || C <= 0 || (pass != 1 && pass != 2)) return VNP_EINVAL;
    if (!mean || !rstd || !gamma || !beta) return VNP_EINVAL;
    if (!go && (!go_row || !go_col)) return VNP_EINVAL;
    if (pass == 1 && (!partial || (gw_partial && (go || !d)))) return VNP_EINVAL;
    if (pass == 2 && (!gq || (d && !gd))) return VNP_EINVAL;
    const int chunk = chunk_points > 0 ? chunk_points : VNP_CHUNK;
    if (!vnp_chunk_ok(chunk) || vnp_cdiv(P, chunk) > 65535) return VNP_EINVAL;
    const float inv = count > 0 ? 1.0f / (float)count : 0.f;
    if (pass == 1)
        hipLaunchKernelGGL(vnp_bn_leaky_bwd1_kernel<1>, vnp_grid_pc(P, C, chunk), dim3(256), 0, st, go, ldgo, go_row, go_col, p, ldp, d, ldd,
                           gq, ldgq, gd, ldgd, P, C, mean, rstd, gamma, beta, negative_slope, (float2*)partial, gw_partial, chunk,
                           (const float*)nullptr, 0.f, (float*)nullptr);
    else
        hipLaunchKernelGGL(vnp_bn_leaky_bwd1_kernel<2>, vnp_grid_pc(P, C, chunk), dim3(256), 0, st, go, ldgo, go_row, go_col, p, ldp, d, ldd,
                           gq, ldgq, gd, ldgd, P, C, mean, rstd, gamma, beta, negative_slope, (float2*)nullptr, (float*)nullptr, chunk,
                           S, inv, (d && pass == 2) ? colsum : (float*)nullptr);
    return vnp_last_error();
}

extern "C" int vnp_vn_bn_leaky_rowdot_fwd(hipStream_t st, const float* p, int ldp, const float* d, int ldd, long long P, int C,
                                          const float* mean, const float* rstd, const float* gamma, const float* beta,
                                          float negative_slope, const float* w, const float* add, int add_group, float* out)
{
    if (!p || !d || !mean || !rstd || !gamma || !beta || !w || !out || P <= 0 || C <= 0) return VNP_EINVAL;
    if (add && add_group <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_bn_leaky_rowdot_kernel, dim3(vnp_cdiv(P, 4)), dim3(256), 0, st, p, ldp, d, ldd, P, C, mean, rstd, gamma, beta,
                       negative_slope, w, add, add_group > 0 ? add_group : 1, out);
    return vnp_last_error();
}

extern "C" int vnp_vn_bn_bwd2(hipStream_t st, float* g, int ldg, const float* p, int ldp, long long P, int C,
                              const float* mean, const float* rstd, const float* gamma, const float* beta,
                              const float* S, long long count)
{
    if (!g || !p || !mean || !rstd || !gamma || !beta || P <= 0 || C <= 0) return VNP_EINVAL;
    const int chunk = vnp_chunk_points_pc(P, C);
    hipLaunchKernelGGL(vnp_bn_bwd2_kernel, vnp_grid_pc(P, C, chunk), dim3(256), 0, st, g, ldg, p, ldp, P, C, mean, rstd, gamma, beta,
                       S, count > 0 ? 1.0f / (float)count : 0.f, chunk);
    return vnp_last_error();
}

extern "C" int vnp_reduce_rows(hipStream_t st, const float* in, int nb, long long batch_stride, int n, long long stride, int W,
                               float* out, long long out_stride)
{
    if (!in || !out || nb <= 0 || n <= 0 || W <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_reduce_rows_kernel, dim3(vnp_cdiv(W, 64), nb), dim3(64 * VNP_RR_LANES), 0, st, in, batch_stride, n, stride, W, out, out_stride);
    return vnp_last_error();
}

// Sum of n partials of interleaved pairs, in [n][C][2] -> S [C][2] (what the BatchNorm backward kernels read) AND planar [2][C]
// (planar[0] = the first member of every pair, planar[1] = the second: the two parameter gradients as contiguous rows, so that no
// strided copy follows the reduction).
extern "C" int vnp_reduce_pairs(hipStream_t st, const float* in, int n, int C, float* S, float* planar)
{
    if (!in || !S || !planar || n <= 0 || C <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_reduce_rows_kernel, dim3(vnp_cdiv(2 * C, 64), 1), dim3(64 * VNP_RR_LANES), 0, st, in, 0LL, n, (long long)2 * C, 2 * C,
                       S, 0LL, planar);
    return vnp_last_error();
}

// ws_val / ws_idx: G * ceil(gs / vnp_chunk_points()) * C elements each
extern "C" int vnp_vn_pool_fwd(hipStream_t st, const float* x, int ldx, const float* d, int ldd, long long G, int gs, int C,
                               int* idx, float* out, int ldo, float* ws_val, int* ws_idx)
{
    if (!x || !d || !idx || !out || !ws_val || !ws_idx || G <= 0 || gs <= 0 || C <= 0 || G > 2147483647LL) return VNP_EINVAL;
    const int nchunk = vnp_cdiv(gs, VNP_CHUNK);
    if (nchunk > 65535) return VNP_EINVAL;
    // gridDim.z is limited to 65535: loop over slices of groups
    for (long long g0 = 0; g0 < G; g0 += 65535) {
        const int ng = (int)((G - g0) < 65535 ? (G - g0) : 65535);
        hipLaunchKernelGGL(vnp_pool_partial_kernel, dim3(vnp_cdiv(C, 64), nchunk, ng), dim3(256), 0, st,
                           x + (size_t)g0 * gs * 3 * ldx, ldx, d + (size_t)g0 * gs * 3 * ldd, ldd, gs, C, nchunk,
                           ws_val + (size_t)g0 * nchunk * C, ws_idx + (size_t)g0 * nchunk * C);
        hipLaunchKernelGGL(vnp_pool_final_kernel, dim3(vnp_cdiv(C, 256), ng), dim3(256), 0, st,
                           x + (size_t)g0 * gs * 3 * ldx, ldx, gs, C, nchunk,
                           (const float*)(ws_val + (size_t)g0 * nchunk * C), (const int*)(ws_idx + (size_t)g0 * nchunk * C),
                           idx + (size_t)g0 * C, out + (size_t)g0 * 3 * ldo, ldo);
    }
    return vnp_last_error();
}

// gx [G*gs*3, ldgx] is zero-filled here, then the selected rows receive gout.
extern "C" int vnp_vn_pool_bwd(hipStream_t st, const float* gout, int ldgo, const int* idx, long long G, int gs, int C,
                               float* gx, int ldgx)
{
    if (!gout || !idx || !gx || G <= 0 || gs <= 0 || C <= 0) return VNP_EINVAL;
    if (ldgx != C) {
        // strided destination: clear row by row
        hipError_t e = hipMemset2DAsync(gx, (size_t)ldgx * 4, 0, (size_t)C * 4, (size_t)G * gs * 3, st);
        if (e != hipSuccess) return (int)e;
    } else {
        hipError_t e = hipMemsetAsync(gx, 0, (size_t)G * gs * 3 * C * 4, st);
        if (e != hipSuccess) return (int)e;
    }
    for (long long g0 = 0; g0 < G; g0 += 65535) {
        const int ng = (int)((G - g0) < 65535 ? (G - g0) : 65535);
        hipLaunchKernelGGL(vnp_pool_bwd_kernel, dim3(vnp_cdiv(C, 256), ng), dim3(256), 0, st,
                           gout + (size_t)g0 * 3 * ldgo, ldgo, idx + (size_t)g0 * C, gs, C, gx + (size_t)g0 * gs * 3 * ldgx, ldgx);
    }
    return vnp_last_error();
}

// gx [G*gs*3, ldgx] += the pool's gradient (NOT zero-filled: gx already holds the gradient from the input's other consumer)
extern "C" int vnp_vn_pool_bwd_acc(hipStream_t st, const float* gout, int ldgo, const int* idx, long long G, int gs, int C,
                                   float* gx, int ldgx)
{
    if (!gout || !idx || !gx || G <= 0 || gs <= 0 || C <= 0) return VNP_EINVAL;
    for (long long g0 = 0; g0 < G; g0 += 65535) {
        const int ng = (int)((G - g0) < 65535 ? (G - g0) : 65535);
        hipLaunchKernelGGL(vnp_pool_bwd_acc_kernel, dim3(vnp_cdiv(C, 256), ng), dim3(256), 0, st,
                           gout + (size_t)g0 * 3 * ldgo, ldgo, idx + (size_t)g0 * C, gs, C, gx + (size_t)g0 * gs * 3 * ldgx, ldgx);
    }
    return vnp_last_error();
}

extern "C" int vnp_vn_mean_pool_fwd(hipStream_t st, const float* x, int ldx, long long G, int gs, int C, float* out, int ldo)
{
    if (!x || !out || G <= 0 || gs <= 0 || C <= 0) return VNP_EINVAL;
    for (long long g0 = 0; g0 < G; g0 += 21845) {
        const int ng = (int)((G - g0) < 21845 ? (G - g0) : 21845);
        hipLaunchKernelGGL(vnp_mean_pool_fwd_kernel, dim3(vnp_cdiv(C, 256), ng * 3), dim3(256), 0, st,
                           x + (size_t)g0 * gs * 3 * ldx, ldx, gs, C, out + (size_t)g0 * 3 * ldo, ldo);
    }
    return vnp_last_error();
}

extern "C" int vnp_vn_mean_pool_bwd(hipStream_t st, const float* gout, int ldgo, long long G, int gs, int C, float* gx, int ldgx)
{
    if (!gout || !gx || G <= 0 || gs <= 0 || C <= 0) return VNP_EINVAL;
    for (long long g0 = 0; g0 < G; g0 += 21845) {
        const int ng = (int)((G - g0) < 21845 ? (G - g0) : 21845);
        hipLaunchKernelGGL(vnp_mean_pool_bwd_kernel, dim3(vnp_cdiv(C, 256), ng * 3), dim3(256), 0, st,
                           gout + (size_t)g0 * 3 * ldgo, ldgo, gs, C, gx + (size_t)g0 * gs * 3 * ldgx, ldgx);
    }
    return vnp_last_error();
}

extern "C" int vnp_fold_fwd(hipStream_t st, const float* bias, int ldb, const float* wseed, const float* wcoarse,
                            const float* seed, int seed_batched, const float* coarse, int B, int Mc, int S, int W,
                            float* out, int ldo)
{
    if (!bias || !wseed || !wcoarse || !seed || !coarse || !out || B <= 0 || Mc <= 0 || S <= 0 || W <= 0) return VNP_EINVAL;
    if ((long long)B * Mc > 2147483647LL / 4) return VNP_EINVAL;
    const long long npc = (long long)B * Mc;
    for (long long g0 = 0; g0 < npc;) {
        // whole samples per launch (gridDim.y <= 65535) so that b = pc / Mc stays valid with the offset pointers
        const long long nb = (npc - g0) / Mc;
        const long long take = nb * Mc < 65535 ? nb * Mc : (65535 / Mc) * (long long)Mc;
        if (take <= 0) return VNP_EINVAL;
        const int b0 = (int)(g0 / Mc);
        hipLaunchKernelGGL(vnp_fold_fwd_kernel, dim3(vnp_cdiv(W, 256), (int)take), dim3(256), 0, st,
                           bias + (size_t)b0 * 3 * ldb, ldb, wseed, wcoarse, seed + (seed_batched ? (size_t)b0 * S * 3 : 0), seed_batched,
                           coarse + (size_t)g0 * 3, Mc, S, W, out + (size_t)g0 * S * 3 * ldo, ldo);
        g0 += take;
    }
    return vnp_last_error();
}

// part: [B*Mc][5][W]
extern "C" int vnp_fold_bwd(hipStream_t st, const float* g, int ldg, const float* seed, int seed_batched, const float* coarse,
                            int B, int Mc, int S, int W, float* part)
{
    if (!g || !seed || !coarse || !part || B <= 0 || Mc <= 0 || S <= 0 || W <= 0) return VNP_EINVAL;
    const long long npc = (long long)B * Mc;
    for (long long g0 = 0; g0 < npc;) {
        const long long nb = (npc - g0) / Mc;
        const long long take = nb * Mc < 65535 ? nb * Mc : (65535 / Mc) * (long long)Mc;
        if (take <= 0) return VNP_EINVAL;
        const int b0 = (int)(g0 / Mc);
        hipLaunchKernelGGL(vnp_fold_bwd_kernel, dim3(vnp_cdiv(W, 256), (int)take), dim3(256), 0, st,
                           g + (size_t)g0 * S * 3 * ldg, ldg, seed + (seed_batched ? (size_t)b0 * S * 3 : 0), seed_batched,
                           coarse + (size_t)g0 * 3, Mc, S, W, part + (size_t)g0 * 5 * W);
        g0 += take;
    }
    return vnp_last_error();
}

extern "C" int vnp_rowdot(hipStream_t st, const float* X, int ldx, const float* w, long long R, int C,
                          const float* add, int add_group, float* out)
{
    if (!X || !w || !out || R <= 0 || C <= 0) return VNP_EINVAL;
    if (add && add_group <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_rowdot_kernel, dim3(vnp_cdiv(R, 4)), dim3(256), 0, st, X, ldx, w, R, C, add, add_group > 0 ? add_group : 1, out);
    return vnp_last_error();
}

extern "C" int vnp_group_sum3(hipStream_t st, const float* in, long long Pc, int S, float* out, int accumulate)
{
    if (!in || !out || Pc <= 0 || S <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_group_sum3_kernel, dim3(vnp_cdiv(Pc * 3, 256)), dim3(256), 0, st, in, Pc, S, out, accumulate);
    return vnp_last_error();
}

// short groups (R <= 128 rows: 16 grid samples per coarse point in the attention decoder, 16384 groups): a WAVE per group, four groups
// per workgroup — one workgroup per group left three of its four row lanes idle on every second load (0.3 TB/s)
__global__ void __launch_bounds__(256) vnp_colsum_small_kernel(const float* __restrict__ X, long long batch_stride, long long row_stride,
                                                              int nb, int R, int C, const float* __restrict__ w, float* __restrict__ partial)
{
    const int cx = threadIdx.x & 63, g = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int c = blockIdx.x * 64 + cx;
    if (g >= nb || c >= C) return;
    const float* base = X + (size_t)g * batch_stride + c;
    const float* ww = w ? w + (size_t)g * R : nullptr;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int r = 0;
    for (; r + 3 < R; r += 4) {
        const float x0 = __builtin_nontemporal_load(&base[(size_t)r * row_stride]), x1 = __builtin_nontemporal_load(&base[(size_t)(r + 1) * row_stride]);
        const float x2 = __builtin_nontemporal_load(&base[(size_t)(r + 2) * row_stride]), x3 = __builtin_nontemporal_load(&base[(size_t)(r + 3) * row_stride]);
        t0 += x0 * (ww ? ww[r] : 1.f); t1 += x1 * (ww ? ww[r + 1] : 1.f); t2 += x2 * (ww ? ww[r + 2] : 1.f); t3 += x3 * (ww ? ww[r + 3] : 1.f);
    }
    for (; r < R; ++r) t0 += __builtin_nontemporal_load(&base[(size_t)r * row_stride]) * (ww ? ww[r] : 1.f);
    partial[(size_t)g * C + c] = (t0 + t1) + (t2 + t3);
}

// partial: [nb][ceil(R / vnp_chunk_points())][C]
extern "C" int vnp_colsum(hipStream_t st, const float* X, long long batch_stride, long long row_stride, int nb, long long R, int C,
                          const float* w, float* partial)
{
    if (!X || !partial || nb <= 0 || R <= 0 || C <= 0 || nb > 65535) return VNP_EINVAL;
    const int nchunk = vnp_cdiv(R, VNP_CHUNK);
    if (nchunk > 65535) return VNP_EINVAL;
    if (R <= 128 && nb >= 64) {
        hipLaunchKernelGGL(vnp_colsum_small_kernel, dim3(vnp_cdiv(C, 64), vnp_cdiv(nb, 4)), dim3(256), 0, st, X, batch_stride, row_stride, nb, (int)R, C, w,
                           partial);
        return vnp_last_error();
    }
    hipLaunchKernelGGL(vnp_colsum_kernel, dim3(vnp_cdiv(C, 64), nchunk, nb), dim3(256), 0, st, X, batch_stride, row_stride, R, C, w, nchunk, partial);
    return vnp_last_error();
}

extern "C" int vnp_rows_to_ref(hipStream_t st, const float* rows, int ld, int B, int C, int N, float* ref)
{
    if (!rows || !ref || B <= 0 || C <= 0 || N <= 0) return VNP_EINVAL;
    const long long total = (long long)B * C * 3 * N;
    hipLaunchKernelGGL(vnp_rows_to_ref_kernel, dim3((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256)), dim3(256), 0, st, rows, ld, B, C, N, ref);
    return vnp_last_error();
}

extern "C" int vnp_ref_to_rows(hipStream_t st, const float* ref, int B, int C, int N, float* rows, int ld)
{
    if (!rows || !ref || B <= 0 || C <= 0 || N <= 0) return VNP_EINVAL;
    const long long total = (long long)B * C * 3 * N;
    hipLaunchKernelGGL(vnp_ref_to_rows_kernel, dim3((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256)), dim3(256), 0, st, ref, B, C, N, rows, ld);
    return vnp_last_error();
}

// BatchNorm statistics from partials whose chunks hold `chunk_points` values each (the fused EdgeConv: 64 queries x k edges)
extern "C" int vnp_vn_bn_finalize_chunked(hipStream_t st, const float* partial, int nchunk, int chunk_points, int C, long long count,
                                          float* mean, float* rstd, float* running_mean, float* running_var, float momentum, float eps)
{
    if (!partial || !mean || !rstd || C <= 0 || count <= 0 || chunk_points <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_bn_finalize_kernel, dim3(vnp_cdiv(C, 64)), dim3(64 * VNP_FIN_LANES), 0, st, (const float2*)partial, nchunk, chunk_points, C, count,
                       mean, rstd, running_mean, running_var, momentum, eps);
    return vnp_last_error();
}

extern "C" int vnp_edge_block_queries(void) { return VNP_EDGE_QB; }

// Fused EdgeConv passes (see vnp_edge_fused_kernel).  u [Pu*3, ldu >= 2C], v [Pq*3, ldv >= 2C], nbr [Pq][k] ids < Pu.
// mode 0: partial [ceil(Pq/64)][C][2].  mode 1: out [Pq*3, ldo].  mode 2: go, partial.  mode 3: go, sums ([C][2] or NULL), count =
// number of edges, gv [Pq*3, ldgv >= 2C]; gu / ldgu / Pu must be NULL / 0 / 0 (the scatter form with float atomics is gone: the source
// points' gradient comes from vnp_edge_fused_gather, a gather over the reverse adjacency).
static int vnp_edge_fused_impl(hipStream_t st, int mode, const float* u, int ldu, const float* v, int ldv, const int* nbr,
                              long long Pq, int k, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                              float negative_slope, float* partial, float* out, int ldo, const float* go, int ldgo,
                              const float* sums, long long count, float* gu, int ldgu, long long Pu, float* gv, int ldgv,
                              const float* wu, const float* wv)
{
    const bool r1 = wu != nullptr;
    if (!u || !v || !nbr || Pq <= 0 || k <= 0 || C <= 0 || mode < 0 || mode > 3) return VNP_EINVAL;
    if (mode >= 1 && (!mean || !rstd || !gamma || !beta)) return VNP_EINVAL;
    if ((mode == 0 || mode == 2) && !partial) return VNP_EINVAL;
    if (mode == 1 && !out) return VNP_EINVAL;
    if (mode >= 2 && !go) return VNP_EINVAL;
    if (mode == 3 && (!gv || gu)) return VNP_EINVAL;
    (void)Pu; (void)ldgu;
    const long long nblk = (Pq + VNP_EDGE_QB - 1) / VNP_EDGE_QB;
    if (nblk > 65535LL * 32768LL) return VNP_EINVAL;
    VnpEdgeArgs a;
    a.u = u; a.ldu = ldu; a.v = v; a.ldv = ldv; a.nbr = nbr; a.Pq = Pq; a.k = k; a.C = C;
    a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.ns = negative_slope;
    a.partial = (float2*)partial; a.out = out; a.ldo = ldo; a.go = go; a.ldgo = ldgo;
    a.sums = sums; a.inv_count = count > 0 ? 1.0f / (float)count : 0.f;
    a.gu = nullptr; a.ldgu = 0; a.gv = gv; a.ldgv = ldgv;
    a.wu = wu; a.wv = wv;
    // gridDim.y <= 65535: slices of query blocks
    for (long long b0 = 0; b0 < nblk; b0 += 65535) {
        const int nb = (int)((nblk - b0) < 65535 ? (nblk - b0) : 65535);
        VnpEdgeArgs s = a;
        const long long qoff = b0 * VNP_EDGE_QB;
        s.v = v + (size_t)qoff * 3 * ldv; s.nbr = nbr + (size_t)qoff * k; s.Pq = (Pq - qoff) < (long long)nb * VNP_EDGE_QB ? (Pq - qoff) : (long long)nb * VNP_EDGE_QB;
        if (partial) s.partial = (float2*)partial + (size_t)b0 * C;
        if (out) s.out = out + (size_t)qoff * 3 * ldo;
        if (go) s.go = go + (size_t)qoff * 3 * ldgo;
        if (gv) s.gv = gv + (size_t)qoff * 3 * ldgv;
        const dim3 grid(vnp_cdiv(C, C <= 32 ? 32 : 64), nb), block(256);
        if (r1) switch (mode) {
        case 0: hipLaunchKernelGGL((vnp_edge_fused_kernel<0, true>), grid, block, 0, st, s); break;
        case 1: hipLaunchKernelGGL((vnp_edge_fused_kernel<1, true>), grid, block, 0, st, s); break;
        case 2: hipLaunchKernelGGL((vnp_edge_fused_kernel<2, true>), grid, block, 0, st, s); break;
        default: hipLaunchKernelGGL((vnp_edge_fused_kernel<3, true>), grid, block, 0, st, s); break;
        }
        else switch (mode) {
        case 0: hipLaunchKernelGGL((vnp_edge_fused_kernel<0, false>), grid, block, 0, st, s); break;
        case 1: hipLaunchKernelGGL((vnp_edge_fused_kernel<1, false>), grid, block, 0, st, s); break;
        case 2: hipLaunchKernelGGL((vnp_edge_fused_kernel<2, false>), grid, block, 0, st, s); break;
        default: hipLaunchKernelGGL((vnp_edge_fused_kernel<3, false>), grid, block, 0, st, s); break;
        }
    }
    return vnp_last_error();
}

extern "C" int vnp_edge_fused(hipStream_t st, int mode, const float* u, int ldu, const float* v, int ldv, const int* nbr,
                              long long Pq, int k, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                              float negative_slope, float* partial, float* out, int ldo, const float* go, int ldgo,
                              const float* sums, long long count, float* gu, int ldgu, long long Pu, float* gv, int ldgv)
{
    return vnp_edge_fused_impl(st, mode, u, ldu, v, ldv, nbr, Pq, k, C, mean, rstd, gamma, beta, negative_slope, partial, out, ldo, go, ldgo,
                               sums, count, gu, ldgu, Pu, gv, ldgv, nullptr, nullptr);
}

// The same passes for a layer whose input is ONE VN channel, the points themselves (conv1 of the encoders: models/dgcnn.py:291,
// dgcnn_group.py:225): xu [Pu*3], xv [Pq*3] coordinates, wu, wv [2C] the columns of [Wu; Wv] (vnp_edge_weights with one input channel).
// The per-point matrices u = x wu^T, v = x wv^T are never formed: an edge reads its neighbour's 12 bytes instead of 24 C.
// Forward: the same bits as vnp_edge_fused on the formed matrices; backward: equal to rounding.
extern "C" int vnp_edge_fused_r1(hipStream_t st, int mode, const float* xu, const float* xv, const float* wu, const float* wv, const int* nbr,
                                 long long Pq, int k, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 float negative_slope, float* partial, float* out, int ldo, const float* go, int ldgo,
                                 const float* sums, long long count, float* gv, int ldgv)
{
    if (!wu || !wv) return VNP_EINVAL;
    return vnp_edge_fused_impl(st, mode, xu, 1, xv, 1, nbr, Pq, k, C, mean, rstd, gamma, beta, negative_slope, partial, out, ldo, go, ldgo,
                               sums, count, nullptr, 0, 0, gv, ldgv, wu, wv);
}

// Reverse adjacency of nbr [Pq][k] (ids < Pu): start [Pu + 1] = exclusive prefix of the in-degrees, rev [Pq * k] = query ids grouped by
// source point (order inside a group unspecified).  ws: (Pu + 1) ints.  Pq * k and Pu below 2^31.
// start [Pu + 1], rev [Pq * k]: for every source point the ids of the queries that list it, ASCENDING (deterministic).
// ws: (Pu + 1) + Pq * k ints (the fill pass's cursors and its unsorted lists).
extern "C" int vnp_edge_reverse(hipStream_t st, const int* nbr, long long Pq, int k, long long Pu, int* start, int* rev, int* ws)
{
    if (!nbr || !start || !rev || !ws || Pq <= 0 || k <= 0 || Pu <= 0 || Pq * k > 2147483647LL || Pu > 2147483646LL) return VNP_EINVAL;
    const long long total = Pq * k;
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)(Pu + 1) * 4, st);
    if (e != hipSuccess) return (int)e;
    int* unsorted = ws + (Pu + 1);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(vnp_edge_rev_hist_kernel, dim3(blocks), dim3(256), 0, st, nbr, total, ws);
    hipLaunchKernelGGL(vnp_edge_rev_scan_kernel, dim3(1), dim3(1024), 0, st, ws, Pu, start);
    hipLaunchKernelGGL(vnp_edge_rev_fill_kernel, dim3(blocks), dim3(256), 0, st, nbr, total, k, (const int*)start, ws, unsorted);
    if ((Pu + 3) / 4 > 2147483647LL) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_edge_rev_sort_kernel, dim3((unsigned)((Pu + 3) / 4)), dim3(256), 0, st, (const int*)start, unsorted, rev, Pu);
    return vnp_last_error();
}

// gu [Pu*3, ldgu >= 2C] of vnp_edge_fused's mode 3 as a gather over the reverse adjacency (start, rev from vnp_edge_reverse): no atomics,
// no zero fill, the same per-edge arithmetic; k = neighbours per query (the mean-pool divisor), count = number of edges.
static int vnp_edge_fused_gather_impl(hipStream_t st, const float* u, int ldu, const float* v, int ldv, const int* start, const int* rev,
                                     long long Pu, int k, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                     float negative_slope, const float* go, int ldgo, const float* sums, long long count, float* gu, int ldgu,
                                     const float* wu, const float* wv)
{
    if (!u || !v || !start || !rev || !mean || !rstd || !gamma || !beta || !go || !gu || Pu <= 0 || k <= 0 || C <= 0) return VNP_EINVAL;
    VnpEdgeArgs a{};
    a.u = u; a.ldu = ldu; a.v = v; a.ldv = ldv; a.k = k; a.C = C;
    a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.ns = negative_slope;
    a.go = go; a.ldgo = ldgo; a.sums = sums; a.inv_count = count > 0 ? 1.0f / (float)count : 0.f;
    a.gu = gu; a.ldgu = ldgu;
    a.wu = wu; a.wv = wv;
    const long long nblk = (Pu + VNP_EDGE_SB - 1) / VNP_EDGE_SB;
    for (long long b0 = 0; b0 < nblk; b0 += 65535) {
        const int nb = (int)((nblk - b0) < 65535 ? (nblk - b0) : 65535);
        VnpEdgeArgs s = a;
        const long long soff = b0 * VNP_EDGE_SB;
        s.u = u + (size_t)soff * 3 * ldu; s.gu = gu + (size_t)soff * 3 * ldgu;
        const long long np = (Pu - soff) < (long long)nb * VNP_EDGE_SB ? (Pu - soff) : (long long)nb * VNP_EDGE_SB;
        const int cw = C <= 32 ? 32 : 64;
        if (wu) hipLaunchKernelGGL((vnp_edge_gather_u_kernel<true>), dim3(vnp_cdiv(C, cw), nb), dim3(256), 0, st, s, start + soff, rev, np);
        else    hipLaunchKernelGGL((vnp_edge_gather_u_kernel<false>), dim3(vnp_cdiv(C, cw), nb), dim3(256), 0, st, s, start + soff, rev, np);
    }
    return vnp_last_error();
}

extern "C" int vnp_edge_fused_gather(hipStream_t st, const float* u, int ldu, const float* v, int ldv, const int* start, const int* rev,
                                     long long Pu, int k, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                     float negative_slope, const float* go, int ldgo, const float* sums, long long count, float* gu, int ldgu)
{
    return vnp_edge_fused_gather_impl(st, u, ldu, v, ldv, start, rev, Pu, k, C, mean, rstd, gamma, beta, negative_slope, go, ldgo, sums, count,
                                      gu, ldgu, nullptr, nullptr);
}

extern "C" int vnp_edge_fused_gather_r1(hipStream_t st, const float* xu, const float* xv, const float* wu, const float* wv, const int* start,
                                        const int* rev, long long Pu, int k, int C, const float* mean, const float* rstd, const float* gamma,
                                        const float* beta, float negative_slope, const float* go, int ldgo, const float* sums, long long count,
                                        float* gu, int ldgu)
{
    if (!wu || !wv) return VNP_EINVAL;
    return vnp_edge_fused_gather_impl(st, xu, 1, xv, 1, start, rev, Pu, k, C, mean, rstd, gamma, beta, negative_slope, go, ldgo, sums, count,
                                      gu, ldgu, wu, wv);
}
