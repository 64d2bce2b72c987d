// Vector-Neuron pointwise / reduction kernels on the rows layout (vnp_common.h):
//   norm statistics + BatchNorm-over-norms + leaky projection (models/vn_layers.py:25-43, 46-74, 107-127),
//   arg-max pooling (models/vn_layers.py:153-167), the FoldingNet rank-2 expansion (models/pcn.py:375-385)
//   and the small column / row reductions their backward passes need.
// All of them are HBM-bound: one thread owns one (point, channel) pair, a wave reads 64 consecutive
// channels of a row (256 contiguous bytes) for each of the 3 Cartesian components.
#include "vn_kernels.h"

// point-loop unrolling of the streaming BatchNorm / leaky passes.  Measured with 2 (loads of two points in flight before the
// first store): decoder-sized maps bwd1 1588 -> 1627 us, bwd2 878 -> 882, fwd 501 -> 469 — neutral, so 1.  These passes already
// sit at the chip's mixed read/write ceiling (scripts/hbm_yardstick.hip: 5.5-5.8 TB/s for 2R1W / 3R2W streams, 6.4-6.9 read-only).
#ifndef VNP_PT_UNROLL_N
#define VNP_PT_UNROLL_N 1
#endif
#define VNP_STR2(x) #x
#define VNP_STR(x) VNP_STR2(x)
#define VNP_PT_UNROLL _Pragma(VNP_STR(unroll VNP_PT_UNROLL_N))

extern "C" int vnp_chunk_points() { return VNP_CHUNK; }

// Points per workgroup of the BatchNorm / leaky passes for a [P points, C channels] map: VNP_CHUNK, halved (down to 16: four points per
// point lane) until the grid has at least 1024 workgroups — a 128-channel layer over 65536 points is only 128 chunks of 1024 and left half
// the chip idle (measured: 1.0-1.5 TB/s on first_conv[0] against 5+ TB/s on the wide layers); the 4096-point layers of the transformer
// blocks stopped at 64 points (256 workgroups, 16 dependent iterations per lane: 20-29 us for 12 MB) until round 4 lowered the floor.  Statistics / gradient partials are per chunk of
// this size: callers size them with this function and pass the value on (vnp_vn_norm_stats, vnp_vn_bn_leaky_bwd1,
// vnp_vn_bn_finalize_chunked).
extern "C" int vnp_chunk_points_pc(long long P, int C)
{
    int chunk = VNP_CHUNK;
    const long long cols = (C + 63) / 64;
    while (chunk > 16 && ((P + chunk - 1) / chunk) * cols < 1024) chunk >>= 1;
    return chunk;
}

// ------------------------------------------------------------------------------------------------
// block = 256 threads = 64 channels x 4 point lanes; grid = (ceil(C/64), nchunk)
// ------------------------------------------------------------------------------------------------

// partial[chunk][c] = (mean, M2) of n = |p| + EPS over the chunk's points (vn_layers.py:121).
// Welford per thread + Chan's pairwise merge: the batch variance of the norms can be 1e4-1e5 times smaller than
// mean^2 (FoldingNet layer 0: every point of a sample shares the dominant global term), where sum / sum-of-squares
// in fp32 loses the variance to cancellation.

__global__ void __launch_bounds__(256) vnp_norm_stats_kernel(const float* __restrict__ p, int ld, long long P, int C,
                                                            float2* __restrict__ partial, int chunk)
{
    __shared__ float2 red[4][64];
    __shared__ int cnt[4];
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
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
// 1024 threads = VNP_FIN_CH channels x VNP_FIN_LANES partial lanes: lane l merges the partials l, l + 64, ... of its channel (Chan, in
// double), then the 64 lanes of a channel merge pairwise in LDS (a fixed tree: deterministic).  (Rounds 1-3: 64 channels x 16 lanes and a
// serial merge of the 16 — four workgroups for a 256-channel layer, 25 us for 512 partials; 16 channels x 64 lanes: 16 workgroups.)
#define VNP_FIN_CH 16
#define VNP_FIN_LANES 64
__global__ void __launch_bounds__(VNP_FIN_CH * VNP_FIN_LANES) vnp_bn_finalize_kernel(const float2* __restrict__ partial, int nchunk, int chunk_points, int C,
                                                             long long count, float* __restrict__ mean, float* __restrict__ rstd,
                                                             float* running_mean, float* running_var, float momentum, float eps)
{
    __shared__ double rn[VNP_FIN_LANES][VNP_FIN_CH], rm[VNP_FIN_LANES][VNP_FIN_CH], rM[VNP_FIN_LANES][VNP_FIN_CH];
    const int cx = threadIdx.x & (VNP_FIN_CH - 1), py = threadIdx.x / VNP_FIN_CH;
    const int c = blockIdx.x * VNP_FIN_CH + cx;
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
    for (int o = VNP_FIN_LANES / 2; o > 0; o >>= 1) {          // lane py absorbs lane py + o
        if (py < o) {
            const double nb = rn[py + o][cx];
            if (nb > 0.0) {
                if (na == 0.0) { na = nb; ma = rm[py + o][cx]; Ma = rM[py + o][cx]; }
                else vnp_chan_merge(na, ma, Ma, nb, rm[py + o][cx], rM[py + o][cx]);
                rn[py][cx] = na; rm[py][cx] = ma; rM[py][cx] = Ma;
            }
        }
        __syncthreads();
    }
    if (py != 0 || c >= C) return;
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
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
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
    // (MODE 2, the store-heavy final pass, measured 1.7 % slower with the scalar index — it sits at the HBM ceiling either way — and keeps the vector one)
    const int cx = threadIdx.x & 63, py = MODE == 2 ? (int)(threadIdx.x >> 6) : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    const long long pt = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
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
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
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

// ================================================================================================ C-ABI

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
    hipLaunchKernelGGL(vnp_bn_finalize_kernel, dim3(vnp_cdiv(C, VNP_FIN_CH)), dim3(VNP_FIN_CH * VNP_FIN_LANES), 0, st, (const float2*)partial, nchunk, VNP_CHUNK, C, count,
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

// BatchNorm statistics from partials whose chunks hold `chunk_points` values each (the fused EdgeConv: 64 queries x k edges)
extern "C" int vnp_vn_bn_finalize_chunked(hipStream_t st, const float* partial, int nchunk, int chunk_points, int C, long long count,
                                          float* mean, float* rstd, float* running_mean, float* running_var, float momentum, float eps)
{
    if (!partial || !mean || !rstd || C <= 0 || count <= 0 || chunk_points <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_bn_finalize_kernel, dim3(vnp_cdiv(C, VNP_FIN_CH)), dim3(VNP_FIN_CH * VNP_FIN_LANES), 0, st, (const float2*)partial, nchunk, chunk_points, C, count,
                       mean, rstd, running_mean, running_var, momentum, eps);
    return vnp_last_error();
}
