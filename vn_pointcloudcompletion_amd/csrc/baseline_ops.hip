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
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
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
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
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
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
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

// ================================================================================================================================
// EdgeConv block of the non-equivariant DGCNN_fps (/root/reference/models/dgcnn.py:101-124 and the layerN blocks):
//   Conv2d(1x1, no bias) on the edge tensor cat(x_k[nbr] - x_q, x_q)  ->  GroupNorm(G)  ->  LeakyReLU(slope)  ->  max over the k neighbours
// with the convolution applied per POINT first (u = W1 x_k, v = (W2 - W1) x_q: e(q, j) = u[nbr[q][j]] + v[q]) and the edge tensor never
// written: every pass recomputes e from u, v and the neighbour table.  Rows are point-major scalars: u [Pk, C], v [Pq, C], nbr [Pq][k]
// global row ids of u; a sample owns nq consecutive queries; GroupNorm statistics are per (sample, group of C / G channels) over the
// nq * k * C / G edge values.
//   mode 0: Welford partials per (block of EGN_QB queries, channel)        -> vnp_edge_gn_finalize: mean / rstd per (sample, group)
//   mode 1: out[q][c] = max_j leaky(gn(e(q, j))[c]), amax[q][c] = the j that attains it (first one)
//   mode 2: backward sums: per (block, channel) dbeta, dgamma, s1 = dy * gamma, s2 = dy * gamma * xhat   (dy lives at the arg-max edge only)
//   mode 3: gv[q][c] = sum_j de(q, j)[c] with de = rstd (dy gamma [j == j*] - m1 - xhat m2), m1 / m2 = the group's s1 / s2 over its element count
//   gather: gu[s][c] = sum over the edges INTO s of de, over the reverse adjacency (vnp_edge_reverse: ascending queries): no atomics
// Outside the measured hot path.
#define EGN_QB 64
struct VnpEdgeGnArgs {
    const float* u; int ldu; const float* v; int ldv; const int* nbr;
    long long Pq; int k, C, G, nq;
    const float* mean; const float* rstd;          // [B][G]
    const float* gamma; const float* beta; float slope;
    float* out; int ldo; unsigned char* amax;      // mode 1 (amax [Pq][C])
    const float* go; int ldgo;                     // modes 2, 3, gather
    float* partial;                                // mode 0: [nblk][C][2]; mode 2: [nblk][C][4]
    const float* sums;                             // mode 3 / gather: [B][C][4] (reduced mode-2 partials of each sample)
    float* gv; int ldgv;
};

__device__ __forceinline__ void egn_group_means(const VnpEdgeGnArgs& a, int b, int c, float& m1, float& m2)
{
    // the group's s1, s2: sums of its C / G channels' column sums, divided by the number of elements of the group in this sample
    const int cg = a.C / a.G, g0 = (c / cg) * cg;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < cg; ++i) { const float* s = a.sums + ((size_t)b * a.C + g0 + i) * 4; s1 += (double)s[2]; s2 += (double)s[3]; }
    const double n = (double)a.nq * a.k * cg;
    m1 = (float)(s1 / n); m2 = (float)(s2 / n);
}

template <int MODE>
__global__ void __launch_bounds__(256) vnp_edge_gn_kernel(const VnpEdgeGnArgs a)
{
    __shared__ float red[4][4][64];
    __shared__ int cnt[4];
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
    const int c = blockIdx.x * 64 + cx, C = a.C, k = a.k;
    const bool live = c < C;
    const int cc = live ? c : 0;
    const int nblk_s = (a.nq + EGN_QB - 1) / EGN_QB;                  // blocks per sample: a block never straddles two samples
    const int b = blockIdx.y / nblk_s, blk = blockIdx.y % nblk_s;
    const long long q0 = (long long)b * a.nq + (long long)blk * EGN_QB;
    const long long q1 = min((long long)(b + 1) * a.nq, q0 + EGN_QB);
    const int cg = C / a.G, g = cc / cg;
    float mu = 0.f, rs = 0.f, ga = 0.f, be = 0.f, m1 = 0.f, m2 = 0.f;
    if (MODE >= 1) { mu = a.mean[(size_t)b * a.G + g]; rs = a.rstd[(size_t)b * a.G + g]; ga = a.gamma[cc]; be = a.beta[cc]; }
    if (MODE == 3) egn_group_means(a, b, cc, m1, m2);
    float wmean = 0.f, wm2 = 0.f; int wk = 0;                          // MODE 0
    float sb = 0.f, sg = 0.f, s1 = 0.f, s2 = 0.f;                      // MODE 2
    for (long long q = q0 + py; q < q1; q += 4) {
        const float vq = a.v[(size_t)q * a.ldv + cc];
        const int* nb = a.nbr + (size_t)q * k;
        if (MODE == 0) {
            for (int j = 0; j < k; ++j) {
                const float e = a.u[(size_t)nb[j] * a.ldu + cc] + vq;
                ++wk; const float dl = e - wmean; wmean += dl / (float)wk; wm2 += dl * (e - wmean);
            }
        } else if (MODE == 1) {
            float best = -INFINITY; int bj = 0;
            for (int j = 0; j < k; ++j) {
                const float e = a.u[(size_t)nb[j] * a.ldu + cc] + vq;
                const float y = (e - mu) * rs * ga + be;
                const float z = y > 0.f ? y : y * a.slope;
                if (z > best) { best = z; bj = j; }
            }
            if (live) { a.out[(size_t)q * a.ldo + c] = best; a.amax[(size_t)q * C + c] = (unsigned char)bj; }
        } else {
            const int js = a.amax[(size_t)q * C + cc];
            const float gq = a.go[(size_t)q * a.ldgo + cc];
            const float es = a.u[(size_t)nb[js] * a.ldu + cc] + vq;
            const float xs = (es - mu) * rs;
            const float ys = xs * ga + be;
            const float dy = gq * (ys > 0.f ? 1.f : a.slope);
            if (MODE == 2) { sb += dy; sg += dy * xs; s1 += dy * ga; s2 += dy * ga * xs; continue; }
            float acc = rs * dy * ga;                                // the arg-max edge's own term
            for (int j = 0; j < k; ++j) {
                const float e = a.u[(size_t)nb[j] * a.ldu + cc] + vq;
                acc -= rs * (m1 + (e - mu) * rs * m2);
            }
            if (live) a.gv[(size_t)q * a.ldgv + c] = acc;
        }
    }
    if (MODE == 0) {
        red[py][0][cx] = wmean; red[py][1][cx] = wm2;
        if (cx == 0) cnt[py] = wk;
        __syncthreads();
        if (py == 0 && live) {
            double na = cnt[0], ma = red[0][0][cx], Ma = red[0][1][cx];
#pragma unroll
            for (int i = 1; i < 4; ++i) vnp_b1_chan_merge(na, ma, Ma, (double)cnt[i], (double)red[i][0][cx], (double)red[i][1][cx]);
            float* p = a.partial + ((size_t)blockIdx.y * C + c) * 2;
            p[0] = (float)ma; p[1] = (float)Ma;
        }
    } else if (MODE == 2) {
        red[py][0][cx] = sb; red[py][1][cx] = sg; red[py][2][cx] = s1; red[py][3][cx] = s2;
        __syncthreads();
        if (py == 0 && live) {
            float* p = a.partial + ((size_t)blockIdx.y * C + c) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] = (red[0][i][cx] + red[1][i][cx]) + (red[2][i][cx] + red[3][i][cx]);
        }
    }
}

// mean / rstd per (sample, group) from the mode-0 partials: one thread per (sample, group) merges its blocks' and channels' partials in
// double (<= 32 blocks x 32 channels at the network's sizes)
__global__ void vnp_edge_gn_finalize_kernel(const float* __restrict__ partial, int B, int nq, int k, int C, int G, float eps,
                                            float* __restrict__ mean, float* __restrict__ rstd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, g = i % G, cg = C / G;
    const int nblk_s = (nq + EGN_QB - 1) / EGN_QB;
    double na = 0.0, ma = 0.0, Ma = 0.0;
    for (int blk = 0; blk < nblk_s; ++blk) {
        const int nqb = min(EGN_QB, nq - blk * EGN_QB);
        for (int c = g * cg; c < (g + 1) * cg; ++c) {
            const float* p = partial + ((size_t)(b * nblk_s + blk) * C + c) * 2;
            vnp_b1_chan_merge(na, ma, Ma, (double)nqb * k, (double)p[0], (double)p[1]);
        }
    }
    mean[i] = (float)ma;
    rstd[i] = (float)(1.0 / sqrt(Ma / na + (double)eps));
}

// gu[s][c] = sum over the edges (q, j) with nbr[q][j] == s of de(q, j)[c]; start / rev: reverse adjacency (queries ascending, a query that
// lists s twice appears twice: its t-th appearance is the t-th j with nbr[q][j] == s).  grid (ceil(C / 64), ceil(Pk / 4)), a wave per source.
__global__ void __launch_bounds__(256) vnp_edge_gn_gather_kernel(const VnpEdgeGnArgs a, const int* __restrict__ start, const int* __restrict__ rev,
                                                                 long long s_base, long long Pk, float* __restrict__ gu, int ldgu)
{
    const int cx = threadIdx.x & 63;
    const long long s = s_base + (long long)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * 64 + cx, C = a.C, k = a.k;
    if (s >= Pk || c >= C) return;
    const int cg = C / a.G, g = c / cg;
    const float us = a.u[(size_t)s * a.ldu + c], ga = a.gamma[c], be = a.beta[c];
    const int e0 = start[s], e1 = start[s + 1];
    double acc = 0.0;
    int prev_q = -1, occ = 0, b_cur = -1;
    float mu = 0.f, rs = 0.f, m1 = 0.f, m2 = 0.f;
    for (int e = e0; e < e1; ++e) {
        const int q = rev[e];
        occ = q == prev_q ? occ + 1 : 0;
        prev_q = q;
        const int b = q / a.nq;
        if (b != b_cur) {                                            // (a source is named by queries of its own sample only, but nothing here relies on it)
            b_cur = b;
            mu = a.mean[(size_t)b * a.G + g]; rs = a.rstd[(size_t)b * a.G + g];
            egn_group_means(a, b, c, m1, m2);
        }
        const int* nb = a.nbr + (size_t)q * k;
        int j = 0, seen = 0;
        for (; j < k; ++j) if (nb[j] == (int)s) { if (seen == occ) break; ++seen; }
        const float ev = us + a.v[(size_t)q * a.ldv + c];
        const float xh = (ev - mu) * rs;
        float de = -rs * (m1 + xh * m2);
        if (j == (int)a.amax[(size_t)q * C + c]) {
            const float y = xh * ga + be;
            de += rs * a.go[(size_t)q * a.ldgo + c] * (y > 0.f ? 1.f : a.slope) * ga;
        }
        acc += (double)de;
    }
    gu[(size_t)s * ldgu + c] = (float)acc;
}

// See the block comment above.  B samples of nq queries each (Pq = B * nq); G divides C; k <= 255.
extern "C" int vnp_edge_gn_max(hipStream_t st, int mode, const float* u, int ldu, const float* v, int ldv, const int* nbr, int B, int nq, int k,
                               int C, int G, const float* mean, const float* rstd, const float* gamma, const float* beta, float slope,
                               float* out, int ldo, unsigned char* amax, const float* go, int ldgo, float* partial, const float* sums,
                               float* gv, int ldgv)
{
    if (!u || !v || !nbr || B <= 0 || nq <= 0 || k <= 0 || k > 255 || C <= 0 || G <= 0 || C % G || mode < 0 || mode > 3) return VNP_EINVAL;
    if (mode >= 1 && (!mean || !rstd || !gamma || !beta)) return VNP_EINVAL;
    if ((mode == 0 || mode == 2) && !partial) return VNP_EINVAL;
    if (mode == 1 && (!out || !amax)) return VNP_EINVAL;
    if (mode >= 2 && (!go || !amax)) return VNP_EINVAL;
    if (mode == 3 && (!gv || !sums)) return VNP_EINVAL;
    const long long nblk = (long long)B * ((nq + EGN_QB - 1) / EGN_QB);
    if (nblk > 65535) return VNP_EINVAL;
    VnpEdgeGnArgs a{};
    a.u = u; a.ldu = ldu; a.v = v; a.ldv = ldv; a.nbr = nbr; a.Pq = (long long)B * nq; a.k = k; a.C = C; a.G = G; a.nq = nq;
    a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.slope = slope;
    a.out = out; a.ldo = ldo; a.amax = amax; a.go = go; a.ldgo = ldgo; a.partial = partial; a.sums = sums; a.gv = gv; a.ldgv = ldgv;
    const dim3 grid(vnp_cdiv(C, 64), (unsigned)nblk), block(256);
    switch (mode) {
    case 0: hipLaunchKernelGGL(vnp_edge_gn_kernel<0>, grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL(vnp_edge_gn_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(vnp_edge_gn_kernel<2>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(vnp_edge_gn_kernel<3>, grid, block, 0, st, a); break;
    }
    return vnp_last_error();
}

// partial: mode-0 output of vnp_edge_gn_max -> mean, rstd [B][G]
extern "C" int vnp_edge_gn_finalize(hipStream_t st, const float* partial, int B, int nq, int k, int C, int G, float eps, float* mean, float* rstd)
{
    if (!partial || !mean || !rstd || B <= 0 || nq <= 0 || k <= 0 || C <= 0 || G <= 0 || C % G) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_edge_gn_finalize_kernel, dim3(vnp_cdiv(B * G, 64)), dim3(64), 0, st, partial, B, nq, k, C, G, eps, mean, rstd);
    return vnp_last_error();
}

// gu [Pk][ldgu]: the neighbour term's gradient, a gather over the reverse adjacency (start [Pk + 1], rev [B * nq * k] from vnp_edge_reverse)
extern "C" int vnp_edge_gn_gather(hipStream_t st, const float* u, int ldu, const float* v, int ldv, const int* nbr, const int* start,
                                  const int* rev, long long Pk, int B, int nq, int k, int C, int G, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, float slope, const unsigned char* amax, const float* go, int ldgo,
                                  const float* sums, float* gu, int ldgu)
{
    if (!u || !v || !nbr || !start || !rev || !mean || !rstd || !gamma || !beta || !amax || !go || !sums || !gu) return VNP_EINVAL;
    if (Pk <= 0 || B <= 0 || nq <= 0 || k <= 0 || k > 255 || C <= 0 || G <= 0 || C % G || (Pk + 3) / 4 > 65535LL * 32768LL) return VNP_EINVAL;
    VnpEdgeGnArgs a{};
    a.u = u; a.ldu = ldu; a.v = v; a.ldv = ldv; a.nbr = nbr; a.Pq = (long long)B * nq; a.k = k; a.C = C; a.G = G; a.nq = nq;
    a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.slope = slope; a.amax = (unsigned char*)amax; a.go = go; a.ldgo = ldgo;
    a.sums = sums;
    const long long nb = (Pk + 3) / 4;
    for (long long b0 = 0; b0 < nb; b0 += 65535) {
        const int n = (int)((nb - b0) < 65535 ? (nb - b0) : 65535);
        hipLaunchKernelGGL(vnp_edge_gn_gather_kernel, dim3(vnp_cdiv(C, 64), n), dim3(256), 0, st, a, start, rev, b0 * 4, Pk, gu, ldgu);
    }
    return vnp_last_error();
}

// ---- y = act(x + bias) on rows [P][C] (bias [C] or NULL; act 0 none / 1 ReLU / 2 LeakyReLU) and its input gradient gx = gy act'(y):
// the bias adds and the ReLU of the baselines' linear heads (the bias gradient is a column sum: vnp_colsum / vnp_reduce_rows)
__global__ void vnp_bias_act_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ bias, long long P, int C, int act, float slope,
                                    float* __restrict__ y, int ldy, const float* __restrict__ gy, int ldgy, float* __restrict__ gx, int ldgx)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * C) return;
    const long long r = i / C; const int c = (int)(i % C);
    const float t = x[r * ldx + c] + (bias ? bias[c] : 0.f);
    if (y) y[r * ldy + c] = vnp_b1_act(t, act, slope);
    if (gx) gx[r * ldgx + c] = gy[r * ldgy + c] * vnp_b1_dact(t, act, slope);
}

// forward: y != NULL; backward: gy, gx != NULL (x is the forward's input again: the pre-activation is recomputed)
extern "C" int vnp_bias_act(hipStream_t st, const float* x, int ldx, const float* bias, long long P, int C, int act, float slope,
                            float* y, int ldy, const float* gy, int ldgy, float* gx, int ldgx)
{
    if (!x || P <= 0 || C <= 0 || act < 0 || act > 2 || (!y && !gx) || (gx && !gy)) return VNP_EINVAL;
    const long long total = P * C;
    if ((total + 255) / 256 > 2147483647LL) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_bias_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, ldx, bias, P, C, act, slope, y, ldy, gy, ldgy,
                       gx, ldgx);
    return vnp_last_error();
}

// ---- out[b][c] = max over the n rows of sample b of x[(b n + i)][c], amax[b][c] = the first i attaining it; backward: gx = go at that row, else 0
__global__ void vnp_rows_max_kernel(const float* __restrict__ x, int ldx, int B, int n, int C, float* __restrict__ out, int* __restrict__ amax)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C;
    float best = -INFINITY; int bi = 0;
    for (int r = 0; r < n; ++r) { const float v = x[((size_t)b * n + r) * ldx + c]; if (v > best) { best = v; bi = r; } }
    out[i] = best; amax[i] = bi;
}
__global__ void vnp_rows_max_bwd_kernel(const float* __restrict__ go, const int* __restrict__ amax, int B, int n, int C, float* __restrict__ gx)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * n * C) return;
    const int c = (int)(i % C); const long long rr = i / C; const int r = (int)(rr % n), b = (int)(rr / n);
    gx[i] = amax[(size_t)b * C + c] == r ? go[(size_t)b * C + c] : 0.f;
}
extern "C" int vnp_rows_max(hipStream_t st, const float* x, int ldx, int B, int n, int C, float* out, int* amax)
{
    if (!x || !out || !amax || B <= 0 || n <= 0 || C <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_rows_max_kernel, dim3(vnp_cdiv(B * C, 256)), dim3(256), 0, st, x, ldx, B, n, C, out, amax);
    return vnp_last_error();
}
extern "C" int vnp_rows_max_bwd(hipStream_t st, const float* go, const int* amax, int B, int n, int C, float* gx)
{
    if (!go || !amax || !gx || B <= 0 || n <= 0 || C <= 0) return VNP_EINVAL;
    const long long total = (long long)B * n * C;
    hipLaunchKernelGGL(vnp_rows_max_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, go, amax, B, n, C, gx);
    return vnp_last_error();
}
