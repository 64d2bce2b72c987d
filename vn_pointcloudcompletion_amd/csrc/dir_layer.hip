// VNLinearLeakyReLU on a map whose every VN vector is parallel to ONE direction per point (models/vn_layers.py:46-74 applied behind
// models/pcn.py:165-173).
//
// The first layer of VN_PointNet acts on a single VN channel — the point x itself — so whatever its weights are, each of its 128 output
// vectors at a point is a multiple of that point's unit direction u = x / |x|:  a[(n, t)][c] = u_t(n) h[n][c]  with one SCALAR per
// (point, channel).  Linear maps keep that form, so the next layer's pre-activations are
//     p[(n, t)][c] = u_t(n) v[n][c]     + bias[g(n) * 3 + t][c]          (c <  C)        v = h [W_f ; W_d]^T   (ONE row per point)
//     d[(n, t)][c] = u_t(n) v[n][C + c] + bias[g(n) * 3 + t][C + c]
// with the per-sample bias carrying the de-duplicated global channels (pcn.py:172).  The rows layout would spend three GEMM rows and
// three stored rows of p | d per point on the same v; here the GEMMs run on points ([P, K] instead of [3P, K]: a third of the FLOPs,
// forward and both gradients) and the four passes of the layer read v — a third of the bytes — and rebuild p, d in registers:
//   MODE 0  Welford partials of n = |p| + eps per chunk of points            (-> vnp_vn_bn_finalize_chunked)
//   MODE 1  out[(n, t)][c] = leaky(bn(p), d)                                  (rows layout: the next layer is a generic one)
//   MODE 2  backward sums S1 = sum g_nhat, S2 = sum g_nhat * xhat per chunk   (= dL/dbeta, dL/dgamma)
//   MODE 3  backward: gv[n][c] = sum_t u_t (dL/dp)_t, gv[n][C + c] = sum_t u_t (dL/dd)_t, and the per-chunk column sums of
//           (dL/dp | dL/dd) per component (-> the bias gradient)
// The per-element arithmetic is that of vnp_bn_leaky_fwd / bwd1<1> / bwd1<2> (vn_ops.hip) term for term; like there, the gradient
// modes use v_rcp_f32 / v_sqrt_f32 and the forward modes IEEE division and sqrtf.  u gets no gradient (it depends on the input only).
// scalar_io (vnp_vn_dir_layer_ex; bias == NULL): the first layer itself — its input is the point, h = |x|, and without a bias its OUTPUT
// is parallel to u again — returns per-point scalars sum_t u_t out_t [P][C] and takes the upstream gradient as scalars (dL/dout_t = u_t g).
// block = 64 channels x 4 point lanes; grid = (ceil(C / 64), ceil(P / chunk)); a chunk never straddles two groups (the host checks).
#include "vnp_common.h"

struct VnpDirArgs {
    const float* v; int ldv;              // [P, 2C]
    const float* u;                       // [P, 3] unit directions
    const float* bias; int ldb;           // [G * 3, 2C] or nullptr
    long long P; int C, ppg, chunk;       // ppg = points per group
    const float* mean; const float* rstd; const float* gamma; const float* beta; float ns;
    float* out; int ldo;                  // MODE 1
    const float* go; int ldgo;            // MODE 2, 3: [P * 3, C]
    float2* partial;                      // MODE 0: (mean, M2); MODE 2: (S1, S2)   [nchunk][C]
    const float* S; float inv_count;      // MODE 3: reduced sums [C][2] (nullptr in eval mode)
    float* gv; int ldgv;                  // MODE 3: [P, 2C]
    float* colsum;                        // MODE 3: [nchunk][3][2C] or nullptr
    int scalar_io;                        // the OUTPUT is parallel to u as well (no bias: the single-channel first layer itself): out and go
                                          // are per-point scalars [P][C] (out = sum_t u_t out_t; dL/dout_t = u_t go) instead of rows [P*3][C]
};

__device__ __forceinline__ float vnp_dl_sq3(float a, float b, float c) { return a * a + b * b + c * c; }

__device__ __forceinline__ void vnp_dl_chan_merge(double& na, double& ma, double& Ma, double nb, double mb, double Mb)
{
    if (nb <= 0.0) return;
    const double n = na + nb, delta = mb - ma;
    ma += delta * nb / n;
    Ma += Mb + delta * delta * na * nb / n;
    na = n;
}

template <int MODE>
__global__ void __launch_bounds__(256) vnp_dir_layer_kernel(const VnpDirArgs a)
{
#define DDIV(x, y) (MODE >= 2 ? (x) * __builtin_amdgcn_rcpf(y) : (x) / (y))
#define DSQRT(x) (MODE >= 2 ? __builtin_amdgcn_sqrtf(x) : sqrtf(x))
    __shared__ float2 red[4][64];
    __shared__ double2 redd[MODE == 2 ? 4 : 1][64];
    __shared__ float red6[MODE == 3 ? 4 : 1][6][64];
    __shared__ int cnt[4];
    // (MODE 3 measured 4.6 % slower with the scalar index and keeps the vector one; modes 0 and 2: -10 %)
    const int cx = threadIdx.x & 63, py = MODE == 3 ? (int)(threadIdx.x >> 6) : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = blockIdx.x * 64 + cx;
    const int C = a.C;
    const bool live = c < C;
    const long long p0 = (long long)blockIdx.y * a.chunk;
    const long long p1 = min(a.P, p0 + a.chunk);
    float mu = 0.f, rs = 0.f, ga = 0.f, be = 0.f;
    if (live && MODE >= 1) { mu = a.mean[c]; rs = a.rstd[c]; ga = a.gamma[c]; be = a.beta[c]; }
    const float m1 = (MODE == 3 && a.S && live) ? a.S[2 * c] * a.inv_count : 0.f;
    const float m2 = (MODE == 3 && a.S && live) ? a.S[2 * c + 1] * a.inv_count : 0.f;
    // the chunk lies inside one group: its bias rows are loop-invariant
    float bp0 = 0.f, bp1 = 0.f, bp2 = 0.f, bd0 = 0.f, bd1 = 0.f, bd2 = 0.f;
    if (live && a.bias) {
        const float* b = a.bias + (size_t)(p0 / a.ppg) * 3 * a.ldb + c;
        bp0 = b[0]; bp1 = b[a.ldb]; bp2 = b[2 * (size_t)a.ldb];
        bd0 = b[C]; bd1 = b[a.ldb + C]; bd2 = b[2 * (size_t)a.ldb + C];
    }
    float wmean = 0.f, wm2 = 0.f; int wk = 0;                 // MODE 0
    double s1 = 0.0, s2 = 0.0;                                // MODE 2
    float cs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};             // MODE 3
    if (live) {
        // U points in flight per thread (their loads are issued before the first dependent arithmetic): with one point per iteration
        // the passes ran at 1.6 - 2.9 TB/s, latency-bound on 65536 points x 1024 channels
        constexpr int U = MODE <= 1 ? 4 : 2;
        for (long long pt0 = p0 + py; pt0 < p1; pt0 += 4 * U) {
            float u0[U], u1[U], u2[U], vp[U], vd[U], g0[U], g1[U], g2[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const long long pt = pt0 + 4 * i;
                const bool ok = pt < p1;
                const long long q = ok ? pt : p0;                  // a valid address; the lane's result is dropped below
                u0[i] = a.u[q * 3]; u1[i] = a.u[q * 3 + 1]; u2[i] = a.u[q * 3 + 2];
                vp[i] = __builtin_nontemporal_load(a.v + (size_t)q * a.ldv + c);
                vd[i] = MODE >= 1 ? __builtin_nontemporal_load(a.v + (size_t)q * a.ldv + C + c) : 0.f;
                if (MODE >= 2) {
                    if (a.scalar_io) {
                        const float gh = __builtin_nontemporal_load(a.go + (size_t)q * a.ldgo + c);
                        g0[i] = u0[i] * gh; g1[i] = u1[i] * gh; g2[i] = u2[i] * gh;
                    } else {
                        const float* rg = a.go + (size_t)q * 3 * a.ldgo + c;
                        g0[i] = __builtin_nontemporal_load(rg); g1[i] = __builtin_nontemporal_load(rg + a.ldgo); g2[i] = __builtin_nontemporal_load(rg + 2 * (size_t)a.ldgo);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const long long pt = pt0 + 4 * i;
                if (pt >= p1) break;
                const float x0 = fmaf(u0[i], vp[i], bp0), x1 = fmaf(u1[i], vp[i], bp1), x2 = fmaf(u2[i], vp[i], bp2);
                const float n = DSQRT(vnp_dl_sq3(x0, x1, x2)) + VNP_EPS;
                if (MODE == 0) {
                    ++wk;
                    const float dl = n - wmean;
                    wmean += dl / (float)wk;
                    wm2 += dl * (n - wmean);
                    continue;
                }
                const float d0 = fmaf(u0[i], vd[i], bd0), d1 = fmaf(u1[i], vd[i], bd1), d2 = fmaf(u2[i], vd[i], bd2);
                const float xh = (n - mu) * rs;
                const float s = DDIV(xh * ga + be, n);
                const float q0 = x0 * s, q1 = x1 * s, q2 = x2 * s;
                const float dot = q0 * d0 + q1 * d1 + q2 * d2;
                if (MODE == 1) {
                    float o0 = q0, o1 = q1, o2 = q2;
                    if (dot < 0.f) {
                        const float al = (1.f - a.ns) * dot / (vnp_dl_sq3(d0, d1, d2) + VNP_EPS);
                        o0 -= al * d0; o1 -= al * d1; o2 -= al * d2;
                    }
                    if (a.scalar_io) { __builtin_nontemporal_store(u0[i] * o0 + u1[i] * o1 + u2[i] * o2, a.out + (size_t)pt * a.ldo + c); continue; }
                    float* o = a.out + (size_t)pt * 3 * a.ldo + c;
                    __builtin_nontemporal_store(o0, o); __builtin_nontemporal_store(o1, o + a.ldo); __builtin_nontemporal_store(o2, o + 2 * (size_t)a.ldo);
                    continue;
                }
                float h0 = g0[i], h1 = g1[i], h2 = g2[i];           // dL/dout, turned into dL/dq and then dL/dp below
                float e0 = 0.f, e1 = 0.f, e2 = 0.f;                 // dL/dd
                if (dot < 0.f) {
                    const float cc = 1.f - a.ns;
                    const float inv = DDIV(1.f, vnp_dl_sq3(d0, d1, d2) + VNP_EPS);
                    const float al = dot * inv;
                    const float gdd = h0 * d0 + h1 * d1 + h2 * d2;
                    const float k1 = cc * gdd * inv;
                    e0 = -cc * (al * h0 + gdd * (q0 * inv - 2.f * al * inv * d0));
                    e1 = -cc * (al * h1 + gdd * (q1 * inv - 2.f * al * inv * d1));
                    e2 = -cc * (al * h2 + gdd * (q2 * inv - 2.f * al * inv * d2));
                    h0 -= k1 * d0; h1 -= k1 * d1; h2 -= k1 * d2;
                }
                const float gs = h0 * x0 + h1 * x1 + h2 * x2;
                const float gnh = DDIV(gs, n);
                if (MODE == 2) { s1 += (double)gnh; s2 += (double)(gnh * xh); continue; }
                // MODE 3: dL/dp = s * dL/dq + gn * p / |p|
                const float pn = n - VNP_EPS;
                const float nh = xh * ga + be;
                const float gn = ga * rs * (gnh - m1 - xh * m2) - DDIV(gs * nh, n * n);
                const float k = pn > 0.f ? DDIV(gn, pn) : 0.f;
                h0 = s * h0 + k * x0; h1 = s * h1 + k * x1; h2 = s * h2 + k * x2;
                float* gr = a.gv + (size_t)pt * a.ldgv + c;
                __builtin_nontemporal_store(u0[i] * h0 + u1[i] * h1 + u2[i] * h2, gr);
                __builtin_nontemporal_store(u0[i] * e0 + u1[i] * e1 + u2[i] * e2, gr + C);
                cs[0] += h0; cs[1] += h1; cs[2] += h2; cs[3] += e0; cs[4] += e1; cs[5] += e2;
            }
        }
    }
    if (MODE == 0) {
        red[py][cx] = make_float2(wmean, wm2);
        if (cx == 0) cnt[py] = (int)((p1 - p0 - py + 3) / 4 > 0 ? (p1 - p0 - py + 3) / 4 : 0);
        __syncthreads();
        if (py == 0 && live) {
            double na = cnt[0], ma = red[0][cx].x, Ma = red[0][cx].y;
#pragma unroll
            for (int j = 1; j < 4; ++j) vnp_dl_chan_merge(na, ma, Ma, (double)cnt[j], (double)red[j][cx].x, (double)red[j][cx].y);
            a.partial[(size_t)blockIdx.y * C + c] = make_float2((float)ma, (float)Ma);
        }
    } else if (MODE == 2) {
        redd[py][cx] = make_double2(s1, s2);
        __syncthreads();
        if (py == 0 && live) {
            const double2 r0 = redd[0][cx], r1 = redd[1][cx], r2 = redd[2][cx], r3 = redd[3][cx];
            a.partial[(size_t)blockIdx.y * C + c] = make_float2((float)((r0.x + r1.x) + (r2.x + r3.x)), (float)((r0.y + r1.y) + (r2.y + r3.y)));
        }
    } else if (MODE == 3 && a.colsum) {
#pragma unroll
        for (int q = 0; q < 6; ++q) red6[py][q][cx] = cs[q];
        __syncthreads();
        if (py == 0 && live) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float* dst = a.colsum + ((size_t)blockIdx.y * 3 + t) * 2 * C;
                dst[c] = (red6[0][t][cx] + red6[1][t][cx]) + (red6[2][t][cx] + red6[3][t][cx]);
                dst[C + c] = (red6[0][3 + t][cx] + red6[1][3 + t][cx]) + (red6[2][3 + t][cx] + red6[3][3 + t][cx]);
            }
        }
    }
#undef DDIV
#undef DSQRT
}

// mode 0: statistics partials [nchunk][C][2]; 1: forward; 2: backward sums [nchunk][C][2]; 3: backward gradients.
// chunk_points must divide points_per_group when bias is given (a chunk takes its bias rows from one group).
// scalar_io != 0 (bias must be NULL): the layer's OUTPUT is parallel to u too, so out [P][ldo] and go [P][ldgo] are per-point scalars
extern "C" int vnp_vn_dir_layer_ex(hipStream_t st, int mode, const float* v, int ldv, const float* u, const float* bias, int ldb,
                                   long long P, int C, int points_per_group, int chunk_points,
                                   const float* mean, const float* rstd, const float* gamma, const float* beta, float negative_slope,
                                   float* out, int ldo, const float* go, int ldgo, float* partial, const float* S, long long count,
                                   float* gv, int ldgv, float* colsum, int scalar_io)
{
    if (scalar_io && bias) return VNP_EINVAL;
    if (!v || !u || P <= 0 || C <= 0 || mode < 0 || mode > 3 || chunk_points < 4 || chunk_points > 1024) return VNP_EINVAL;
    if (bias && (points_per_group <= 0 || points_per_group % chunk_points != 0 || P % points_per_group != 0)) return VNP_EINVAL;
    if (mode > 0 && (!mean || !rstd || !gamma || !beta)) return VNP_EINVAL;
    if ((mode == 0 || mode == 2) && !partial) return VNP_EINVAL;
    if (mode == 1 && !out) return VNP_EINVAL;
    if (mode >= 2 && !go) return VNP_EINVAL;
    if (mode == 3 && !gv) return VNP_EINVAL;
    if (vnp_cdiv(P, chunk_points) > 65535) return VNP_EINVAL;
    VnpDirArgs a;
    a.v = v; a.ldv = ldv; a.u = u; a.bias = bias; a.ldb = ldb; a.P = P; a.C = C;
    a.ppg = points_per_group > 0 ? points_per_group : 1; a.chunk = chunk_points;
    a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.ns = negative_slope;
    a.out = out; a.ldo = ldo; a.go = go; a.ldgo = ldgo; a.partial = (float2*)partial;
    a.S = S; a.inv_count = count > 0 ? 1.0f / (float)count : 0.f; a.gv = gv; a.ldgv = ldgv; a.colsum = colsum;
    a.scalar_io = scalar_io ? 1 : 0;
    dim3 grid(vnp_cdiv(C, 64), vnp_cdiv(P, chunk_points)), block(256);
    if (mode == 0) hipLaunchKernelGGL(vnp_dir_layer_kernel<0>, grid, block, 0, st, a);
    else if (mode == 1) hipLaunchKernelGGL(vnp_dir_layer_kernel<1>, grid, block, 0, st, a);
    else if (mode == 2) hipLaunchKernelGGL(vnp_dir_layer_kernel<2>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(vnp_dir_layer_kernel<3>, grid, block, 0, st, a);
    return vnp_last_error();
}

extern "C" int vnp_vn_dir_layer(hipStream_t st, int mode, const float* v, int ldv, const float* u, const float* bias, int ldb,
                                long long P, int C, int points_per_group, int chunk_points,
                                const float* mean, const float* rstd, const float* gamma, const float* beta, float negative_slope,
                                float* out, int ldo, const float* go, int ldgo, float* partial, const float* S, long long count,
                                float* gv, int ldgv, float* colsum)
{
    return vnp_vn_dir_layer_ex(st, mode, v, ldv, u, bias, ldb, P, C, points_per_group, chunk_points, mean, rstd, gamma, beta, negative_slope,
                               out, ldo, go, ldgo, partial, S, count, gv, ldgv, colsum, 0);
}

// u[n] = x[n] / |x[n]|  (0 for the origin); r (nullable) [P] = |x[n]|
__global__ void __launch_bounds__(256) vnp_unit_dirs_kernel(const float* __restrict__ xyz, long long P, float* __restrict__ u, float* __restrict__ rr)
{
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= P) return;
    const float x0 = xyz[n * 3], x1 = xyz[n * 3 + 1], x2 = xyz[n * 3 + 2];
    const float r = sqrtf(vnp_dl_sq3(x0, x1, x2));
    const float inv = r > 0.f ? 1.f / r : 0.f;
    u[n * 3] = x0 * inv; u[n * 3 + 1] = x1 * inv; u[n * 3 + 2] = x2 * inv;
    if (rr) rr[n] = r;
}

extern "C" int vnp_unit_dirs_r(hipStream_t st, const float* xyz, long long P, float* u, float* r)
{
    if (!xyz || !u || P <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_unit_dirs_kernel, dim3(vnp_cdiv(P, 256)), dim3(256), 0, st, xyz, P, u, r);
    return vnp_last_error();
}

extern "C" int vnp_unit_dirs(hipStream_t st, const float* xyz, long long P, float* u) { return vnp_unit_dirs_r(st, xyz, P, u, nullptr); }

// h[n][c] = sum_t u_t(n) a[(n, t)][c]   (the scalar of a map whose vectors are parallel to u);   expand = 0
// a[(n, t)][c] (+)= u_t(n) h[n][c]       (its adjoint, and the way back to the rows layout);       expand = 1 (accumulate: 2)
__global__ void __launch_bounds__(256) vnp_dir_contract_kernel(float* __restrict__ rows, int ldr, const float* __restrict__ u,
                                                              float* __restrict__ h, int ldh, long long P, int C, int expand)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    if (c >= C) return;
    for (long long n = (long long)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); n < P; n += (long long)gridDim.y * 4) {
        const float u0 = u[n * 3], u1 = u[n * 3 + 1], u2 = u[n * 3 + 2];
        float* r = rows + (size_t)n * 3 * ldr + c;
        if (expand) {
            const float hv = h[(size_t)n * ldh + c];
            if (expand == 2) { r[0] += u0 * hv; r[ldr] += u1 * hv; r[2 * (size_t)ldr] += u2 * hv; }
            else { r[0] = u0 * hv; r[ldr] = u1 * hv; r[2 * (size_t)ldr] = u2 * hv; }
        } else
            h[(size_t)n * ldh + c] = u0 * r[0] + u1 * r[ldr] + u2 * r[2 * (size_t)ldr];
    }
}

extern "C" int vnp_dir_contract(hipStream_t st, float* rows, int ldr, const float* u, float* h, int ldh, long long P, int C, int expand)
{
    if (!rows || !u || !h || P <= 0 || C <= 0 || expand < 0 || expand > 2) return VNP_EINVAL;
    const long long by = (P + 3) / 4;
    hipLaunchKernelGGL(vnp_dir_contract_kernel, dim3(vnp_cdiv(C, 64), (unsigned)(by < 16384 ? by : 16384)), dim3(256), 0, st,
                       rows, ldr, u, h, ldh, P, C, expand);
    return vnp_last_error();
}

// ------------------------------------------------------------------------------------------------ VNLinear -> VNMaxPool on scalars
// maxpool1 behind the single-channel first layer (models/pcn.py:168-169, models/vn_layers.py:153-167): the pooled map is u (x) y with
// y = h W^T, its direction u (x) y' with y' = h (Wd W)^T, so a point's score <x, d> = y y' |u|^2 = y y' needs no components at all.
// y [P][ldy]: columns [0, C) = y, [C, 2C) = y'.  Group g = points [g gs, (g + 1) gs).  idx[g][c] = first arg-max over the group of
// y y' (the origin, u = 0, has y = 0: score 0, as in the rows form), out[(g*3+t)][c] = u[n*][t] y[n*][c].
// block = 64 channels x 16 point lanes; grid (ceil(C / 64), G)
__global__ void __launch_bounds__(1024) vnp_dir_pool_fwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ u, int gs, int C,
                                                               int* __restrict__ idx, float* __restrict__ out, int ldo)
{
    __shared__ float rv[16][64];
    __shared__ int ri[16][64];
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
    const int c = blockIdx.x * 64 + cx;
    const long long g = blockIdx.y;
    float best = -INFINITY; int bi = 0x7fffffff;
    if (c < C) {
        const float* base = y + (size_t)g * gs * ldy + c;
        for (int n = py; n < gs; n += 16) {
            const float s = base[(size_t)n * ldy] * base[(size_t)n * ldy + C];
            if (s > best || bi == 0x7fffffff) { best = s; bi = n; }
        }
    }
    rv[py][cx] = best; ri[py][cx] = bi;
    __syncthreads();
    if (py == 0 && c < C) {
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const float v = rv[q][cx]; const int i = ri[q][cx];
            if (i != 0x7fffffff && (v > best || (v == best && i < bi) || bi == 0x7fffffff)) { best = v; bi = i; }
        }
        idx[(size_t)g * C + c] = bi;
        const size_t n = (size_t)g * gs + bi;
        const float yv = y[n * ldy + c];
#pragma unroll
        for (int t = 0; t < 3; ++t) out[((size_t)g * 3 + t) * ldo + c] = u[n * 3 + t] * yv;
    }
}

extern "C" int vnp_dir_pool_fwd(hipStream_t st, const float* y, int ldy, const float* u, long long G, int gs, int C, int* idx, float* out, int ldo)
{
    if (!y || !u || !idx || !out || G <= 0 || G > 65535 || gs <= 0 || C <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_dir_pool_fwd_kernel, dim3(vnp_cdiv(C, 64), (unsigned)G), dim3(1024), 0, st, y, ldy, u, gs, C, idx, out, ldo);
    return vnp_last_error();
}

// backward of y = h W^T under the pool: dL/dy has ONE entry per (group, channel), gy = sum_t u[n*][t] gout[(g*3+t)][c] at n* = idx[g][c].
// gW[c][:] = sum_g gy(g, c) h[n*(g, c)][:]                                     grid (C), 256 threads over Ci
__global__ void __launch_bounds__(256) vnp_dir_pool_bwd_w_kernel(const float* __restrict__ gout, int ldgo, const float* __restrict__ u,
                                                                const int* __restrict__ idx, const float* __restrict__ h, int ldh,
                                                                int G, int gs, int C, int Ci, float* __restrict__ gW, int ldgw)
{
    const int c = blockIdx.x;
    for (int k = threadIdx.x; k < Ci; k += 256) {
        float acc = 0.f;
#pragma unroll 8                                     // eight groups' index -> row gathers in flight
        for (int g = 0; g < G; ++g) {
            const size_t n = (size_t)g * gs + idx[(size_t)g * C + c];
            const float gy = u[n * 3] * gout[((size_t)g * 3) * ldgo + c] + u[n * 3 + 1] * gout[((size_t)g * 3 + 1) * ldgo + c] +
                             u[n * 3 + 2] * gout[((size_t)g * 3 + 2) * ldgo + c];
            acc += gy * h[n * ldh + k];
        }
        gW[(size_t)c * ldgw + k] = acc;
    }
}

// gh[n][:] = sum over the channels that picked point n of gy W[c][:]; `order`[g][j] lists the channels of group g sorted by their
// picked point, `pts`[g][j] that point (ascending): block (n, g) owns point n, finds its run in `pts` by bisection and writes the point's
// row — the sum over the run, or zeros for a point no channel picked: every row written exactly once, no atomics, no zero fill beforehand
// (as vnp_pool_lin_bwd)
__global__ void __launch_bounds__(128) vnp_dir_pool_bwd_x_kernel(const float* __restrict__ gout, int ldgo, const float* __restrict__ u,
                                                                const int* __restrict__ order, const int* __restrict__ pts,
                                                                const float* __restrict__ W, int ldw, int gs, int C, int Ci,
                                                                float* __restrict__ gh, int ldgh)
{
    const int n0 = blockIdx.x * 4, g = blockIdx.y;                     // 4 consecutive points per block: one bisection, then a walk
    const int n1 = n0 + 4 < gs ? n0 + 4 : gs;
    const int* pg = pts + (size_t)g * C;
    int lo = 0, hi = C;                               // first j with pg[j] >= n0
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (pg[mid] < n0) lo = mid + 1; else hi = mid; }
    const int* og = order + (size_t)g * C;
    int j1 = lo;
    for (int nloc = n0; nloc < n1; ++nloc) {
        const int j0 = j1;
        while (j1 < C && pg[j1] == nloc) ++j1;        // an empty run stores zeros below
        const size_t n = (size_t)g * gs + nloc;
        const float u0 = u[n * 3], u1 = u[n * 3 + 1], u2 = u[n * 3 + 2];
        for (int k = threadIdx.x; k < Ci; k += 128) {
            float acc = 0.f;
            for (int j = j0; j < j1; ++j) {
                const int c = og[j];
                const float gy = u0 * gout[((size_t)g * 3) * ldgo + c] + u1 * gout[((size_t)g * 3 + 1) * ldgo + c] + u2 * gout[((size_t)g * 3 + 2) * ldgo + c];
                acc += gy * W[(size_t)c * ldw + k];
            }
            gh[n * ldgh + k] = acc;
        }
    }
}

// gW (nullable) [C][ldgw], gh (nullable) [G gs][ldgh] (every row written); order / pts [G][C] int32 from a stable sort of idx per group
extern "C" int vnp_dir_pool_bwd(hipStream_t st, const float* gout, int ldgo, const float* u, const int* idx, const int* order, const int* pts,
                                const float* h, int ldh, const float* W, int ldw, long long G, int gs, int C, int Ci,
                                float* gW, int ldgw, float* gh, int ldgh)
{
    if (!gout || !u || !idx || G <= 0 || G > 65535 || gs <= 0 || C <= 0 || Ci <= 0) return VNP_EINVAL;
    if (gW) {
        if (!h) return VNP_EINVAL;
        hipLaunchKernelGGL(vnp_dir_pool_bwd_w_kernel, dim3(C), dim3(256), 0, st, gout, ldgo, u, idx, h, ldh, (int)G, gs, C, Ci, gW, ldgw);
    }
    if (gh) {
        if (!order || !pts || !W) return VNP_EINVAL;
        hipLaunchKernelGGL(vnp_dir_pool_bwd_x_kernel, dim3(vnp_cdiv(gs, 4), (unsigned)G), dim3(128), 0, st, gout, ldgo, u, order, pts, W, ldw, gs, C, Ci, gh, ldgh);
    }
    return vnp_last_error();
}
