// Fused EdgeConv (models/dgcnn.py:250-278 + the conv blocks of VN_DGCNN_fps and of the VN-PoinTr grouper) without its edge tensor, the
// reverse adjacency its gradient gather walks, and their C-ABI (vnp_edge_fused*, vnp_edge_reverse).
#include "vn_kernels.h"

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

// CW (template): channel lanes.  With 64 a query lane IS a wave: its index, the query, the neighbour list and every row base are SCALARS
// (v_readfirstlane of the wave index) — the neighbour ids arrive through scalar loads, a gather's address is one SGPR base + the lane's
// column, and the 64-bit per-lane address arithmetic of every edge (round 5: a third of the pass's vector instructions) is gone.
// base (a scalar where the caller's row is one) + the lane's BYTE offset as an unsigned 32-bit value: the form the saddr addressing of
// global_load / global_store takes (SGPR pair + one VGPR), instead of a 64-bit per-lane pointer built for every access
__device__ __forceinline__ float vnp_ldo(const float* base, unsigned boff) { return *(const float*)((const char*)base + boff); }
__device__ __forceinline__ void vnp_sto(float* base, unsigned boff, float v) { *(float*)((char*)base + boff) = v; }

template <int MODE, bool R1 = false, int CW = 64>
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
    constexpr int NQ = 256 / CW;
    const int cx = threadIdx.x & (CW - 1), py = CW == 64 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)threadIdx.x / CW;
    int bx, by;
    vnp_xcd_block(bx, by);
    const int c = bx * CW + cx;
    const bool live = c < C;
    const int cc = live ? c : 0;
    const unsigned cb = (unsigned)cc * 4u, cbs = (unsigned)c * 4u;          // the lane's column as a byte offset (loads: clamped; stores: live lanes only)
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
            // (row bases first, the lane's column last: with a scalar query the bases stay in SGPRs and a load is base + one 32-bit lane offset)
            const float* rv = a.v + (size_t)q * 3 * a.ldv;
            vp0 = vnp_ldo(rv, cb); vp1 = vnp_ldo(rv + a.ldv, cb); vp2 = vnp_ldo(rv + 2 * a.ldv, cb);
            vd0 = vnp_ldo(rv + C, cb); vd1 = vnp_ldo(rv + a.ldv + C, cb); vd2 = vnp_ldo(rv + 2 * a.ldv + C, cb);
        }
        float ge0 = 0.f, ge1 = 0.f, ge2 = 0.f;
        if (MODE >= 2) {
            const float* rg = a.go + (size_t)q * 3 * a.ldgo;
            ge0 = vnp_ldo(rg, cb) * kinv; ge1 = vnp_ldo(rg + a.ldgo, cb) * kinv; ge2 = vnp_ldo(rg + 2 * a.ldgo, cb) * kinv;     // vnp_mean_pool_bwd: g / gs (gradients: 1 ulp, as EDIV)
        }
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, h0 = 0.f, h1 = 0.f, h2 = 0.f;                 // MODE 1: sum of outputs; MODE 3: gv (p | d)
        const int* nb = a.nbr + (size_t)q * k;
#pragma unroll 4
        for (int j = 0; j < k; ++j) {
            const size_t urow = (size_t)nb[j] * 3;
            const float* ru = R1 ? a.u + urow : a.u + urow * a.ldu;                       // row base (a scalar when the query is one)
            float x0, x1, x2, t0 = 0.f, t1 = 0.f, t2 = 0.f;
            if (R1) {
                t0 = ru[0]; t1 = ru[1]; t2 = ru[2];                                          // the neighbour's coordinates: 12 bytes, not 24 C
                x0 = vnp_mul_alone(t0, wup) + vp0; x1 = vnp_mul_alone(t1, wup) + vp1; x2 = vnp_mul_alone(t2, wup) + vp2;
            } else { x0 = vnp_ldo(ru, cb) + vp0; x1 = vnp_ldo(ru + a.ldu, cb) + vp1; x2 = vnp_ldo(ru + 2 * a.ldu, cb) + vp2; }
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
            else { d0 = vnp_ldo(ru + C, cb) + vd0; d1 = vnp_ldo(ru + a.ldu + C, cb) + vd1; d2 = vnp_ldo(ru + 2 * a.ldu + C, cb) + vd2; }
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
            float* w = a.out + (size_t)q * 3 * a.ldo;
            vnp_sto(w, cbs, o0 / kf); vnp_sto(w + a.ldo, cbs, o1 / kf); vnp_sto(w + 2 * a.ldo, cbs, o2 / kf);        // vnp_mean_pool_fwd: s / gs
        }
        if (MODE == 3 && live) {
            float* w = a.gv + (size_t)q * 3 * a.ldgv;
            vnp_sto(w, cbs, o0); vnp_sto(w + a.ldgv, cbs, o1); vnp_sto(w + 2 * a.ldgv, cbs, o2);
            vnp_sto(w + C, cbs, h0); vnp_sto(w + a.ldgv + C, cbs, h1); vnp_sto(w + 2 * a.ldgv + C, cbs, h2);
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
template <bool R1 = false, int CW = 64>
__global__ void __launch_bounds__(256) vnp_edge_gather_u_kernel(const VnpEdgeArgs a, const int* __restrict__ rstart, const int* __restrict__ rev,
                                                               long long Pu)
{
    constexpr int MODE = 3;
    __shared__ double red[7 * 6 * 32];               // [NQ - 1][6][CW]: 64 channel lanes x 4 edge lanes, or 32 x 8 for C <= 32
    const int C = a.C;
    constexpr int NQ = 256 / CW;
    const int cx = threadIdx.x & (CW - 1), py = CW == 64 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)threadIdx.x / CW;
    int bx, by;
    vnp_xcd_block(bx, by);
    const int c = bx * CW + cx;
    const bool live = c < C;
    const int cc = live ? c : 0;
    const unsigned cb = (unsigned)cc * 4u, cbs = (unsigned)c * 4u;
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
            const float* ru = a.u + (size_t)sp * 3 * a.ldu;
            up0 = vnp_ldo(ru, cb); up1 = vnp_ldo(ru + a.ldu, cb); up2 = vnp_ldo(ru + 2 * a.ldu, cb);
            ud0 = vnp_ldo(ru + C, cb); ud1 = vnp_ldo(ru + a.ldu + C, cb); ud2 = vnp_ldo(ru + 2 * a.ldu + C, cb);
        }
        double F0 = 0.0, F1 = 0.0, F2 = 0.0, E0 = 0.0, E1 = 0.0, E2 = 0.0;
        const int e1 = rstart[sp + 1];
#pragma unroll 2
        for (int e = rstart[sp] + py; e < e1; e += NQ) {
            const size_t q = (size_t)rev[e];
            const float* rv = R1 ? a.v + q * 3 : a.v + q * 3 * a.ldv;
            const float* rg = a.go + q * 3 * a.ldgo;
            float x0, x1, x2, d0, d1, d2;
            if (R1) {
                const float c0 = rv[0], c1 = rv[1], c2 = rv[2];
                x0 = up0 + vnp_mul_alone(c0, wvp); x1 = up1 + vnp_mul_alone(c1, wvp); x2 = up2 + vnp_mul_alone(c2, wvp);
                d0 = ud0 + vnp_mul_alone(c0, wvd); d1 = ud1 + vnp_mul_alone(c1, wvd); d2 = ud2 + vnp_mul_alone(c2, wvd);
            } else {
                x0 = up0 + vnp_ldo(rv, cb); x1 = up1 + vnp_ldo(rv + a.ldv, cb); x2 = up2 + vnp_ldo(rv + 2 * a.ldv, cb);
                d0 = ud0 + vnp_ldo(rv + C, cb); d1 = ud1 + vnp_ldo(rv + a.ldv + C, cb); d2 = ud2 + vnp_ldo(rv + 2 * a.ldv + C, cb);
            }
            const float ge0 = vnp_ldo(rg, cb) * kinv, ge1 = vnp_ldo(rg + a.ldgo, cb) * kinv, ge2 = vnp_ldo(rg + 2 * a.ldgo, cb) * kinv;
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
            float* w = a.gu + (size_t)sp * 3 * a.ldgu;
            vnp_sto(w, cbs, (float)F0); vnp_sto(w + a.ldgu, cbs, (float)F1); vnp_sto(w + 2 * a.ldgu, cbs, (float)F2);
            vnp_sto(w + C, cbs, (float)E0); vnp_sto(w + a.ldgu + C, cbs, (float)E1); vnp_sto(w + 2 * a.ldgu + C, cbs, (float)E2);
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
#define VNP_EDGE_GO(R1V, CWV) switch (mode) { \
        case 0: hipLaunchKernelGGL((vnp_edge_fused_kernel<0, R1V, CWV>), grid, block, 0, st, s); break; \
        case 1: hipLaunchKernelGGL((vnp_edge_fused_kernel<1, R1V, CWV>), grid, block, 0, st, s); break; \
        case 2: hipLaunchKernelGGL((vnp_edge_fused_kernel<2, R1V, CWV>), grid, block, 0, st, s); break; \
        default: hipLaunchKernelGGL((vnp_edge_fused_kernel<3, R1V, CWV>), grid, block, 0, st, s); break; }
        if (C <= 32) { if (r1) VNP_EDGE_GO(true, 32) else VNP_EDGE_GO(false, 32) }
        else { if (r1) VNP_EDGE_GO(true, 64) else VNP_EDGE_GO(false, 64) }
#undef VNP_EDGE_GO
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
        if (cw == 32) {
            if (wu) hipLaunchKernelGGL((vnp_edge_gather_u_kernel<true, 32>), dim3(vnp_cdiv(C, cw), nb), dim3(256), 0, st, s, start + soff, rev, np);
            else    hipLaunchKernelGGL((vnp_edge_gather_u_kernel<false, 32>), dim3(vnp_cdiv(C, cw), nb), dim3(256), 0, st, s, start + soff, rev, np);
        } else {
            if (wu) hipLaunchKernelGGL((vnp_edge_gather_u_kernel<true, 64>), dim3(vnp_cdiv(C, cw), nb), dim3(256), 0, st, s, start + soff, rev, np);
            else    hipLaunchKernelGGL((vnp_edge_gather_u_kernel<false, 64>), dim3(vnp_cdiv(C, cw), nb), dim3(256), 0, st, s, start + soff, rev, np);
        }
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

