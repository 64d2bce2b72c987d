// FoldingNet's first layer end to end without its pre-activation tensor (models/pcn.py:335-346, 375-385; the attention decoder's
// folding stages, models/pcn.py:479-492, in the per-group form): kernels and C-ABI of vnp_fold_fused / vnp_group_fold_fused.
#include "vn_kernels.h"

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
__global__ void __launch_bounds__(256, (MODE == 3 && PK && !GRP) ? 3 : 1) vnp_fold_fused_kernel(const VnpFoldArgs a)
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
    // the wave index as a SCALAR: the coarse point, the sample, the seed pointer and every row base derive from it, so their arithmetic
    // and the loads of coordinates / seeds run on the scalar unit and a gradient load is one SGPR base + the lane's column (round 5:
    // with a per-lane wave index all of it was 64-bit vector arithmetic held in registers across the seed loop)
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    // the packed MODE 3 body keeps these ten double sums in its own LDS slots (redd[py][r][h][cx], touched once per coarse point) instead
    // of in 20 registers that would be live across the whole seed loop: 202 -> 178 VGPRs
    constexpr bool ACC_LDS = MODE == 3 && PK;
    if (ACC_LDS) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 5; ++r) redd[py][r][h][cx] = 0.0;
    }
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
                for (int r = 0; r < 5; ++r) {
                    if (ACC_LDS) redd[py][r][h][cx] += (double)vacc[h][r][0] + (double)vacc[h][r][1];
                    else acc[h][r] += (double)vacc[h][r][0] + (double)vacc[h][r][1];
                }
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
        if (!ACC_LDS) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 5; ++r) redd[py][r][h][cx] = acc[h][r];
        }
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

