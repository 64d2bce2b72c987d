// Kernels for the VN transformer blocks (models/transformer.py:25-106, models/pointr/vn_pointr.py:101-160, 366-412):
//   VNLayerNorm (models/vn_layers.py:129-150)  : LayerNorm over the CHANNEL axis of the vector norms, per point
//   rows <-> heads layout change               : [B*N*3, C] rows  <->  [B, H, N, (C/H)*3] tokens (transformer.py:91-93,103)
//   row softmax forward / backward             : attn = softmax(q k^T * scale) (transformer.py:98-99)
// The two attention matrix products run on the batched GEMM (vnp_gemm_batched: fp32 results, see gemm.hip).  All kernels are HBM-bound.
#include "vnp_common.h"

#define LN_EPS 1e-5f

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per point (4 points per block).  stats[pt] = (mean, rstd) of the C norms of that point.
// a1 / a2 (optional): the normalised tensor is s = x + a1 (+ a2) — a block's input plus its position embedding plus the previous block's
// MLP branch — and s is written to sum_out (the residual stream) by the lanes that then read it back for the norms.
__global__ void __launch_bounds__(256) vnp_ln_fwd_kernel(const float* __restrict__ x, int ldx, long long P, int C,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ out, int ldo, float2* __restrict__ stats,
                                                        const float* __restrict__ a1, int lda1, const float* __restrict__ a2, int lda2,
                                                        float* sum_out, int lds)
{
    const long long pt = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pt >= P) return;
    const float* r = x + (size_t)pt * 3 * ldx;
    if (a1) {
        float* so = sum_out + (size_t)pt * 3 * lds;
        const float* p1 = a1 + (size_t)pt * 3 * lda1;
        const float* p2 = a2 ? a2 + (size_t)pt * 3 * lda2 : nullptr;
        for (int c = lane; c < C; c += 64)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float v = r[t * ldx + c] + p1[t * lda1 + c];
                if (p2) v += p2[t * lda2 + c];
                so[t * lds + c] = v;
            }
        r = so; ldx = lds;                 // each lane re-reads only what it wrote itself
    }
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += sqrtf(r[c] * r[c] + r[ldx + c] * r[ldx + c] + r[2 * ldx + c] * r[2 * ldx + c]) + VNP_EPS;
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float n = sqrtf(r[c] * r[c] + r[ldx + c] * r[ldx + c] + r[2 * ldx + c] * r[2 * ldx + c]) + VNP_EPS;
        v += (n - mean) * (n - mean);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)C + LN_EPS);
    float* o = out + (size_t)pt * 3 * ldo;
    for (int c = lane; c < C; c += 64) {
        const float x0 = r[c], x1 = r[ldx + c], x2 = r[2 * ldx + c];
        const float n = sqrtf(x0 * x0 + x1 * x1 + x2 * x2) + VNP_EPS;
        const float sc = ((n - mean) * rstd * gamma[c] + beta[c]) / n;
        o[c] = x0 * sc; o[ldo + c] = x1 * sc; o[2 * ldo + c] = x2 * sc;
    }
    if (lane == 0 && stats) stats[pt] = make_float2(mean, rstd);
}

// Backward: gx, and per-block partial sums of dgamma / dbeta (part[block][2][C]).
__global__ void __launch_bounds__(256) vnp_ln_bwd_kernel(const float* __restrict__ go, int ldgo, const float* __restrict__ x, int ldx,
                                                        long long P, int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float2* __restrict__ stats, float* __restrict__ gx, int ldgx,
                                                        float* __restrict__ part, const float* __restrict__ gres, int ldgres)
{
    extern __shared__ float sm[];                 // [4 waves][2][C]
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long long pt = (long long)blockIdx.x * 4 + w;
    float* my = sm + (size_t)w * 2 * C;
    for (int c = lane; c < C; c += 64) { my[c] = 0.f; my[C + c] = 0.f; }
    if (pt < P) {
        const float* r = x + (size_t)pt * 3 * ldx;
        const float* g = go + (size_t)pt * 3 * ldgo;
        const float mean = stats[pt].x, rstd = stats[pt].y;
        // sums over channels of gamma*g_nl and gamma*g_nl*xhat
        float a = 0.f, b = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float x0 = r[c], x1 = r[ldx + c], x2 = r[2 * ldx + c];
            const float n = sqrtf(x0 * x0 + x1 * x1 + x2 * x2) + VNP_EPS;
            const float xh = (n - mean) * rstd;
            const float gnl = (g[c] * x0 + g[ldgo + c] * x1 + g[2 * ldgo + c] * x2) / n;
            a += gamma[c] * gnl; b += gamma[c] * gnl * xh;
            my[c] = gnl * xh; my[C + c] = gnl;               // dgamma, dbeta contributions of this point
        }
        a = wave_sum(a) / (float)C; b = wave_sum(b) / (float)C;
        float* o = gx + (size_t)pt * 3 * ldgx;
        // gres (optional): the gradient arriving on the residual path of the same tensor, added here instead of by a separate launch
        const float* gr = gres ? gres + (size_t)pt * 3 * ldgres : nullptr;
        for (int c = lane; c < C; c += 64) {
            const float x0 = r[c], x1 = r[ldx + c], x2 = r[2 * ldx + c];
            const float pn = sqrtf(x0 * x0 + x1 * x1 + x2 * x2), n = pn + VNP_EPS;
            const float xh = (n - mean) * rstd;
            const float nl = xh * gamma[c] + beta[c];
            const float g0 = g[c], g1 = g[ldgo + c], g2 = g[2 * ldgo + c];
            const float gs = g0 * x0 + g1 * x1 + g2 * x2;
            const float gnl = gs / n;
            const float gn = rstd * (gamma[c] * gnl - a - xh * b) - gs * nl / (n * n);
            const float sc = nl / n, kk = pn > 0.f ? gn / pn : 0.f;
            float o0 = sc * g0 + kk * x0, o1 = sc * g1 + kk * x1, o2 = sc * g2 + kk * x2;
            if (gr) { o0 += gr[c]; o1 += gr[ldgres + c]; o2 += gr[2 * ldgres + c]; }
            o[c] = o0; o[ldgx + c] = o1; o[2 * ldgx + c] = o2;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256)
        part[(size_t)blockIdx.x * 2 * C + i] = (sm[i] + sm[2 * C + i]) + (sm[4 * C + i] + sm[6 * C + i]);
}

// heads[((b*H + h)*N + n)*Dh + cl*3 + t] = rows[((b*N + n)*3 + t)*ld + h*(C/H) + cl],  Dh = (C/H)*3
// rows [(b, n, t)][C] <-> heads [(b, h)][n][cl * 3 + t] (C = H ch): for one token (b, n) the three rows hold [t][h][cl] and the H head
// slices hold [h][cl][t] — a transpose of the two inner index pairs inside 3 C contiguous floats per side.  A workgroup stages PN tokens
// in LDS: every global read is a contiguous run of C floats (a row) and every global write a contiguous run of 3 ch floats (a head
// slice), or the converse; the permuted LDS accesses have a stride of 3 or ch words.  (Rounds 1-2 did the permutation in the global
// address of one side: every 128-byte line of that side was touched by three waves at three times, and the counters showed 3x its bytes
// through the fabric — profiles/r03_cfg5_summary.json: 57 MB written for 19 MB read.)
template <bool TO_HEADS>
__global__ void __launch_bounds__(256) vnp_head_permute_kernel(const float* __restrict__ src, float* __restrict__ dst, int ld, int B, int N, int C, int H, int PN,
                                                              int ldh, int hg, long long gstride)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [PN][3][C] in the rows' order
    const int ch = C / H, Dh = ch * 3, per = 3 * C;
    const long long tok0 = (long long)blockIdx.x * PN, ntok = (long long)B * N;
    const int cnt = (int)(ntok - tok0 < PN ? ntok - tok0 : PN);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // a wave walks whole rows / whole head slices, its lanes the contiguous floats inside: the only divisions by run-time values are
    // per slice, not per element (with one flat index per element the kernel was bound by integer division, not by memory)
    const float* rowsp = TO_HEADS ? src : dst;
    const bool vec = (C & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)rowsp) & 15) == 0;
    auto rows_side = [&](bool load) {
        if (vec) {                                                    // float4 per lane, several rows in flight per wave
            const int c4n = C >> 2;
#pragma unroll 2
            for (int pt = wave; pt < cnt * 3; pt += 4) {
                const size_t g = ((size_t)tok0 * 3 + pt) * ld;
                for (int c4 = lane; c4 < c4n; c4 += 64) {
                    if (load) *(vnp_f4*)(tile + pt * C + c4 * 4) = *(const vnp_f4*)(src + g + c4 * 4);
                    else *(vnp_f4*)(dst + g + c4 * 4) = *(const vnp_f4*)(tile + pt * C + c4 * 4);
                }
            }
            return;
        }
        for (int pt = wave; pt < cnt * 3; pt += 4) {                  // row pt = 3 p + t of the workgroup's tokens: C contiguous floats
            const size_t g = ((size_t)tok0 * 3 + pt) * ld;
            for (int c = lane; c < C; c += 64) {
                if (load) tile[pt * C + c] = src[g + c];
                else dst[g + c] = tile[pt * C + c];
            }
        }
    };
    // heads side: lane = channel c = h ch + cl of a token; it moves the three components [t] of that channel — three LDS words a row
    // (C words) apart, consecutive lanes on consecutive words: conflict-free — to / from the 12 contiguous bytes [cl][0..2] of head h's
    // slice.  (The first LDS form of this kernel walked a slice with lane = cl * 3 + t: three lanes per bank, SQ_LDS_BANK_CONFLICT at two
    // thirds of its LDS cycles.)  No division per element: (h, cl) advance with c.
    auto heads_side = [&](bool load) {
        const int pad = ldh - Dh;
        for (int p = wave; p < cnt; p += 4) {
            const long long tok = tok0 + p;
            const int b = (int)(tok / N), n = (int)(tok - (long long)b * N);
            // heads in groups of hg (q | k | v side by side in the rows: one head tensor [B * hg][N][ldh] per group, gstride floats apart;
            // hg = H: one group): head h = gi hg + hl lives at gi gstride + ((b hg + hl) N + n) ldh
            const size_t g0 = ((size_t)b * hg * N + n) * ldh;
            auto head_off = [&](int hh) { const int gi = hh / hg; return (size_t)gi * gstride + (size_t)(hh - gi * hg) * N * ldh; };
            const float* tp = tile + p * per;
            int h = lane / ch, cl = lane - h * ch;
#pragma unroll 2
            for (int c = lane; c < C; c += 64) {
                const size_t g = g0 + head_off(h) + cl * 3;
                if (load) {
                    const float v0 = src[g], v1 = src[g + 1], v2 = src[g + 2];
                    tile[p * per + c] = v0; tile[p * per + C + c] = v1; tile[p * per + 2 * C + c] = v2;
                } else {
                    dst[g] = tp[c]; dst[g + 1] = tp[C + c]; dst[g + 2] = tp[2 * C + c];
                }
                cl += 64;
                while (cl >= ch) { cl -= ch; ++h; }
            }
            if (!load && pad > 0)
                for (int i = lane; i < H * pad; i += 64) { const int hh = i / pad, r = i - hh * pad; dst[g0 + head_off(hh) + Dh + r] = 0.f; }
        }
    };
    if (TO_HEADS) { rows_side(true); __syncthreads(); heads_side(false); }
    else          { heads_side(true); __syncthreads(); rows_side(false); }
}

// FEATURE-major head tensors: hf[(b, h)][r][n], r = cl * 3 + t < Dh, with rows [Dh, FR) zero.  They put the head dimension (48 * 3 = 144
// in Attention_VN_FoldingNet: neither a multiple of 128 nor of 32) on M of the attention's products, padded to whole row tiles, instead of
// on N / K where the bounds-checked kernels had to take it: out^T = v^T P^T, dv^T = dout^T P, dq^T = k^T dS^T, dk^T = q^T dS.
// A workgroup = one (b, h) and 64 tokens, staged in LDS as [token][Dh + 1]: the rows side moves runs of ch contiguous floats, the
// feature-major side runs of 64 consecutive tokens.
template <bool TO_HEADS>
__global__ void __launch_bounds__(256) vnp_head_permute_fm_kernel(const float* __restrict__ src, float* __restrict__ dst, int ld, int B, int N, int C, int H,
                                                                 int FR, int TT, int vec, int TS)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [TT][TS] in the rows' order [t][cl], TT <= 64 tokens
    const int ch = C / H, Dh = ch * 3;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int n0 = blockIdx.x * TT, cnt = N - n0 < TT ? N - n0 : TT;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // vec (ch % 4 == 0, 16-byte aligned rows; TS % 4 == 0 with TS / 4 odd): every LDS access is 16 bytes wide and conflict-free —
    //   rows side: a (token, component) segment is ch contiguous floats of one row; a wave moves 64 / (ch / 4) segments per instruction
    //     as float4 (ds_write_b128: 8 contiguous lanes = 128 contiguous bytes);
    //   feature-major side: lane = token reads four consecutive words [t][cl .. cl + 3] of its own row (ds_read_b128: the 16 lanes
    //     of a group are 16 distinct tokens mod 16, their addresses n * TS words apart with TS / 4 odd: 16 distinct 16-byte units) and
    //     stores them to the four feature rows (cl + j) * 3 + t, each a run of TT consecutive tokens.
    // (The first form of this kernel — scalar LDS words at a stride of 3 ch + 1 — had SQ_LDS_BANK_CONFLICT at half of its LDS cycles.)
    // Loops are unrolled so that several loads are in flight per wave.
    if (vec) {
        const int q4 = ch >> 2, per = 64 / q4, sl = lane / q4, quad = lane - sl * q4;
        const bool on = sl < per;
        const int nseg = cnt * 3;
        auto rows_side = [&](bool load) {
#pragma unroll 4
            for (int s0 = wave * per; s0 < nseg; s0 += 4 * per) {
                const int seg = s0 + sl;
                if (on && seg < nseg) {
                    const int n = seg / 3, t = seg - n * 3;
                    const size_t g = (((size_t)b * N + n0 + n) * 3 + t) * ld + h * ch + quad * 4;
                    float* tp = tile + n * TS + t * ch + quad * 4;
                    if (load) *(vnp_f4*)tp = *(const vnp_f4*)(src + g);
                    else *(vnp_f4*)(dst + g) = *(const vnp_f4*)tp;
                }
            }
        };
        auto fm_side = [&](bool load) {
            const int ngrp = Dh >> 2;                                   // groups of four consecutive words [t][cl .. cl + 3]
#pragma unroll 3
            for (int gi = wave; gi < ngrp; gi += 4) {
                const int w = gi * 4, t = w / ch, cl = w - t * ch;
                const size_t g = ((size_t)bh * FR + cl * 3 + t) * N + n0 + lane;      // feature row of word 0; words j: + 3 j rows
                if (lane < cnt) {
                    float* tp = tile + lane * TS + w;
                    if (load) {
                        vnp_f4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = src[g + (size_t)3 * j * N];
                        *(vnp_f4*)tp = v;
                    } else {
                        const vnp_f4 v = *(const vnp_f4*)tp;
#pragma unroll
                        for (int j = 0; j < 4; ++j) dst[g + (size_t)3 * j * N] = v[j];
                    }
                }
            }
            if (!load) for (int r = Dh + wave; r < FR; r += 4) if (lane < cnt) dst[((size_t)bh * FR + r) * N + n0 + lane] = 0.f;
        };
        if (TO_HEADS) { rows_side(true); __syncthreads(); fm_side(false); }
        else          { fm_side(true); __syncthreads(); rows_side(false); }
        return;
    }
    // general form (any ch): scalar words
    auto rows_side = [&](bool load) {
        for (int seg = wave; seg < cnt * 3; seg += 4) {
            const int n = seg / 3, t = seg - n * 3;
            const size_t g = (((size_t)b * N + n0 + n) * 3 + t) * ld + h * ch;
            for (int cl = lane; cl < ch; cl += 64) {
                if (load) tile[n * TS + t * ch + cl] = src[g + cl];
                else dst[g + cl] = tile[n * TS + t * ch + cl];
            }
        }
    };
    auto fm_side = [&](bool load) {
        const int nr = load ? Dh : FR;
        for (int r = wave; r < nr; r += 4) {
            const size_t g = ((size_t)bh * FR + r) * N + n0;
            const int cl = r / 3, t = r - cl * 3;
            if (lane < cnt) {
                if (load) tile[lane * TS + t * ch + cl] = src[g + lane];
                else dst[g + lane] = r < Dh ? tile[lane * TS + t * ch + cl] : 0.f;
            }
        }
    };
    if (TO_HEADS) { rows_side(true); __syncthreads(); fm_side(false); }
    else          { fm_side(true); __syncthreads(); rows_side(false); }
}

// tokens per workgroup: ~24 KB of LDS, at least one token, at most eight
static inline int vnp_head_permute_pn(int C) { const int pn = 6144 / (3 * C); return pn < 1 ? 1 : pn > 8 ? 8 : pn; }

// rows of up to 2048 elements (L % 64 == 0) stay in registers between the passes: one read and one write of the matrix instead of three
// reads and two writes; same element-to-lane assignment and summation order as the streaming form below
template <int NV>
__global__ void __launch_bounds__(256) vnp_softmax_fwd_reg_kernel(float* __restrict__ S, long long R, float scale)
{
    const long long r = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    float* row = S + (size_t)r * (NV * 64);
    float v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = __builtin_nontemporal_load(&row[lane + 64 * i]) * scale; mx = fmaxf(mx, v[i]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = expf(v[i] - mx); s += v[i]; }
    const float inv = 1.0f / wave_sum(s);
#pragma unroll
    for (int i = 0; i < NV; ++i) row[lane + 64 * i] = v[i] * inv;
}

template <int NV>
__global__ void __launch_bounds__(256) vnp_softmax_bwd_reg_kernel(const float* __restrict__ Pm, float* __restrict__ dP, long long R, float scale)
{
    const long long r = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    const float* p = Pm + (size_t)r * (NV * 64);
    float* d = dP + (size_t)r * (NV * 64);
    float pv[NV], dv[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { pv[i] = __builtin_nontemporal_load(&p[lane + 64 * i]); dv[i] = __builtin_nontemporal_load(&d[lane + 64 * i]); s += dv[i] * pv[i]; }
    s = wave_sum(s);
#pragma unroll
    for (int i = 0; i < NV; ++i) d[lane + 64 * i] = pv[i] * (dv[i] - s) * scale;
}

// in-place row softmax of S[R][L] * scale; one wave per row
__global__ void __launch_bounds__(256) vnp_softmax_fwd_kernel(float* __restrict__ S, long long R, int L, float scale)
{
    const long long r = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    float* row = S + (size_t)r * L;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, row[j] * scale);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float s = 0.f;
    for (int j = lane; j < L; j += 64) { const float e = expf(row[j] * scale - mx); row[j] = e; s += e; }
    const float inv = 1.0f / wave_sum(s);
    for (int j = lane; j < L; j += 64) row[j] *= inv;
}

// dS = P * (dP - sum_j dP*P) * scale, written over dP
__global__ void __launch_bounds__(256) vnp_softmax_bwd_kernel(const float* __restrict__ Pm, float* __restrict__ dP, long long R, int L, float scale)
{
    const long long r = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    const float* p = Pm + (size_t)r * L;
    float* d = dP + (size_t)r * L;
    float s = 0.f;
    for (int j = lane; j < L; j += 64) s += d[j] * p[j];
    s = wave_sum(s);
    for (int j = lane; j < L; j += 64) d[j] = p[j] * (d[j] - s) * scale;
}

extern "C" int vnp_vn_layernorm_fwd(hipStream_t st, const float* x, int ldx, long long P, int C, const float* gamma, const float* beta,
                                    float* out, int ldo, float* stats)
{
    if (!x || !gamma || !beta || !out || P <= 0 || C <= 0) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_ln_fwd_kernel, dim3(vnp_cdiv(P, 4)), dim3(256), 0, st, x, ldx, P, C, gamma, beta, out, ldo, (float2*)stats,
                       (const float*)nullptr, 0, (const float*)nullptr, 0, (float*)nullptr, 0);
    return vnp_last_error();
}

// s = x + a1 (+ a2, may be NULL) -> sum_out (the residual stream), out = VNLayerNorm(s): the adds of a transformer block's input
// (x + pos, models/pointr/vn_pointr.py:700; vn_x + x_2 of the previous block, :410) inside the pass that normalises it
extern "C" int vnp_vn_add_layernorm_fwd(hipStream_t st, const float* x, int ldx, const float* a1, int lda1, const float* a2, int lda2,
                                        long long P, int C, const float* gamma, const float* beta, float* sum_out, int lds,
                                        float* out, int ldo, float* stats)
{
    if (!x || !a1 || !sum_out || !gamma || !beta || !out || P <= 0 || C <= 0 || sum_out == x || sum_out == a1 || sum_out == a2) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_ln_fwd_kernel, dim3(vnp_cdiv(P, 4)), dim3(256), 0, st, x, ldx, P, C, gamma, beta, out, ldo, (float2*)stats,
                       a1, lda1, a2, lda2, sum_out, lds);
    return vnp_last_error();
}

// part: ceil(P/4) * 2 * C floats ([block][dgamma | dbeta][C]); the caller reduces it over the blocks
extern "C" int vnp_vn_layernorm_bwd(hipStream_t st, const float* go, int ldgo, const float* x, int ldx, long long P, int C,
                                    const float* gamma, const float* beta, const float* stats, float* gx, int ldgx, float* part)
{
    if (!go || !x || !gamma || !beta || !stats || !gx || !part || P <= 0 || C <= 0 || C > 4096) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_ln_bwd_kernel, dim3(vnp_cdiv(P, 4)), dim3(256), (size_t)8 * C * 4, st, go, ldgo, x, ldx, P, C, gamma, beta,
                       (const float2*)stats, gx, ldgx, part, (const float*)nullptr, 0);
    return vnp_last_error();
}

// gx = (backward of VNLayerNorm) + gres: the normalised tensor also feeds the block's skip connection, whose gradient gres [P*3][C]
// arrives separately; adding it here replaces the launch autograd would spend on the sum
extern "C" int vnp_vn_layernorm_bwd_add(hipStream_t st, const float* go, int ldgo, const float* x, int ldx, long long P, int C,
                                        const float* gamma, const float* beta, const float* stats, const float* gres, int ldgres,
                                        float* gx, int ldgx, float* part)
{
    if (!go || !x || !gamma || !beta || !stats || !gres || !gx || !part || P <= 0 || C <= 0 || C > 4096) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_ln_bwd_kernel, dim3(vnp_cdiv(P, 4)), dim3(256), (size_t)8 * C * 4, st, go, ldgo, x, ldx, P, C, gamma, beta,
                       (const float2*)stats, gx, ldgx, part, gres, ldgres);
    return vnp_last_error();
}

// heads [(b, h)][n][ldh] token-major with ldh >= 3 C / H; the columns [3 C / H, ldh) are written as zeros (K padding of q k^T / dout v^T)
extern "C" int vnp_rows_to_heads_pad(hipStream_t st, const float* rows, int ld, int B, int N, int C, int H, float* heads, int ldh)
{
    if (!rows || !heads || B <= 0 || N <= 0 || C <= 0 || H <= 0 || C % H || ld < C || C > 5000 || ldh < 3 * (C / H)) return VNP_EINVAL;   // one token (3 C floats) must fit 64 KB of LDS
    const int pn = vnp_head_permute_pn(C);
    const long long blocks = ((long long)B * N + pn - 1) / pn;
    if (blocks > 2147483647LL) return VNP_EINVAL;
    hipLaunchKernelGGL((vnp_head_permute_kernel<true>), dim3((unsigned)blocks), dim3(256), (size_t)pn * 3 * C * sizeof(float), st,
                       rows, heads, ld, B, N, C, H, pn, ldh, H, 0LL);
    return vnp_last_error();
}

extern "C" int vnp_rows_to_heads(hipStream_t st, const float* rows, int ld, int B, int N, int C, int H, float* heads)
{
    if (H <= 0 || C <= 0 || C % H) return VNP_EINVAL;
    return vnp_rows_to_heads_pad(st, rows, ld, B, N, C, H, heads, 3 * (C / H));
}

extern "C" int vnp_heads_to_rows(hipStream_t st, const float* heads, int B, int N, int C, int H, float* rows, int ld)
{
    if (!rows || !heads || B <= 0 || N <= 0 || C <= 0 || H <= 0 || C % H || ld < C || C > 5000) return VNP_EINVAL;          // one token (3 C floats) must fit 64 KB of LDS
    const int pn = vnp_head_permute_pn(C);
    const long long blocks = ((long long)B * N + pn - 1) / pn;
    if (blocks > 2147483647LL) return VNP_EINVAL;
    hipLaunchKernelGGL((vnp_head_permute_kernel<false>), dim3((unsigned)blocks), dim3(256), (size_t)pn * 3 * C * sizeof(float), st,
                       heads, rows, ld, B, N, C, H, pn, 3 * (C / H), H, 0LL);
    return vnp_last_error();
}

// G matrices side by side in the rows (q | k | v of one projection GEMM: rows [B*N*3][G*C], any row stride) <-> G head tensors
// heads[g] = [B*H][N][3 C / H], contiguous one after the other: the three (or G) permutes of an attention in one launch
static int vnp_heads_grouped(hipStream_t st, bool to_heads, const float* src, float* dst, int ld, int B, int N, int C, int H, int G)
{
    if (!src || !dst || B <= 0 || N <= 0 || C <= 0 || H <= 0 || G <= 0 || C % H || ld < G * C || (long long)G * C > 5000) return VNP_EINVAL;
    const int Ct = G * C, Ht = G * H;
    const int pn = vnp_head_permute_pn(Ct);
    const long long blocks = ((long long)B * N + pn - 1) / pn;
    if (blocks > 2147483647LL) return VNP_EINVAL;
    const long long gstride = (long long)B * H * N * 3 * (C / H);
    if (to_heads) hipLaunchKernelGGL((vnp_head_permute_kernel<true>), dim3((unsigned)blocks), dim3(256), (size_t)pn * 3 * Ct * sizeof(float), st,
                                     src, dst, ld, B, N, Ct, Ht, pn, 3 * (C / H), H, gstride);
    else          hipLaunchKernelGGL((vnp_head_permute_kernel<false>), dim3((unsigned)blocks), dim3(256), (size_t)pn * 3 * Ct * sizeof(float), st,
                                     src, dst, ld, B, N, Ct, Ht, pn, 3 * (C / H), H, gstride);
    return vnp_last_error();
}

extern "C" int vnp_rows_to_heads_grouped(hipStream_t st, const float* rows, int ld, int B, int N, int C, int H, int G, float* heads)
{
    return vnp_heads_grouped(st, true, rows, heads, ld, B, N, C, H, G);
}

extern "C" int vnp_heads_to_rows_grouped(hipStream_t st, const float* heads, int B, int N, int C, int H, int G, float* rows, int ld)
{
    return vnp_heads_grouped(st, false, heads, rows, ld, B, N, C, H, G);
}

// feature-major heads hf [(b, h)][fr][N], fr >= 3 C / H (rows beyond the head dimension are written as zeros) <-> rows [(b, n, t)][C]
static int vnp_heads_fm(hipStream_t st, bool to_heads, const float* src, float* dst, int ld, int B, int N, int C, int H, int fr)
{
    if (!src || !dst || B <= 0 || N <= 0 || C <= 0 || H <= 0 || C % H || ld < C || fr < 3 * (C / H) || (long long)B * H > 65535) return VNP_EINVAL;
    const int Dh = 3 * (C / H), ch = C / H;
    const float* rows = to_heads ? src : dst;
    const int vec = (ch % 4 == 0 && ch / 4 <= 64 && ld % 4 == 0 && ((uintptr_t)rows & 15) == 0) ? 1 : 0;
    // LDS row stride in words: vec: a multiple of 4 whose quarter is odd (see the kernel); else 3 ch + 1 (odd: lane = token is conflict-free)
    int ts = Dh + 1;
    if (vec) { ts = Dh + 4; if (((ts >> 2) & 1) == 0) ts += 4; }
    int tt = 64;                                                     // tokens per workgroup: as many (64, 32, 16) as fit 64 KB of LDS
    while (tt > 16 && (size_t)tt * ts * sizeof(float) > 65536) tt >>= 1;
    const size_t lds = (size_t)tt * ts * sizeof(float);
    if (lds > 65536) return VNP_EINVAL;
    dim3 grid((unsigned)((N + tt - 1) / tt), (unsigned)(B * H));
    if (to_heads) hipLaunchKernelGGL((vnp_head_permute_fm_kernel<true>), grid, dim3(256), lds, st, src, dst, ld, B, N, C, H, fr, tt, vec, ts);
    else          hipLaunchKernelGGL((vnp_head_permute_fm_kernel<false>), grid, dim3(256), lds, st, src, dst, ld, B, N, C, H, fr, tt, vec, ts);
    return vnp_last_error();
}

extern "C" int vnp_rows_to_heads_fm(hipStream_t st, const float* rows, int ld, int B, int N, int C, int H, float* heads, int fr)
{
    return vnp_heads_fm(st, true, rows, heads, ld, B, N, C, H, fr);
}

extern "C" int vnp_heads_fm_to_rows(hipStream_t st, const float* heads, int fr, int B, int N, int C, int H, float* rows, int ld)
{
    return vnp_heads_fm(st, false, heads, rows, ld, B, N, C, H, fr);
}

extern "C" int vnp_softmax_fwd(hipStream_t st, float* S, long long R, int L, float scale)
{
    if (!S || R <= 0 || L <= 0) return VNP_EINVAL;
#define VNP_SM_FWD(NV) hipLaunchKernelGGL((vnp_softmax_fwd_reg_kernel<NV>), dim3(vnp_cdiv(R, 4)), dim3(256), 0, st, S, R, scale)
    if (L == 64 * 16) VNP_SM_FWD(16);
    else if (L == 64 * 8) VNP_SM_FWD(8);
    else if (L == 64 * 4) VNP_SM_FWD(4);
    else if (L == 64 * 2) VNP_SM_FWD(2);
    else if (L == 64 * 32) VNP_SM_FWD(32);
    else hipLaunchKernelGGL(vnp_softmax_fwd_kernel, dim3(vnp_cdiv(R, 4)), dim3(256), 0, st, S, R, L, scale);
#undef VNP_SM_FWD
    return vnp_last_error();
}

extern "C" int vnp_softmax_bwd(hipStream_t st, const float* Pm, float* dP, long long R, int L, float scale)
{
    if (!Pm || !dP || R <= 0 || L <= 0) return VNP_EINVAL;
#define VNP_SM_BWD(NV) hipLaunchKernelGGL((vnp_softmax_bwd_reg_kernel<NV>), dim3(vnp_cdiv(R, 4)), dim3(256), 0, st, Pm, dP, R, scale)
    if (L == 64 * 16) VNP_SM_BWD(16);
    else if (L == 64 * 8) VNP_SM_BWD(8);
    else if (L == 64 * 4) VNP_SM_BWD(4);
    else if (L == 64 * 2) VNP_SM_BWD(2);
    else if (L == 64 * 32) VNP_SM_BWD(32);
    else hipLaunchKernelGGL(vnp_softmax_bwd_kernel, dim3(vnp_cdiv(R, 4)), dim3(256), 0, st, Pm, dP, R, L, scale);
#undef VNP_SM_BWD
    return vnp_last_error();
}
