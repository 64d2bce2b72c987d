// The FoldingNet rank-2 expansion in its materialising form (models/pcn.py:375-385), the small row / column reductions the backward passes
// need, and the layout conversions between the reference's [B, C, 3, N] tensors and the rows layout: kernels and C-ABI.
#include "vn_kernels.h"

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
    const long long r = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
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


// ================================================================================================ C-ABI
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
    const int cx = threadIdx.x & 63, g = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
