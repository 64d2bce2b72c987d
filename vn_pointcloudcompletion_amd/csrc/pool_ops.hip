// Arg-max / mean pooling of VN feature maps on the rows layout (models/vn_layers.py:153-171) and the sparse backward of a VNLinear whose
// only consumer is the arg-max pool (vnp_pool_lin_bwd): kernels and C-ABI.
#include "vn_kernels.h"

// ------------------------------------------------------------------------------------------------ pooling
// VNMaxPool (vn_layers.py:153-167): per group of gs consecutive points and channel c, argmax_n <x, d>, first max wins.
// Stage 1: grid (ceil(C/64), nchunk, G): best (value, local index) of each chunk of the group.
__global__ void __launch_bounds__(256) vnp_pool_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ d, int ldd,
                                                              int gs, int C, int nchunk, float* __restrict__ pval, int* __restrict__ pidx)
{
    __shared__ float rv[4][64];
    __shared__ int ri[4][64];
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
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


// ---- stable sort of every group's arg-max indices: what the sparse pool backward walks --------------------------------------------------
// idx [G][C] (values in [0, gs)) -> pts [G][C] = the values ascending, order [G][C] = the positions they came from, ties in position order
// (torch.sort(stable=True) per row, which this replaces together with its two dtype conversions).  One workgroup per group sorts the
// combined keys value * C2 + position (C2 = C rounded up to a power of two) in LDS.
#define VNP_SORT_MAXC 4096
__global__ void __launch_bounds__(1024) vnp_sort_groups_kernel(const int* __restrict__ idx, int C, int shift, int* __restrict__ pts, int* __restrict__ order)
{
    __shared__ int keys[VNP_SORT_MAXC];
    const size_t g = blockIdx.x;
    for (int j = threadIdx.x; j < C; j += 1024) keys[j] = (idx[g * C + j] << shift) | j;
    __syncthreads();
    vnp_block_sort_asc(keys, C);
    for (int j = threadIdx.x; j < C; j += 1024) {
        const int v = keys[j];
        pts[g * C + j] = v >> shift;
        order[g * C + j] = v & ((1 << shift) - 1);
    }
}

// ================================================================================================ C-ABI
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

// idx [G][C] int32 with values in [0, gs): pts / order as described at vnp_sort_groups_kernel.  C <= 4096, gs * C2 < 2^31.
extern "C" int vnp_sort_groups(hipStream_t st, const int* idx, long long G, int C, int gs, int* pts, int* order)
{
    if (!idx || !pts || !order || G <= 0 || G > 2147483647LL || C <= 0 || C > VNP_SORT_MAXC || gs <= 0) return VNP_EINVAL;
    int shift = 0;
    while ((1 << shift) < C) ++shift;
    if (((long long)gs << shift) > 2147483647LL) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_sort_groups_kernel, dim3((unsigned)G), dim3(1024), 0, st, idx, C, shift, pts, order);
    return vnp_last_error();
}
