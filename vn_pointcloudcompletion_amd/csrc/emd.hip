// Earth Mover's Distance (auction-style approximate matching): replaces the reference's emd_cuda extension
// (extensions/earth_movers_distance/emd_kernel.cu: approxmatch :25-158, matchcost :200-243, matchcostgrad2 :286-327,
// matchcostgrad1 :333-355).
//
// The reference runs ONE 512-thread block per sample through all 10 annealing levels x 3 passes (32 blocks busy at
// most).  Here every pass is its own launch over (tiles of 64 points x samples), four threads per point, so B = 16 samples of 1024 points
// fill 256 workgroups and the 30 dependent passes cost a few microseconds each; the arithmetic (level schedule -4^j, j = 7..-1,
// then 0; integer n/m multipliers; the 1e-9 guards; fmax/fmin clamps; __expf; l/k summation order) is the reference's.
// match is [B][m][n] (match[b][l][k]), temp is [B][(n+m)*2] = remainL[n] | remainR[m] | ratioL[n] | ratioR[m].
#include "vnp_common.h"

#define EMD_T 256
#define EMD_TILE 1024

__device__ __forceinline__ float emd_d2(float x1, float y1, float z1, float x2, float y2, float z2)
{
    return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
}

__global__ void emd_init_kernel(int n, int m, float* __restrict__ temp, float multiL, float multiR)
{
    float* t = temp + (size_t)blockIdx.y * (n + m) * 2;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) t[j] = multiL;
    if (j < m) t[n + j] = multiR;
}

// The three reductions of a level — pass (a): sum over l per k, pass (b): sum over k per l, the drain of pass (c): sum over l per k — share
// one skeleton: a workgroup takes EMD_ROWS "row" points and ALL "column" points of its sample; the column cloud (with its per-point weight)
// is staged in LDS tile by tile, thread (row r, part q) sums the q-th quarter of every tile, and the EMD_PARTS partial sums of a row meet in
// LDS in a fixed order.  (Rounds 1-3: one thread per row walking all 1024 columns — 64 workgroups for B = 16, 1024 points, 30 us per pass; with
// 64 rows x 4 parts a pass is 256 workgroups of 256-step loops.)  The sums differ from the one-thread order by fp32 re-association only.
#define EMD_ROWS 64
#define EMD_PARTS 4
// returns (valid for q == 0 threads) sum_j exp(level * |row - col_j|^2) * w_j over all ncol column points; `scale` multiplies every term
// BEFORE the sum (pass (c): w = exp(..) * ratioL[k] * ratioR[l], emd_kernel.cu:139-141)
__device__ __forceinline__ float emd_row_sum(const float* __restrict__ pc, const float* __restrict__ wc, int ncol, float x, float y, float z,
                                             float level, float scale, bool scaled, float4* buf, float (*red)[EMD_ROWS])
{
#pragma clang fp contract(off)
    const int r = threadIdx.x & (EMD_ROWS - 1), q = threadIdx.x / EMD_ROWS;
    float s = 0.f;
    for (int c0 = 0; c0 < ncol; c0 += EMD_TILE) {
        const int cend = min(ncol, c0 + EMD_TILE) - c0;
        __syncthreads();
        for (int c = threadIdx.x; c < cend; c += EMD_T)
            buf[c] = make_float4(pc[(c0 + c) * 3], pc[(c0 + c) * 3 + 1], pc[(c0 + c) * 3 + 2], wc[c0 + c]);
        __syncthreads();
        const int per = (cend + EMD_PARTS - 1) / EMD_PARTS;
        const int j0 = q * per, j1 = min(cend, j0 + per);
        for (int j = j0; j < j1; ++j) {
            const float4 p = buf[j];                              // (same address across the 64 lanes of a wave: LDS broadcast)
            const float e = __expf(level * emd_d2(x, y, z, p.x, p.y, p.z));
            s += scaled ? e * scale * p.w : e * p.w;
        }
    }
    red[q][r] = s;
    __syncthreads();
    float t = red[0][r];
#pragma unroll
    for (int i = 1; i < EMD_PARTS; ++i) t += red[i][r];
    return t;
}

// pass (a): ratioL[k] = remainL[k] / (1e-9 + sum_l exp(level*d) * remainR[l])          emd_kernel.cu:51-84
__global__ void __launch_bounds__(EMD_T) emd_pass_a_kernel(int n, int m, const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                          float* __restrict__ temp, float level)
{
    __shared__ float4 buf[EMD_TILE];
    __shared__ float red[EMD_PARTS][EMD_ROWS];
    const int b = blockIdx.y, k = blockIdx.x * EMD_ROWS + (threadIdx.x & (EMD_ROWS - 1));
    float* t = temp + (size_t)b * (n + m) * 2;
    const float* remainL = t; const float* remainR = t + n; float* ratioL = t + n + m;
    const float* p1 = xyz1 + (size_t)b * n * 3; const float* p2 = xyz2 + (size_t)b * m * 3;
    float x1 = 0, y1 = 0, z1 = 0;
    if (k < n) { x1 = p1[k * 3]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; }
    const float suml = 1e-9f + emd_row_sum(p2, remainR, m, x1, y1, z1, level, 1.f, false, buf, red);
    if (threadIdx.x < EMD_ROWS && k < n) ratioL[k] = remainL[k] / suml;
}

// pass (b): per l: sumr = remainR[l] * sum_k exp(level*d)*ratioL[k]; ratioR, remainR update     emd_kernel.cu:86-119
__global__ void __launch_bounds__(EMD_T) emd_pass_b_kernel(int n, int m, const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                          float* __restrict__ temp, float level)
{
    __shared__ float4 buf[EMD_TILE];
    __shared__ float red[EMD_PARTS][EMD_ROWS];
    const int b = blockIdx.y, l = blockIdx.x * EMD_ROWS + (threadIdx.x & (EMD_ROWS - 1));
    float* t = temp + (size_t)b * (n + m) * 2;
    float* remainR = t + n; const float* ratioL = t + n + m; float* ratioR = t + n + m + n;
    const float* p1 = xyz1 + (size_t)b * n * 3; const float* p2 = xyz2 + (size_t)b * m * 3;
    float x2 = 0, y2 = 0, z2 = 0;
    if (l < m) { x2 = p2[l * 3]; y2 = p2[l * 3 + 1]; z2 = p2[l * 3 + 2]; }
    float sumr = emd_row_sum(p1, ratioL, n, x2, y2, z2, level, 1.f, false, buf, red);
    if (threadIdx.x < EMD_ROWS && l < m) {
        const float r = remainR[l];
        sumr *= r;
        const float consumption = fminf(r / (sumr + 1e-9f), 1.0f);
        ratioR[l] = consumption * r;
        remainR[l] = fmaxf(0.0f, r - sumr);
    }
}

// pass (c): match[l][k] += exp(level*d)*ratioL[k]*ratioR[l]; remainL[k] = max(0, remainL[k] - sum_l w)     :121-154
// Two launches: the match update is a read-modify-write of the whole [m][n] matrix (128 MB of traffic per level at B = 16, n = m = 1024) and
// wants every CU — a workgroup per (tile of k, chunk of EMD_LCH rows l, sample); the row sums that drain remainL are a reduction over
// all l per k and keep the one-thread-per-k form (and the l order) of the other passes.  As ONE kernel of n / 256 x B workgroups with
// 1024 dependent read-modify-writes per thread the pass took 252 us per level; the weights w are the same expression in both kernels
// (no contraction), so match and remainL have the bits of the single-kernel form.
#define EMD_LCH 32
__global__ void __launch_bounds__(EMD_T) emd_pass_c_match_kernel(int n, int m, const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                                const float* __restrict__ temp, float* __restrict__ match, float level)
{
#pragma clang fp contract(off)
    const int b = blockIdx.z, k = blockIdx.x * EMD_T + threadIdx.x;
    if (k >= n) return;
    const float* t = temp + (size_t)b * (n + m) * 2;
    const float* ratioL = t + n + m; const float* ratioR = t + n + m + n;
    const float* p1 = xyz1 + (size_t)b * n * 3; const float* p2 = xyz2 + (size_t)b * m * 3;
    float* mt = match + (size_t)b * n * m;
    const float x1 = p1[k * 3], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2], rl = ratioL[k];
    const int l0 = blockIdx.y * EMD_LCH, l1 = min(m, l0 + EMD_LCH);
    for (int l = l0; l < l1; ++l) {                              // p2[l], ratioR[l]: wave-uniform (scalar) loads
        const float w = __expf(level * emd_d2(x1, y1, z1, p2[l * 3], p2[l * 3 + 1], p2[l * 3 + 2])) * rl * ratioR[l];
        mt[(size_t)l * n + k] += w;
    }
}

__global__ void __launch_bounds__(EMD_T) emd_pass_c_sum_kernel(int n, int m, const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                              float* __restrict__ temp, float level)
{
    __shared__ float4 buf[EMD_TILE];
    __shared__ float red[EMD_PARTS][EMD_ROWS];
    const int b = blockIdx.y, k = blockIdx.x * EMD_ROWS + (threadIdx.x & (EMD_ROWS - 1));
    float* t = temp + (size_t)b * (n + m) * 2;
    float* remainL = t; const float* ratioL = t + n + m; const float* ratioR = t + n + m + n;
    const float* p1 = xyz1 + (size_t)b * n * 3; const float* p2 = xyz2 + (size_t)b * m * 3;
    float x1 = 0, y1 = 0, z1 = 0, rl = 0;
    if (k < n) { x1 = p1[k * 3]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; rl = ratioL[k]; }
    const float suml = emd_row_sum(p2, ratioR, m, x1, y1, z1, level, rl, true, buf, red);      // the weights of emd_pass_c_match_kernel, term for term
    if (threadIdx.x < EMD_ROWS && k < n) remainL[k] = fmaxf(0.0f, remainL[k] - suml);
}

// partial[chunk][b] = sum over the chunk's l and all k of match[l][k]*d(k,l)        emd_kernel.cu:200-243
__global__ void __launch_bounds__(EMD_T) emd_cost_kernel(int n, int m, const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                        const float* __restrict__ match, int lchunk, int B, float* __restrict__ partial)
{
    __shared__ float red[EMD_T];
    const int b = blockIdx.y;
    const int lbeg = blockIdx.x * lchunk, lend = min(m, lbeg + lchunk);
    const float* p1 = xyz1 + (size_t)b * n * 3; const float* p2 = xyz2 + (size_t)b * m * 3;
    const float* mt = match + (size_t)b * n * m;
    float s = 0;
    for (int k = threadIdx.x; k < n; k += EMD_T) {
        const float x1 = p1[k * 3], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];
        for (int l = lbeg; l < lend; ++l)
            s += emd_d2(x1, y1, z1, p2[l * 3], p2[l * 3 + 1], p2[l * 3 + 2]) * mt[(size_t)l * n + k];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = EMD_T / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * B + b] = red[0];
}

__global__ void emd_cost_final_kernel(const float* __restrict__ partial, int nchunk, int B, float* __restrict__ cost)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0;
    for (int c = 0; c < nchunk; ++c) s += partial[(size_t)c * B + b];
    cost[b] = s;
}

// grad1[k] = gc * sum_l 2*match[l][k]*(x1[k]-x2[l])          emd_kernel.cu:333-355
__global__ void __launch_bounds__(EMD_T) emd_grad1_kernel(int n, int m, const float* __restrict__ gc, const float* __restrict__ xyz1,
                                                         const float* __restrict__ xyz2, const float* __restrict__ match, float* __restrict__ grad1)
{
    __shared__ float buf[EMD_TILE * 3];
    const int b = blockIdx.y, k = blockIdx.x * EMD_T + threadIdx.x;
    const float* p1 = xyz1 + (size_t)b * n * 3; const float* p2 = xyz2 + (size_t)b * m * 3;
    const float* mt = match + (size_t)b * n * m;
    float x1 = 0, y1 = 0, z1 = 0, dx = 0, dy = 0, dz = 0;
    if (k < n) { x1 = p1[k * 3]; y1 = p1[k * 3 + 1]; z1 = p1[k * 3 + 2]; }
    for (int l0 = 0; l0 < m; l0 += EMD_TILE) {
        const int lend = min(m, l0 + EMD_TILE) - l0;
        __syncthreads();
        for (int i = threadIdx.x; i < lend * 3; i += EMD_T) buf[i] = p2[(size_t)l0 * 3 + i];
        __syncthreads();
        if (k < n)
            for (int l = 0; l < lend; ++l) {
                const float d = mt[(size_t)(l0 + l) * n + k] * 2;
                dx += (x1 - buf[l * 3]) * d; dy += (y1 - buf[l * 3 + 1]) * d; dz += (z1 - buf[l * 3 + 2]) * d;
            }
    }
    if (k < n) {
        const float g = gc[b];
        grad1[((size_t)b * n + k) * 3] = dx * g; grad1[((size_t)b * n + k) * 3 + 1] = dy * g; grad1[((size_t)b * n + k) * 3 + 2] = dz * g;
    }
}

// grad2[l] = gc * sum_k 2*match[l][k]*(x2[l]-x1[k]); one wave per l          emd_kernel.cu:286-327
__global__ void __launch_bounds__(256) emd_grad2_kernel(int n, int m, const float* __restrict__ gc, const float* __restrict__ xyz1,
                                                       const float* __restrict__ xyz2, const float* __restrict__ match, float* __restrict__ grad2)
{
    const int b = blockIdx.y, l = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (l >= m) return;
    const float* p1 = xyz1 + (size_t)b * n * 3; const float* p2 = xyz2 + ((size_t)b * m + l) * 3;
    const float* mt = match + (size_t)b * n * m + (size_t)l * n;
    const float x2 = p2[0], y2 = p2[1], z2 = p2[2];
    float sx = 0, sy = 0, sz = 0;
    for (int k = lane; k < n; k += 64) {
        const float d = mt[k] * 2;
        sx += (x2 - p1[k * 3]) * d; sy += (y2 - p1[k * 3 + 1]) * d; sz += (z2 - p1[k * 3 + 2]) * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
    if (lane == 0) {
        const float g = gc[b];
        float* o = grad2 + ((size_t)b * m + l) * 3;
        o[0] = sx * g; o[1] = sy * g; o[2] = sz * g;
    }
}

// emd_cuda.approxmatch_forward(xyz1[B,n,3], xyz2[B,m,3]) -> match[B,m,n]  (emd.cpp:8-13); temp: B*(n+m)*2 floats
extern "C" int vnp_emd_approxmatch(hipStream_t st, int B, int n, int m, const float* xyz1, const float* xyz2, float* match, float* temp)
{
    if (!xyz1 || !xyz2 || !match || !temp || B <= 0 || n <= 0 || m <= 0 || B > 65535 || vnp_cdiv(m, EMD_LCH) > 65535) return VNP_EINVAL;
    hipError_t e = hipMemsetAsync(match, 0, (size_t)B * n * m * 4, st);
    if (e != hipSuccess) return (int)e;
    const float multiL = n >= m ? 1.f : (float)(m / n), multiR = n >= m ? (float)(n / m) : 1.f;      // integer division, :29-35
    hipLaunchKernelGGL(emd_init_kernel, dim3(vnp_cdiv(n > m ? n : m, 256), B), dim3(256), 0, st, n, m, temp, multiL, multiR);
    for (int j = 7; j >= -2; --j) {
        const float level = j == -2 ? 0.f : -powf(4.0f, (float)j);
        hipLaunchKernelGGL(emd_pass_a_kernel, dim3(vnp_cdiv(n, EMD_ROWS), B), dim3(EMD_T), 0, st, n, m, xyz1, xyz2, temp, level);
        hipLaunchKernelGGL(emd_pass_b_kernel, dim3(vnp_cdiv(m, EMD_ROWS), B), dim3(EMD_T), 0, st, n, m, xyz1, xyz2, temp, level);
        hipLaunchKernelGGL(emd_pass_c_match_kernel, dim3(vnp_cdiv(n, EMD_T), vnp_cdiv(m, EMD_LCH), B), dim3(EMD_T), 0, st, n, m, xyz1, xyz2,
                           (const float*)temp, match, level);
        hipLaunchKernelGGL(emd_pass_c_sum_kernel, dim3(vnp_cdiv(n, EMD_ROWS), B), dim3(EMD_T), 0, st, n, m, xyz1, xyz2, temp, level);
    }
    return vnp_last_error();
}

// emd_cuda.matchcost_forward -> cost[B] (emd.cpp:15-19); ws: ceil(m/64)*B floats
extern "C" int vnp_emd_matchcost(hipStream_t st, int B, int n, int m, const float* xyz1, const float* xyz2, const float* match,
                                 float* cost, float* ws)
{
    if (!xyz1 || !xyz2 || !match || !cost || !ws || B <= 0 || n <= 0 || m <= 0 || B > 65535) return VNP_EINVAL;
    const int lchunk = 64, nchunk = vnp_cdiv(m, lchunk);
    hipLaunchKernelGGL(emd_cost_kernel, dim3(nchunk, B), dim3(EMD_T), 0, st, n, m, xyz1, xyz2, match, lchunk, B, ws);
    hipLaunchKernelGGL(emd_cost_final_kernel, dim3(vnp_cdiv(B, 64)), dim3(64), 0, st, (const float*)ws, nchunk, B, cost);
    return vnp_last_error();
}

// emd_cuda.matchcost_backward(grad_cost[B], ...) -> grad1[B,n,3], grad2[B,m,3] (emd.cpp:21-27)
extern "C" int vnp_emd_matchcost_bwd(hipStream_t st, int B, int n, int m, const float* grad_cost, const float* xyz1, const float* xyz2,
                                     const float* match, float* grad1, float* grad2)
{
    if (!grad_cost || !xyz1 || !xyz2 || !match || !grad1 || !grad2 || B <= 0 || n <= 0 || m <= 0 || B > 65535) return VNP_EINVAL;
    hipLaunchKernelGGL(emd_grad1_kernel, dim3(vnp_cdiv(n, EMD_T), B), dim3(EMD_T), 0, st, n, m, grad_cost, xyz1, xyz2, match, grad1);
    hipLaunchKernelGGL(emd_grad2_kernel, dim3(vnp_cdiv(m, 4), B), dim3(256), 0, st, n, m, grad_cost, xyz1, xyz2, match, grad2);
    return vnp_last_error();
}
