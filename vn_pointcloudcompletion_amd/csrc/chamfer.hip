// Chamfer distance for 3-D point clouds: replaces the reference's NmDistanceKernel / NmDistanceGradKernel
// (extensions/chamfer_distance/chamfer3D.cu:12-134, 155-174).
//
// Forward, one direction (queries q[B,n,3] against references r[B,m,3]): squared distance to, and index of, the
// nearest reference; direct fp32 (r-q)^2 form written as fmaf(z,z,fmaf(y,y,x*x)); the FIRST minimum wins.
// The reference launches 32x16 blocks that each walk the whole reference cloud.  Here the work is cut in both
// dimensions: a block owns QPT*256 queries (held in registers) and ONE slice of the reference cloud, staged once
// into LDS as padded float4 (every lane reads the same address -> LDS broadcast, no bank conflicts), and merges
// its result into a packed 64-bit word per query with atomicMin:  (float bits of d) << 32 | index.  d >= 0, so the
// unsigned order of the bits is the order of the floats, and among equal distances the smaller index wins —
// exactly the reference's strict '<' scan.  A second tiny kernel unpacks (dist, idx).
// This is VALU-bound (~9.5 lane-op-equivalents per pair), not HBM-bound: ~12 B per point are read once per block.
#include "vnp_common.h"

#define CH_THREADS 256
#define CH_QPT 4                 // queries per thread
#define CH_TILE 1024             // reference points staged per LDS tile (16 KB)

// Measured notes on the inner loop (kept for the next reader): (1) packing the arithmetic (v_pk_add/mul/fma_f32) halves the
// instruction count but not the time — on gfx950 a wave64 v_pk_*_f32 issues in 4 cycles, the bound is 32 lane-ops / clk /
// SIMD either way, and the extra v_mov splats made it 8 % slower; (2) tracking only chunk minima (v_min3) and re-scanning
// a chunk when some lane improved was slower too: with 256 queries per wave some lane improves in 40-60 % of 8-reference
// chunks.  The plain form below — 3 sub, 1 mul, 2 fma, 1 cmp, 2 select per pair, 4 queries per thread sharing each LDS
// broadcast read — runs 18.3 G pairs/step in ~3.1 ms = ~70 % of the VALU issue bound.
__global__ void __launch_bounds__(CH_THREADS) vnp_chamfer_nn_kernel(const float* __restrict__ q, int n, const float* __restrict__ r, int m,
                                                                   int slice, unsigned long long* __restrict__ packed)
{
    __shared__ float4 tile[CH_TILE];
    const int b = blockIdx.z;
    const float* qb = q + (size_t)b * n * 3;
    const float* rb = r + (size_t)b * m * 3;
    const int k_begin = blockIdx.y * slice;
    const int k_end = min(m, k_begin + slice);

    float qx[CH_QPT], qy[CH_QPT], qz[CH_QPT], best[CH_QPT];
    int bi[CH_QPT];
    const int j0 = blockIdx.x * (CH_THREADS * CH_QPT) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < CH_QPT; ++u) {
        const int j = min(j0 + u * CH_THREADS, n - 1);      // clamp: out-of-range lanes redo the last point, never stored
        qx[u] = qb[j * 3 + 0]; qy[u] = qb[j * 3 + 1]; qz[u] = qb[j * 3 + 2];
        best[u] = INFINITY; bi[u] = 0;
    }
    for (int k0 = k_begin; k0 < k_end; k0 += CH_TILE) {
        const int cnt = min(CH_TILE, k_end - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += CH_THREADS) {
            const float* s = rb + (size_t)(k0 + i) * 3;
            tile[i] = make_float4(s[0], s[1], s[2], 0.f);
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < cnt; ++i) {
            const float4 t = tile[i];
#pragma unroll
            for (int u = 0; u < CH_QPT; ++u) {
                const float dx = t.x - qx[u], dy = t.y - qy[u], dz = t.z - qz[u];
                const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                if (d < best[u]) { best[u] = d; bi[u] = k0 + i; }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < CH_QPT; ++u) {
        const int j = j0 + u * CH_THREADS;
        if (j < n && k_begin < k_end) {
            const unsigned long long v = ((unsigned long long)__float_as_uint(best[u]) << 32) | (unsigned)bi[u];
            atomicMin(&packed[(size_t)b * n + j], v);
        }
    }
}

__global__ void vnp_chamfer_unpack_kernel(const unsigned long long* __restrict__ packed, long long total,
                                          float* __restrict__ dist, int* __restrict__ idx)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const unsigned long long v = packed[i];
    dist[i] = __uint_as_float((unsigned)(v >> 32));
    idx[i] = (int)(unsigned)(v & 0xffffffffu);
}

// grad_xyz1[j] += g*(x1 - x2[idx]); grad_xyz2[idx] -= g*(x1 - x2[idx]), g = 2*grad_dist[j]   (chamfer3D.cu:155-174)
__global__ void vnp_chamfer_grad_kernel(int n, const float* __restrict__ xyz1, int m, const float* __restrict__ xyz2,
                                        const float* __restrict__ grad_dist, const int* __restrict__ idx,
                                        float* __restrict__ grad1, float* __restrict__ grad2)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const size_t p = ((size_t)b * n + j) * 3;
    const int j2 = idx[(size_t)b * n + j];
    const size_t qq = ((size_t)b * m + j2) * 3;
    const float g = grad_dist[(size_t)b * n + j] * 2.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const float v = g * (xyz1[p + t] - xyz2[qq + t]);
        atomicAdd(&grad1[p + t], v);
        atomicAdd(&grad2[qq + t], -v);
    }
}

static int vnp_chamfer_one_direction(hipStream_t st, int B, int n, const float* q, int m, const float* r,
                                     float* dist, int* idx, unsigned long long* packed)
{
    hipError_t e = hipMemsetAsync(packed, 0xff, (size_t)B * n * 8, st);
    if (e != hipSuccess) return (int)e;
    const int qblocks = vnp_cdiv(n, CH_THREADS * CH_QPT);
    // slice the reference cloud so that the launch has >= ~2048 blocks, in whole LDS tiles
    int nslice = vnp_cdiv(2048, qblocks * B);
    const int max_slices = vnp_cdiv(m, CH_TILE);
    if (nslice > max_slices) nslice = max_slices;
    if (nslice < 1) nslice = 1;
    int slice = vnp_cdiv(vnp_cdiv(m, nslice), CH_TILE) * CH_TILE;
    nslice = vnp_cdiv(m, slice);
    hipLaunchKernelGGL(vnp_chamfer_nn_kernel, dim3(qblocks, nslice, B), dim3(CH_THREADS), 0, st, q, n, r, m, slice, packed);
    const long long total = (long long)B * n;
    hipLaunchKernelGGL(vnp_chamfer_unpack_kernel, dim3(vnp_cdiv(total, 256)), dim3(256), 0, st,
                       (const unsigned long long*)packed, total, dist, idx);
    return vnp_last_error();
}

// Replaces chamfer_3D.forward(xyz1, xyz2, dist1, dist2, idx1, idx2) (chamfer_cuda.cpp:17-19, chamfer3D.cu:136-154).
// ws: scratch of max(n, m) * B * 8 bytes.  Outputs need not be zeroed.
extern "C" int vnp_chamfer_fwd(hipStream_t st, int B, int n, const float* xyz1, int m, const float* xyz2,
                               float* dist1, int* idx1, float* dist2, int* idx2, void* ws)
{
    if (B <= 0 || n <= 0 || m <= 0 || !xyz1 || !xyz2 || !dist1 || !idx1 || !dist2 || !idx2 || !ws) return VNP_EINVAL;
    if (B > 65535) return VNP_EINVAL;
    int rc = vnp_chamfer_one_direction(st, B, n, xyz1, m, xyz2, dist1, idx1, (unsigned long long*)ws);
    if (rc) return rc;
    return vnp_chamfer_one_direction(st, B, m, xyz2, n, xyz1, dist2, idx2, (unsigned long long*)ws);
}

// Replaces chamfer_3D.backward (chamfer_cuda.cpp:22-26, chamfer3D.cu:176-195).  grad_xyz1/2 are zeroed here.
extern "C" int vnp_chamfer_bwd(hipStream_t st, int B, int n, const float* xyz1, int m, const float* xyz2,
                               const float* grad_dist1, const int* idx1, const float* grad_dist2, const int* idx2,
                               float* grad_xyz1, float* grad_xyz2)
{
    if (B <= 0 || n <= 0 || m <= 0 || !xyz1 || !xyz2 || !grad_dist1 || !idx1 || !grad_dist2 || !idx2 || !grad_xyz1 || !grad_xyz2)
        return VNP_EINVAL;
    if (B > 65535) return VNP_EINVAL;
    hipError_t e = hipMemsetAsync(grad_xyz1, 0, (size_t)B * n * 12, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(grad_xyz2, 0, (size_t)B * m * 12, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(vnp_chamfer_grad_kernel, dim3(vnp_cdiv(n, 256), B), dim3(256), 0, st, n, xyz1, m, xyz2, grad_dist1, idx1, grad_xyz1, grad_xyz2);
    hipLaunchKernelGGL(vnp_chamfer_grad_kernel, dim3(vnp_cdiv(m, 256), B), dim3(256), 0, st, m, xyz2, n, xyz1, grad_dist2, idx2, grad_xyz2, grad_xyz1);
    return vnp_last_error();
}
