// Definitions shared by the Vector-Neuron kernel files (vn_ops.hip: BatchNorm / leaky / pooling passes; fold_fused.hip: FoldingNet's first
// layer end to end; edge_ops.hip: fused EdgeConv and the reverse adjacency).
#pragma once
#include "vnp_common.h"

#define VNP_CHUNK 1024         // points (rows) per block in the chunked kernels (also exported via vnp_chunk_points)
// gradient kernels use v_rcp_f32 / v_sqrt_f32 (1 ulp each) where the forward kernels keep IEEE division and sqrtf
#define VNP_GRAD_HW 1

__device__ __forceinline__ float vnp_sq3(float a, float b, float c) { return a * a + b * b + c * c; }

// (count, mean, M2) of two sets of values merged (Chan): Welford per thread + pairwise merges — the batch variance of the norms can be
// 1e4-1e5 times smaller than mean^2 (FoldingNet layer 0: every point of a sample shares the dominant global term), where sum /
// sum-of-squares in fp32 loses the variance to cancellation.
__device__ __forceinline__ void vnp_chan_merge(double& na, double& ma, double& Ma, double nb, double mb, double Mb)
{
    if (nb <= 0.0) return;
    const double n = na + nb, delta = mb - ma;
    ma += delta * nb / n;
    Ma += Mb + delta * delta * na * nb / n;
    na = n;
}

static inline dim3 vnp_grid_pc(long long P, int C, int chunk) { return dim3(vnp_cdiv(C, 64), vnp_cdiv(P, chunk)); }
static inline bool vnp_chunk_ok(int chunk) { return chunk >= 4 && chunk <= VNP_CHUNK; }

// Ascending sort of data[0 .. L) by one workgroup: the bitonic network in its all-ascending form (first stage of every merge mirrors the
// block, i ^ (2 j - 1); the others i ^ j; the smaller value always goes to the smaller index), so positions >= L behave as +infinity
// without being stored.  `data` is LDS or global memory (a workgroup's own global accesses are ordered by __syncthreads).
__device__ inline void vnp_block_sort_asc(int* data, int L)
{
    int n2 = 1;
    while (n2 < L) n2 <<= 1;
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool first = j == (k >> 1);
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                const int p = first ? (i ^ (k - 1)) : (i ^ j);
                if (p > i && p < L) {                             // (i < p < L: both real; p >= L is +infinity and stays where it is)
                    const int a = data[i], c = data[p];
                    if (c < a) { data[i] = c; data[p] = a; }
                }
            }
            __syncthreads();
        }
    }
}
