// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chains, 157 TF/s peak).
//
// Every Vector-Neuron linear map of the reference (models/vn_layers.py:21,36,65,69,162: x.transpose(1,-1) ->
// nn.Linear -> transpose back) is, in the rows layout of vnp_common.h, one plain row-major GEMM
//     Y[P*3, Cout] = X[P*3, Cin] * W[Cout, Cin]^T
// and its two gradients are the NN (dX = dY * W) and TN (dW = dY^T * X, split over K) forms.  One kernel
// template covers the three; an operand is either K-contiguous ("KC": fragments read from LDS as
// ds_read_b128 along K) or M/N-contiguous (fragments read as ds_read_b32).
//
// Tiling (256 threads = 4 waves as 2(M) x 2(N)): block tile BM x BN x BK = 192x128x32 or 128x128x32,
// each wave (BM/2)x64 = TM x 2 MFMA tiles of 32x32 -> 96 / 64 accumulator registers, 2 blocks per CU.
// Global -> registers -> LDS staging with the next K-tile's loads issued before the current tile's MFMAs
// (one LDS buffer, two barriers per K-tile; the co-resident block covers the barrier gaps).
// K order inside an MFMA pair is permuted identically for A and B (lane half h takes k = 8q+4h+s), so a lane's
// four consecutive K values come from ONE 16-byte LDS read.
#include "vnp_common.h"

struct VnpGemmArgs {
    const float* A; const float* B; float* C;
    const float* A2; const float* B2;   // optional second product accumulated into the same tile (K2 > 0)
    const float* bias;                   // optional: C[r][c] += bias[(r / rows_per_group) * 3 + r % 3][c]
    int M, N, K, K2;
    int lda, ldb, ldc, lda2, ldb2, ldbias;
    int rows_per_group;
    int accumulate;                      // C += result instead of C = result
    int tiles_per_split;                 // K-tiles handled by one blockIdx.y
    long long slab;                      // elements between the output slabs of consecutive splits
};

template <int R, int BK, bool KC, int NV>
__device__ __forceinline__ void vnp_load_tile(vnp_f4 (&reg)[NV], const float* __restrict__ G, int ld,
                                              int r0, int rmax, int k0, int kmax, bool vec_ok, int tid)
{
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = tid + 256 * j;
        int r, k;
        if (KC) { r = f / (BK / 4); k = (f % (BK / 4)) * 4; }
        else    { k = f / (R / 4);  r = (f % (R / 4)) * 4; }
        const int gr = r0 + r, gk = k0 + k;
        vnp_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (KC) {
            if (gr < rmax && gk < kmax) {
                const float* src = G + (size_t)gr * ld + gk;
                if (vec_ok && gk + 3 < kmax) v = *(const vnp_f4*)src;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gk + e < kmax) v[e] = src[e];
                }
            }
        } else {
            if (gk < kmax && gr < rmax) {
                const float* src = G + (size_t)gk * ld + gr;
                if (vec_ok && gr + 3 < rmax) v = *(const vnp_f4*)src;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gr + e < rmax) v[e] = src[e];
                }
            }
        }
        reg[j] = v;
    }
}

template <int R, int BK, bool KC, int NV, int LDS_LD>
__device__ __forceinline__ void vnp_store_tile(const vnp_f4 (&reg)[NV], float* S, int tid)
{
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = tid + 256 * j;
        int r, k;
        if (KC) { r = f / (BK / 4); k = (f % (BK / 4)) * 4; *(vnp_f4*)&S[r * LDS_LD + k] = reg[j]; }
        else    { k = f / (R / 4);  r = (f % (R / 4)) * 4; *(vnp_f4*)&S[k * LDS_LD + r] = reg[j]; }
    }
}

template <int BM, int BN, int BK, bool AKC, bool BKC>
__global__ void __launch_bounds__(256, 2) vnp_gemm_kernel(const VnpGemmArgs p)
{
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int LDAS = AKC ? BK + 4 : BM;       // +4 floats: conflict-free ds_read_b128 (stride 36 = 4*9, 9 odd)
    constexpr int LDBS = BKC ? BK + 4 : BN;
    constexpr int ASZ = AKC ? BM * LDAS : BK * LDAS;
    constexpr int BSZ = BKC ? BN * LDBS : BK * LDBS;
    constexpr int AV = BM * BK / 4 / 256, BV = BN * BK / 4 / 256;
    __shared__ __attribute__((aligned(16))) float smem[ASZ + BSZ];
    float* As = smem;
    float* Bs = smem + ASZ;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give every XCD a
    // contiguous run of tiles (column tile fastest) -> the co-resident blocks of one L2 share A/B panels.
    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int nwg = nbm * nbn;
    int wg;
    {
        const int id = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = id & 7;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    }
    const int m0 = (wg / nbn) * BM, n0 = (wg % nbn) * BN;

    const int nk1 = (p.K + BK - 1) / BK;
    const int nk2 = p.K2 > 0 ? (p.K2 + BK - 1) / BK : 0;
    const int nk = nk1 + nk2;
    const int kt0 = blockIdx.y * p.tiles_per_split;
    const int kt1 = min(nk, kt0 + p.tiles_per_split);

    const bool va1 = ((p.lda & 3) == 0) && ((((uintptr_t)p.A) & 15) == 0);
    const bool vb1 = ((p.ldb & 3) == 0) && ((((uintptr_t)p.B) & 15) == 0);
    const bool va2 = nk2 && ((p.lda2 & 3) == 0) && ((((uintptr_t)p.A2) & 15) == 0);
    const bool vb2 = nk2 && ((p.ldb2 & 3) == 0) && ((((uintptr_t)p.B2) & 15) == 0);

    vnp_f4 ra[AV], rb[BV];
    vnp_f16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto fetch = [&](int kt) {
        if (kt < nk1) {
            vnp_load_tile<BM, BK, AKC, AV>(ra, p.A, p.lda, m0, p.M, kt * BK, p.K, va1, tid);
            vnp_load_tile<BN, BK, BKC, BV>(rb, p.B, p.ldb, n0, p.N, kt * BK, p.K, vb1, tid);
        } else {
            vnp_load_tile<BM, BK, AKC, AV>(ra, p.A2, p.lda2, m0, p.M, (kt - nk1) * BK, p.K2, va2, tid);
            vnp_load_tile<BN, BK, BKC, BV>(rb, p.B2, p.ldb2, n0, p.N, (kt - nk1) * BK, p.K2, vb2, tid);
        }
    };
    auto stage = [&]() {
        vnp_store_tile<BM, BK, AKC, AV, LDAS>(ra, As, tid);
        vnp_store_tile<BN, BK, BKC, BV, LDBS>(rb, Bs, tid);
    };

    if (kt0 < kt1) {
        fetch(kt0);
        stage();
        __syncthreads();
    }
    for (int kt = kt0; kt < kt1; ++kt) {
        if (kt + 1 < kt1) fetch(kt + 1);
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            float a[TM][4], b[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / 2) + i * 32 + li;
                if (AKC) {
                    const vnp_f4 v = *(const vnp_f4*)&As[row * LDAS + 8 * q + 4 * lh];
                    a[i][0] = v[0]; a[i][1] = v[1]; a[i][2] = v[2]; a[i][3] = v[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) a[i][s] = As[(8 * q + 4 * lh + s) * LDAS + row];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * (BN / 2) + j * 32 + li;
                if (BKC) {
                    const vnp_f4 v = *(const vnp_f4*)&Bs[col * LDBS + 8 * q + 4 * lh];
                    b[j][0] = v[0]; b[j][1] = v[1]; b[j][2] = v[2]; b[j][3] = v[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) b[j][s] = Bs[(8 * q + 4 * lh + s) * LDBS + col];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < kt1) {
            stage();
            __syncthreads();
        }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* Cz = p.C + (size_t)blockIdx.y * p.slab;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (row < p.M && col < p.N) {
                    float v = acc[i][j][e];
                    if (p.bias) v += p.bias[(size_t)((row / p.rows_per_group) * 3 + row % 3) * p.ldbias + col];
                    float* dst = Cz + (size_t)row * p.ldc + col;
                    if (p.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

// out[r][c] (+)= sum_z ws[z][r][c]; ws slabs are dense [M, N]
__global__ void vnp_reduce_slabs_kernel(const float* __restrict__ ws, int nsplit, long long slab, float* __restrict__ C,
                                        int M, int N, int ldc, int accumulate)
{
    const long long total = (long long)M * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += ws[(size_t)z * slab + i];
        const int r = (int)(i / N), c = (int)(i % N);
        float* dst = C + (size_t)r * ldc + c;
        *dst = accumulate ? *dst + s : s;
    }
}

// ---------------------------------------------------------------------------------------------- C-ABI
// op_a: 0 = A is [M, K] row-major (K contiguous), 1 = A is [K, M] row-major (M contiguous)
// op_b: 0 = B is [N, K] row-major (K contiguous), 1 = B is [K, N] row-major (N contiguous)
// nsplit > 1: split-K; `ws` must hold nsplit * M * N floats; the result is reduced into C afterwards.
extern "C" int vnp_gemm(hipStream_t stream, int op_a, int op_b, int M, int N, int K,
                        const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                        const float* A2, int lda2, const float* B2, int ldb2, int K2,
                        const float* bias, int ldbias, int rows_per_group,
                        int accumulate, int nsplit, float* ws)
{
    if (M <= 0 || N <= 0 || K < 0 || !A || !B || !C) return VNP_EINVAL;
    if (K2 > 0 && (!A2 || !B2)) return VNP_EINVAL;
    if (bias && rows_per_group <= 0) return VNP_EINVAL;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 1 && !ws) return VNP_EINVAL;
    constexpr int BK = 32;
    VnpGemmArgs p;
    p.A = A; p.B = B; p.A2 = A2; p.B2 = B2; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.K2 = K2 > 0 ? K2 : 0;
    p.lda = lda; p.ldb = ldb; p.lda2 = lda2; p.ldb2 = ldb2; p.ldbias = ldbias;
    p.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
    const int nk = vnp_cdiv(K, BK) + (p.K2 ? vnp_cdiv(p.K2, BK) : 0);
    if (nsplit > nk) nsplit = nk > 0 ? nk : 1;
    p.tiles_per_split = vnp_cdiv(nk > 0 ? nk : 1, nsplit);
    nsplit = vnp_cdiv(nk > 0 ? nk : 1, p.tiles_per_split);
    if (nsplit > 1) {
        if (bias) return VNP_EINVAL;
        p.C = ws; p.ldc = N; p.slab = (long long)M * N; p.accumulate = 0;
    } else {
        p.C = C; p.ldc = ldc; p.slab = 0; p.accumulate = accumulate;
    }
    const bool big_m = (op_a == 0);      // activations on the M side: 192-row tiles; weight gradients: 128
    dim3 block(256);
    if (op_a == 0 && op_b == 0) {
        dim3 grid(vnp_cdiv(M, 192) * vnp_cdiv(N, 128), nsplit);
        hipLaunchKernelGGL((vnp_gemm_kernel<192, 128, BK, true, true>), grid, block, 0, stream, p);
    } else if (op_a == 0 && op_b == 1) {
        dim3 grid(vnp_cdiv(M, 192) * vnp_cdiv(N, 128), nsplit);
        hipLaunchKernelGGL((vnp_gemm_kernel<192, 128, BK, true, false>), grid, block, 0, stream, p);
    } else if (op_a == 1 && op_b == 1) {
        dim3 grid(vnp_cdiv(M, 128) * vnp_cdiv(N, 128), nsplit);
        hipLaunchKernelGGL((vnp_gemm_kernel<128, 128, BK, false, false>), grid, block, 0, stream, p);
    } else {
        dim3 grid(vnp_cdiv(M, 128) * vnp_cdiv(N, 128), nsplit);
        hipLaunchKernelGGL((vnp_gemm_kernel<128, 128, BK, false, true>), grid, block, 0, stream, p);
    }
    (void)big_m;
    int rc = vnp_last_error();
    if (rc) return rc;
    if (nsplit > 1) {
        const long long total = (long long)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(vnp_reduce_slabs_kernel, dim3(blocks), dim3(256), 0, stream,
                           (const float*)ws, nsplit, (long long)M * N, C, M, N, ldc, accumulate);
        rc = vnp_last_error();
    }
    return rc;
}
