// fp32 GEMM on the CDNA4 matrix cores.  Two kernels with the same tiling, arguments and results to fp32 rounding:
//   * gemm_split_kernel.h (default): every operand is split exactly into three bf16 terms on its way into LDS and six
//     v_mfma_f32_16x16x32_bf16 products are accumulated in fp32 — error vs fp64 at or below the fp32 kernel's
//     (tests/test_gpu_gemm.py), 1.45-1.75x its speed;
//   * gemm_kernel.h (vnp_gemm_set_mode(1) / VNP_GEMM_MODE=fp32): v_mfma_f32_32x32x2_f32, 157 TF/s peak.
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
#include "gemm_host.h"

// 0 = bf16x3 split (fp32-grade), 1 = fp32 MFMA.  The library's only process-wide setting: an atomic switch that every vnp_gemm*
// call reads once on entry (a concurrent vnp_gemm_set_mode affects later calls, never a launch in progress).
static std::atomic<int> g_vnp_gemm_mode{0};

extern "C" int vnp_gemm_set_mode(int mode)
{
    if (mode != 0 && mode != 1) return VNP_EINVAL;
    g_vnp_gemm_mode.store(mode, std::memory_order_relaxed);
    return 0;
}

extern "C" int vnp_gemm_get_mode(void) { return g_vnp_gemm_mode.load(std::memory_order_relaxed); }

thread_local VnpProfReq g_vnp_prof = {nullptr, nullptr, nullptr};

extern "C" int vnp_gemm_profile_next(hipEvent_t e0, hipEvent_t e1, int* kind)
{
    if (!e0 || !e1) return VNP_EINVAL;
    g_vnp_prof = {e0, e1, kind};
    return 0;
}

// Drop a pending request (its events are about to be destroyed, or the armed call took a path that brackets no kernel).
extern "C" int vnp_gemm_profile_cancel(void)
{
    g_vnp_prof = {nullptr, nullptr, nullptr};
    return 0;
}


// out[r][c] (+)= sum_z ws[z][r][c]; ws slabs are dense [M, N].  A workgroup = 64 consecutive outputs x 4 slab lanes: lane q sums the
// slabs z = q, q + 4, ... with four loads in flight, the four partial sums meet in LDS in a fixed order (deterministic).  (One thread
// per output walking all nsplit slabs was a dependent chain of up to 256 loads: 19 us per call at 1 TB/s, 109 calls per step in config 5.)
// The slabs are summed in DOUBLE (the kernel is bound by its loads): a weight gradient over 10^5 - 10^6 rows then carries the fp32
// rounding of its K-slices only, which average out over the slabs, instead of another sqrt(nsplit) 2^-24 of the running total —
// it matters where the gradient is multiplied on into a cancelling product (ops.WeightProductFn).
__global__ void __launch_bounds__(256) vnp_reduce_slabs_kernel(const float* __restrict__ ws, int nsplit, long long slab, float* __restrict__ C,
                                                               int M, int N, int ldc, int accumulate, const float* __restrict__ resid, int ldr)
{
    __shared__ double part[4][64];
    const long long total = (long long)M * N;
    const int ox = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
        const long long i = base + ox;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (i < total) {
            const float* src = ws + i;
            int z = q;
            for (; z + 12 < nsplit; z += 16) {
                const float v0 = src[(size_t)z * slab], v1 = src[(size_t)(z + 4) * slab], v2 = src[(size_t)(z + 8) * slab], v3 = src[(size_t)(z + 12) * slab];
                s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
            }
            for (; z < nsplit; z += 4) s0 += (double)src[(size_t)z * slab];
        }
        part[q][ox] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (q == 0 && i < total) {
            const float s = (float)((part[0][ox] + part[1][ox]) + (part[2][ox] + part[3][ox]));
            const int r = (int)(i / N), c = (int)(i % N);
            float* dst = C + (size_t)r * ldc + c;
            *dst = accumulate ? (resid ? resid[(size_t)r * ldr + c] : *dst) + s : s;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- degenerate shapes
// K = 1 (a layer whose only per-point input channel is one VN feature: the attention-folding stages), N = 1 or M = 1 (its
// gradients, VNLinear(C -> 1)) are outer products, matrix-vector products and weighted column sums: HBM-bound streaming
// kernels, not tensor-core work (the MFMA kernel spends a full 32-deep K tile on them: 0.95 ms instead of 0.4 for the 1.6 GB
// outer product of config 4).
// C[r][c] (+)= a[r] * b[c] (+ per-sample VN bias)
// VEC4 (N % 4 == 0, unit-stride b, 16-byte aligned C / bias rows): a thread owns four consecutive columns: 16-byte stores, a wave writes 1 KB per row
template <bool VEC4>
__global__ void __launch_bounds__(256) vnp_gemm_rank1_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, long long bstride,
                                                            float* __restrict__ C, int ldc, int M, int N, const float* __restrict__ bias,
                                                            int ldbias, int rows_per_group, int accumulate)
{
    if (VEC4) {
        const int c = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
        if (c >= N) return;
        const vnp_f4 bv = *(const vnp_f4*)(B + c);
        const int r0 = blockIdx.y * 64 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r1 = min(M, (int)(blockIdx.y + 1) * 64);
        int grp = 0, rem = 0, t = 0;
        if (bias) { grp = r0 / rows_per_group; rem = r0 - grp * rows_per_group; t = r0 % 3; }
        for (int r = r0; r < r1; r += 4) {
            vnp_f4 v = A[(size_t)r * lda] * bv;
            if (bias) {
                v += *(const vnp_f4*)(bias + (size_t)(grp * 3 + t) * ldbias + c);
                rem += 4; t += 1;
                while (rem >= rows_per_group) { rem -= rows_per_group; ++grp; }
                if (t >= 3) t -= 3;
            }
            vnp_f4* dst = (vnp_f4*)(C + (size_t)r * ldc + c);
            if (accumulate) v += *dst;
            __builtin_nontemporal_store(v, dst);
        }
        return;
    }
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    if (c >= N) return;
    const float bv = B[(size_t)c * bstride];
    // a thread's rows are r0, r0 + 4, ...: group index and component advance incrementally (one 32-bit division per thread instead of a
    // 64-bit division and a modulo per element: the kernel is a pure 0.8 GB store stream in config 4 and ran at half the store rate)
    const int r0 = blockIdx.y * 64 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r1 = min(M, (int)(blockIdx.y + 1) * 64);
    int grp = 0, rem = 0, t = 0;
    if (bias) { grp = r0 / rows_per_group; rem = r0 - grp * rows_per_group; t = r0 % 3; }
    for (int r = r0; r < r1; r += 4) {
        float v = A[(size_t)r * lda] * bv;
        if (bias) {
            v += bias[(size_t)(grp * 3 + t) * ldbias + c];
            rem += 4; t += 1;                                  // (r + 4) % 3 = (r % 3 + 1) % 3
            while (rem >= rows_per_group) { rem -= rows_per_group; ++grp; }
            if (t >= 3) t -= 3;
        }
        float* dst = C + (size_t)r * ldc + c;
        if (accumulate) v += *dst;
        __builtin_nontemporal_store(v, dst);
    }
}

// C[r] (+)= sum_k A[r][k] * b[k]   (A K-contiguous; one wave per row)
__global__ void __launch_bounds__(256) vnp_gemm_matvec_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, long long bstride,
                                                             float* __restrict__ C, int ldc, int M, int K, int accumulate)
{
    const long long r = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (r >= M) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    if (bstride == 1 && (K & 3) == 0 && (lda & 3) == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0) {          // 16-byte loads: 1 KB per wave and step
        for (int k = lane * 4; k < K; k += 256) {
            const vnp_f4 av = __builtin_nontemporal_load((const vnp_f4*)(A + (size_t)r * lda + k));
            const vnp_f4 bv = *(const vnp_f4*)(B + k);
            s = fmaf(av[3], bv[3], fmaf(av[2], bv[2], fmaf(av[1], bv[1], fmaf(av[0], bv[0], s))));
        }
    } else
        for (int k = lane; k < K; k += 64) s = fmaf(__builtin_nontemporal_load(&A[(size_t)r * lda + k]), B[(size_t)k * bstride], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) C[(size_t)r * ldc] = accumulate ? C[(size_t)r * ldc] + s : s;
}

// slab[z][c] = sum over the K-rows of split z of X[k][c] * w[k]   (X C-contiguous: weighted column sums)
__global__ void __launch_bounds__(256) vnp_gemm_wcolsum_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ w, long long wstride,
                                                              int K, int Ccols, int rows_per_split, float* __restrict__ slab)
{
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
    const int c = blockIdx.x * 64 + cx;
    const long long k0 = (long long)blockIdx.y * rows_per_split, k1 = min((long long)K, k0 + rows_per_split);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                  // four rows in flight per thread
    if (c < Ccols) {
        long long k = k0 + py;
        for (; k + 12 < k1; k += 16) {
            const float x0 = __builtin_nontemporal_load(&X[(size_t)k * ldx + c]), x1 = __builtin_nontemporal_load(&X[(size_t)(k + 4) * ldx + c]);
            const float x2 = __builtin_nontemporal_load(&X[(size_t)(k + 8) * ldx + c]), x3 = __builtin_nontemporal_load(&X[(size_t)(k + 12) * ldx + c]);
            s0 = fmaf(x0, w[(size_t)k * wstride], s0); s1 = fmaf(x1, w[(size_t)(k + 4) * wstride], s1);
            s2 = fmaf(x2, w[(size_t)(k + 8) * wstride], s2); s3 = fmaf(x3, w[(size_t)(k + 12) * wstride], s3);
        }
        for (; k < k1; k += 4) s0 = fmaf(__builtin_nontemporal_load(&X[(size_t)k * ldx + c]), w[(size_t)k * wstride], s0);
    }
    const float s = (s0 + s1) + (s2 + s3);
    red[py][cx] = s;
    __syncthreads();
    if (py == 0 && c < Ccols) slab[(size_t)blockIdx.y * Ccols + c] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}

// ---- tile choice for activations on M (A [M, K], aligned shapes).  A launch is a number of ROUNDS of workgroups (two resident per
// CU), and for the mid-sized products of the transformer configs (12288 .. 49152 rows) the fill of the last round decides the time.
// Measured on M x N x K grids of such shapes (scripts/experiments/gemm_tile_choice_bench.hip, profiles/r03_gemm_tile_choice.txt), in units
// of one lone 192-row workgroup per CU:  a round that fills at most ONE slot per CU costs 1.0 / 0.77 / 0.68 with 192- / 128- / 96-row
// tiles, a round that uses both slots 1.45 / 1.08 / 0.95; the 512-thread wide kernel (192 x 256, one per CU) 1.38 per round of 256.
// The cheapest height that divides M wins (192 unless another is clearly cheaper): 96 rows up to one 192-row tile per CU (-10..-35 %),
// 128 rows where 192 would spill a nearly empty second round (N = 1152: -14 %), the wide tile for big N % 256 == 0 products (-2..-5 %).
// Returns the row-tile height (192 also when no aligned height divides M: the bounds-checked kernel); wide = the 192 x 256 kernel.
static int vnp_choose_tile(int op_b, int M, int N, int nsplit, int nbatch, bool has_bias, bool& wide)
{
    wide = false;
    int bm = 192;
    const long long sb = (long long)nsplit * nbatch, slots = 2LL * vnp_num_cus();
    auto cost = [&](int h, double c1, double c2) -> double {
        if (M % h) return 1e30;
        const long long it = (long long)(M / h) * (N / 128) * sb, full = it / slots, r = it % slots;
        return (double)full * c2 + (r == 0 ? 0.0 : 2 * r <= slots ? c1 : c2);
    };
    double best = cost(192, 1.0, 1.45);
    const double c128 = cost(128, 0.77, 1.08);
    if (c128 < 0.97 * best) { best = c128; bm = 128; }
    if (!has_bias && nbatch == 1) {           // (the 96-row instantiations carry no bias epilogue and no batch walk)
        const double c96 = cost(96, 0.68, 0.95);
        if (c96 < 0.97 * best) { best = c96; bm = 96; }
    }
    if (op_b == 0 && M % 192 == 0 && N % 256 == 0) {
        const long long items_w = (long long)(M / 192) * (N / 256) * sb;
        const double cw = (double)((items_w + vnp_num_cus() - 1) / vnp_num_cus()) * 1.38;
        if (cw < 0.99 * best) { best = cw; bm = 192; wide = true; }
    }
    return best >= 1e30 ? 192 : bm;
}

// What vnp_gemm would run for an aligned split-mode product with A [M, K] (op_a = 0): the row-tile height 192 / 128 / 96, or 256 for the
// 512-thread 192 x 256 kernel (tests and DESIGN.md quote it; N % 128 == 0 and K % 32 == 0 assumed).
extern "C" int vnp_gemm_tile_rows(int op_b, int M, int N, int nsplit, int nbatch, int has_bias)
{
    if (M <= 0 || N <= 0 || N % 128) return VNP_EINVAL;
    bool wide = false;
    const int bm = vnp_choose_tile(op_b, M, N, nsplit < 1 ? 1 : nsplit, nbatch < 1 ? 1 : nbatch, has_bias != 0, wide);
    return wide ? 256 : bm;
}

// ---------------------------------------------------------------------------------------------- C-ABI
// op_a: 0 = A is [M, K] row-major (K contiguous), 1 = A is [K, M] row-major (M contiguous)
// op_b: 0 = B is [N, K] row-major (K contiguous), 1 = B is [K, N] row-major (N contiguous)
// nsplit > 1: split-K; `ws` must hold nsplit * M * N floats; the result is reduced into C afterwards.
static int vnp_gemm_impl(hipStream_t stream, int op_a, int op_b, int M, int N, int K,
                         const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                         const float* bias, int ldbias, int rows_per_group,
                         int accumulate, int nsplit, float* ws,
                         int nbatch, long long bsa, long long bsb, long long bsc,
                         const unsigned* amax_a = nullptr, const unsigned* amax_b = nullptr, const float* resid = nullptr, int ldr = 0)
{
    if (M <= 0 || N <= 0 || K < 0 || !A || !B || !C) return VNP_EINVAL;
    if (resid && (!accumulate || nbatch > 1 || ldr < N)) return VNP_EINVAL;
    if (bias && rows_per_group <= 0) return VNP_EINVAL;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 1 && !ws) return VNP_EINVAL;
    if (nbatch < 1 || (nbatch > 1 && nsplit > 1)) return VNP_EINVAL;
    if (nbatch == 1 && K >= 1 && !resid) {
        // outer product: A [M,1] (op_a = 0) x b [N]
        if (K == 1 && op_a == 0 && M <= 2147483647 / 2) {
            const long long bstr = op_b == 0 ? ldb : 1;
            const bool v4 = bstr == 1 && N % 4 == 0 && ldc % 4 == 0 && (uintptr_t)C % 16 == 0 && (uintptr_t)B % 16 == 0 &&
                            (!bias || (ldbias % 4 == 0 && (uintptr_t)bias % 16 == 0));
            vnp_prof_begin(stream);
            if (v4) VNP_KLAUNCH(vnp_gemm_rank1_kernel<true>, dim3(vnp_cdiv(N, 256), vnp_cdiv(M, 64)), dim3(256), 0, stream, A, lda, B, bstr,
                                       C, ldc, M, N, bias, ldbias, rows_per_group > 0 ? rows_per_group : 1, accumulate);
            else VNP_KLAUNCH(vnp_gemm_rank1_kernel<false>, dim3(vnp_cdiv(N, 64), vnp_cdiv(M, 64)), dim3(256), 0, stream, A, lda, B, bstr,
                                    C, ldc, M, N, bias, ldbias, rows_per_group > 0 ? rows_per_group : 1, accumulate);
            vnp_prof_end(stream, 2);
            return vnp_last_error();
        }
        // matrix-vector: A [M,K] K-contiguous, one output column
        if (N == 1 && op_a == 0 && !bias) {
            vnp_prof_begin(stream);
            VNP_KLAUNCH(vnp_gemm_matvec_kernel, dim3(vnp_cdiv(M, 4)), dim3(256), 0, stream, A, lda, B, (long long)(op_b == 0 ? 1 : ldb),
                               C, ldc, M, K, accumulate);
            vnp_prof_end(stream, 2);
            return vnp_last_error();
        }
        // weighted column sums: C[m][0] = sum_k A[k][m] B[k][0]  or  C[0][n] = sum_k A[k][0] B[k][n]   (both operands [K, .])
        if (op_a == 1 && op_b == 1 && (N == 1 || M == 1) && !bias && K > 4096) {
            const float* X = N == 1 ? A : B; const int ldx = N == 1 ? lda : ldb, cols = N == 1 ? M : N;
            const float* wv = N == 1 ? B : A; const long long ws_ = N == 1 ? ldb : lda;
            int ns = nsplit > 1 && ws ? nsplit : 1;
            if (ns > 65535) ns = 65535;
            const int rps = vnp_cdiv(K, ns);
            ns = vnp_cdiv(K, rps);
            float* slab = ws;
            if (ns > 1) {
            vnp_prof_begin(stream);
            VNP_KLAUNCH(vnp_gemm_wcolsum_kernel, dim3(vnp_cdiv(cols, 64), ns), dim3(256), 0, stream, X, ldx, wv, ws_, K, cols, rps, slab);
            vnp_prof_end(stream, 2);
            int rc2 = vnp_last_error();
            if (rc2) return rc2;
            int blocks = vnp_cdiv(cols, 64);
            hipLaunchKernelGGL(vnp_reduce_slabs_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)slab, ns, (long long)cols, C,
                               M, N, ldc, accumulate, (const float*)nullptr, 0);
            return vnp_last_error();
            }                                                  // without split-K scratch the MFMA path below handles it
        }
    }
    constexpr int BK = 32;
    VnpGemmArgs p{};
    p.A = A; p.B = B; p.bias = bias;
    p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldbias = ldbias;
    p.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
    p.bsa = bsa; p.bsb = bsb; p.bsc = bsc;
    const int nk = vnp_cdiv(K, BK) > 0 ? vnp_cdiv(K, BK) : 1;
    if (nsplit > nk) nsplit = nk;
    p.tiles_per_split = vnp_cdiv(nk, nsplit);
    nsplit = vnp_cdiv(nk, p.tiles_per_split);
    if (nsplit > 1) {
        if (bias) return VNP_EINVAL;
        p.C = ws; p.ldc = N; p.slab = (long long)M * N; p.accumulate = 0;
    } else {
        p.C = C; p.ldc = ldc; p.slab = 0; p.accumulate = accumulate;
        p.resid = resid; p.ldr = ldr;
    }
    p.nsplit = nsplit; p.nbatch = nbatch;
    p.xcd_split = (nsplit > 1 && nsplit % 8 == 0 && nbatch == 1 && vnp_gemm_get_mode() == 0) ? 1 : 0;
    p.col_group = 8;                     // measured: +3 % at N = 2048 (B panels of 8 column tiles stay in the XCD's L2)
    p.pool_val = nullptr; p.pool_idx = nullptr; p.pool_x = nullptr; p.pool_gs = 0;
    const bool split = vnp_gemm_get_mode() == 0;
    const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) &&
                         (bsa % 4 == 0) && (bsb % 4 == 0);
    bool wide = false;
    const bool tileable = op_a == 0 && split && aligned && N % 128 == 0 && K % BK == 0 && K > 0 && !amax_a;
    const int bm = tileable ? vnp_choose_tile(op_b, M, N, nsplit, nbatch, bias != nullptr, wide) : (op_a == 0 ? 192 : 128);
    const long long items = (long long)vnp_cdiv(M, bm) * vnp_cdiv(N, 128) * nsplit * nbatch;
    if (items > 2147483647LL) return VNP_EINVAL;
    const bool fast = aligned && (M % bm == 0) && (N % 128 == 0) && (K % BK == 0) && K > 0;
    // two fp16 planes / three products when the caller supplies max |A| and max |B| (vnp_gemm_ex); aligned shapes with K >= 64 only
    const bool h2 = split && fast && amax_a && amax_b && K >= 64;
    p.amax_a = amax_a; p.amax_b = amax_b;
    // resident workgroups per CU: 2 for the 192-row tiles (~240 VGPRs), 3 for the aligned 128-row weight-gradient kernel (151)
    const long long cap = (long long)vnp_num_cus() * ((!split && bm == 128 && fast && op_b == 1) ? 3 : 2);
    dim3 grid((unsigned)(items < cap ? items : cap)), block(256);
#define VNP_LAUNCH(BM_, AKC_, BKC_)                                                                                    \
    do {                                                                                                                \
        if (h2) VNP_KLAUNCH((vnp_gemm_split_kernel<BM_, 128, AKC_, BKC_, true, 0, 0, 1>), grid, block, 0, stream, p); \
        else if (split) {                                                                                               \
            if (fast) VNP_KLAUNCH((vnp_gemm_split_kernel<BM_, 128, AKC_, BKC_, true>), grid, block, 0, stream, p);  \
            else      VNP_KLAUNCH((vnp_gemm_split_kernel<BM_, 128, AKC_, BKC_, false>), grid, block, 0, stream, p); \
        } else {                                                                                                        \
            if (fast) VNP_KLAUNCH((vnp_gemm_kernel<BM_, 128, BK, AKC_, BKC_, true>), grid, block, 0, stream, p);    \
            else      VNP_KLAUNCH((vnp_gemm_kernel<BM_, 128, BK, AKC_, BKC_, false>), grid, block, 0, stream, p);   \
        }                                                                                                               \
    } while (0)
    vnp_prof_begin(stream);
    if (wide) {
        // both operands K-contiguous, N % 256 == 0: the 512-thread kernel with the 192 x 256 tile — the same arithmetic bit for bit,
        // 0.7x the staging work per MFMA (scripts/experiments/gemm_wide_bench.hip)
        const long long items_w = (long long)(M / 192) * (N / 256) * nsplit * nbatch;
        VNP_KLAUNCH((vnp_gemm_split_wide_kernel<0>), dim3((unsigned)(items_w < vnp_num_cus() ? items_w : vnp_num_cus())), dim3(512), 0, stream, p);
    } else if (tileable && fast && bm == 96) {
        if (op_b == 0) VNP_KLAUNCH((vnp_gemm_split_kernel<96, 128, true, true, true>), grid, block, 0, stream, p);
        else           VNP_KLAUNCH((vnp_gemm_split_kernel<96, 128, true, false, true>), grid, block, 0, stream, p);
    } else if (tileable && fast && bm == 128) {
        if (op_b == 0) VNP_KLAUNCH((vnp_gemm_split_kernel<128, 128, true, true, true>), grid, block, 0, stream, p);
        else           VNP_KLAUNCH((vnp_gemm_split_kernel<128, 128, true, false, true>), grid, block, 0, stream, p);
    }
    else if (op_a == 0 && op_b == 0) VNP_LAUNCH(192, true, true);
    else if (op_a == 0 && op_b == 1) VNP_LAUNCH(192, true, false);
    else if (op_a == 1 && op_b == 1 && split && fast && !h2 && !bias && M % 256 == 0 &&
             (long long)(M / 256) * (N / 128) * nsplit * nbatch >= vnp_num_cus() &&
             ((long long)(M / 256) * (N / 128) * nsplit * nbatch % vnp_num_cus() == 0 ||
              (long long)(M / 256) * (N / 128) * nsplit * nbatch >= 4LL * vnp_num_cus())) {
        // weight gradients whose 256 x 128 tiles fill whole rounds of one 512-thread workgroup per CU: 0.75x the staging work per MFMA,
        // evenly divided (vnp_gemm_split_wide_tn_kernel); measured +2.5 .. 4 % on the step's shapes, same bits
        const long long items_w = (long long)(M / 256) * (N / 128) * nsplit * nbatch;
        VNP_KLAUNCH((vnp_gemm_split_wide_tn_kernel<0>), dim3((unsigned)(items_w < vnp_num_cus() ? items_w : vnp_num_cus())), dim3(512), 0, stream, p);
    }
    else if (op_a == 1 && op_b == 1) VNP_LAUNCH(128, false, false);
    else                             VNP_LAUNCH(128, false, true);
#undef VNP_LAUNCH
    vnp_prof_end(stream, h2 ? 5 : split ? 0 : 1);
    int rc = vnp_last_error();
    if (rc) return rc;
    if (nsplit > 1) {
        const long long total = (long long)M * N;
        int blocks = (int)((total + 63) / 64 > 16384 ? 16384 : (total + 63) / 64);
        hipLaunchKernelGGL(vnp_reduce_slabs_kernel, dim3(blocks), dim3(256), 0, stream,
                           (const float*)ws, nsplit, (long long)M * N, C, M, N, ldc, accumulate, resid, ldr);
        rc = vnp_last_error();
    }
    return rc;
}

// ---- [p | d] = x [Wf; Wd]^T with the BatchNorm statistics of p taken in the epilogue --------------------------------
// merges 32 consecutive 32-point Welford partials into the 1024-point chunks vnp_vn_bn_finalize expects
__global__ void __launch_bounds__(256) vnp_stats_merge32_kernel(const float2* __restrict__ part32, int C, float2* __restrict__ out)
{
    __shared__ double rm[4][64], rM[4][64];
    const int cx = threadIdx.x & 63, py = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
    const int c = blockIdx.x * 64 + cx;
    double na = 0.0, ma = 0.0, Ma = 0.0;
    if (c < C) {
        const float2* src = part32 + ((size_t)blockIdx.y * 32 + py * 8) * C + c;
        for (int k = 0; k < 8; ++k) {
            const float2 v = src[(size_t)k * C];
            if (na == 0.0) { na = 32.0; ma = v.x; Ma = v.y; }
            else { const double n = na + 32.0, delta = (double)v.x - ma; ma += delta * 32.0 / n; Ma += (double)v.y + delta * delta * na * 32.0 / n; na = n; }
        }
    }
    rm[py][cx] = ma; rM[py][cx] = Ma;
    __syncthreads();
    if (py == 0 && c < C) {
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            const double nb = 256.0, n = na + nb, delta = rm[j][cx] - ma;
            ma += delta * nb / n; Ma += rM[j][cx] + delta * delta * na * nb / n; na = n;
        }
        out[(size_t)blockIdx.y * C + c] = make_float2((float)ma, (float)Ma);
    }
}

extern "C" long long vnp_gemm_vn_stats_ws_floats(long long M, int stats_cols) { return (M / 96) * (long long)stats_cols * 2; }

// C[M, N] = A[M, K] B[N, K]^T (+ per-sample VN bias), and for the first stats_cols columns the Welford partials
// (mean, M2) of n = |row triple| + eps per 1024 points in part1024 [M/3072][stats_cols][2] — what vnp_vn_norm_stats would
// have produced from C, without reading C back.  ws: vnp_gemm_vn_stats_ws_floats(M, stats_cols) floats.
// VNP_EINVAL unless split GEMM mode, M % 3072 == 0, N % 128 == 0, K % 32 == 0, stats_cols % 64 == 0, 16-byte alignment.
// amax_a / amax_b (both or neither): device words with the bit pattern of max |A|, max |B| (vnp_absmax) -> two fp16 planes, three products
extern "C" int vnp_gemm_vn_stats_ex(hipStream_t st, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                    const float* bias, int ldbias, int rows_per_group, int stats_cols, float* part1024, float* ws,
                                    const unsigned* amax_a, const unsigned* amax_b)
{
    if (!A || !B || !C || !part1024 || !ws || M <= 0 || N <= 0 || K <= 0 || stats_cols <= 0) return VNP_EINVAL;
    if (vnp_gemm_get_mode() != 0 || M % 3072 || N % 128 || K % 32 || stats_cols % 64 || stats_cols > N || lda % 4 || ldb % 4 ||
        (uintptr_t)A % 16 || (uintptr_t)B % 16 || (bias && rows_per_group <= 0) || (bias && rows_per_group % 3))
        return VNP_EINVAL;
    VnpGemmArgs p{};
    p.A = A; p.B = B; p.C = C; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldbias = ldbias;
    p.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
    p.tiles_per_split = K / 32; p.nsplit = 1; p.nbatch = 1; p.col_group = 8;
    p.stats_part = (float2*)ws; p.stats_cols = stats_cols;
    const long long items = (long long)(M / 192) * (N / 128);
    const long long cap = (long long)vnp_num_cus() * 2;
    dim3 grid((unsigned)(items < cap ? items : cap)), block(256);
    if ((amax_a == nullptr) != (amax_b == nullptr)) return VNP_EINVAL;
    p.amax_a = amax_a; p.amax_b = amax_b;
    const long long items_w = (long long)(M / 192) * (N / 256);
    vnp_prof_begin(st);
    if (amax_a && K >= 64) VNP_KLAUNCH((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 2, 1>), grid, block, 0, st, p);
    else if (N % 256 == 0 &&                                     // the wide tile where its rounds are cheaper (vnp_gemm_impl)
             (double)((items_w + vnp_num_cus() - 1) / vnp_num_cus()) * 1.38 <
             0.99 * ((double)(items / cap) * 1.45 + (items % cap == 0 ? 0.0 : 2 * (items % cap) <= cap ? 1.0 : 1.45)))
        VNP_KLAUNCH((vnp_gemm_split_wide_kernel<2>), dim3((unsigned)(items_w < vnp_num_cus() ? items_w : vnp_num_cus())), dim3(512), 0, st, p);
    else VNP_KLAUNCH((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 2>), grid, block, 0, st, p);
    vnp_prof_end(st, (amax_a && K >= 64) ? 5 : 0);
    hipLaunchKernelGGL(vnp_stats_merge32_kernel, dim3(vnp_cdiv(stats_cols, 64), M / 3072), dim3(256), 0, st,
                       (const float2*)ws, stats_cols, (float2*)part1024);
    return vnp_last_error();
}

extern "C" int vnp_gemm_vn_stats(hipStream_t st, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                 const float* bias, int ldbias, int rows_per_group, int stats_cols, float* part1024, float* ws)
{
    return vnp_gemm_vn_stats_ex(st, M, N, K, A, lda, B, ldb, C, ldc, bias, ldbias, rows_per_group, stats_cols, part1024, ws, nullptr, nullptr);
}

// ---- max |x| of a matrix, as the bit pattern of the fp32 value (non-negative floats order like unsigned integers; a NaN ranks above
// Inf, so a poisoned operand is visible in the word).  out is overwritten.
__global__ void __launch_bounds__(256) vnp_absmax_kernel(const float* __restrict__ x, long long ld, long long rows, int cols4, unsigned* __restrict__ out)
{
    unsigned m = 0;
    const long long total = rows * cols4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cols4; const int c = (int)(i - r * cols4);
        const vnp_f4 v = __builtin_nontemporal_load((const vnp_f4*)(x + r * ld) + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const unsigned b = __float_as_uint(v[e]) & 0x7fffffffu; m = b > m ? b : m; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// out[0] = bits of max |x[r][c]| over rows x cols (cols % 4 == 0, ld % 4 == 0, x 16-byte aligned; else VNP_EINVAL)
extern "C" int vnp_absmax(hipStream_t st, const float* x, long long ld, long long rows, int cols, unsigned* out)
{
    if (!x || !out || rows <= 0 || cols <= 0 || cols % 4 || ld % 4 || ld < cols || (uintptr_t)x % 16) return VNP_EINVAL;
    hipError_t e = hipMemsetAsync(out, 0, 4, st);
    if (e != hipSuccess) return (int)e;
    const long long total = rows * (cols / 4);
    long long blocks = (total + 255) / 256;
    const long long cap = (long long)vnp_num_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(vnp_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, ld, rows, cols / 4, out);
    return vnp_last_error();
}

extern "C" int vnp_gemm(hipStream_t stream, int op_a, int op_b, int M, int N, int K,
                        const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                        const float* bias, int ldbias, int rows_per_group,
                        int accumulate, int nsplit, float* ws)
{
    return vnp_gemm_impl(stream, op_a, op_b, M, N, K, A, lda, B, ldb, C, ldc, bias, ldbias, rows_per_group,
                         accumulate, nsplit, ws, 1, 0, 0, 0);
}

// vnp_gemm with the operands' absolute maxima (device words: bit pattern of max |A| and of max |B| over the WHOLE operand, from
// vnp_absmax or from the kernel that produced the operand): aligned shapes with K >= 64 in the default GEMM mode then run with two fp16
// planes and three MFMA products instead of three bf16 planes and six (gemm_split_kernel.h, PREC = 1), every other shape as vnp_gemm does.
// The words must be upper bounds of the operands' magnitudes: an element above the word overflows fp16 and gives Inf / NaN.
extern "C" int vnp_gemm_ex(hipStream_t stream, int op_a, int op_b, int M, int N, int K,
                           const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                           const float* bias, int ldbias, int rows_per_group,
                           int accumulate, int nsplit, float* ws, const unsigned* amax_a, const unsigned* amax_b)
{
    if ((amax_a == nullptr) != (amax_b == nullptr)) return VNP_EINVAL;
    return vnp_gemm_impl(stream, op_a, op_b, M, N, K, A, lda, B, ldb, C, ldc, bias, ldbias, rows_per_group,
                         accumulate, nsplit, ws, 1, 0, 0, 0, amax_a, amax_b);
}

// C = op(A) op(B) + R: a residual connection taken in the epilogue (R [M][N], row stride ldr; R may be C itself).  The matrix kernels
// only (no K = 1 / N = 1 / M = 1 streaming forms: those shapes run on the bounds-checked matrix kernel here).
extern "C" int vnp_gemm_add(hipStream_t stream, int op_a, int op_b, int M, int N, int K,
                            const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                            const float* R, int ldr, int nsplit, float* ws)
{
    if (!R) return VNP_EINVAL;
    return vnp_gemm_impl(stream, op_a, op_b, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, 0, 1, nsplit, ws, 1, 0, 0, 0, nullptr, nullptr, R, ldr);
}

// nbatch independent products C_i = op(A_i) op(B_i) with element strides bsa / bsb / bsc (attention: one per (sample, head))
extern "C" int vnp_gemm_batched(hipStream_t stream, int op_a, int op_b, int M, int N, int K,
                                const float* A, int lda, long long bsa, const float* B, int ldb, long long bsb,
                                float* C, int ldc, long long bsc, int nbatch, int accumulate)
{
    return vnp_gemm_impl(stream, op_a, op_b, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, 0,
                         accumulate, 1, nullptr, nbatch, bsa, bsb, bsc);
}
