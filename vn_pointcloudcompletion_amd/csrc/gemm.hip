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
#include "gemm_split_kernel.h"
#include "gemm_split_wide_kernel.h"
#include "screen_wide_kernel.h"
#include "screen_xwide_kernel.h"

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

// Measurement hook (bench.py): the NEXT matrix-product launch made by the calling host thread records e0 / e1 around its ONE dominant
// kernel on the launch stream — not around the slab reduction, the pool's candidate passes or anything else the entry point enqueues —
// and stores what that kernel was in *kind (host memory):
//   0 six-product bf16 split kernel, 1 fp32-input MFMA kernel, 2 streaming kernel of a degenerate shape (no MFMA),
//   3 screening pass, one fp16 product, 4 screening pass, three bf16 products, 5 two fp16 planes / three products (vnp_gemm_ex).
// Thread-local and consumed by that launch; a launch without a pending request costs one branch.
struct VnpProfReq { hipEvent_t e0, e1; int* kind; };
static thread_local VnpProfReq g_vnp_prof = {nullptr, nullptr, nullptr};

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

// Every matrix-product entry point holds one of these: a request that the call did not consume (argument error, a path without a
// bracketed kernel) is dropped when the call returns instead of arming some later, unrelated launch with stale event handles.
struct VnpProfScope { ~VnpProfScope() { g_vnp_prof = {nullptr, nullptr, nullptr}; } };

static inline void vnp_prof_begin(hipStream_t st) { if (g_vnp_prof.e0) (void)hipEventRecord(g_vnp_prof.e0, st); }
static inline void vnp_prof_end(hipStream_t st, int kind)
{
    if (!g_vnp_prof.e0) return;
    (void)hipEventRecord(g_vnp_prof.e1, st);
    if (g_vnp_prof.kind) *g_vnp_prof.kind = kind;
    g_vnp_prof = {nullptr, nullptr, nullptr};
}

static int vnp_num_cus()
{
    static std::atomic<int> cached{0};             // every GPU of a node is the same part: one value serves all devices
    int n = cached.load(std::memory_order_relaxed);
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cached.store(n, std::memory_order_relaxed);
    }
    return n;
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
    const int ox = threadIdx.x & 63, q = threadIdx.x >> 6;
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
        const int r0 = blockIdx.y * 64 + (threadIdx.x >> 6), r1 = min(M, (int)(blockIdx.y + 1) * 64);
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
    const int r0 = blockIdx.y * 64 + (threadIdx.x >> 6), r1 = min(M, (int)(blockIdx.y + 1) * 64);
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
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
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
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
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
            if (v4) hipLaunchKernelGGL(vnp_gemm_rank1_kernel<true>, dim3(vnp_cdiv(N, 256), vnp_cdiv(M, 64)), dim3(256), 0, stream, A, lda, B, bstr,
                                       C, ldc, M, N, bias, ldbias, rows_per_group > 0 ? rows_per_group : 1, accumulate);
            else hipLaunchKernelGGL(vnp_gemm_rank1_kernel<false>, dim3(vnp_cdiv(N, 64), vnp_cdiv(M, 64)), dim3(256), 0, stream, A, lda, B, bstr,
                                    C, ldc, M, N, bias, ldbias, rows_per_group > 0 ? rows_per_group : 1, accumulate);
            vnp_prof_end(stream, 2);
            return vnp_last_error();
        }
        // matrix-vector: A [M,K] K-contiguous, one output column
        if (N == 1 && op_a == 0 && !bias) {
            vnp_prof_begin(stream);
            hipLaunchKernelGGL(vnp_gemm_matvec_kernel, dim3(vnp_cdiv(M, 4)), dim3(256), 0, stream, A, lda, B, (long long)(op_b == 0 ? 1 : ldb),
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
            hipLaunchKernelGGL(vnp_gemm_wcolsum_kernel, dim3(vnp_cdiv(cols, 64), ns), dim3(256), 0, stream, X, ldx, wv, ws_, K, cols, rps, slab);
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
        if (h2) hipLaunchKernelGGL((vnp_gemm_split_kernel<BM_, 128, AKC_, BKC_, true, 0, 0, 1>), grid, block, 0, stream, p); \
        else if (split) {                                                                                               \
            if (fast) hipLaunchKernelGGL((vnp_gemm_split_kernel<BM_, 128, AKC_, BKC_, true>), grid, block, 0, stream, p);  \
            else      hipLaunchKernelGGL((vnp_gemm_split_kernel<BM_, 128, AKC_, BKC_, false>), grid, block, 0, stream, p); \
        } else {                                                                                                        \
            if (fast) hipLaunchKernelGGL((vnp_gemm_kernel<BM_, 128, BK, AKC_, BKC_, true>), grid, block, 0, stream, p);    \
            else      hipLaunchKernelGGL((vnp_gemm_kernel<BM_, 128, BK, AKC_, BKC_, false>), grid, block, 0, stream, p);   \
        }                                                                                                               \
    } while (0)
    vnp_prof_begin(stream);
    if (wide) {
        // both operands K-contiguous, N % 256 == 0: the 512-thread kernel with the 192 x 256 tile — the same arithmetic bit for bit,
        // 0.7x the staging work per MFMA (scripts/experiments/gemm_wide_bench.hip)
        const long long items_w = (long long)(M / 192) * (N / 256) * nsplit * nbatch;
        hipLaunchKernelGGL((vnp_gemm_split_wide_kernel<0>), dim3((unsigned)(items_w < vnp_num_cus() ? items_w : vnp_num_cus())), dim3(512), 0, stream, p);
    } else if (tileable && fast && bm == 96) {
        if (op_b == 0) hipLaunchKernelGGL((vnp_gemm_split_kernel<96, 128, true, true, true>), grid, block, 0, stream, p);
        else           hipLaunchKernelGGL((vnp_gemm_split_kernel<96, 128, true, false, true>), grid, block, 0, stream, p);
    } else if (tileable && fast && bm == 128) {
        if (op_b == 0) hipLaunchKernelGGL((vnp_gemm_split_kernel<128, 128, true, true, true>), grid, block, 0, stream, p);
        else           hipLaunchKernelGGL((vnp_gemm_split_kernel<128, 128, true, false, true>), grid, block, 0, stream, p);
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
        hipLaunchKernelGGL((vnp_gemm_split_wide_tn_kernel<0>), dim3((unsigned)(items_w < vnp_num_cus() ? items_w : vnp_num_cus())), dim3(512), 0, stream, p);
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

// ---- VNLinear + VNMaxPool in one pass (models/pcn.py:173-174: second_conv[1] -> maxpool2) -------------------------------
// Stage 2 of the fused pool: reduce the per-32-point partials of a group (ascending order keeps the first maximum).
__global__ void vnp_pool_fused_final_kernel(const float* __restrict__ pval, const int* __restrict__ pidx, const float* __restrict__ px,
                                            int nparts, int C, int* __restrict__ idx, float* __restrict__ out, int ldo,
                                            const int* __restrict__ pred)
{
    if (pred && *pred == 0) return;                       // predicated launch (see VnpGemmArgs::pred)
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const long long g = blockIdx.y;
    size_t slot = (size_t)g * nparts * C + c;
    float best = pval[slot];
    size_t bs = slot;
    for (int k = 1; k < nparts; ++k) {
        slot += C;
        const float v = pval[slot];
        if (v > best) { best = v; bs = slot; }
    }
    idx[(size_t)g * C + c] = pidx[bs];
#pragma unroll
    for (int t = 0; t < 3; ++t) out[((size_t)g * 3 + t) * ldo + c] = px[3 * bs + t];
}

extern "C" long long vnp_linear_pool_ws_bytes(long long G, int gs, int C)
{
    if (G <= 0 || gs <= 0 || C <= 0 || gs % 32) return 0;
    return G * (gs / 32) * (long long)C * 20;
}

// out[g*3+t][c] = (W f)[(g*gs + n*)*3 + t][c] with n* = argmax_n <W f, W' f>(n, c) over the gs points of group g, idx[g][c] = n*.
// x [G*gs*3, K] rows layout; Wcat [2C, K]: per 32 channels, 32 rows of W then 32 rows of W' (W' = Wd W for VNMaxPool's
// direction map); ws: vnp_linear_pool_ws_bytes(G, gs, C).  Requirements (else VNP_EINVAL, callers fall back to
// vnp_gemm + vnp_vn_pool_fwd): split GEMM mode, (G*gs*3) % 192 == 0, gs % 32 == 0, C % 64 == 0, K % 32 == 0, 16-byte aligned.
extern "C" int vnp_linear_pool_fwd(hipStream_t st, const float* x, int ldx, const float* Wcat, int ldw, long long G, int gs, int C,
                                   int K, int* idx, float* out, int ldo, void* ws)
{
    if (!x || !Wcat || !idx || !out || !ws || G <= 0 || gs <= 0 || C <= 0 || K <= 0) return VNP_EINVAL;
    const long long M = G * gs * 3;
    if (vnp_gemm_get_mode() != 0 || M % 192 || gs % 32 || C % 64 || K % 32 || ldx % 4 || ldw % 4 || M > 2147483647LL ||
        (uintptr_t)x % 16 || (uintptr_t)Wcat % 16 || G > 65535)
        return VNP_EINVAL;
    const long long nslot = G * (gs / 32) * (long long)C;
    VnpGemmArgs p{};
    p.A = x; p.B = Wcat; p.C = nullptr; p.bias = nullptr;
    p.M = (int)M; p.N = 2 * C; p.K = K;
    p.lda = ldx; p.ldb = ldw; p.ldc = 0; p.rows_per_group = 1;
    p.tiles_per_split = K / 32; p.nsplit = 1; p.nbatch = 1;
    p.pool_val = (float*)ws;
    p.pool_idx = (int*)((char*)ws + nslot * 4);
    p.pool_x = (float*)((char*)ws + nslot * 8);
    p.pool_gs = gs;
    p.col_group = 8;
    const long long items = (M / 192) * (2LL * C / 128);
    if (items > 2147483647LL) return VNP_EINVAL;
    const long long cap = (long long)vnp_num_cus() * 2;
    dim3 grid((unsigned)(items < cap ? items : cap)), block(256);
    vnp_prof_begin(st);
    hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 1>), grid, block, 0, st, p);
    vnp_prof_end(st, 0);
    hipLaunchKernelGGL(vnp_pool_fused_final_kernel, dim3(vnp_cdiv(C, 256), (unsigned)G), dim3(256), 0, st,
                       (const float*)p.pool_val, (const int*)p.pool_idx, (const float*)p.pool_x, gs / 32, C, idx, out, ldo,
                       (const int*)nullptr);
    return vnp_last_error();
}

// VNLinear -> VNMaxPool where the linear output has ANOTHER consumer too (encoder first_conv[1] -> maxpool1 and second_conv[0]): as
// vnp_linear_pool_fwd, and y = x W^T [G*gs*3, C] is stored as well (EPI = 5); the direction d = W' x still never exists in memory and
// the separate direction GEMM + pool pass over y and d are gone.  ws: vnp_linear_pool_ws_bytes(G, gs, C).
extern "C" int vnp_linear_pool_store_fwd(hipStream_t st, const float* x, int ldx, const float* Wcat, int ldw, long long G, int gs, int C,
                                         int K, float* y, int ldy, int* idx, float* out, int ldo, void* ws)
{
    if (!x || !Wcat || !y || !idx || !out || !ws || G <= 0 || gs <= 0 || C <= 0 || K <= 0) return VNP_EINVAL;
    const long long M = G * gs * 3;
    if (vnp_gemm_get_mode() != 0 || M % 192 || gs % 32 || C % 64 || K % 32 || ldx % 4 || ldw % 4 || ldy < C || M > 2147483647LL ||
        (uintptr_t)x % 16 || (uintptr_t)Wcat % 16 || G > 65535)
        return VNP_EINVAL;
    const long long nslot = G * (gs / 32) * (long long)C;
    VnpGemmArgs p{};
    p.A = x; p.B = Wcat; p.C = y; p.bias = nullptr;
    p.M = (int)M; p.N = 2 * C; p.K = K;
    p.lda = ldx; p.ldb = ldw; p.ldc = ldy; p.rows_per_group = 1;
    p.tiles_per_split = K / 32; p.nsplit = 1; p.nbatch = 1;
    p.pool_val = (float*)ws;
    p.pool_idx = (int*)((char*)ws + nslot * 4);
    p.pool_x = (float*)((char*)ws + nslot * 8);
    p.pool_gs = gs;
    p.col_group = 8;
    const long long items = (M / 192) * (2LL * C / 128);
    if (items > 2147483647LL) return VNP_EINVAL;
    const long long cap = (long long)vnp_num_cus() * 2;
    dim3 grid((unsigned)(items < cap ? items : cap)), block(256);
    vnp_prof_begin(st);
    hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 5>), grid, block, 0, st, p);
    vnp_prof_end(st, 0);
    hipLaunchKernelGGL(vnp_pool_fused_final_kernel, dim3(vnp_cdiv(C, 256), (unsigned)G), dim3(256), 0, st,
                       (const float*)p.pool_val, (const int*)p.pool_idx, (const float*)p.pool_x, gs / 32, C, idx, out, ldo,
                       (const int*)nullptr);
    return vnp_last_error();
}

// ---- the same pool with an exact SCREENED arg-max ---------------------------------------------------------------------------
// The 6-product GEMM above spends 1.65 TFLOP (config 2) to pick ONE point per (sample, channel).  Here a 3-product pass (half
// the MFMA work, no l plane) brackets every score s6(n, c) = <x6, d6> the 6-product kernels would produce between rigorous
// bounds (EPI = 3 epilogue); a point can only be the arg-max if its upper bound reaches the largest lower bound of its group,
// which leaves ~1-2 candidates per (sample, channel); the candidates — (point, channel) PAIRS, not tiles — are then evaluated
// exactly: 16 pairs per wave as the DIAGONAL of a 16x16 MFMA tile whose A rows are the pairs' feature rows and whose B rows are
// the pairs' weight rows, with the 6-product sequence and K order of vnp_gemm_split_kernel, so every x6 / d6 / score has the bits
// the full kernels produce (an MFMA output element depends only on its own A row, B row and accumulator).  The arg-max over the
// candidates (ties to the lowest point) is therefore the arg-max of the full computation: indices and pooled vectors are
// bit-identical to vnp_linear_pool_fwd and to vnp_gemm + vnp_vn_pool_fwd (tests/test_gpu_layers.py).  If the candidate list
// overflows or any bound is not finite (NaN / Inf inputs), a device-side flag is raised and the full kernel — enqueued
// unconditionally behind it as a predicated launch — recomputes everything; without the flag it exits at once (~10 us).

// upper bound of |row|_2 for rows of K floats (one wave per row); inflated past its own fp32 rounding
// h16 (optional): the row rounded to fp16 (round to nearest even; |x| > 65504 -> Inf), [rows][K] halves — the operand image the
// one-product screening pass reads
__global__ void __launch_bounds__(256) vnp_row_norm_kernel(const float* __restrict__ x, int ld, long long rows, int K, float* __restrict__ out,
                                                          unsigned short* __restrict__ h16)
{
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* src = x + (size_t)r * ld;
    float s = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const vnp_f4 v = __builtin_nontemporal_load((const vnp_f4*)(src + k));
        s = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], s))));
        if (h16) *(vnp_u2*)(h16 + (size_t)r * K + k) = vnp_u2{vnp_pack_f16(v[0], v[1]), vnp_pack_f16(v[2], v[3])};
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[r] = sqrtf(s) * (1.0f + 2.0f * (float)K * 5.97e-8f) * 1.0001f;
}

struct VnpScreenHead { int count; int flag; int cap; int pad; };     // head of the workspace (device memory)
#define VNP_POOL_SCREEN_FP16 1
#define VNP_POOL_SCREEN_NO_XWIDE 2
#define VNP_POOL_SCREEN_NARROW 4

__global__ void vnp_pool_screen_init_kernel(VnpScreenHead* head, int cap) { head->count = 0; head->flag = 0; head->cap = cap; head->pad = 0; }

// per (group, channel): L = largest lower bound; every block whose largest upper bound reaches L contributes its top point, its
// second point when that one's upper bound reaches L too — or all 32 of its points when even the third-largest does.  A workgroup = 64 channels x 4 quarters of the
// group's 32-point blocks; each thread keeps its quarter's slots in registers (one pass over memory), the four quarters meet in
// LDS, and the workgroup appends ALL its pairs with one atomicAdd, so the candidates of neighbouring channels are contiguous in
// the list (coalesced reads in the exact and final passes).  grid (C/64, G), 256 threads; nparts <= 4 * SEL_MAXQ.
#define SEL_MAXQ 16
__global__ void __launch_bounds__(256) vnp_pool_screen_select_kernel(const float4* __restrict__ slots, const int* __restrict__ slot_idx,
                                                                     int nparts, int C, int gs,
                                                                     VnpScreenHead* __restrict__ head, int2* __restrict__ seg, int2* __restrict__ list)
{
    __shared__ float sL[4][64];
    __shared__ int sbad[4][64], scnt[4][64], sbase;
    const int cx = threadIdx.x & 63, qy = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const long long g = blockIdx.y;
    const int per = (nparts + 3) / 4, k0 = qy * per, k1 = min(nparts, k0 + per);
    const float4* sl = slots + (size_t)g * nparts * C + c;
    float4 v[SEL_MAXQ];
    float L = -INFINITY;
    int bad = 0;
#pragma unroll
    for (int j = 0; j < SEL_MAXQ; ++j) {
        if (k0 + j < k1) {
            v[j] = sl[(size_t)(k0 + j) * C];
            L = fmaxf(L, v[j].x);
            bad |= !(v[j].x < INFINITY) || !(v[j].y < INFINITY) || !(v[j].z < INFINITY) || !(v[j].w < INFINITY);
        }
    }
    sL[qy][cx] = L; sbad[qy][cx] = bad;
    __syncthreads();
    L = fmaxf(fmaxf(sL[0][cx], sL[1][cx]), fmaxf(sL[2][cx], sL[3][cx]));
    bad = sbad[0][cx] | sbad[1][cx] | sbad[2][cx] | sbad[3][cx];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < SEL_MAXQ; ++j)
        if (k0 + j < k1 && v[j].y >= L) cnt += (v[j].w >= L) ? 32 : (v[j].z >= L) ? 2 : 1;
    scnt[qy][cx] = cnt;
    __syncthreads();
    // exclusive prefix over (channel, quarter) in channel-major order: thread (cx, qy) starts after all pairs of channels < cx and
    // of quarters < qy of its own channel
    int before = 0, total = 0, mine_total = 0;
    for (int cc = 0; cc < 64; ++cc) {
        const int t = scnt[0][cc] + scnt[1][cc] + scnt[2][cc] + scnt[3][cc];
        if (cc < cx) before += t;
        if (cc == cx) mine_total = t;
        total += t;
    }
    for (int q = 0; q < qy; ++q) before += scnt[q][cx];
    int anybad = bad;
    anybad = __syncthreads_or(anybad || mine_total <= 0);
    if (threadIdx.x == 0) {
        int base = -1;
        if (anybad) atomicOr(&head->flag, 1);
        else {
            base = atomicAdd(&head->count, total);
            if (base + total > head->cap) { atomicOr(&head->flag, 1); base = -1; }
        }
        sbase = base;
    }
    __syncthreads();
    const int base = sbase;
    if (qy == 0) seg[(size_t)g * C + c] = base < 0 ? make_int2(0, 0) : make_int2(base + before, mine_total);
    if (base < 0) return;
    int w = base + before;
    const int pbase = (int)(g * gs);
#pragma unroll
    for (int j = 0; j < SEL_MAXQ; ++j) {
        if (k0 + j < k1 && v[j].y >= L) {
            if (v[j].w >= L) { for (int i = 0; i < 32; ++i) list[w++] = make_int2(pbase + (k0 + j) * 32 + i, c); }
            else {
                const int packed = slot_idx[((size_t)g * nparts + k0 + j) * C + c];
                list[w++] = make_int2(pbase + (packed & 0xffff), c);
                if (v[j].z >= L) list[w++] = make_int2(pbase + (packed >> 16), c);
            }
        }
    }
}

// ---- exact evaluation of the candidate pairs ---------------------------------------------------------------------------------------
// The pairs are counting-sorted by BIN = 5 consecutive points (hist -> scan -> scatter).  One wave evaluates one TILE = up to 16 pairs of
// one bin: the A operand of its MFMAs holds the bin's 15 feature rows (5 points x 3 components, row 3 s + t), the B operands the W and W'
// rows of the tile's 16 pairs, so  C[3 s + t][j] = <f[(5 bin + s, t)], W_(c_j)>  and pair j reads its x6 / d6 from the three rows of its
// own point — 48 of the 256 tile entries are useful instead of the 16 of a pure diagonal arrangement, and a bin's feature rows are
// converted once per tile instead of once per pair.  Product sequence and K order are those of vnp_gemm_split_kernel (K-tiles ascending;
// per K-tile (l h)(h l)(m m)(m h)(h m)(h h); A = features, B = weights), so x6, d6 and the score (the pool kernels' fmaf form) have the bits
// the full kernels produce: an MFMA output element depends only on its own A row, B row and accumulator.
#define SCR_BIN 5

__global__ void vnp_pool_screen_hist_kernel(const VnpScreenHead* __restrict__ head, const int2* __restrict__ list, int* __restrict__ hist)
{
    const int count = head->flag ? 0 : head->count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) atomicAdd(&hist[list[i].x / SCR_BIN], 1);
}

// hist[n] (counts) -> start[n + 1] (exclusive prefix of the counts), cursor[n] (a copy for the scatter pass), tstart[n + 1] (exclusive
// prefix of the tiles ceil(count / 16)); one workgroup of 1024 threads, contiguous chunks + an LDS scan of the chunk totals
__global__ void __launch_bounds__(1024) vnp_pool_screen_scan_kernel(const int* __restrict__ hist, int n, int* __restrict__ start,
                                                                    int* __restrict__ cursor, int* __restrict__ tstart)
{
    __shared__ int tot[1024], ttot[1024];
    const int t = threadIdx.x, per = (n + 1023) / 1024, lo = min(n, t * per), hi = min(n, lo + per);
    int s = 0, ts = 0;
    for (int i = lo; i < hi; ++i) { const int v = hist[i]; s += v; ts += (v + 15) >> 4; }
    tot[t] = s; ttot[t] = ts;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? tot[t - o] : 0, tv = t >= o ? ttot[t - o] : 0;
        __syncthreads();
        tot[t] += v; ttot[t] += tv;
        __syncthreads();
    }
    int run = tot[t] - s, trun = ttot[t] - ts;
    for (int i = lo; i < hi; ++i) {
        const int v = hist[i];
        start[i] = run; cursor[i] = run; tstart[i] = trun;
        run += v; trun += (v + 15) >> 4;
    }
    if (t == 1023) { start[n] = tot[1023]; tstart[n] = ttot[1023]; }
}

__global__ void vnp_pool_screen_scatter_kernel(const VnpScreenHead* __restrict__ head, const int2* __restrict__ list, int* __restrict__ cursor,
                                               int* __restrict__ order)
{
    const int count = head->flag ? 0 : head->count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) order[atomicAdd(&cursor[list[i].x / SCR_BIN], 1)] = i;
}

__global__ void __launch_bounds__(256) vnp_pool_screen_exact_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ Wcat, int ldw,
                                                                    int K, long long npts, const VnpScreenHead* __restrict__ head,
                                                                    const int2* __restrict__ list, const int* __restrict__ order,
                                                                    const int* __restrict__ start, const int* __restrict__ tstart, int nbins,
                                                                    float4* __restrict__ res)
{
    if (head->flag) return;
    const int lane = threadIdx.x & 63, l16 = lane & 15, lq = lane >> 4;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= tstart[nbins]) return;
    // the bin of this tile: largest b with tstart[b] <= tile (wave-uniform binary search)
    int lo = 0, hi = nbins;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tstart[mid] <= tile) lo = mid; else hi = mid; }
    const int bin = lo;
    const int first = start[bin] + 16 * (tile - tstart[bin]);
    const int npair = min(16, start[bin + 1] - first);
    const int mine = order[first + min(l16, npair - 1)];              // lanes past the tile's last pair repeat it
    const int2 pc = list[mine];
    // A row of this lane: component r % 3 of point 5 bin + r / 3 (r = lane & 15; row 15 is unused and repeats row 14)
    const int r = min(l16, 14);
    long long pa = (long long)bin * SCR_BIN + r / 3;
    if (pa >= npts) pa = npts - 1;
    const float* f0 = x + ((size_t)pa * 3 + r % 3) * ldx + lq * 8;
    const float* w0 = Wcat + (size_t)((pc.y >> 5) * 64 + (pc.y & 31)) * ldw + lq * 8;
    const float* w1 = w0 + (size_t)32 * ldw;
    vnp_f4 ax = {0.f, 0.f, 0.f, 0.f}, ad = {0.f, 0.f, 0.f, 0.f};
    auto planes = [](const vnp_f4 v0, const vnp_f4 v1, vnp_bf8 (&pl)[3]) {
        unsigned h[4], m[4], l[4];
        vnp_split_pair(v0[0], v0[1], h[0], m[0], l[0]);
        vnp_split_pair(v0[2], v0[3], h[1], m[1], l[1]);
        vnp_split_pair(v1[0], v1[1], h[2], m[2], l[2]);
        vnp_split_pair(v1[2], v1[3], h[3], m[3], l[3]);
        pl[0] = __builtin_bit_cast(vnp_bf8, vnp_u4{h[0], h[1], h[2], h[3]});
        pl[1] = __builtin_bit_cast(vnp_bf8, vnp_u4{m[0], m[1], m[2], m[3]});
        pl[2] = __builtin_bit_cast(vnp_bf8, vnp_u4{l[0], l[1], l[2], l[3]});
    };
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    const float* src[3] = {f0, w0, w1};
    vnp_f4 cur[3][2], nxt[3][2];
#pragma unroll
    for (int o = 0; o < 3; ++o) { cur[o][0] = *(const vnp_f4*)src[o]; cur[o][1] = *(const vnp_f4*)(src[o] + 4); }
    for (int k0 = 0; k0 < K; k0 += 32) {
        if (k0 + 32 < K) {                          // next K-step's operands in flight during this step's MFMAs
#pragma unroll
            for (int o = 0; o < 3; ++o) { nxt[o][0] = *(const vnp_f4*)(src[o] + k0 + 32); nxt[o][1] = *(const vnp_f4*)(src[o] + k0 + 36); }
        }
        vnp_bf8 a[3], bw[3], bd[3];
        planes(cur[0][0], cur[0][1], a);
        planes(cur[1][0], cur[1][1], bw);
        planes(cur[2][0], cur[2][1], bd);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            ax = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q]], bw[PB[q]], ax, 0, 0, 0);
            ad = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q]], bd[PB[q]], ad, 0, 0, 0);
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) { cur[o][0] = nxt[o][0]; cur[o][1] = nxt[o][1]; }
    }
    // C layout: lane (lq, col j = lane & 15) holds rows 4 lq + e of column j.  Pair j needs rows 3 s + t (s = its point's slot in the bin):
    // row R sits in lane (R >> 2) * 16 + j, register R & 3 — fetched with one shuffle per register and selected.
    const int sl = pc.x - bin * SCR_BIN;                                          // 0 .. 4
    float xv[3], dv[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int R = 3 * sl + t, srcl = (R >> 2) * 16 + l16, e = R & 3;
        const float x0 = __shfl(ax[0], srcl), x1 = __shfl(ax[1], srcl), x2 = __shfl(ax[2], srcl), x3 = __shfl(ax[3], srcl);
        const float d0 = __shfl(ad[0], srcl), d1 = __shfl(ad[1], srcl), d2 = __shfl(ad[2], srcl), d3 = __shfl(ad[3], srcl);
        xv[t] = e == 0 ? x0 : e == 1 ? x1 : e == 2 ? x2 : x3;
        dv[t] = e == 0 ? d0 : e == 1 ? d1 : e == 2 ? d2 : d3;
    }
    if (lq == 0 && l16 < npair) {
        float4 o;
        o.x = xv[0]; o.y = xv[1]; o.z = xv[2];
        o.w = fmaf(xv[2], dv[2], fmaf(xv[1], dv[1], xv[0] * dv[0]));      // the pool kernels' score form
        res[mine] = o;
    }
}

// per (group, channel): arg-max over its candidates' exact scores (strictly greater wins, ties to the lowest point: the first maximum
// of an ascending scan, as vnp_vn_pool_fwd), index and pooled vector out.  Eight lanes share a pair (a pair has 1-3 candidates as a rule
// but a whole 32-point block now and then: one thread per pair made the block wait for its longest loop).  Skipped when the flag is up
// (the fallback writes the outputs).  grid (ceil(C/32), G), 256 threads = 32 channels x 8 lanes
__global__ void __launch_bounds__(256) vnp_pool_screen_final_kernel(const VnpScreenHead* __restrict__ head, const int2* __restrict__ seg,
                                                                    const int2* __restrict__ list, const float4* __restrict__ res, int C, int gs,
                                                                    int* __restrict__ idx, float* __restrict__ out, int ldo)
{
    if (head->flag) return;
    const int sub = threadIdx.x & 7;
    const int c = blockIdx.x * 32 + (threadIdx.x >> 3);
    const long long g = blockIdx.y;
    const bool live = c < C;
    const int2 sg = live ? seg[(size_t)g * C + c] : make_int2(0, 0);
    float best = 0.f; int bp = 0x7fffffff; float bx = 0.f, by = 0.f, bz = 0.f;
    for (int i = sg.x + sub; i < sg.x + sg.y; i += 8) {
        const float4 r = res[i];
        const int pt = list[i].x;
        if (bp == 0x7fffffff || r.w > best || (r.w == best && pt < bp)) { best = r.w; bp = pt; bx = r.x; by = r.y; bz = r.z; }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const float ob = __shfl_xor(best, o), ox = __shfl_xor(bx, o), oy = __shfl_xor(by, o), oz = __shfl_xor(bz, o);
        const int op = __shfl_xor(bp, o);
        if (op != 0x7fffffff && (bp == 0x7fffffff || ob > best || (ob == best && op < bp))) { best = ob; bp = op; bx = ox; by = oy; bz = oz; }
    }
    if (live && sub == 0) {
        idx[(size_t)g * C + c] = bp - (int)(g * gs);
        out[((size_t)g * 3 + 0) * ldo + c] = bx;
        out[((size_t)g * 3 + 1) * ldo + c] = by;
        out[((size_t)g * 3 + 2) * ldo + c] = bz;
    }
}

static inline long long vnp_align256(long long v) { return (v + 255) / 256 * 256; }

// candidate capacity: 32 per (group, channel) on average (measured means: 1.7 / 3.6 (bf16x3 / fp16 pass) on uniform random clouds with
// random-init weights, 8 / 21 on surface-like clouds, where neighbouring points have nearly equal scores)
static inline long long vnp_screen_cap(long long G, int C) { long long c = 32 * G * C; return c > 2000000000LL ? 2000000000LL : c; }

extern "C" long long vnp_linear_pool_screened_ws_bytes(long long G, int gs, int C, int K)
{
    if (G <= 0 || gs <= 0 || C <= 0 || K <= 0 || gs % 32) return 0;
    const long long nslot = G * (gs / 32) * (long long)C, cap = vnp_screen_cap(G, C);
    return 256 + vnp_align256(G * gs * 3 * (long long)K * 2) + vnp_align256(2LL * C * K * 2) + vnp_align256(G * gs * 3 * 4) + vnp_align256(2LL * C * 4) + vnp_align256(nslot * 16) + vnp_align256(nslot * 4) + vnp_align256(G * C * 8) +
           vnp_align256(cap * 8) + vnp_align256(cap * 16) + 4 * vnp_align256((G * gs / SCR_BIN + 2) * 4) + vnp_align256(cap * 4) +
           vnp_align256(vnp_linear_pool_ws_bytes(G, gs, C));
}

// Same contract, arguments and results (bit for bit) as vnp_linear_pool_fwd; ws: vnp_linear_pool_screened_ws_bytes(G, gs, C, K) bytes.
// After the call ws begins with {int candidates, int fallback_flag, int capacity} (diagnostics; device memory).
extern "C" int vnp_linear_pool_screened_fwd(hipStream_t st, const float* x, int ldx, const float* Wcat, int ldw, long long G, int gs, int C,
                                            int K, int* idx, float* out, int ldo, void* ws, int flags)
{
    if (!x || !Wcat || !idx || !out || !ws || G <= 0 || gs <= 0 || C <= 0 || K <= 0 || (flags & ~7)) return VNP_EINVAL;
    const long long M = G * gs * 3;
    if (vnp_gemm_get_mode() != 0 || M % 192 || gs % 32 || C % 64 || K % 32 || ldx % 4 || ldw % 4 || M > 2147483647LL ||
        (uintptr_t)x % 16 || (uintptr_t)Wcat % 16 || G > 65535 || G * gs > 2147483647LL || gs / 32 > 4 * SEL_MAXQ)
        return VNP_EINVAL;
    const long long nslot = G * (gs / 32) * (long long)C, cap = vnp_screen_cap(G, C);
    char* w = (char*)ws;
    VnpScreenHead* head = (VnpScreenHead*)w;       w += 256;
    unsigned short* x16 = (unsigned short*)w;      w += vnp_align256(M * (long long)K * 2);          // fp16 images of x and Wcat
    unsigned short* w16 = (unsigned short*)w;      w += vnp_align256(2LL * C * K * 2);
    float* rnorm = (float*)w;                      w += vnp_align256(M * 4);
    float* cnorm = (float*)w;                      w += vnp_align256(2LL * C * 4);
    float4* slots = (float4*)w;                    w += vnp_align256(nslot * 16);
    int* slot_idx = (int*)w;                       w += vnp_align256(nslot * 4);
    int2* seg = (int2*)w;                          w += vnp_align256(G * C * 8);
    int2* list = (int2*)w;                         w += vnp_align256(cap * 8);
    float4* res = (float4*)w;                      w += vnp_align256(cap * 16);
    const int nbins = (int)((G * gs + SCR_BIN - 1) / SCR_BIN);
    const long long binb = vnp_align256((G * gs / SCR_BIN + 2) * 4);
    int* hist = (int*)w;                           w += binb;
    int* start = (int*)w;                          w += binb;
    int* cursor = (int*)w;                         w += binb;
    int* tstart = (int*)w;                         w += binb;
    int* order = (int*)w;                          w += vnp_align256(cap * 4);
    void* ws_full = (void*)w;
    hipLaunchKernelGGL(vnp_pool_screen_init_kernel, dim3(1), dim3(1), 0, st, head, (int)cap);
    const bool f16pass = (flags & VNP_POOL_SCREEN_FP16) != 0 && K % 64 == 0;          // the fp16 pass walks K in steps of 64
    hipLaunchKernelGGL(vnp_row_norm_kernel, dim3((unsigned)vnp_cdiv(M, 4)), dim3(256), 0, st, x, ldx, M, K, rnorm, f16pass ? x16 : nullptr);
    hipLaunchKernelGGL(vnp_row_norm_kernel, dim3((unsigned)vnp_cdiv(2LL * C, 4)), dim3(256), 0, st, Wcat, ldw, 2LL * C, K, cnorm,
                       f16pass ? w16 : nullptr);
    VnpGemmArgs p{};
    p.A = x; p.B = Wcat; p.C = nullptr; p.bias = nullptr;
    p.M = (int)M; p.N = 2 * C; p.K = K;
    p.lda = ldx; p.ldb = ldw; p.ldc = 0; p.rows_per_group = 1;
    p.tiles_per_split = K / 32; p.nsplit = 1; p.nbatch = 1;
    p.pool_val = (float*)slots; p.pool_idx = slot_idx; p.pool_gs = gs; p.col_group = 8;
    p.row_norm = rnorm; p.col_norm = cnorm;
    const long long items = (M / 192) * (2LL * C / 128);
    if (items > 2147483647LL) return VNP_EINVAL;
    const long long capwg = (long long)vnp_num_cus() * 2;
    dim3 grid((unsigned)(items < capwg ? items : capwg)), block(256);
    vnp_prof_begin(st);
    if (f16pass) {
        // ONE fp16 product per operand pair (11 significant bits against bf16's 8): |w f - w~ f~| <= (2^-10 + 2^-22) |w||f| +
        // 2^-25 (|w| + |f|) (the second term: operands under fp16's normal range; an operand over 65504 becomes Inf and raises the
        // fallback flag through a non-finite bound), plus the worst-case fp32 rounding of both accumulations (K and 6K terms)
        p.screen_eps = 9.78e-4f + 7.2f * (float)K * 5.9605e-8f;
        p.screen_abs = 3.0e-8f * sqrtf((float)K) * 1.01f;
        p.A = (const float*)x16; p.B = (const float*)w16; p.lda = K; p.ldb = K;          // the pass reads the fp16 images (halves)
        const bool narrow = (flags & VNP_POOL_SCREEN_NARROW) != 0;
        if ((2 * C) % 256 == 0 && M % 384 == 0 && !narrow && !(flags & VNP_POOL_SCREEN_NO_XWIDE)) {
            // 384 x 256 tiles, operands by LDS-DMA straight into the two 80 KB stages (screen_xwide_kernel.h): 0.71x the bytes into the
            // CU and 0.7x the fragment reads per MFMA of the 192 x 256 pass, no staging registers, no ds_write pass; same slots.
            // maxpool2 (196608 x 4096 x 1024): 1.85-1.95 -> 1.56-1.61 ms (profiles/r04_screen_ablation.txt)
            const long long iw = (M / 384) * (2LL * C / 256);
            const long long cw = vnp_num_cus();
            p.col_group = 16;                       // measured 1 / 2 / 4 / 16 column tiles per XCD run: 1.68 / 1.61 / 1.61 / 1.57 ms
            constexpr int xwide_lds = 2 * (2 * 384 * 64 + 2 * 256 * 64);          // two stages of a 384 x 256 x 64 fp16 tile: all 160 KB
            static std::atomic<unsigned long long> xwide_ok{0};
            hipError_t ex = vnp_allow_dynamic_lds(xwide_ok, (const void*)vnp_screen_f16_xwide_kernel<0>, xwide_lds);
            if (ex != hipSuccess) return (int)ex;
            hipLaunchKernelGGL(vnp_screen_f16_xwide_kernel<0>, dim3((unsigned)(iw < cw ? iw : cw)), dim3(512), xwide_lds, st, p);
        } else if ((2 * C) % 256 == 0 && !narrow) {
            // 192 x 256 tiles on eight waves, one workgroup per CU: 0.70x the operand bytes per MFMA (screen_wide_kernel.h); same slots
            const long long iw = (M / 192) * (2LL * C / 256);
            const long long cw = vnp_num_cus();
            // tile walk of an XCD: two column tiles (2 x 512 KB of W images) stay in its L2 while the A panels stream through; with
            // eight (4 MB = the whole L2) the panels evicted them: 1.99 -> 1.89 ms (measured 1 / 2 / 4 / 8 / 16: 1.90 1.89 1.95 1.99 2.00)
            p.col_group = 2;
            constexpr int wide_lds = 2 * (2 * (192 * 64 + 128) + 2 * (256 * 64 + 128));          // two stages of a 192 x 256 x 64 fp16 tile
            static std::atomic<unsigned long long> wide_ok{0};
            hipError_t ew = vnp_allow_dynamic_lds(wide_ok, (const void*)vnp_screen_f16_wide_kernel, wide_lds);
            if (ew != hipSuccess) return (int)ew;
            hipLaunchKernelGGL(vnp_screen_f16_wide_kernel, dim3((unsigned)(iw < cw ? iw : cw)), dim3(512), wide_lds, st, p);
        } else
            hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 4>), grid, block, 0, st, p);
    } else {
        // THREE bf16 products: dropped (h l)+(l h)+(m m)+... <= 3.05 * 2^-18 of |w||f| per term, plus the worst-case fp32 rounding of
        // BOTH accumulations in any summation order (3K and 6K terms, unit roundoff 2^-24, terms <= 1.01 |w||f|)
        p.screen_eps = 1.17e-5f + 9.2f * (float)K * 5.9605e-8f;
        p.screen_abs = 0.f;
        hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 3>), grid, block, 0, st, p);
    }
    vnp_prof_end(st, f16pass ? 3 : 4);
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)nbins * 4, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(vnp_pool_screen_select_kernel, dim3(C / 64, (unsigned)G), dim3(256), 0, st,
                       (const float4*)slots, (const int*)slot_idx, gs / 32, C, gs, head, seg, list);
    hipLaunchKernelGGL(vnp_pool_screen_hist_kernel, dim3(1024), dim3(256), 0, st, (const VnpScreenHead*)head, (const int2*)list, hist);
    hipLaunchKernelGGL(vnp_pool_screen_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)hist, nbins, start, cursor, tstart);
    hipLaunchKernelGGL(vnp_pool_screen_scatter_kernel, dim3(1024), dim3(256), 0, st, (const VnpScreenHead*)head, (const int2*)list, cursor, order);
    // at most cap / 16 + nbins tiles (every bin's last tile may be partial); waves past the real tile count exit at once
    hipLaunchKernelGGL(vnp_pool_screen_exact_kernel, dim3((unsigned)vnp_cdiv(cap / 16 + nbins, 4)), dim3(256), 0, st, x, ldx, Wcat, ldw, K,
                       G * gs, (const VnpScreenHead*)head, (const int2*)list, (const int*)order, (const int*)start, (const int*)tstart, nbins, res);
    hipLaunchKernelGGL(vnp_pool_screen_final_kernel, dim3(vnp_cdiv(C, 32), (unsigned)G), dim3(256), 0, st,
                       (const VnpScreenHead*)head, (const int2*)seg, (const int2*)list, (const float4*)res, C, gs, idx, out, ldo);
    // predicated fallback: the full 6-product pool, runs only if the flag was raised
    VnpGemmArgs q{};
    q.A = x; q.B = Wcat; q.M = (int)M; q.N = 2 * C; q.K = K; q.lda = ldx; q.ldb = ldw; q.rows_per_group = 1;
    q.tiles_per_split = K / 32; q.nsplit = 1; q.nbatch = 1; q.col_group = 8; q.pool_gs = gs;
    q.pool_val = (float*)ws_full;
    q.pool_idx = (int*)((char*)ws_full + nslot * 4);
    q.pool_x = (float*)((char*)ws_full + nslot * 8);
    q.pred = &head->flag;
    hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 1>), grid, block, 0, st, q);
    hipLaunchKernelGGL(vnp_pool_fused_final_kernel, dim3(vnp_cdiv(C, 256), (unsigned)G), dim3(256), 0, st,
                       (const float*)q.pool_val, (const int*)q.pool_idx, (const float*)q.pool_x, gs / 32, C, idx, out, ldo,
                       (const int*)&head->flag);
    return vnp_last_error();
}

// ---- [p | d] = x [Wf; Wd]^T with the BatchNorm statistics of p taken in the epilogue --------------------------------
// merges 32 consecutive 32-point Welford partials into the 1024-point chunks vnp_vn_bn_finalize expects
__global__ void __launch_bounds__(256) vnp_stats_merge32_kernel(const float2* __restrict__ part32, int C, float2* __restrict__ out)
{
    __shared__ double rm[4][64], rM[4][64];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
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
    if (amax_a && K >= 64) hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 2, 1>), grid, block, 0, st, p);
    else if (N % 256 == 0 &&                                     // the wide tile where its rounds are cheaper (vnp_gemm_impl)
             (double)((items_w + vnp_num_cus() - 1) / vnp_num_cus()) * 1.38 <
             0.99 * ((double)(items / cap) * 1.45 + (items % cap == 0 ? 0.0 : 2 * (items % cap) <= cap ? 1.0 : 1.45)))
        hipLaunchKernelGGL((vnp_gemm_split_wide_kernel<2>), dim3((unsigned)(items_w < vnp_num_cus() ? items_w : vnp_num_cus())), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 2>), grid, block, 0, st, p);
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
