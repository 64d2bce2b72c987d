// Dev probe: time vnp_gemm_split_kernel on the step's three big shapes.  Built twice — as is, and with
// -DVNP_EXPERIMENT_CHEAP_SPLIT (the fp32 -> 3 x bf16 conversion replaced by ~3 bit operations per pair: wrong results, same loads,
// LDS traffic and MFMAs) — to price the in-loop conversion, i.e. the most that operands pre-split in memory could save.
//   hipcc --offload-arch=gfx950 -O3 -I vn_pointcloudcompletion_amd/csrc scripts/gemm_variants_probe.hip -o scripts/gemm_variants_probe.bin
#include "gemm_split_kernel.h"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static unsigned* g_amax = nullptr;      // [2]: bits of max |A|, max |B| (PREC >= 1)
template <int BM, bool AKC, bool BKC, int EPI, int PREC = 0>
static float run(const char* name, int M, int N, int K, int nsplit, const float* A, const float* B, float* C, float* ws)
{
    VnpGemmArgs p{};
    p.amax_a = g_amax; p.amax_b = g_amax + 1;
    p.A = A; p.B = B; p.M = M; p.N = N; p.K = K;
    p.lda = AKC ? K : M; p.ldb = BKC ? K : N; p.ldc = N; p.rows_per_group = 1; p.col_group = 8;
    p.tiles_per_split = K / 32 / nsplit; p.nsplit = nsplit; p.nbatch = 1; p.slab = (long long)M * N;
    p.C = nsplit > 1 ? ws : C;
    const long long items = (long long)(M / BM) * (N / 128) * nsplit;
    dim3 grid((unsigned)(items < 512 ? items : 512)), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((vnp_gemm_split_kernel<BM, 128, AKC, BKC, true, 0, EPI, PREC>), grid, block, 0, 0, p);
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((vnp_gemm_split_kernel<BM, 128, AKC, BKC, true, 0, EPI, PREC>), grid, block, 0, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("%-28s M=%8d N=%5d K=%8d split=%3d  %8.3f ms  %7.1f TF/s fp32-equivalent\n", name, M, N, K, nsplit, ms, 2.0 * M * N * K / ms * 1e-9);
    return ms;
}

int main()
{
    const size_t nA = (size_t)1572864 * 512, nC = (size_t)1572864 * 512;
    float *A, *B, *C, *ws;
    CK(hipMalloc(&A, nA * 4)); CK(hipMalloc(&B, nA * 4)); CK(hipMalloc(&C, nC * 4)); CK(hipMalloc(&ws, (size_t)6144 * 2048 * 16 + (size_t)128 * 512 * 256 * 4 + (size_t)16 * 2048 * 512 * 4));
    std::vector<float> h(nA);
    srand(1);
    for (auto& v : h) v = ((float)(rand() & 0xffffff) / 8388608.f - 1.f);
    CK(hipMemcpy(A, h.data(), nA * 4, hipMemcpyHostToDevice));
    for (auto& v : h) v = ((float)(rand() & 0xffffff) / 8388608.f - 1.f) * 0.05f;
    CK(hipMemcpy(B, h.data(), nA * 4, hipMemcpyHostToDevice));
    const unsigned am[2] = {0x3f800000u, 0x3d4ccccdu};            // 1.0, 0.05
    CK(hipMalloc(&g_amax, 8)); CK(hipMemcpy(g_amax, am, 8, hipMemcpyHostToDevice));
#ifdef VNP_EXPERIMENT_CHEAP_SPLIT
    printf("== conversion replaced by 3 bit-ops per pair (wrong results) ==\n");
#else
    printf("== production kernel ==\n");
#endif
#define ALL(PREC, label)                                                                                              \
    do {                                                                                                              \
        float t = 0;                                                                                                  \
        printf("-- %s --\n", label);                                                                                  \
        t += run<192, true, true, 0, PREC>("enc fwd NT [p|d]", 196608, 2048, 512, 1, A, B, C, ws);                    \
        t += run<192, true, true, 0, PREC>("dec fwd NT [p|d]", 1572864, 512, 256, 1, A, B, C, ws);                    \
        t += run<192, true, false, 0, PREC>("enc dgrad NN", 196608, 512, 2048, 1, A, B, C, ws);                       \
        t += run<192, true, false, 0, PREC>("dec dgrad NN", 1572864, 256, 512, 1, A, B, C, ws);                       \
        t += run<128, false, false, 0, PREC>("enc wgrad TN", 2048, 512, 196608, 16, A, B, C, ws);                     \
        t += run<128, false, false, 0, PREC>("dec wgrad TN", 512, 256, 1572864, 128, A, B, C, ws);                    \
        printf("sum %.3f ms\n", t);                                                                                   \
    } while (0)
    {   // the one-fp16-product screening pass of the pool (EPI = 4: fp16 operand images, bracketing epilogue): maxpool2's shape
        VnpGemmArgs p{};
        const int M = 196608, N = 4096, K = 1024;
        p.A = A; p.B = B; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.rows_per_group = 1; p.col_group = 8;       // (as halves)
        p.tiles_per_split = K / 32; p.nsplit = 1; p.nbatch = 1; p.pool_gs = 2048;
        p.pool_val = ws; p.pool_idx = (int*)C; p.row_norm = C + (1 << 24); p.col_norm = C + (1 << 25); p.screen_eps = 1e-3f; p.screen_abs = 1e-8f;
        dim3 grid(512), block(256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 4>), grid, block, 0, 0, p);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 4>), grid, block, 0, 0, p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("pool screening pass (one fp16 product) M=%d N=%d K=%d: %.3f ms\n", M, N, K, ms / 10);
    }
    ALL(0, "three bf16 planes, six products");
    ALL(1, "two fp16 planes, three products, per-K-tile partial sums added on the VALU");
    return 0;
}
