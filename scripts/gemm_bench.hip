// Tuning harness (dev tool, not part of the library): times variants of vnp_gemm_kernel in one process, interleaved.
//   hipcc --offload-arch=gfx950 -O3 -I vn_pointcloudcompletion_amd/csrc scripts/gemm_bench.hip -o gpurun_out/gemm_bench
#include "gemm_kernel.h"
#include "gemm_wave_private.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int BM, int BN, int BK, bool AKC, bool BKC, bool FAST, int PER_CU = 2>
float run(const VnpGemmArgs& p0, int iters)
{
    VnpGemmArgs p = p0;
    const int nk = (p.K + BK - 1) / BK;
    p.tiles_per_split = nk; p.nsplit = 1; p.nbatch = 1; p.bsa = p.bsb = p.bsc = 0;
    int items = vnp_cdiv(p.M, BM) * vnp_cdiv(p.N, BN);
    dim3 grid(items < PER_CU * 256 ? items : PER_CU * 256), block(256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((vnp_gemm_kernel<BM, BN, BK, AKC, BKC, FAST>), grid, block, 0, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((vnp_gemm_kernel<BM, BN, BK, AKC, BKC, FAST>), grid, block, 0, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int BK>
float run_wp(const VnpGemmArgs& p0, int iters)
{
    VnpGemmArgs p = p0;
    int items = (p.M / 192) * (p.N / 128);
    dim3 grid(items < 512 ? items : 512), block(256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((vnp_gemm_wp_kernel<BK>), grid, block, 0, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((vnp_gemm_wp_kernel<BK>), grid, block, 0, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main()
{
    struct Shape { int M, N, K; const char* name; };
    Shape shapes[] = {{196608, 2048, 2048, "pool2 dir"}, {196608, 2048, 1024, "enc 1024->2048"}, {196608, 1024, 512, "enc L3"},
                      {196608, 512, 128, "enc 128->512"}, {1572864, 256, 256, "dec 256->256"}};
    size_t maxA = 0, maxB = 0, maxC = 0;
    for (auto& s : shapes) {
        maxA = std::max(maxA, (size_t)s.M * s.K); maxB = std::max(maxB, (size_t)s.N * s.K); maxC = std::max(maxC, (size_t)s.M * s.N);
    }
    float *A, *B, *C;
    CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&B, maxB * 4)); CK(hipMalloc(&C, maxC * 4));
    std::vector<float> h(std::max(maxA, maxB));
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    CK(hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), maxB * 4, hipMemcpyHostToDevice));
    for (auto& s : shapes) {
        VnpGemmArgs p{};
        p.A = A; p.B = B; p.C = C; p.M = s.M; p.N = s.N; p.K = s.K; p.lda = s.K; p.ldb = s.K; p.ldc = s.N; p.rows_per_group = 1;
        const double fl = 2.0 * s.M * s.N * s.K;
        const int it = 5;
        printf("%-16s M=%d N=%d K=%d\n", s.name, s.M, s.N, s.K);
        for (int round = 0; round < 2; ++round) {
            float t;
            t = run<192, 128, 32, true, true, true, 2>(p, it);  printf("  r%d 192x128x32 fast  2/CU %8.3f ms %7.1f TF\n", round, t, fl / t * 1e-9);
            t = run_wp<16>(p, it); printf("  r%d wave-private BK16    %8.3f ms %7.1f TF\n", round, t, fl / t * 1e-9);
            t = run<192, 128, 32, true, true, false, 2>(p, it); printf("  r%d 192x128x32 gen   2/CU %8.3f ms %7.1f TF\n", round, t, fl / t * 1e-9);
            t = run<128, 128, 32, true, true, true, 3>(p, it);  printf("  r%d 128x128x32 fast  3/CU %8.3f ms %7.1f TF\n", round, t, fl / t * 1e-9);
        }
    }
    return 0;
}
