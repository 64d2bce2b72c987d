// Dev harness (round 3): vnp_gemm_split_kernel<128, 128, false, false> (256 threads, two workgroups per CU) against vnp_gemm_split_wide_tn_kernel
// (512 threads, 256 x 128 tile) on the step's weight-gradient shapes: time, TF/s fp32-equivalent, bit-for-bit comparison of the slabs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vn_pointcloudcompletion_amd/csrc scripts/experiments/gemm_wide_tn_bench.hip -o scripts/experiments/gemm_wide_tn_bench.bin
#include "gemm_split_wide_kernel.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename F>
float time_it(F launch, int iters)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

void run_case(const char* name, int M, int N, int K, int nsplit, const float* A, const float* B, float* C1, float* C2)
{
    VnpGemmArgs p{};
    p.A = A; p.B = B; p.M = M; p.N = N; p.K = K;
    p.lda = M; p.ldb = N; p.ldc = N; p.rows_per_group = 1;
    p.tiles_per_split = (K / 32 + nsplit - 1) / nsplit; p.nsplit = nsplit; p.nbatch = 1; p.slab = (long long)M * N;
    p.col_group = 8; p.xcd_split = nsplit % 8 == 0 ? 1 : 0;
    const int items1 = (M / 128) * (N / 128) * nsplit, items2 = (M / 256) * (N / 128) * nsplit;
    dim3 g1(items1 < 512 ? items1 : 512), g2(items2 < 256 ? items2 : 256);
    const double fl = 2.0 * M * N * K;
    VnpGemmArgs p1 = p, p2 = p; p1.C = C1; p2.C = C2;
    const size_t outn = (size_t)M * N * nsplit;
    CK(hipMemset(C1, 0xff, outn * 4)); CK(hipMemset(C2, 0x7f, outn * 4));
    printf("%-22s TN M=%d N=%d K=%d split=%d\n", name, M, N, K, nsplit);
    for (int round = 0; round < 2; ++round) {
        float t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_kernel<128, 128, false, false, true>), g1, dim3(256), 0, 0, p1); }, 10);
        printf("  r%d 128x128 / 256 threads   %8.3f ms %7.1f TF-equiv\n", round, t, fl / t * 1e-9);
        t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_wide_tn_kernel<0>), g2, dim3(512), 0, 0, p2); }, 10);
        printf("  r%d 256x128 / 512 threads   %8.3f ms %7.1f TF-equiv\n", round, t, fl / t * 1e-9);
    }
    std::vector<float> h1(outn), h2(outn);
    CK(hipMemcpy(h1.data(), C1, outn * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), C2, outn * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < outn; ++i) diff += memcmp(&h1[i], &h2[i], 4) != 0;
    printf("  slabs: %zu of %zu words differ%s\n", diff, outn, diff ? "   <-- MISMATCH" : " (bit-identical)");
}

int main()
{
    const size_t maxA = (size_t)1572864 * 512, maxC = (size_t)64 << 20;
    float *A, *B, *C1, *C2;
    CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&B, maxA * 4)); CK(hipMalloc(&C1, maxC * 4)); CK(hipMalloc(&C2, maxC * 4));
    std::vector<float> hA(maxA);
    srand(1);
    auto rnd = [] { float u = (float)(rand() & 0xffffff) / 8388608.f - 1.f; int e = rand() % 13 - 6; return ldexpf(u, e); };
    for (auto& v : hA) v = rnd();
    CK(hipMemcpy(A, hA.data(), maxA * 4, hipMemcpyHostToDevice));
    for (auto& v : hA) v = rnd();
    CK(hipMemcpy(B, hA.data(), maxA * 4, hipMemcpyHostToDevice));
    run_case("dec wgrad", 512, 256, 1572864, 64, A, B, C1, C2);
    run_case("cfg4 fold wgrad", 256, 256, 786432, 128, A, B, C1, C2);
    run_case("cfg4 768x768", 768, 768, 49152, 14, A, B, C1, C2);
    run_case("cfg4 1536x384", 1536, 384, 49152, 14, A, B, C1, C2);
    run_case("enc 2048x128", 2048, 128, 65536, 32, A, B, C1, C2);
    run_case("cfg5 512x128", 512, 128, 12288, 96, A, B, C1, C2);
    run_case("odd split", 512, 384, 4000 * 32, 13, A, B, C1, C2);
    return 0;
}
