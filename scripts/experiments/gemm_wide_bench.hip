// Dev harness (round 3): vnp_gemm_split_kernel<192, 128> (256 threads, two workgroups per CU) against vnp_gemm_split_wide_kernel (512 threads,
// 192 x 256 tile) on the step's NT shapes: time, TF/s fp32-equivalent, bit-for-bit comparison of the outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vn_pointcloudcompletion_amd/csrc scripts/experiments/gemm_wide_bench.hip -o scripts/experiments/gemm_wide_bench.bin
#include "gemm_split_wide_kernel.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static bool g_pmc = false;
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

template <int EPI>
void run_case(const char* name, int M, int N, int K, int nsplit, int cg, const float* A, const float* B, float* C1, float* C2, float* S1, float* S2)
{
    VnpGemmArgs p{};
    p.A = A; p.B = B; p.M = M; p.N = N; p.K = K;
    p.lda = K; p.ldb = K; p.ldc = N; p.rows_per_group = 1;
    p.tiles_per_split = K / 32 / nsplit; p.nsplit = nsplit; p.nbatch = 1; p.slab = (long long)M * N;
    p.col_group = cg; p.xcd_split = 0;
    const int items1 = (M / 192) * (N / 128) * nsplit, items2 = (M / 192) * (N / 256) * nsplit;
    dim3 g1(items1 < 512 ? items1 : 512), g2(items2 < 256 ? items2 : 256);
    const double fl = 2.0 * M * N * K;
    VnpGemmArgs p1 = p, p2 = p; p1.C = C1; p2.C = C2;
    if (EPI == 2) { p1.stats_part = (float2*)S1; p2.stats_part = (float2*)S2; p1.stats_cols = p2.stats_cols = N / 2; }
    if (g_pmc) {          // counter runs (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE): each kernel twice, nothing else
        for (int i = 0; i < 2; ++i) {
            hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 0, EPI>), g1, dim3(256), 0, 0, p1);
            hipLaunchKernelGGL((vnp_gemm_split_wide_kernel<EPI>), g2, dim3(512), 0, 0, p2);
        }
        CK(hipDeviceSynchronize());
        return;
    }
    const size_t outn = (size_t)M * N * nsplit;
    CK(hipMemset(C1, 0xff, outn * 4)); CK(hipMemset(C2, 0x7f, outn * 4));
    printf("%-16s NT M=%d N=%d K=%d split=%d EPI=%d\n", name, M, N, K, nsplit, EPI);
    for (int round = 0; round < 2; ++round) {
        float t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 0, EPI>), g1, dim3(256), 0, 0, p1); }, 10);
        printf("  r%d 192x128 / 256 threads   %8.3f ms %7.1f TF-equiv\n", round, t, fl / t * 1e-9);
        t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_wide_kernel<EPI>), g2, dim3(512), 0, 0, p2); }, 10);
        printf("  r%d 192x256 / 512 threads   %8.3f ms %7.1f TF-equiv\n", round, t, fl / t * 1e-9);
        if (round == 1) {
            t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_wide_kernel<EPI, 16>), g2, dim3(512), 0, 0, p2); }, 10);
            printf("     192x256 without the output stores %8.3f ms\n", t);
            t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 16, EPI>), g1, dim3(256), 0, 0, p1); }, 10);
            printf("     192x128 without the output stores %8.3f ms\n", t);
        }
    }
    std::vector<float> h1(outn), h2(outn);
    CK(hipMemcpy(h1.data(), C1, outn * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), C2, outn * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < outn; ++i) diff += memcmp(&h1[i], &h2[i], 4) != 0;
    printf("  outputs: %zu of %zu words differ%s\n", diff, outn, diff ? "   <-- MISMATCH" : " (bit-identical)");
}

int main(int argc, char** argv)
{
    g_pmc = argc > 1 && !strcmp(argv[1], "pmc");
    const size_t maxA = (size_t)1572864 * 512, maxC = (size_t)1572864 * 512;
    float *A, *B, *C1, *C2, *S1, *S2;
    CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&B, maxA * 4)); CK(hipMalloc(&C1, maxC * 4)); CK(hipMalloc(&C2, maxC * 4));
    CK(hipMalloc(&S1, 1 << 26)); CK(hipMalloc(&S2, 1 << 26));          // EPI = 2: [M / 96][stats_cols] float2 partials
    CK(hipMemset(S1, 0, 1 << 26)); CK(hipMemset(S2, 0, 1 << 26));
    std::vector<float> hA(maxA);
    srand(1);
    auto rnd = [] { float u = (float)(rand() & 0xffffff) / 8388608.f - 1.f; int e = rand() % 13 - 6; return ldexpf(u, e); };
    for (auto& v : hA) v = rnd();
    CK(hipMemcpy(A, hA.data(), maxA * 4, hipMemcpyHostToDevice));
    for (auto& v : hA) v = rnd();
    CK(hipMemcpy(B, hA.data(), maxA * 4, hipMemcpyHostToDevice));
    run_case<0>("dec 256->512", 1572864, 512, 256, 1, 8, A, B, C1, C2, S1, S2);
    run_case<2>("dec 256->512 + BatchNorm statistics", 1572864, 512, 256, 1, 8, A, B, C1, C2, S1, S2);
    if (g_pmc) return 0;
    run_case<0>("dec dgrad as NT", 1572864, 256, 512, 1, 8, A, B, C1, C2, S1, S2);
    run_case<0>("enc 1024->2048", 196608, 2048, 1024, 1, 8, A, B, C1, C2, S1, S2);
    run_case<0>("enc 1024->2048 cg4", 196608, 2048, 1024, 1, 4, A, B, C1, C2, S1, S2);
    run_case<0>("odd items", 192 * 37, 256 * 3, 96, 1, 8, A, B, C1, C2, S1, S2);
    return 0;
}
