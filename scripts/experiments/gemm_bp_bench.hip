// Dev harness (round 5): vnp_gemm_split_wide_kernel (B split in the loop) against vnp_gemm_split_wide_bp_kernel (B's bf16 planes split once,
// moved global -> LDS by DMA) on the step's NT shapes: time, TF/s fp32-equivalent, bit-for-bit comparison of the outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vn_pointcloudcompletion_amd/csrc -I scripts/experiments scripts/experiments/gemm_bp_bench.hip -o scripts/experiments/gemm_bp_bench.bin
#include "gemm_split_wide_bp_kernel.h"
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

constexpr int BP_LDS = 3 * 192 * 64 + 2 * 3 * 256 * 64;

template <int EPI>
void run_case(const char* name, int M, int N, int K, int cg, const float* A, const float* B, unsigned short* Bp, float* C1, float* C2, float* S1, float* S2)
{
    VnpGemmArgs p{};
    p.A = A; p.B = B; p.M = M; p.N = N; p.K = K;
    p.lda = K; p.ldb = K; p.ldc = N; p.rows_per_group = 1;
    p.tiles_per_split = K / 32; p.nsplit = 1; p.nbatch = 1; p.slab = (long long)M * N;
    p.col_group = cg; p.xcd_split = 0;
    const int items = (M / 192) * (N / 256);
    dim3 g(items < 256 ? items : 256);
    const double fl = 2.0 * M * N * K;
    VnpGemmArgs p1 = p, p2 = p; p1.C = C1; p2.C = C2;
    if (EPI == 2) { p1.stats_part = (float2*)S1; p2.stats_part = (float2*)S2; p1.stats_cols = p2.stats_cols = N / 2; }
    const size_t outn = (size_t)M * N;
    CK(hipMemset(C1, 0xff, outn * 4)); CK(hipMemset(C2, 0x7f, outn * 4));
    CK(hipFuncSetAttribute((const void*)vnp_gemm_split_wide_bp_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, BP_LDS));
    printf("%-16s NT M=%d N=%d K=%d EPI=%d\n", name, M, N, K, EPI);
    float tsp = time_it([&] { hipLaunchKernelGGL(vnp_split_planes_kernel, dim3(512), dim3(256), 0, 0, B, K, (long long)N, K, Bp, K, (long long)N * K); }, 10);
    printf("  split of B into planes          %8.4f ms\n", tsp);
    for (int round = 0; round < 3; ++round) {
        float t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_wide_kernel<EPI>), g, dim3(512), 0, 0, p1); }, 10);
        printf("  r%d wide (B split in the loop)   %8.3f ms %7.1f TF-equiv\n", round, t, fl / t * 1e-9);
        t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_wide_bp_kernel<EPI>), g, dim3(512), BP_LDS, 0, p2, (const unsigned short*)Bp, (long long)N * K, K); }, 10);
        printf("  r%d wide (B planes by DMA)       %8.3f ms %7.1f TF-equiv\n", round, t, fl / t * 1e-9);
    }
    std::vector<float> h1(outn), h2(outn);
    CK(hipMemcpy(h1.data(), C1, outn * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), C2, outn * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < outn; ++i) diff += memcmp(&h1[i], &h2[i], 4) != 0;
    printf("  outputs: %zu of %zu words differ%s\n", diff, outn, diff ? "   <-- MISMATCH" : " (bit-identical)");
    if (EPI == 2) {
        const size_t sn = (size_t)(M / 96) * (N / 2) * 2;
        std::vector<float> s1(sn), s2(sn);
        CK(hipMemcpy(s1.data(), S1, sn * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s2.data(), S2, sn * 4, hipMemcpyDeviceToHost));
        size_t d2 = 0;
        for (size_t i = 0; i < sn; ++i) d2 += memcmp(&s1[i], &s2[i], 4) != 0;
        printf("  statistics partials: %zu of %zu words differ\n", d2, sn);
    }
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const size_t maxA = (size_t)1572864 * 512, maxC = (size_t)1572864 * 512;
    float *A, *B, *C1, *C2, *S1, *S2;
    unsigned short* Bp;
    CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&B, (size_t)2048 * 2048 * 4)); CK(hipMalloc(&C1, maxC * 4)); CK(hipMalloc(&C2, maxC * 4));
    CK(hipMalloc(&Bp, (size_t)2048 * 2048 * 6));
    CK(hipMalloc(&S1, 1 << 26)); CK(hipMalloc(&S2, 1 << 26));
    CK(hipMemset(S1, 0, 1 << 26)); CK(hipMemset(S2, 0, 1 << 26));
    std::vector<float> hA(maxA);
    srand(1);
    auto rnd = [] { float u = (float)(rand() & 0xffffff) / 8388608.f - 1.f; int e = rand() % 13 - 6; return ldexpf(u, e); };
    for (auto& v : hA) v = rnd();
    CK(hipMemcpy(A, hA.data(), maxA * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < (size_t)2048 * 2048; ++i) hA[i] = rnd();
    CK(hipMemcpy(B, hA.data(), (size_t)2048 * 2048 * 4, hipMemcpyHostToDevice));
    run_case<0>("dec 256->512", 1572864, 512, 256, 8, A, B, Bp, C1, C2, S1, S2);
    run_case<2>("dec 256->512 + BatchNorm statistics", 1572864, 512, 256, 8, A, B, Bp, C1, C2, S1, S2);
    run_case<0>("dec dgrad as NT", 1572864, 256, 512, 8, A, B, Bp, C1, C2, S1, S2);
    run_case<0>("enc 1024->2048", 196608, 2048, 1024, 8, A, B, Bp, C1, C2, S1, S2);
    run_case<0>("odd items", 192 * 37, 256 * 3, 96, 8, A, B, Bp, C1, C2, S1, S2);
    return 0;
}
