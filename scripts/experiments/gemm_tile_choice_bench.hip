// Dev harness (round 3): which row-tile height (192 / 128 / 96) is fastest for the mid-sized aligned NT products of the transformer
// configs (M = 12288 or 49152 rows), where a launch is one or two rounds of workgroups and the last round's fill decides the time?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vn_pointcloudcompletion_amd/csrc scripts/experiments/gemm_tile_choice_bench.hip -o scripts/experiments/gemm_tile_choice_bench.bin
#include "gemm_split_wide_kernel.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename F>
float time_it(F launch, int iters)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int BM>
float run(int M, int N, int K, const float* A, const float* B, float* C)
{
    VnpGemmArgs p{};
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.rows_per_group = 1;
    p.tiles_per_split = K / 32; p.nsplit = 1; p.nbatch = 1; p.col_group = 8;
    if (M % BM) return -1.f;
    const int items = (M / BM) * (N / 128);
    dim3 g(items < 512 ? items : 512);
    return time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_kernel<BM, 128, true, true, true>), g, dim3(256), 0, 0, p); }, 50);
}

float run_wide(int M, int N, int K, const float* A, const float* B, float* C)
{
    if (M % 192 || N % 256) return -1.f;
    VnpGemmArgs p{};
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.rows_per_group = 1;
    p.tiles_per_split = K / 32; p.nsplit = 1; p.nbatch = 1; p.col_group = 8;
    const int items = (M / 192) * (N / 256);
    dim3 g(items < 256 ? items : 256);
    return time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_wide_kernel<0>), g, dim3(512), 0, 0, p); }, 50);
}

int main()
{
    const size_t n = (size_t)49152 * 2048;
    float *A, *B, *C;
    CK(hipMalloc(&A, n * 4)); CK(hipMalloc(&B, n * 4)); CK(hipMalloc(&C, n * 4));
    std::vector<float> h(n);
    srand(1);
    for (auto& v : h) v = (float)(rand() & 0xffff) / 32768.f - 1.f;
    CK(hipMemcpy(A, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), n * 4, hipMemcpyHostToDevice));
    const int Ms[] = {12288, 49152, 24576, 6144};
    const int Ns[] = {128, 256, 384, 512, 768, 1024, 1152, 1536, 2048};
    const int Ks[] = {128, 384, 1024};
    printf("%7s %5s %5s | items192 | %9s %9s %9s %9s | best\n", "M", "N", "K", "192 us", "128 us", "96 us", "wide us");
    for (int M : Ms) for (int N : Ns) for (int K : Ks) {
        const float t192 = run<192>(M, N, K, A, B, C), t128 = run<128>(M, N, K, A, B, C), t96 = run<96>(M, N, K, A, B, C), tw = run_wide(M, N, K, A, B, C);
        float best = 1e9f; const char* bn = "";
        if (t192 > 0 && t192 < best) { best = t192; bn = "192"; }
        if (t128 > 0 && t128 < best) { best = t128; bn = "128"; }
        if (t96 > 0 && t96 < best) { best = t96; bn = "96"; }
        if (tw > 0 && tw < best) { best = tw; bn = "wide"; }
        printf("%7d %5d %5d | %8d | %9.1f %9.1f %9.1f %9.1f | %s\n", M, N, K, (M / 192) * (N / 128), t192 * 1e3, t128 * 1e3, t96 * 1e3, tw * 1e3, bn);
    }
    return 0;
}
