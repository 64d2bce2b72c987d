// Tuning / accuracy harness (dev tool): fp32-MFMA kernel vs bf16x3-split kernel, same process, same operands.
//   hipcc --offload-arch=gfx950 -O3 -I vn_pointcloudcompletion_amd/csrc scripts/gemm_split_bench.hip -o scripts/gemm_split_bench.bin
#include "gemm_split_kernel.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

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


// one case: op_a/op_b as in vnp_gemm; nsplit slabs are summed on the host for the accuracy sample
template <int BM, bool AKC, bool BKC>
void run_case(const char* name, int M, int N, int K, int nsplit, const float* A, const float* B, float* C1, float* C2,
              const std::vector<float>& hA, const std::vector<float>& hB)
{
    VnpGemmArgs p{};
    p.A = A; p.B = B; p.M = M; p.N = N; p.K = K;
    p.lda = AKC ? K : M; p.ldb = BKC ? K : N; p.ldc = N; p.rows_per_group = 1;
    p.tiles_per_split = K / 32 / nsplit; p.nsplit = nsplit; p.nbatch = 1; p.slab = (long long)M * N;
    const int items = (M / BM) * (N / 128) * nsplit;
    dim3 grid(items < 512 ? items : 512), block(256);
    const double fl = 2.0 * M * N * K;
    printf("%-16s %c%c M=%d N=%d K=%d split=%d\n", name, AKC ? 'N' : 'T', BKC ? 'T' : 'N', M, N, K, nsplit);
    VnpGemmArgs p1 = p, p2 = p; p1.C = C1; p2.C = C2;
    for (int round = 0; round < 2; ++round) {
        float t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_kernel<BM, 128, 32, AKC, BKC, true>), grid, block, 0, 0, p1); }, 5);
        printf("  r%d fp32 mfma    %8.3f ms %7.1f TF\n", round, t, fl / t * 1e-9);
        t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_kernel<BM, 128, AKC, BKC, true>), grid, block, 0, 0, p2); }, 5);
        printf("  r%d bf16x3 split %8.3f ms %7.1f TF-equiv\n", round, t, fl / t * 1e-9);
        for (int cgv : {4, 8, 16}) {
            if (cgv >= N / 128) continue;
            VnpGemmArgs p4 = p2; p4.col_group = cgv;
            t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_kernel<BM, 128, AKC, BKC, true>), grid, block, 0, 0, p4); }, 5);
            printf("  r%d   col_group %2d %8.3f ms %7.1f TF-equiv\n", round, cgv, t, fl / t * 1e-9);
        }
    }
    {   // in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6): d(s_memtime) / d(s_memrealtime) x 100 MHz
        unsigned long long* dbg;
        CK(hipMalloc(&dbg, 1024 * 16));
        VnpGemmArgs p3 = p2; p3.bias = (const float*)dbg;
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((vnp_gemm_split_kernel<BM, 128, AKC, BKC, true, 8>), grid, block, 0, 0, p3);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(2 * grid.x);
        CK(hipMemcpy(h.data(), dbg, grid.x * 16, hipMemcpyDeviceToHost));
        std::vector<double> mhz;
        for (unsigned i = 0; i < grid.x; ++i) if (h[2 * i + 1]) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        if (!mhz.empty()) printf("  in-kernel clock after 200 launches: median %.0f MHz (min %.0f, max %.0f), block life %.3f ms\n",
                                 mhz[mhz.size() / 2], mhz.front(), mhz.back(), h[1] / 100e3);
        CK(hipFree(dbg));
    }
    const int NS = 1000;
    double e1 = 0, e2 = 0, ref2 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < NS; ++i) {
        const size_t r = (size_t)rand() % M, c = (size_t)rand() % N;
        double v1 = 0, v2 = 0;
        for (int z = 0; z < nsplit; ++z) {
            float a, b;
            CK(hipMemcpy(&a, C1 + (size_t)z * M * N + r * N + c, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&b, C2 + (size_t)z * M * N + r * N + c, 4, hipMemcpyDeviceToHost));
            v1 += a; v2 += b;
        }
        double ref = 0, mag = 0;
        for (int k = 0; k < K; ++k) {
            const double x = AKC ? hA[r * K + k] : hA[(size_t)k * M + r];
            const double y = BKC ? hB[c * K + k] : hB[(size_t)k * N + c];
            ref += x * y; mag += fabs(x * y);
        }
        e1 += (v1 - ref) * (v1 - ref); e2 += (v2 - ref) * (v2 - ref); ref2 += ref * ref;
        m1 = std::max(m1, fabs(v1 - ref) / mag); m2 = std::max(m2, fabs(v2 - ref) / mag);
    }
    printf("  error vs fp64: fp32-mfma rel-rms %.3e  max|err|/sum|xy| %.3e ; split rel-rms %.3e  max %.3e\n",
           sqrt(e1 / ref2), m1, sqrt(e2 / ref2), m2);
}

int main(int argc, char** argv)
{
    const size_t maxA = (size_t)1572864 * 512, maxB = (size_t)1572864 * 512, maxC = (size_t)196608 * 4096;
    float *A, *B, *C1, *C2;
    CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&B, maxB * 4)); CK(hipMalloc(&C1, maxC * 4)); CK(hipMalloc(&C2, maxC * 4));
    std::vector<float> hA(maxA), hB(maxB);
    srand(1);
    // wide dynamic range: uniform magnitudes times 2^[-6, 6]
    auto rnd = [] { float u = (float)(rand() & 0xffffff) / 8388608.f - 1.f; int e = rand() % 13 - 6; return ldexpf(u, e); };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    CK(hipMemcpy(A, hA.data(), maxA * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), maxB * 4, hipMemcpyHostToDevice));
    run_case<192, true, true>("enc 1024->2048", 196608, 2048, 1024, 1, A, B, C1, C2, hA, hB);
    run_case<192, true, true>("pool x|d N=4096", 196608, 4096, 1024, 1, A, B, C1, C2, hA, hB);
    run_case<192, true, true>("dec 256->512", 1572864, 512, 256, 1, A, B, C1, C2, hA, hB);
    run_case<192, true, false>("enc dgrad NN", 196608, 512, 2048, 1, A, B, C1, C2, hA, hB);
    run_case<192, true, false>("dec dgrad NN", 1572864, 256, 512, 1, A, B, C1, C2, hA, hB);
    run_case<128, false, false>("enc wgrad TN", 2048, 512, 196608, 16, A, B, C1, C2, hA, hB);
    run_case<128, false, false>("dec wgrad TN", 512, 256, 1572864, 128, A, B, C1, C2, hA, hB);
    run_case<128, false, true>("TT", 2048, 512, 4096, 1, A, B, C1, C2, hA, hB);
    return 0;
}
