// Dev harness (round 3): the library's vnp_gemm_split_kernel (256 threads, two workgroups per CU) against two restructurings of the same
// arithmetic — vnp_gemm_split_pp_kernel (512 threads, halves in anti-phase) and vnp_gemm_split_db_kernel (512 threads, two LDS stages, the
// split interleaved into the wave's own MFMAs) — on the step's shapes: time, TF/s fp32-equivalent, bit-for-bit comparison of the outputs,
// in-kernel clock, per-segment stamps and ablations.  Results: profiles/r03_gemm_segments.txt.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vn_pointcloudcompletion_amd/csrc -I scripts/experiments \
//         scripts/experiments/gemm_variants_bench.hip -o scripts/experiments/gemm_variants_bench.bin
#include "gemm_split_pp_kernel.h"
#include "gemm_split_db_kernel.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

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

template <int BM, bool AKC, bool BKC>
void run_case(const char* name, int M, int N, int K, int nsplit, int xcd_split, const float* A, const float* B, float* C1, float* C2, float* C3)
{
    VnpGemmArgs p{};
    p.A = A; p.B = B; p.M = M; p.N = N; p.K = K;
    p.lda = AKC ? K : M; p.ldb = BKC ? K : N; p.ldc = N; p.rows_per_group = 1;
    p.tiles_per_split = K / 32 / nsplit; p.nsplit = nsplit; p.nbatch = 1; p.slab = (long long)M * N;
    p.col_group = 8; p.xcd_split = xcd_split;
    const int items = (M / BM) * (N / 128) * nsplit;
    dim3 g1(items < 512 ? items : 512), g2((items + 1) / 2 < 256 ? (items + 1) / 2 : 256);
    const double fl = 2.0 * M * N * K;
    VnpGemmArgs p1 = p, p2 = p, p3 = p; p1.C = C1; p2.C = C2; p3.C = C3;
    const size_t outn = (size_t)M * N * nsplit;
    CK(hipMemset(C1, 0xff, outn * 4)); CK(hipMemset(C2, 0x7f, outn * 4)); CK(hipMemset(C3, 0x3f, outn * 4));
    printf("%-16s %c%c M=%d N=%d K=%d split=%d\n", name, AKC ? 'N' : 'T', BKC ? 'T' : 'N', M, N, K, nsplit);
    if (g_pmc) {
        for (int i = 0; i < 2; ++i) {
            hipLaunchKernelGGL((vnp_gemm_split_kernel<BM, 128, AKC, BKC, true>), g1, dim3(256), 0, 0, p1);
            if constexpr (AKC == BKC) hipLaunchKernelGGL((vnp_gemm_split_db_kernel<BM, 128, AKC>), dim3(items < 256 ? items : 256), dim3(512), 0, 0, p3);
        }
        CK(hipDeviceSynchronize());
        return;
    }
    for (int round = 0; round < 2; ++round) {
        float t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_kernel<BM, 128, AKC, BKC, true>), g1, dim3(256), 0, 0, p1); }, 10);
        printf("  r%d 256-thread   %8.3f ms %7.1f TF-equiv\n", round, t, fl / t * 1e-9);
        if (round == 0) {
            t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_pp_kernel<BM, 128, AKC, BKC, true>), g2, dim3(512), 0, 0, p2); }, 10);
            printf("  r%d ping-pong    %8.3f ms %7.1f TF-equiv\n", round, t, fl / t * 1e-9);
        }
        if constexpr (AKC == BKC) {
            dim3 g3(items < 256 ? items : 256);
            t = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_db_kernel<BM, 128, AKC>), g3, dim3(512), 0, 0, p3); }, 10);
            printf("  r%d double-buf   %8.3f ms %7.1f TF-equiv\n", round, t, fl / t * 1e-9);
            if (round == 1 && BM == 192) {
                auto abl = [&](const char* what, auto var) {
                    constexpr int V = decltype(var)::value;
                    const float ta = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_db_kernel<BM, 128, AKC, 0, V>), g3, dim3(512), 0, 0, p2); }, 10);
                    printf("     ablation %-44s %8.3f ms (%+5.1f %%)\n", what, ta, (ta / t - 1.f) * 100.f);
                };
                abl("no global loads in the loop", std::integral_constant<int, 64>{});
                abl("global loads issued, never waited for", std::integral_constant<int, 512>{});
            }
            if (false) {          // ablations of the double-buffered kernel (wrong results by construction): what each piece of a K-step costs
                auto abl = [&](const char* what, auto var) {
                    constexpr int V = decltype(var)::value;
                    const float ta = time_it([&] { hipLaunchKernelGGL((vnp_gemm_split_db_kernel<BM, 128, AKC, 0, V>), g3, dim3(512), 0, 0, p2); }, 10);
                    printf("     ablation %-44s %8.3f ms (%+5.1f %%)\n", what, ta, (ta / t - 1.f) * 100.f);
                };
                abl("no split arithmetic", std::integral_constant<int, 16>{});
                abl("no LDS writes", std::integral_constant<int, 32>{});
                abl("no split arithmetic, no LDS writes", std::integral_constant<int, 48>{});
                abl("no global loads in the loop", std::integral_constant<int, 64>{});
                abl("no MFMAs", std::integral_constant<int, 128>{});
                abl("no fragment reads", std::integral_constant<int, 256>{});
                abl("no fragment reads, no LDS writes", std::integral_constant<int, 288>{});
                abl("MFMAs + barrier only", std::integral_constant<int, 16 + 32 + 64 + 256>{});
                abl("MFMAs + fragment reads only", std::integral_constant<int, 16 + 32 + 64>{});
                abl("everything but MFMAs and split", std::integral_constant<int, 16 + 128>{});
            }
        }
    }
    {   // in-kernel clock and workgroup life in shader cycles (MI355X_MICROARCH.md, DVFS give-back item 6), after a run of back-to-back launches
        unsigned long long* dbg;
        CK(hipMalloc(&dbg, 1024 * 16));
        auto report = [&](const char* what, unsigned nb) {
            std::vector<unsigned long long> h(2 * nb);
            CK(hipMemcpy(h.data(), dbg, nb * 16, hipMemcpyDeviceToHost));
            std::vector<double> mhz, cyc;
            for (unsigned i = 0; i < nb; ++i) if (h[2 * i + 1]) { mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0); cyc.push_back((double)h[2 * i]); }
            std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
            if (!mhz.empty()) printf("  %-12s in-kernel clock median %.0f MHz (min %.0f, max %.0f); workgroup life median %.0f kcycles = %.0f cycles per K-step and tile\n",
                                     what, mhz[mhz.size() / 2], mhz.front(), mhz.back(), cyc[cyc.size() / 2] * 1e-3,
                                     cyc[cyc.size() / 2] / ((double)items / nb * (K / 32 / nsplit)));
        };
        VnpGemmArgs pd = p1; pd.bias = (const float*)dbg;
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((vnp_gemm_split_kernel<BM, 128, AKC, BKC, true, 8>), g1, dim3(256), 0, 0, pd);
        CK(hipDeviceSynchronize());
        report("256-thread", g1.x);
        if constexpr (AKC == BKC) {
            dim3 g3(items < 256 ? items : 256);
            VnpGemmArgs pe = p3; pe.bias = (const float*)dbg;
            for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((vnp_gemm_split_db_kernel<BM, 128, AKC, 0, 8>), g3, dim3(512), 0, 0, pe);
            CK(hipDeviceSynchronize());
            report("double-buf", g3.x);
        }
        CK(hipFree(dbg));
    }
    std::vector<float> h1(outn), h2(outn);
    CK(hipMemcpy(h1.data(), C1, outn * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), C2, outn * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < outn; ++i) diff += memcmp(&h1[i], &h2[i], 4) != 0;
    printf("  ping-pong outputs: %zu of %zu words differ%s\n", diff, outn, diff ? "   <-- MISMATCH" : " (bit-identical)");
    CK(hipMemcpy(h2.data(), C3, outn * 4, hipMemcpyDeviceToHost));
    diff = 0;
    for (size_t i = 0; i < outn; ++i) diff += memcmp(&h1[i], &h2[i], 4) != 0;
    printf("  double-buffered outputs: %zu of %zu words differ%s\n", diff, outn, diff ? "   <-- MISMATCH" : " (bit-identical)");
}

int main(int argc, char** argv)
{
    g_pmc = argc > 1 && !strcmp(argv[1], "pmc");          // counter runs: two launches of each kernel on two shapes, nothing else
    const size_t maxA = (size_t)1572864 * 512, maxB = (size_t)1572864 * 512, maxC = (size_t)1572864 * 512;
    float *A, *B, *C1, *C2, *C3;
    CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&B, maxB * 4)); CK(hipMalloc(&C1, maxC * 4)); CK(hipMalloc(&C2, maxC * 4)); CK(hipMalloc(&C3, maxC * 4));
    std::vector<float> hA(maxA);
    srand(1);
    auto rnd = [] { float u = (float)(rand() & 0xffffff) / 8388608.f - 1.f; int e = rand() % 13 - 6; return ldexpf(u, e); };
    for (auto& v : hA) v = rnd();
    CK(hipMemcpy(A, hA.data(), maxA * 4, hipMemcpyHostToDevice));
    for (auto& v : hA) v = rnd();
    CK(hipMemcpy(B, hA.data(), maxB * 4, hipMemcpyHostToDevice));
    run_case<192, true, true>("dec 256->512", 1572864, 512, 256, 1, 0, A, B, C1, C2, C3);
    if (!g_pmc) run_case<192, true, false>("dec dgrad NN", 1572864, 256, 512, 1, 0, A, B, C1, C2, C3);
    if (!g_pmc) run_case<192, true, true>("dec dgrad as NT", 1572864, 256, 512, 1, 0, A, B, C1, C2, C3);
    run_case<128, false, false>("dec wgrad TN", 512, 256, 1572864, 128, 1, A, B, C1, C2, C3);
    if (g_pmc) return 0;
    run_case<192, true, true>("enc 1024->2048", 196608, 2048, 1024, 1, 0, A, B, C1, C2, C3);
    run_case<128, true, true>("pt 128->2048", 65536, 2048, 128, 1, 0, A, B, C1, C2, C3);
    run_case<192, true, true>("odd items", 192 * 37, 128 * 3, 96, 1, 0, A, B, C1, C2, C3);
    return 0;
}
