// Ablations of the arg-max pool's screening pass (vnp_screen_f16_wide_kernel) on maxpool2's shape: what each piece of a 64-wide K-step costs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vn_pointcloudcompletion_amd/csrc -I scripts/experiments scripts/experiments/screen_bench.hip -o scripts/experiments/screen_bench.bin
#include "screen_wide_kernel.h"
#include "screen_variants.h"
#include "screen_xwide_kernel.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <typename F> float time_it(F launch, int iters)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}
int main()
{
    const int M = 196608, N = 4096, K = 1024;
    unsigned short *A, *B; float *rn, *cn; float4* slots; int* sidx;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&rn, M * 4)); CK(hipMalloc(&cn, N * 4));
    const size_t nslot = (size_t)(M / 96) * (N / 2);
    CK(hipMalloc(&slots, nslot * 16)); CK(hipMalloc(&sidx, nslot * 4));
    std::vector<unsigned short> h((size_t)M * K);
    srand(1);
    for (auto& v : h) v = (unsigned short)(0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15));          // fp16 values of magnitude ~0.1-0.5, both signs
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    std::vector<float> ones(M, 1.f);
    CK(hipMemcpy(rn, ones.data(), M * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(cn, ones.data(), N * 4, hipMemcpyHostToDevice));
    VnpGemmArgs p{};
    p.A = (const float*)A; p.B = (const float*)B; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.rows_per_group = 1;
    p.tiles_per_split = K / 32; p.nsplit = 1; p.nbatch = 1; p.col_group = 8; p.pool_gs = 2048;
    p.pool_val = (float*)slots; p.pool_idx = sidx; p.row_norm = rn; p.col_norm = cn; p.screen_eps = 1e-3f; p.screen_abs = 1e-6f;
    constexpr int lds = 2 * (2 * (192 * 64 + 128) + 2 * (256 * 64 + 128));
    const dim3 grid(256), block(512);
#define RUN(name, kern) do { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        const float t = time_it([&] { hipLaunchKernelGGL(kern, grid, block, lds, 0, p); }, 10); \
        printf("%-52s %8.3f ms  %6.0f TF/s of one fp16 product\n", name, t, 2.0 * M * N * K / t * 1e-9); } while (0)
    // the 384 x 256 tile (LDS-DMA staging): time, then its slots against the 192 x 256 kernel's, bit for bit
    {
        constexpr int xlds = 2 * (2 * 384 * 64 + 2 * 256 * 64);
        CK(hipFuncSetAttribute((const void*)vnp_screen_f16_xwide_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, xlds));
        std::vector<float> ref(nslot * 4), got(nslot * 4);
        std::vector<int> refi(nslot), goti(nslot);
        CK(hipMemset(slots, 0xff, nslot * 16)); CK(hipMemset(sidx, 0xff, nslot * 4));
        CK(hipFuncSetAttribute((const void*)vnp_screen_f16_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(vnp_screen_f16_wide_kernel, grid, block, lds, 0, p);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ref.data(), slots, nslot * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(refi.data(), sidx, nslot * 4, hipMemcpyDeviceToHost));
        for (int cg : {2, 1, 4, 16}) {
            p.col_group = cg;
            CK(hipMemset(slots, 0xff, nslot * 16)); CK(hipMemset(sidx, 0xff, nslot * 4));
            const float t = time_it([&] { hipLaunchKernelGGL(vnp_screen_f16_xwide_kernel<0>, grid, dim3(512), xlds, 0, p); }, 10);
            CK(hipMemcpy(got.data(), slots, nslot * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(goti.data(), sidx, nslot * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < nslot * 4; ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
            for (size_t i = 0; i < nslot; ++i) bad += refi[i] != goti[i];
            printf("%-52s %8.3f ms  %6.0f TF/s   col_group %2d   words differing from the 192 x 256 kernel: %zu\n", "384 x 256 tile, LDS-DMA (xwide)", t,
                   2.0 * M * N * K / t * 1e-9, cg, bad);
        }
        p.col_group = 4;
#define XRUN(name, V) do { CK(hipFuncSetAttribute((const void*)vnp_screen_f16_xwide_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, xlds)); \
        const float t = time_it([&] { hipLaunchKernelGGL(vnp_screen_f16_xwide_kernel<V>, grid, dim3(512), xlds, 0, p); }, 10); \
        printf("xwide %-46s %8.3f ms  %6.0f TF/s\n", name, t, 2.0 * M * N * K / t * 1e-9); } while (0)
        XRUN("as it is", 0);
        XRUN("no DMA in the K loop", 1);
        XRUN("fragments read once per item", 2);
        XRUN("no DMA, fragments once (MFMAs + barriers)", 3);
        XRUN("no MFMAs", 4);
        XRUN("no MFMAs, no DMA (fragment reads + barriers)", 5);
        p.col_group = 8;
    }
    for (int round = 0; round < 2; ++round) {
        RUN("library kernel", vnp_screen_f16_wide_kernel);
        RUN("copy, VAR = 0", vnp_screen_f16_wide_var_kernel<0>);
        RUN("no global loads in the loop", vnp_screen_f16_wide_var_kernel<64>);
        RUN("no LDS writes in the loop", vnp_screen_f16_wide_var_kernel<32>);
        RUN("no fragment reads in the loop", vnp_screen_f16_wide_var_kernel<256>);
        RUN("no global loads, no LDS writes", vnp_screen_f16_wide_var_kernel<96>);
        RUN("MFMAs + barrier only", vnp_screen_f16_wide_var_kernel<352>);
        RUN("no MFMAs", vnp_screen_f16_wide_var_kernel<128>);
    }
    return 0;
}
