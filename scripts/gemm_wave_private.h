// EXPERIMENT (dev only): NT fp32-MFMA GEMM where every wave stages its OWN A (96 x BK) and B (64 x BK) tiles in a private
// LDS region: no workgroup barriers at all, waves drift freely (2x the L2->LDS traffic of the shared-tile kernel).
#pragma once
#include "gemm_kernel.h"

template <int BK>
__global__ void __launch_bounds__(256, 2) vnp_gemm_wp_kernel(const VnpGemmArgs p)
{
    constexpr int WM = 96, WN = 64, TM = 3, TN = 2;
    constexpr int LDS_ = BK + 4;
    constexpr int AV = WM * BK / 4 / 64, BV = WN * BK / 4 / 64;       // float4 per lane
    __shared__ __attribute__((aligned(16))) float smem[4 * (WM + WN) * LDS_];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    float* As = smem + wave * (WM + WN) * LDS_;
    float* Bs = As + WM * LDS_;
    const int nbn = p.N / 128, nbm = p.M / 192;
    const int nwg = nbm * nbn;
    const int nk = p.K / BK;
    // per-lane offsets: f = lane + 64 j -> row f / (BK/4), k4 = f % (BK/4); linear in j: row step 64/(BK/4)
    const int r0 = lane / (BK / 4), k4 = (lane % (BK / 4)) * 4;
    const int oa = r0 * p.lda + k4, sa = (64 / (BK / 4)) * p.lda;
    const int ob = r0 * p.ldb + k4, sb = (64 / (BK / 4)) * p.ldb;
    vnp_f4 ra[AV], rb[BV];
    int m0, n0;
    auto set_item = [&](int t) {
        const int q = nwg >> 3, r = nwg & 7, x = t & 7;
        const int wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3);
        m0 = (wg / nbn) * 192 + wm * WM; n0 = (wg % nbn) * 128 + wn * WN;
    };
    auto fetch = [&](int fm0, int fn0, int kt) {
        const float* pa = p.A + (size_t)fm0 * p.lda + kt * BK;
        const float* pb = p.B + (size_t)fn0 * p.ldb + kt * BK;
#pragma unroll
        for (int j = 0; j < AV; ++j) ra[j] = *(const vnp_f4*)(pa + oa + j * sa);
#pragma unroll
        for (int j = 0; j < BV; ++j) rb[j] = *(const vnp_f4*)(pb + ob + j * sb);
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < AV; ++j) *(vnp_f4*)&As[(r0 + j * (64 / (BK / 4))) * LDS_ + k4] = ra[j];
#pragma unroll
        for (int j = 0; j < BV; ++j) *(vnp_f4*)&Bs[(r0 + j * (64 / (BK / 4))) * LDS_ + k4] = rb[j];
    };
    int w = blockIdx.x;
    if (w >= nwg) return;
    set_item(w);
    fetch(m0, n0, 0);
    stage();
    while (true) {
        vnp_f16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int wnext = w + gridDim.x;
        const bool more = wnext < nwg;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) fetch(m0, n0, kt + 1);
            else if (more) { const int cm = m0, cn = n0; set_item(wnext); fetch(m0, n0, 0); m0 = cm; n0 = cn; }
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) {
                float a[TM][4], b[TN][4];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const vnp_f4 v = *(const vnp_f4*)&As[(i * 32 + li) * LDS_ + 8 * q + 4 * lh];
                    a[i][0] = v[0]; a[i][1] = v[1]; a[i][2] = v[2]; a[i][3] = v[3];
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const vnp_f4 v = *(const vnp_f4*)&Bs[(j * 32 + li) * LDS_ + 8 * q + 4 * lh];
                    b[j][0] = v[0]; b[j][1] = v[1]; b[j][2] = v[2]; b[j][3] = v[3];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
            }
            if (kt + 1 < nk) stage();
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                float* drow = p.C + (size_t)row * p.ldc;
#pragma unroll
                for (int j = 0; j < TN; ++j) drow[n0 + j * 32 + li] = acc[i][j][e];
            }
        if (!more) break;
        w = wnext;
        set_item(w);
        stage();
    }
}
