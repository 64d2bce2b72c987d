// The one-fp16-product screening pass of the exact screened arg-max pool (vnp_gemm_split_kernel<.., EPI = 4>, gemm_split_kernel.h) on
// 192 x 256 tiles and eight waves (2 (M) x 4 (N), 512 threads, one workgroup per CU), two LDS stages, ONE barrier per 64-wide K step.
//
// A tile twice as wide reads the A panel once for twice the columns — (192 + 256) / (192 x 256) operand rows per output instead of
// (192 + 128) / (192 x 128), 0.70x the L2 bytes per MFMA — and the second stage removes one of the two barriers per step.  Measured on
// maxpool2's shape (196608 x 4096 x 1024, A/B on one device, VNP_SCREEN_WIDE=0 selects the 256-thread pass): the pool 3.21 -> 3.13 ms
// with the wide tile alone, -> 3.05 ms with two stages.  That is less than the byte count promised: per 32-wide K-tile a wave issues 24
// MFMAs (384 pipe cycles) behind 10 ds_read_b128 (8 waves x 10 KB = 640 LDS cycles per CU at 128 B/clk), so the LDS read port limits the
// pass as much as L2 does, and a 96 x 64 wave tile cannot amortise its fragment reads further without more accumulators than two waves
// per SIMD leave room for.  Same operands (the fp16 images written by vnp_row_norm_kernel), same LDS layout per 32-wide half-tile, same
// MFMA sequence per output element and the same epilogue, so the slots it writes are bit-identical to the 256-thread pass's
// (tests/test_gpu_layers.py: every screened-pool test with 2C % 256 == 0 runs it).
#pragma once
#include "gemm_split_kernel.h"

__global__ void __launch_bounds__(512, 1) vnp_screen_f16_wide_kernel(const VnpGemmArgs p)
{
    constexpr int BM = 192, BN = 256, BK2 = 64;
    constexpr int TM = 6, TN = 4;                               // 16x16 MFMA tiles per wave: 96 x 64
    constexpr int APL = BM * 64, BPL = BN * 64;                 // bytes per 32-wide half-tile
    // the second half sits 64 bytes past the first: a ds_write_b128 is served in groups of 8 contiguous lanes over 32 banks (128 bytes), and
    // the 8 lanes of a group write one row's four slots for BOTH halves — 128 bytes apart (rounds 2-3) they fell on the same banks
    // (SQ_LDS_BANK_CONFLICT 0.28 of the LDS cycles of an LDS-bound kernel); the fragment reads only see a uniform shift
    constexpr int APL2 = APL + 64, BPL2 = BPL + 64;
    constexpr int AH = BM / 64, BH = BN / 64;                   // 16-byte loads per thread and step
    // two stages (2 x 57 KB of dynamic LDS): step kt computes from stage kt & 1 while the next step's operands go into the other one
    constexpr int STAGE = 2 * APL2 + 2 * BPL2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    if (p.pred && *p.pred == 0) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nbn = p.N / BN, nbm = p.M / BM;
    const int total = nbm * nbn;
    const int nkt = p.K / BK2;
    const unsigned short* A16 = (const unsigned short*)p.A;
    const unsigned short* B16 = (const unsigned short*)p.B;

    int m0, n0;
    auto set_item = [&](int w) {                            // XCD-aware walk: each XCD a contiguous run of tiles, column tile fastest inside a group
        const int q = total >> 3, r = total & 7, x = w & 7;
        const int wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (w >> 3);
        const int cg = (p.col_group > 0 && p.col_group < nbn) ? p.col_group : nbn;
        const int full = (nbn / cg) * cg * nbm;
        int row, col;
        if (wg < full) { const int g = wg / (nbm * cg), rem = wg - g * (nbm * cg); row = rem / cg; col = g * cg + rem % cg; }
        else { const int wl = nbn - (nbn / cg) * cg, rem = wg - full; row = rem / wl; col = (nbn / cg) * cg + rem % wl; }
        m0 = row * BM; n0 = col * BN;
    };

    int offA[AH], offB[BH];
#pragma unroll
    for (int j = 0; j < AH; ++j) {
        // LDS row rho holds component t of point (16 i3 + r16) of its 32-point half: MFMA row-tile 3 i3 + t, row r16 (as in EPI = 1)
        const int rho = (tid >> 3) + 64 * j, half = rho / 96, within = rho % 96, i = within >> 4, r16 = within & 15;
        offA[j] = (half * 96 + ((i / 3) * 16 + r16) * 3 + i % 3) * p.lda + (tid & 7) * 8;
    }
#pragma unroll
    for (int j = 0; j < BH; ++j) offB[j] = ((tid >> 3) + 64 * j) * p.ldb + (tid & 7) * 8;
    const int sub = (tid >> 2) & 1, q4 = tid & 3;               // which 32-wide half, and the 16-byte slot inside it
    const int l16 = lane & 15, lq = lane >> 4;
    int a_rd[TM], b_rd[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_rd[i] = vnp_split_row_off<BM, true>(wm * 96 + i * 16 + l16, lq);
#pragma unroll
    for (int j = 0; j < TN; ++j) b_rd[j] = vnp_split_row_off<BN, true>(wn * 64 + j * 16 + l16, lq);

    vnp_u4 ha[AH], hb[BH];
    auto fetch = [&](int fm0, int fn0, int kt) {
        const unsigned short* srcA = A16 + (size_t)fm0 * p.lda + kt * BK2;
        const unsigned short* srcB = B16 + (size_t)fn0 * p.ldb + kt * BK2;
#pragma unroll
        for (int j = 0; j < AH; ++j) ha[j] = *(const vnp_u4*)(srcA + offA[j]);
#pragma unroll
        for (int j = 0; j < BH; ++j) hb[j] = *(const vnp_u4*)(srcB + offB[j]);
    };
    auto write = [&](int stage) {
        char* As = smem + stage * STAGE;
        char* Bs = As + 2 * APL2;
#pragma unroll
        for (int j = 0; j < AH; ++j) *(vnp_u4*)(As + sub * APL2 + vnp_split_row_off<BM, true>((tid >> 3) + 64 * j, q4)) = ha[j];
#pragma unroll
        for (int j = 0; j < BH; ++j) *(vnp_u4*)(Bs + sub * BPL2 + vnp_split_row_off<BN, true>((tid >> 3) + 64 * j, q4)) = hb[j];
    };

    int w = blockIdx.x;
    if (w >= total) return;
    set_item(w);
    fetch(m0, n0, 0);
    write(0);
    __syncthreads();
    int stage = 0;
    while (true) {
        vnp_f4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        const int wnext = w + gridDim.x;
        const bool more = wnext < total;
        const int cm0 = m0, cn0 = n0;
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 1 < nkt) fetch(cm0, cn0, kt + 1);
            else if (more) { set_item(wnext); fetch(m0, n0, 0); }          // the next item's first step, behind this item's last MFMAs
            const char* As = smem + stage * STAGE;
            const char* Bs = As + 2 * APL2;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                vnp_h8 b[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = __builtin_bit_cast(vnp_h8, *(const vnp_u4*)(Bs + s * BPL2 + b_rd[j]));
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const vnp_h8 a = __builtin_bit_cast(vnp_h8, *(const vnp_u4*)(As + s * APL2 + a_rd[i]));
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[j], acc[i][j], 0, 0, 0);
                }
            }
            // the other stage was last read in the previous step, whose barrier every wave has passed
            if (kt + 1 < nkt || more) write(stage ^ 1);
            __syncthreads();
            stage ^= 1;
        }

        // ---- epilogue: the bounds of vnp_gemm_split_kernel's screen_epilogue, term for term (see there for the derivation)
        {
            const int pt0 = cm0 / 3 + wm * 32;
            const int grp = pt0 / 32;
            const int loc0 = pt0 % p.pool_gs;
            const int Cch = p.N >> 1;
            const int rbase = cm0 + wm * 96;
            float nf[2][4][3];
#pragma unroll
            for (int i3 = 0; i3 < 2; ++i3)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < 3; ++t) nf[i3][e][t] = p.row_norm[rbase + (16 * i3 + 4 * lq + e) * 3 + t];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int colx = cn0 + wn * 64 + jj * 16 + l16;
                const float cnx = p.col_norm[colx], cnd = p.col_norm[colx + 32];
                const float Ex = p.screen_eps * cnx, Ed = p.screen_eps * cnd;
                const float Ax = p.screen_abs * cnx, Ad = p.screen_abs * cnd, An = p.screen_abs;
                float lmax = -INFINITY, u1 = -INFINITY, u2 = -INFINITY, u3 = -INFINITY;
                int i1 = 0, i2 = 0;
                bool bad = false;
                auto insert = [&](float up, int ix) {
                    if (up > u1) { u3 = u2; u2 = u1; i2 = i1; u1 = up; i1 = ix; }
                    else if (up > u2) { u3 = u2; u2 = up; i2 = ix; }
                    else u3 = fmaxf(u3, up);
                };
#pragma unroll
                for (int i3 = 0; i3 < 2; ++i3)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = acc[3 * i3][jj][e], x1 = acc[3 * i3 + 1][jj][e], x2 = acc[3 * i3 + 2][jj][e];
                        const float d0 = acc[3 * i3][jj + 2][e], d1 = acc[3 * i3 + 1][jj + 2][e], d2 = acc[3 * i3 + 2][jj + 2][e];
                        const float st = fmaf(x2, d2, fmaf(x1, d1, x0 * d0));
                        const float ex0 = fmaf(Ex, nf[i3][e][0], fmaf(An, nf[i3][e][0], Ax)), ed0 = fmaf(Ed, nf[i3][e][0], fmaf(An, nf[i3][e][0], Ad));
                        const float ex1 = fmaf(Ex, nf[i3][e][1], fmaf(An, nf[i3][e][1], Ax)), ed1 = fmaf(Ed, nf[i3][e][1], fmaf(An, nf[i3][e][1], Ad));
                        const float ex2 = fmaf(Ex, nf[i3][e][2], fmaf(An, nf[i3][e][2], Ax)), ed2 = fmaf(Ed, nf[i3][e][2], fmaf(An, nf[i3][e][2], Ad));
                        float err = fmaf(fabsf(x0), ed0, fmaf(fabsf(d0), ex0, ex0 * ed0));
                        err += fmaf(fabsf(x1), ed1, fmaf(fabsf(d1), ex1, ex1 * ed1));
                        err += fmaf(fabsf(x2), ed2, fmaf(fabsf(d2), ex2, ex2 * ed2));
                        const float mag = fabsf(x0 * d0) + fabsf(x1 * d1) + fabsf(x2 * d2);
                        err = fmaf(mag, 4.8e-7f, err) * 1.001f + 1e-37f;
                        bad = bad || !(err < INFINITY) || !(fabsf(st) < INFINITY);
                        lmax = fmaxf(lmax, st - err);
                        insert(st + err, 16 * i3 + 4 * lq + e);
                    }
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    const float ol = __shfl_xor(lmax, o), ou1 = __shfl_xor(u1, o), ou2 = __shfl_xor(u2, o), ou3 = __shfl_xor(u3, o);
                    const int oi1 = __shfl_xor(i1, o), oi2 = __shfl_xor(i2, o);
                    const bool ob = __shfl_xor((int)bad, o) != 0;
                    lmax = fmaxf(lmax, ol);
                    insert(ou1, oi1);
                    insert(ou2, oi2);
                    u3 = fmaxf(u3, ou3);
                    bad = bad || ob;
                }
                if (lq == 0) {
                    const int ch = (cn0 >> 1) + wn * 32 + jj * 16 + l16;
                    const size_t slot = (size_t)grp * Cch + ch;
                    float4 o4;
                    o4.x = bad ? INFINITY : lmax; o4.y = bad ? INFINITY : u1; o4.z = bad ? INFINITY : u2; o4.w = bad ? INFINITY : u3;
                    ((float4*)p.pool_val)[slot] = o4;
                    p.pool_idx[slot] = (loc0 + i1) | ((loc0 + i2) << 16);
                }
            }
        }
        if (!more) break;
        w = wnext;              // m0 / n0 already are the next item's, and its first step is in LDS
    }
}
