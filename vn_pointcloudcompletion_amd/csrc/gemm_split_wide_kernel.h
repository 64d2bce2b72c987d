// vnp_gemm_split_kernel's arithmetic (three exact bf16 planes, six v_mfma_f32_16x16x32_bf16 products per fp32 product, same LDS image,
// same product and K order per output element, same epilogues: bit-identical results) on a 192 x 256 block tile shared by the EIGHT waves
// of one 512-thread workgroup (2 (M) x 4 (N), each wave 96 x 64 as in the 256-thread kernel), one workgroup per CU.
//
// Why: per MFMA the wide tile stages (192 + 256) / (192 x 256) operand rows instead of (192 + 128) / (192 x 128) — 0.70x the global
// loads, the fp32 -> 3 x bf16 split work and the LDS writes — while the fragment reads per MFMA stay what they are.  Round 3's study of
// the K-step (profiles/r03_gemm_segments.txt) found the pieces of a K-step adding up on top of the MFMA time whatever
// overlaps them on paper; this kernel makes three of the pieces smaller.  One LDS buffer (84 KB), two barriers per K-step, the next
// K-tile's global loads in flight during the MFMAs: the structure of the 256-thread kernel.
//
// Both operands K-contiguous (A [M, K], B [N, K]); aligned shapes only: M % 192 == 0, N % 256 == 0, K % 32 == 0, 16-byte rows.
// EPI = 0 (store / bias / accumulate / split-K slabs) and EPI = 2 (store + BatchNorm statistics).
#pragma once
#include "gemm_split_kernel.h"

template <int EPI = 0, int VAR = 0>
__global__ void __launch_bounds__(512) vnp_gemm_split_wide_kernel(const VnpGemmArgs p)
{
    constexpr int BM = 192, BN = 256, BK = 32;
    constexpr int TM = 6, TN = 4;                               // 16x16 MFMA tiles per wave: 96 x 64
    constexpr int AV = BM / 64, BV = BN / 64;                   // float4 per thread and K-step: row tid / 8 + 64 j, k = 4 (tid % 8)
    constexpr int APL = BM * 64, BPL = BN * 64;                 // bytes per plane
    static_assert(EPI == 0 || EPI == 2, "plain or BatchNorm-statistics epilogue");
    __shared__ __attribute__((aligned(16))) char smem[3 * APL + 3 * BPL];
    char* As = smem;
    char* Bs = smem + 3 * APL;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    const int nbn = p.N / BN, nbm = p.M / BM;
    const int nwg = nbm * nbn;
    const int total = nwg * p.nsplit * p.nbatch;
    const int nk = p.K / BK;
    if (p.pred && *p.pred == 0) return;            // predicated launch (uniform over the grid)

    const int oa = (tid >> 3) * p.lda + (tid & 7) * 4, sa = 64 * p.lda;
    const int ob = (tid >> 3) * p.ldb + (tid & 7) * 4, sb = 64 * p.ldb;
    int perm_off[AV];                       // EPI = 2: element offset of the global row that LDS row tid / 8 + 64 j holds
    if (EPI == 2) {
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            const int rho = (tid >> 3) + 64 * j, half = rho / 96, within = rho % 96, i = within >> 4, r16 = within & 15;
            perm_off[j] = (half * 96 + ((i / 3) * 16 + r16) * 3 + i % 3) * p.lda + (tid & 7) * 4;
        }
    }

    int m0, n0, kt0, kt1, split, batch;
    auto set_item = [&](int w) {                            // vnp_gemm_split_kernel's XCD-aware walk
        int sbi, wg;
        if (p.xcd_split) {
            const int j = w >> 3;
            wg = j % nwg; sbi = (j / nwg) * 8 + (w & 7);
        } else {
            sbi = w / nwg;
            const int t = w - sbi * nwg;
            const int q = nwg >> 3, r = nwg & 7, x = t & 7;
            wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3);
        }
        const int cg = (p.col_group > 0 && p.col_group < nbn) ? p.col_group : nbn;
        const int full = (nbn / cg) * cg * nbm;
        int row, col;
        if (wg < full) { const int g = wg / (nbm * cg), rem = wg - g * (nbm * cg); row = rem / cg; col = g * cg + rem % cg; }
        else { const int wl = nbn - (nbn / cg) * cg, rem = wg - full; row = rem / wl; col = (nbn / cg) * cg + rem % wl; }
        m0 = row * BM; n0 = col * BN;
        split = sbi % p.nsplit; batch = sbi / p.nsplit;
        kt0 = split * p.tiles_per_split;
        kt1 = min(nk, kt0 + p.tiles_per_split);
    };

    vnp_f4 ra[AV], rb[BV];
    auto fetch = [&](int fm0, int fn0, int fbatch, int kt) {
        const float* blkA = p.A + (size_t)fbatch * p.bsa + (size_t)fm0 * p.lda + kt * BK;
        const float* blkB = p.B + (size_t)fbatch * p.bsb + (size_t)fn0 * p.ldb + kt * BK;
#pragma unroll
        for (int j = 0; j < AV; ++j) ra[j] = *(const vnp_f4*)(blkA + (EPI == 2 ? perm_off[j] : oa + j * sa));
#pragma unroll
        for (int j = 0; j < BV; ++j) rb[j] = *(const vnp_f4*)(blkB + ob + j * sb);
    };
    vnp_u2 pa[AV][3], pb[BV][3];
    auto convert = [&]() {
        vnp_split_convert<true, AV, 3, false>(ra, pa);
        vnp_split_convert<true, BV, 3, false>(rb, pb);
    };
    const int wr_a = vnp_split_row_off<BM, true>(tid >> 3, (tid & 7) >> 1) + (tid & 1) * 8;          // rows r + 64 j share the swizzle
    const int wr_b = vnp_split_row_off<BN, true>(tid >> 3, (tid & 7) >> 1) + (tid & 1) * 8;
    auto write = [&]() {
#pragma unroll
        for (int j = 0; j < AV; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) *(vnp_u2*)(As + wr_a + j * 64 * 64 + q * APL) = pa[j][q];
#pragma unroll
        for (int j = 0; j < BV; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) *(vnp_u2*)(Bs + wr_b + j * 64 * 64 + q * BPL) = pb[j][q];
    };

    const int l16 = lane & 15, lq = lane >> 4;
    int a_rd[TM], b_rd[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_rd[i] = vnp_split_row_off<BM, true>(wm * 96 + i * 16 + l16, lq);
#pragma unroll
    for (int j = 0; j < TN; ++j) b_rd[j] = vnp_split_row_off<BN, true>(wn * 64 + j * 16 + l16, lq);

    int w = blockIdx.x;
    if (w >= total) return;
    unsigned long long t_clk = 0, t_real = 0;
    if (VAR & 8) { t_clk = __builtin_amdgcn_s_memtime(); t_real = __builtin_amdgcn_s_memrealtime(); }
    set_item(w);
    fetch(m0, n0, batch, kt0);
    convert();
    write();
    __syncthreads();

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

        for (int kt = kt0; kt < kt1; ++kt) {
            if (kt + 1 < kt1) fetch(m0, n0, batch, kt + 1);
            else if (more) {
                const int cm0 = m0, cn0 = n0, ck0 = kt0, ck1 = kt1, cs = split, cb = batch;
                set_item(wnext);
                fetch(m0, n0, batch, kt0);
                m0 = cm0; n0 = cn0; kt0 = ck0; kt1 = ck1; split = cs; batch = cb;
            }
            {
                vnp_bf8 b[TN][3];
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        b[j][pl] = __builtin_bit_cast(vnp_bf8, *(const vnp_u4*)(Bs + pl * BPL + b_rd[j]));
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
                constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    vnp_bf8 a[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        a[pl] = __builtin_bit_cast(vnp_bf8, *(const vnp_u4*)(As + pl * APL + a_rd[i]));
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[t]], b[j][PB[t]], acc[i][j], 0, 0, 0);
                }
            }
            __syncthreads();
            if (kt + 1 < kt1) {
                convert();
                write();
                __syncthreads();
            }
        }

        vnp_split_epilogue<BM, BN, true, VAR, EPI, TM, TN>(p, acc, m0, n0, split, batch, wm, wn, lane);

        if (!more) break;
        w = wnext;
        set_item(w);
        convert();
        write();                 // the next item's first K-tile (fetched before the last MFMAs) -> LDS
        __syncthreads();
    }
    if ((VAR & 8) && tid == 0) {
        unsigned long long* dbg = (unsigned long long*)p.bias;          // diagnostic build: bias carries the stamp buffer
        dbg[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_clk;
        dbg[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t_real;
    }
}

// ---- the weight-gradient form (both operands ROW-contiguous: A [K, M], B [K, N], C = A^T B) on a 256 x 128 tile shared by the eight waves
// of one 512-thread workgroup (4 (M) x 2 (N), each wave 64 x 64 as in the 256-thread 128 x 128 kernel).  Same arithmetic, LDS image (per
// 128-row block: vnp_split_row_off<128, false>), products and K order per output element: bit-identical results.
// Per K-step a workgroup stages (256 + 128) x 32 elements for 256 x 128 x 32 products — 0.75x the global loads, split arithmetic and LDS
// writes per MFMA of two 128 x 128 workgroups — and the work divides evenly: every thread transposes one 4 (k) x 4 (m) register block of
// A, and HALF a block of B (two of the four k-rows: one packed pair per row and plane, ds_write_b32), instead of one block of each.
// Aligned shapes only: M % 256 == 0, N % 128 == 0, K % 32 == 0, 16-byte rows.  Plain epilogue (store / accumulate / split-K slabs).
template <int VAR = 0>
__global__ void __launch_bounds__(512) vnp_gemm_split_wide_tn_kernel(const VnpGemmArgs p)
{
    constexpr int BM = 256, BN = 128, BK = 32;
    constexpr int TM = 4, TN = 4;                               // 16x16 MFMA tiles per wave: 64 x 64
    constexpr int APL = BM * 64, BPL = BN * 64, TBLK = 128 * 64;   // bytes per plane; one 128-row block inside a plane
    __shared__ __attribute__((aligned(16))) char smem[3 * APL + 3 * BPL];
    char* As = smem;
    char* Bs = smem + 3 * APL;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int nbn = p.N / BN, nbm = p.M / BM;
    const int nwg = nbm * nbn;
    const int total = nwg * p.nsplit * p.nbatch;
    const int nk = p.K / BK;
    if (p.pred && *p.pred == 0) return;

    // staging maps: t = tid & 255 picks the 4 (k) x 4 (row) block of a 128-row panel as in the 256-thread kernel, hb = tid >> 8 picks
    // A's 128-row block (whole blocks) and B's pair of k-rows (half blocks)
    const int t = tid & 255, hb = tid >> 8;
    const int kq = vnp_split_kq(t), m4 = vnp_split_m4(t);
    const int oa = kq * 4 * p.lda + hb * 128 + m4 * 4;
    const int ob = (kq * 4 + 2 * hb) * p.ldb + m4 * 4;

    int m0, n0, kt0, kt1, split, batch;
    auto set_item = [&](int w) {                            // vnp_gemm_split_kernel's walk (no column groups: N / 128 is small here)
        int sbi, wg;
        if (p.xcd_split) {
            const int j = w >> 3;
            wg = j % nwg; sbi = (j / nwg) * 8 + (w & 7);
        } else {
            sbi = w / nwg;
            const int tt = w - sbi * nwg;
            const int q = nwg >> 3, r = nwg & 7, x = tt & 7;
            wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (tt >> 3);
        }
        m0 = (wg / nbn) * BM; n0 = (wg % nbn) * BN;
        split = sbi % p.nsplit; batch = sbi / p.nsplit;
        kt0 = split * p.tiles_per_split;
        kt1 = min(nk, kt0 + p.tiles_per_split);
    };

    vnp_f4 ra[4], rb[2];
    auto fetch = [&](int fm0, int fn0, int fbatch, int kt) {
        const float* blkA = p.A + (size_t)fbatch * p.bsa + fm0 + (size_t)kt * BK * p.lda + oa;
        const float* blkB = p.B + (size_t)fbatch * p.bsb + fn0 + (size_t)kt * BK * p.ldb + ob;
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = *(const vnp_f4*)(blkA + j * p.lda);
#pragma unroll
        for (int j = 0; j < 2; ++j) rb[j] = *(const vnp_f4*)(blkB + j * p.ldb);
    };
    vnp_u2 pa[4][3];
    unsigned pb[4][3];
    auto convert = [&]() {
        vnp_split_convert<false, 4, 3, false>(ra, pa);
#pragma unroll
        for (int e = 0; e < 4; ++e) vnp_split_pair(rb[0][e], rb[1][e], pb[e][0], pb[e][1], pb[e][2]);
    };
    int wa[4], wb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        wa[e] = hb * TBLK + vnp_split_row_off<128, false>(m4 * 4 + e, kq >> 1) + (kq & 1) * 8;
        wb[e] = vnp_split_row_off<128, false>(m4 * 4 + e, kq >> 1) + (kq & 1) * 8 + hb * 4;
    }
    auto write = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                *(vnp_u2*)(As + wa[e] + q * APL) = pa[e][q];
                *(unsigned*)(Bs + wb[e] + q * BPL) = pb[e][q];
            }
    };

    const int l16 = lane & 15, lq = lane >> 4;
    int a_rd[TM], b_rd[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_rd[i] = (wm >> 1) * TBLK + vnp_split_row_off<128, false>((wm & 1) * 64 + i * 16 + l16, lq);
#pragma unroll
    for (int j = 0; j < TN; ++j) b_rd[j] = vnp_split_row_off<128, false>(wn * 64 + j * 16 + l16, lq);

    int w = blockIdx.x;
    if (w >= total) return;
    set_item(w);
    fetch(m0, n0, batch, kt0);
    convert();
    write();
    __syncthreads();

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

        for (int kt = kt0; kt < kt1; ++kt) {
            if (kt + 1 < kt1) fetch(m0, n0, batch, kt + 1);
            else if (more) {
                const int cm0 = m0, cn0 = n0, ck0 = kt0, ck1 = kt1, cs = split, cb = batch;
                set_item(wnext);
                fetch(m0, n0, batch, kt0);
                m0 = cm0; n0 = cn0; kt0 = ck0; kt1 = ck1; split = cs; batch = cb;
            }
            {
                vnp_bf8 b[TN][3];
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        b[j][pl] = __builtin_bit_cast(vnp_bf8, *(const vnp_u4*)(Bs + pl * BPL + b_rd[j]));
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
                constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    vnp_bf8 a[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        a[pl] = __builtin_bit_cast(vnp_bf8, *(const vnp_u4*)(As + pl * APL + a_rd[i]));
#pragma unroll
                    for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[tt]], b[j][PB[tt]], acc[i][j], 0, 0, 0);
                }
            }
            __syncthreads();
            if (kt + 1 < kt1) {
                convert();
                write();
                __syncthreads();
            }
        }

        // wave (wm, wn) owns rows m0 + 64 wm .. + 63: the shared epilogue with a 128-row "half" of 64 rows per wm step
        vnp_split_epilogue<128, BN, true, VAR, 0, TM, TN>(p, acc, m0, n0, split, batch, wm, wn, lane);

        if (!more) break;
        w = wnext;
        set_item(w);
        convert();
        write();
        __syncthreads();
    }
}
