// vnp_gemm_split_wide_kernel (192 x 256 tile, eight waves, six bf16 products per fp32 product) for a B operand whose three bf16 planes
// ALREADY EXIST in memory — a weight matrix, split once per optimizer step by vnp_split_planes instead of once per tile and K-step by
// every workgroup that uses it.  B's planes go global -> LDS by DMA (global_load_lds_dwordx4, source-side swizzle: the image the
// fragment reads expect), two stages of 48 KB, the next K-step's six instructions per wave issued before this step's MFMAs; A keeps the
// register path (load, split, ds_write) into its single 36 KB stage.  Per K-step a thread splits 3 float4 instead of 7 (66 VALU
// instructions and 9 LDS writes instead of 154 and 21) and holds no B staging registers.
//
// Why: round 3's stamps (profiles/r03_gemm_segments.txt) put the staging segment — the stretch of a K-step in which no MFMA of the
// workgroup runs — at ~1600 of ~8000 cycles; more than half of it is B's share of the split.
// MEASURED (round 5, scripts/experiments/gemm_bp_bench.hip, profiles/r05_gemm_bplanes.txt): bit-identical outputs and statistics
// partials on every shape, and NO gain: 2.29 / 2.26 / 2.14 / 3.77 ms against 2.29 / 2.28 / 2.15 / 3.79 for the kernel that splits B in the loop
// (decoder 256 -> 512, the same with the statistics epilogue, the decoder's input gradient, the encoder's 1024 -> 2048).  Removing 57 % of the
// split arithmetic, 57 % of the LDS writes and all of B's staging registers (235 -> 199 VGPRs) moves nothing: the loop is not bound by its
// staging work.  NOT in the library (kept here as the record of the experiment).
// Same planes (vnp_split_pair on the same fp32 values), same LDS image, same product and K order per output element, same epilogues as
// vnp_gemm_split_wide_kernel: bit-identical results.  Aligned shapes only: M % 192 == 0, N % 256 == 0, K % 32 == 0, 16-byte rows.
#pragma once
#include "gemm_split_wide_kernel.h"
#include "screen_xwide_kernel.h"          // vnp_glds16 / vnp_glds_wait / vnp_lane_now

// fp32 [rows, cols] (row stride ld) -> three bf16 planes [rows, cols] (row stride ldp, plane stride plane): the planes vnp_split_pair forms
__global__ void __launch_bounds__(256) vnp_split_planes_kernel(const float* __restrict__ X, int ld, long long rows, int cols,
                                                               unsigned short* __restrict__ P, int ldp, long long plane)
{
    const long long pairs_per_row = cols / 2;
    const long long n = rows * pairs_per_row;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / pairs_per_row;
        const int c = (int)(i - r * pairs_per_row) * 2;
        const float2 v = *(const float2*)(X + r * ld + c);
        unsigned H, M, L;
        vnp_split_pair(v.x, v.y, H, M, L);
        unsigned* d = (unsigned*)(P + r * ldp + c);
        d[0] = H;
        *(unsigned*)((unsigned short*)d + plane) = M;
        *(unsigned*)((unsigned short*)d + 2 * plane) = L;
    }
}

template <int EPI = 0, int VAR = 0>
__global__ void __launch_bounds__(512) vnp_gemm_split_wide_bp_kernel(const VnpGemmArgs p, const unsigned short* Bp, long long bp_plane, int ldbp)
{
    constexpr int BM = 192, BN = 256, BK = 32;
    constexpr int TM = 6, TN = 4;
    constexpr int AV = BM / 64;
    constexpr int APL = BM * 64, BPL = BN * 64;                 // bytes per plane
    constexpr int BST = 3 * BPL;                                // one B stage
    static_assert(EPI == 0 || EPI == 2, "plain or BatchNorm-statistics epilogue");
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [A: 3 planes][B stage 0: 3 planes][B stage 1]
    char* As = smem;
    char* Bs = smem + 3 * APL;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int nbn = p.N / BN, nbm = p.M / BM;
    const int nwg = nbm * nbn;
    const int total = nwg * p.nsplit * p.nbatch;
    const int nk = p.K / BK;
    if (p.pred && *p.pred == 0) return;

    const int oa = (tid >> 3) * p.lda + (tid & 7) * 4, sa = 64 * p.lda;
    int perm_off[AV];
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

    vnp_f4 ra[AV];
    auto fetch = [&](int fm0, int fbatch, int kt) {
        const float* blkA = p.A + (size_t)fbatch * p.bsa + (size_t)fm0 * p.lda + kt * BK;
#pragma unroll
        for (int j = 0; j < AV; ++j) ra[j] = *(const vnp_f4*)(blkA + (EPI == 2 ? perm_off[j] : oa + j * sa));
    };
    // B by DMA: a wave-instruction fills 16 consecutive LDS rows x 64 bytes of one plane.  Wave w owns B rows 32 w .. 32 w + 31 (two
    // pieces per plane): lane ln writes physical slot ln & 3 of row ln >> 2 of the piece and fetches the logical slot that the image
    // keeps there (vnp_split_row_off: slot q of a row sits at q ^ ((0 - (row >> 2)) & 3), and (row >> 2) & 3 = (ln >> 4) & 3 here).
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned dstB = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + 3 * APL + wave * 32 * 64));
    auto stage_b = [&](int st, int fn0, int kt) {
        const int ln = vnp_lane_now();
        const int rl = ln >> 2, ps = ln & 3;
        const int qsrc = ps ^ ((0 - (rl >> 2)) & 3);
        const unsigned voff = (unsigned)((rl * ldbp + qsrc * 8) * 2);
        const char* src = (const char*)(Bp + (size_t)(fn0 + wave * 32) * ldbp + kt * BK);          // wave-uniform
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                vnp_glds16(src + ((size_t)q * bp_plane + (size_t)j * 16 * ldbp) * 2, voff, st * BST + dstB + q * BPL + j * 1024);
    };
    vnp_u2 pa[AV][3];
    const int wr_a = vnp_split_row_off<BM, true>(tid >> 3, (tid & 7) >> 1) + (tid & 1) * 8;
    auto convert_write = [&]() {
        vnp_split_convert<true, AV, 3, false>(ra, pa);
#pragma unroll
        for (int j = 0; j < AV; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) *(vnp_u2*)(As + wr_a + j * 64 * 64 + q * APL) = pa[j][q];
    };

    const int l16 = lane & 15, lq = lane >> 4;
    int a_rd[TM], b_rd[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_rd[i] = vnp_split_row_off<BM, true>(wm * 96 + i * 16 + l16, lq);
#pragma unroll
    for (int j = 0; j < TN; ++j) b_rd[j] = vnp_split_row_off<BN, true>(wn * 64 + j * 16 + l16, lq);

    int w = blockIdx.x;
    if (w >= total) return;
    set_item(w);
    fetch(m0, batch, kt0);
    stage_b(0, n0, kt0);
    convert_write();
    vnp_glds_wait();
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
        const int cm0 = m0, cn0 = n0, ck1 = kt1, cs = split, cb = batch;

        for (int kt = kt0; kt < ck1; ++kt) {
            // the other B stage was last read in the previous step, whose closing barrier every wave has passed
            if (kt + 1 < ck1) { fetch(cm0, cb, kt + 1); stage_b(stage ^ 1, cn0, kt + 1); }
            else if (more) { set_item(wnext); fetch(m0, batch, kt0); stage_b(stage ^ 1, n0, kt0); }
            {
                const char* Bc = Bs + stage * BST;
                vnp_bf8 b[TN][3];
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        b[j][pl] = __builtin_bit_cast(vnp_bf8, *(const vnp_u4*)(Bc + pl * BPL + b_rd[j]));
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
            if (kt + 1 < ck1 || more) {
                __syncthreads();                     // every wave has read this step's A stage (and this B stage)
                convert_write();                     // the next step's A -> LDS
                vnp_glds_wait();                     // this wave's share of the next step's B has landed ...
                __syncthreads();                     // ... every wave's has, and A is written
                stage ^= 1;
            }
        }

        vnp_split_epilogue<BM, BN, true, VAR, EPI, TM, TN>(p, acc, cm0, cn0, cs, cb, wm, wn, lane);

        if (!more) break;
        w = wnext;               // m0 / n0 / kt0 / kt1 / split / batch already are the next item's, and its first step is in LDS
    }
}
