// vnp_gemm_split_kernel's arithmetic (gemm_split_kernel.h: three exact bf16 planes, six v_mfma_f32_16x16x32_bf16 products per fp32
// product, same tile, same LDS image, same product and K order, same epilogues: bit-identical results) in a 512-thread workgroup whose
// two halves are held in ANTI-PHASE.
//
// Why.  The 256-thread kernel runs two workgroups per CU, i.e. two waves per SIMD, and every K-step of a workgroup is a matrix segment
// (144 MFMAs per wave, 2304 pipe cycles) followed by a staging segment (wait for the prefetched tile, ~220 VALU operations that split it
// into bf16 planes, 30 ds_write_b64, two barriers).  Nothing orders the two workgroups of a CU against each other, and the measured state
// is lockstep: both waves of a SIMD are in their matrix segment together (each gets the pipe every other slot — 55 % of wave time
// "issue-stalled", profiles/r02) and then both are in their staging segment together with the matrix pipe idle: 47-54 % MFMA-busy
// (profiles/r02_bench_summary.json `pipes`), ~7800 cycles per K-step where the pipe needs 4608.
// Here waves 0-3 (half 0) and waves 4-7 (half 1) of ONE workgroup own one tile each — wave w and wave w + 4 share a SIMD
// (MI355X_MICROARCH.md, "Two waves per SIMD") — and alternate roles between workgroup barriers:
//
//     interval A:  half 0  MFMA segment of its K-step   |  half 1  epilogue (at an item end), split + LDS write of its next K-step,
//                                                       |          global loads of the K-step after that
//     interval B:  half 1  MFMA segment                 |  half 0  staging
//
// so the matrix pipe of a SIMD always has exactly one wave feeding it while its partner's VALU / LDS-write / VMEM work issues in the
// gaps (an MFMA 16x16x32 holds the vector issue port for 8 of its 16 cycles).  The global loads of K-step k + 1 are issued at the END
// of the staging segment of step k (not at the start of the matrix segment), so they have a whole round to land; the barriers are raw
// s_barrier behind an lgkmcnt(0) wait — they do not drain the loads in flight.
//
// Work distribution: workgroup b, half h walks the items b + G (2 j + h), G = gridDim.x (so item -> XCD affinity is the 256-thread
// kernel's: w % 8).  Both halves execute the same number of barriers: the number of rounds is the larger K-step count of the two.
#pragma once
#include "gemm_split_kernel.h"

struct VnpSplitItem { int m0, n0, kt0, kt1, split, batch; };

__device__ __forceinline__ void vnp_pp_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DIAG = 1 (scripts/gemm_pp_bench.hip only; the library instantiates DIAG = 0): per-wave s_memtime totals of the segments go to the buffer
// p.bias points to (no bias in that build): [workgroup][wave][8] = staging: epilogue, wait for the prefetched tile, split + LDS write + next
// loads; barrier behind staging; matrix segment; barrier behind it; K-steps; -
template <int BM, int BN, bool AKC, bool BKC, bool FAST, int EPI = 0, int DIAG = 0, int PRIO = 0>
__global__ void __launch_bounds__(512, 2) vnp_gemm_split_pp_kernel(const VnpGemmArgs p)
{
    constexpr int BK = 32;
    constexpr int TM = BM / 32, TN = BN / 32;
    constexpr int AV = BM * BK / 4 / 256, BV = BN * BK / 4 / 256;
    constexpr int APL = BM * 64, BPL = BN * 64;                 // bytes per plane
    constexpr int NPL = EPI == 3 ? 2 : 3;                       // EPI = 3: products (m h), (h m), (h h) only: no l plane
    constexpr int T0 = EPI == 3 ? 3 : 0;
    constexpr int HSZ = 3 * APL + 3 * BPL;                      // LDS of one half
    static_assert(EPI == 0 || (BM == 192 && BN == 128 && AKC && BKC && FAST), "the VN epilogues need the aligned 192x128 NT kernel");
    static_assert(EPI != 4, "the fp16 screening pass has its own kernels");
    __shared__ __attribute__((aligned(16))) char smem[2 * HSZ];

    if (p.pred && *p.pred == 0) return;            // predicated launch (uniform over the grid)

    const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));          // wave-uniform: role state lives in SGPRs
    const int tid = threadIdx.x & 255;             // thread index inside the half: every map below is the 256-thread kernel's
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    char* As = smem + half * HSZ;
    char* Bs = As + 3 * APL;

    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int nwg = nbm * nbn;
    const int total = nwg * p.nsplit * p.nbatch;
    const int nk = (p.K + BK - 1) / BK;
    const int G = gridDim.x;

    const bool va = FAST || (((p.lda & 3) == 0) && ((((uintptr_t)p.A) & 15) == 0) && ((p.bsa & 3) == 0));
    const bool vb = FAST || (((p.ldb & 3) == 0) && ((((uintptr_t)p.B) & 15) == 0) && ((p.bsb & 3) == 0));
    int oa, sa, ob, sb;
    vnp_split_offsets<BM, AKC>(oa, sa, p.lda, tid);
    vnp_split_offsets<BN, BKC>(ob, sb, p.ldb, tid);
    int perm_off[AV];                       // EPI >= 1: element offset of the global row that LDS row tid/8 + 32 j holds
    if (EPI >= 1) {
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            const int rho = (tid >> 3) + 32 * j, hf = rho / 96, within = rho % 96, i = within >> 4, r16 = within & 15;
            perm_off[j] = (hf * 96 + ((i / 3) * 16 + r16) * 3 + i % 3) * p.lda + (tid & 7) * 4;
        }
    }

    auto item_of = [&](int w) {                            // vnp_gemm_split_kernel's XCD-aware walk
        VnpSplitItem it;
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
        it.m0 = row * BM; it.n0 = col * BN;
        it.split = sbi % p.nsplit; it.batch = sbi / p.nsplit;
        it.kt0 = it.split * p.tiles_per_split;
        it.kt1 = min(nk, it.kt0 + p.tiles_per_split);
        return it;
    };

    // K-steps of half h of this workgroup (both halves must run the same number of rounds)
    auto count_steps = [&](int h) {
        int s = 0;
        const int first = blockIdx.x + h * G;
        if (nk % p.tiles_per_split == 0) {
            if (first < total) s = ((total - first + 2 * G - 1) / (2 * G)) * p.tiles_per_split;
        } else
            for (int w = first; w < total; w += 2 * G) { const VnpSplitItem it = item_of(w); s += it.kt1 - it.kt0; }
        return s;
    };
    const int rounds = max(count_steps(0), count_steps(1));

    vnp_f4 ra[AV], rb[BV];
    auto fetch = [&](const VnpSplitItem& it, int kt) {
        const float* pA = p.A + (size_t)it.batch * p.bsa;
        const float* pB = p.B + (size_t)it.batch * p.bsb;
        const int k0 = kt * BK;
        const bool full = FAST || (k0 + BK <= p.K);
        if (EPI >= 1) {
            const float* blk = pA + (size_t)it.m0 * p.lda + k0;
#pragma unroll
            for (int j = 0; j < AV; ++j) ra[j] = *(const vnp_f4*)(blk + perm_off[j]);
        } else if (FAST || (va && full && it.m0 + BM <= p.M))
            vnp_load_tile_fast<AV>(ra, AKC ? pA + (size_t)it.m0 * p.lda + k0 : pA + it.m0 + (size_t)k0 * p.lda, oa, sa);
        else vnp_split_load_generic<BM, AKC, AV>(ra, pA, p.lda, it.m0, p.M, k0, p.K, va, tid);
        if (FAST || (vb && full && it.n0 + BN <= p.N))
            vnp_load_tile_fast<BV>(rb, BKC ? pB + (size_t)it.n0 * p.ldb + k0 : pB + it.n0 + (size_t)k0 * p.ldb, ob, sb);
        else vnp_split_load_generic<BN, BKC, BV>(rb, pB, p.ldb, it.n0, p.N, k0, p.K, vb, tid);
    };

    const int l16 = lane & 15, lq = lane >> 4;
    int a_rd[TM], b_rd[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_rd[i] = vnp_split_row_off<BM, AKC>(wm * (BM / 2) + i * 16 + l16, lq);
#pragma unroll
    for (int j = 0; j < TN; ++j) b_rd[j] = vnp_split_row_off<BN, BKC>(wn * (BN / 2) + j * 16 + l16, lq);

    vnp_f4 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();

    // ---- role state (wave-uniform)
    int w_ld = blockIdx.x + half * G;               // item whose K-step sits in ra / rb
    bool regs_full = w_ld < total;
    VnpSplitItem it_ld{}, it_lds{}, it_epi{};
    int kt_ld = 0;
    bool lds_full = false, lds_last = false, epi_pending = false;
    if (regs_full) { it_ld = item_of(w_ld); kt_ld = it_ld.kt0; fetch(it_ld, kt_ld); }

    // staging segment: epilogue of the item whose last K-step was computed in the previous segment; registers -> planes -> LDS; next loads
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&]() { return DIAG ? __builtin_amdgcn_s_memtime() : 0ull; };
    auto stage = [&]() {
        __builtin_amdgcn_s_setprio(PRIO);
        const unsigned long long s0 = stamp();
        if (epi_pending) {
            vnp_split_epilogue<BM, BN, FAST, DIAG ? 8 : 0, EPI, TM, TN>(p, acc, it_epi.m0, it_epi.n0, it_epi.split, it_epi.batch, wm, wn, lane);
            zero_acc();
            epi_pending = false;
        }
        const unsigned long long s1 = stamp();
        if (DIAG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dg[0] += s1 - s0; const unsigned long long s2 = stamp(); dg[1] += s2 - s1; }
        const unsigned long long s3 = stamp();
        if (regs_full) {
            vnp_u2 pa[AV][3], pb[BV][3];
            vnp_split_convert<AKC, AV, NPL, false>(ra, pa);
            vnp_split_convert<BKC, BV, NPL, false>(rb, pb);
            vnp_split_write<BM, AKC, AV, NPL>(pa, As, tid);
            vnp_split_write<BN, BKC, BV, NPL>(pb, Bs, tid);
            it_lds = it_ld;
            lds_last = kt_ld + 1 == it_ld.kt1;
            lds_full = true;
            if (!lds_last) { ++kt_ld; fetch(it_ld, kt_ld); }
            else {
                w_ld += 2 * G;
                regs_full = w_ld < total;
                if (regs_full) { it_ld = item_of(w_ld); kt_ld = it_ld.kt0; fetch(it_ld, kt_ld); }
            }
        }
        if (DIAG) dg[2] += stamp() - s3;
        __builtin_amdgcn_s_setprio(0);
    };

    // matrix segment: the six products of the K-step in LDS (vnp_gemm_split_kernel's sequence)
    auto compute = [&]() {
        if (!lds_full) return;
        vnp_bf8 b[TN][3];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                b[j][pl] = __builtin_bit_cast(vnp_bf8, *(const vnp_u4*)(Bs + pl * BPL + b_rd[j]));
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
        // A fragments one row-tile ahead: with ONE wave feeding the SIMD's matrix pipe nobody else covers a ds_read's latency, so the
        // reads of row-tile i + 1 are issued before the 24 MFMAs of row-tile i (the 256-thread kernel leaves that to its SIMD partner)
        vnp_bf8 a[2][3];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
            a[0][pl] = __builtin_bit_cast(vnp_bf8, *(const vnp_u4*)(As + pl * APL + a_rd[0]));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i + 1 < TM) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    a[(i + 1) & 1][pl] = __builtin_bit_cast(vnp_bf8, *(const vnp_u4*)(As + pl * APL + a_rd[i + 1]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = T0; t < 6; ++t)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 1][PA[t]], b[j][PB[t]], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_full = false;
        if (lds_last) { epi_pending = true; it_epi = it_lds; }
    };

    // ONE instruction stream for both halves, half 1 one barrier behind: a barrier counts arrivals, not program counters, so
    //   half 0:      S #1 C #2 S #3 C #4 ...          half 1:  #1 S #2 C #3 S #4 C ...
    // puts half 0's matrix segment beside half 1's staging segment between any two barriers, and the other way round in the next interval.
    if (half == 1) vnp_pp_barrier();
    for (int r = 0; r < rounds; ++r) {
        stage();
        const unsigned long long t1 = stamp();
        vnp_pp_barrier();
        const unsigned long long t2 = stamp();
        if (DIAG && lds_full) dg[6] += 1;
        compute();
        const unsigned long long t3 = stamp();
        vnp_pp_barrier();
        if (DIAG) { dg[3] += t2 - t1; dg[4] += t3 - t2; dg[5] += stamp() - t3; }
    }
    if (DIAG && lane == 0) {
        unsigned long long* dbg = (unsigned long long*)p.bias + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) dbg[k] = dg[k];
    }
    if (half == 0) vnp_pp_barrier();                // the barrier half 1 took first
    if (epi_pending)                                // the half's last item
        vnp_split_epilogue<BM, BN, FAST, DIAG ? 8 : 0, EPI, TM, TN>(p, acc, it_epi.m0, it_epi.n0, it_epi.split, it_epi.batch, wm, wn, lane);
}
