// The one-fp16-product screening pass of the exact screened arg-max pool on 384 x 256 tiles and eight waves (4 (M) x 2 (N), 512 threads,
// one workgroup per CU, two waves per SIMD): every wave owns a 96 x 128 tile = 6 x 8 MFMA tiles of 16 x 16 (192 accumulator registers).
// (A 192 x 128 wave tile on four waves — 384 accumulators, one wave per SIMD — halves the fragment reads once more, but the compiler
// keeps MFMA accumulators in ONE register class of 256 and spills 329 registers; it would need hand-assigned AGPR / VGPR tiles.)
//
// Why (round 3's ablations of vnp_screen_f16_wide_kernel, scripts/experiments/screen_bench.hip): that pass is bound by moving operand
// bytes through the CU — global -> registers -> LDS -> fragments — not by the matrix pipe (MFMAs + barriers alone 1.08 ms of 1.85; WITHOUT
// the MFMAs 1.53 ms).  Per MFMA this tile moves
//   * (384 + 256) / (384 x 256) operand rows per output instead of (192 + 256) / (192 x 256): 0.71x the bytes into the CU and into LDS,
//   * 14 ds_read_b128 per 48 MFMAs instead of 10 per 24: 0.7x the fragment reads (a 96 x 128 wave tile re-uses each fragment 6 or 8
//     times instead of 4 or 6),
// and the operands go from global memory STRAIGHT into LDS (global_load_lds_dwordx4: no staging registers — there are none left beside
// 192 accumulators at two waves per SIMD — and no ds_write pass).  An LDS-DMA instruction writes 64 lanes x 16 bytes = 1 KB contiguously,
// i.e. 16 rows of one 32-wide K slice in the [row][64 bytes] image the fragment reads expect; the image's XOR swizzle of the 16-byte slots
// is applied on the SOURCE side (lane -> which slot of its row it fetches).  K is walked in 64-wide steps (two slices) through two LDS
// stages of 80 KB — all 160 KB of the CU — with the next step's DMA issued before this step's MFMAs and retired by the step's barrier.
// Same operands (the fp16 images written by vnp_row_norm_kernel), same LDS image per 32-wide K slice, same MFMA sequence per output
// element (k ascending, one 16x16x32 product per slice) and the same epilogue arithmetic as vnp_screen_f16_wide_kernel /
// vnp_gemm_split_kernel<.., EPI = 4>, so the slots it writes are bit-identical to theirs
// (tests/test_gpu_layers.py::test_screen_pass_tiles_write_the_same_slots).
#pragma once
#include "gemm_split_kernel.h"

// LDS-DMA: 16 bytes per lane from (sbase + voff) to the 64 consecutive 16-byte units at LDS byte address lds_dst (wave-uniform).
// Inline assembly rather than __builtin_amdgcn_global_load_lds: the builtin takes a 64-bit per-lane pointer, and hipcc kept every such
// pointer as a register PAIR across the K loop — beside 192 accumulators they spilled, and the reload (a scratch LOAD, counted by vmcnt)
// made it wait for the DMA issued just before: the steps serialised.  Here a lane holds one 32-bit offset per operand, the base is a
// scalar pair.  M0 (the destination base) is written in the statement that uses it; the compiler does not count this load, so the caller
// waits with vnp_glds_wait() before the barrier that publishes the stage.  (The instruction's immediate offset would move BOTH addresses.)
__device__ __forceinline__ void vnp_glds16(const void* sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void vnp_glds_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// the lane id, opaque to the optimiser: what is derived from it is computed where it is used, not hoisted out of a loop and kept
__device__ __forceinline__ int vnp_lane_now()
{
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}

// VAR: 0 = the kernel; ablations for scripts/experiments/screen_bench.hip (timing only, wrong results): 1 no DMA inside the K loop,
// 2 fragments read once per item, 4 no MFMAs
template <int VAR = 0>
__global__ void __launch_bounds__(512, 1) vnp_screen_f16_xwide_kernel(const VnpGemmArgs p)
{
    constexpr int BM = 384, BN = 256, BK = 64;
    constexpr int TM = 6, TN = 8;                               // 16x16 MFMA tiles per wave: 96 x 128
    constexpr int APL = BM * 64, BPL = BN * 64;                 // bytes of one 32-wide K slice of the tile's operands
    constexpr int STAGE = 2 * APL + 2 * BPL;                    // 80 KB: slices [A k0-31][A k32-63][B k0-31][B k32-63]; two stages = the CU's LDS
    extern __shared__ __attribute__((aligned(16))) char smem[];

    if (p.pred && *p.pred == 0) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // a scalar: what derives from it stays out of the vector registers
    const int wm = wave >> 1, wn = wave & 1;                    // 4 x 2
    const int nbn = p.N / BN, nbm = p.M / BM;
    const int total = nbm * nbn;
    const int nkt = p.K / BK;
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

    // DMA: a wave-instruction fills 16 consecutive LDS rows x 64 bytes of one slice.  Wave w owns A rows 48 w .. 48 w + 47 (three
    // instructions per slice) and B rows 32 w .. 32 w + 31 (two): LDS row 48 w + 16 j + rl of the A image is MFMA row-tile 3 i3 + t with
    // i3 = w & 1, t = j of the 32-point group w >> 1, i.e. global row (w >> 1) * 96 + ((w & 1) * 16 + rl) * 3 + j — the three instructions
    // differ by ONE global row, a scalar offset, so a lane needs one 32-bit source offset for A and one for B.  The image keeps logical
    // slot q of a row at physical slot q ^ ((0 - (row >> 2)) & 3) (vnp_split_row_off); (row >> 2) & 3 = (rl >> 2) & 3 for every row above,
    // and the lane that writes physical slot ps fetches logical slot ps ^ that.
    // (recomputed from the lane id at every use — a handful of VALU operations per K step — instead of living in registers across the
    // MFMA loop, where there is no room for them: see vnp_lane_now)
    auto dma_offsets = [&](unsigned& offA, unsigned& offB) {
        const int ln = vnp_lane_now();
        const int rl = ln >> 2, ps = ln & 3;
        const int qsrc = ps ^ ((0 - (rl >> 2)) & 3);
        offA = (unsigned)((((wave >> 1) * 96 + ((wave & 1) * 16 + rl) * 3) * p.lda + qsrc * 8) * 2);
        offB = (unsigned)(((wave * 32 + rl) * p.ldb + qsrc * 8) * 2);
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned dstA = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + wave * 48 * 64));
    const unsigned dstB = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + 2 * APL + wave * 32 * 64));
    // fragment rows wm * 96 + 16 i + l16 (wn * 128 + 16 j + l16): the swizzle depends on (row >> 2) & 3 = (l16 >> 2) & 3 only, so a
    // lane's offset inside a 16-row block, f_rd, serves every fragment; the block's origin is wave-uniform
    const int a_blk = wm * 96 * 64, b_blk = 2 * APL + wn * 128 * 64;

    auto stage_in = [&](int st, int fm0, int fn0, int kt) {      // 10 DMA instructions per wave: one K step of the tile into stage st
        const char* srcA = (const char*)(A16 + (size_t)fm0 * p.lda + kt * BK);          // wave-uniform
        const char* srcB = (const char*)(B16 + (size_t)fn0 * p.ldb + kt * BK);
        const unsigned S = st * STAGE;
        unsigned offA, offB;
        dma_offsets(offA, offB);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            vnp_glds16(srcA + (size_t)j * p.lda * 2, offA, S + dstA + j * 1024);
            vnp_glds16(srcA + (size_t)j * p.lda * 2 + 64, offA, S + dstA + APL + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            vnp_glds16(srcB + (size_t)j * 16 * p.ldb * 2, offB, S + dstB + j * 1024);
            vnp_glds16(srcB + (size_t)j * 16 * p.ldb * 2 + 64, offB, S + dstB + BPL + j * 1024);
        }
    };

    int w = blockIdx.x;
    if (w >= total) return;
    set_item(w);
    stage_in(0, m0, n0, 0);
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
        const int cm0 = m0, cn0 = n0;
        for (int kt = 0; kt < nkt; ++kt) {
            // the other stage was last read in the previous step, whose barrier every wave has passed: refill it behind this step's MFMAs
            if (kt + 1 < nkt) { if (!(VAR & 1)) stage_in(stage ^ 1, cm0, cn0, kt + 1); }
            else if (more) { set_item(wnext); stage_in(stage ^ 1, m0, n0, 0); }          // the next item's first step
            const int fl = vnp_lane_now();
            const char* S = smem + stage * STAGE + vnp_split_row_off<BM, true>(fl & 15, fl >> 4);
            const int a_rd = a_blk, b_rd = b_blk;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // the six A fragments stay in registers, the eight B fragments stream through (24 + 2 x 4 registers beside 192 accumulators)
                vnp_h8 a[TM];
                const char* F = (VAR & 2) ? smem + vnp_split_row_off<BM, true>(fl & 15, fl >> 4) : S;           // (ablation 2: loop-invariant addresses, the compiler hoists the reads)
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = __builtin_bit_cast(vnp_h8, *(const vnp_u4*)(F + a_rd + s * APL + i * 1024));
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const vnp_h8 b = __builtin_bit_cast(vnp_h8, *(const vnp_u4*)(F + b_rd + s * BPL + j * 1024));
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        if (VAR & 4) acc[i][j][0] += (float)a[i][0] * (float)b[j & 7];
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b, acc[i][j], 0, 0, 0);
                    }
                }
            }
            vnp_glds_wait();                                   // this wave's DMA into the other stage has landed ...
            __syncthreads();                                   // ... every wave's has, and this step's fragment reads are done
            stage ^= 1;
        }

        // ---- epilogue: the bounds of vnp_gemm_split_kernel's screen_epilogue, term for term (see there for the derivation), for the
        // wave's 32-point group and its two 64-column groups (g: 32 x-columns, then their 32 d-columns)
        const int Cch = p.N >> 1;
        {
            // (the tile origin passes through an empty asm statement: everything below depends on it, so the compiler cannot compute the
            // epilogue's per-lane addresses BEFORE the K loop and carry them through it — beside 192 accumulators they were spilled, and
            // every reload is a scratch load whose wait also drains the DMA)
            int em0 = cm0, en0 = cn0;
            asm volatile("" : "+s"(em0), "+s"(en0));
            const int eln = vnp_lane_now(), elq = eln >> 4, el16 = eln & 15;
            const int pt0 = em0 / 3 + wm * 32;
            const int grp = pt0 / 32;
            const int loc0 = pt0 % p.pool_gs;
            const int rbase = em0 + wm * 96;
            float nf[2][4][3];
#pragma unroll
            for (int i3 = 0; i3 < 2; ++i3)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < 3; ++t) nf[i3][e][t] = p.row_norm[rbase + (16 * i3 + 4 * elq + e) * 3 + t];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int colx = en0 + wn * 128 + g * 64 + jj * 16 + el16;
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
                            const int ri = 3 * i3, cx = 4 * g + jj, cd = 4 * g + jj + 2;
                            const float x0 = acc[ri][cx][e], x1 = acc[ri + 1][cx][e], x2 = acc[ri + 2][cx][e];
                            const float d0 = acc[ri][cd][e], d1 = acc[ri + 1][cd][e], d2 = acc[ri + 2][cd][e];
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
                            insert(st + err, 16 * i3 + 4 * elq + e);
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
                    if (elq == 0) {
                        const int ch = (en0 >> 1) + wn * 64 + g * 32 + jj * 16 + el16;
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
