// fp32 GEMM on the bf16 matrix pipe by exact operand splitting (shared by gemm.hip and scripts/gemm_split_bench.hip).
//
// Every fp32 value is x = h + m + l exactly, with h, m, l bf16 (8 significant bits each, round-to-nearest at each
// level: 24 = 3 x 8 bits, and bf16 has fp32's exponent range, so no scaling is needed).  A product of two bf16 values is
// exact in fp32, so   x*y = hh + (hm + mh) + (hl + lh + mm) + (ml + lm) + ll .  The kernel issues the first SIX terms as
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation (smallest first); the dropped terms are bounded by
// |ml + lm + ll| <= 2^-26 |x*y| — four times below the rounding of ONE fp32 multiply-add — so the result is fp32-grade:
// measured against fp64 its error is at or below the fp32-input MFMA kernel's (tests/test_gpu_gemm.py::
// test_split_is_fp32_grade; scripts/gemm_split_bench.hip).  Six bf16 products per fp32 product put the roof at 2500 / 6 = 417 TFLOP/s of
// fp32-equivalent work, against 157 for the fp32-input MFMA.
// Limits (documented, tested): an operand element that is Inf/NaN gives NaN (Inf - Inf in the remainder); |x| above
// bf16's largest finite value 3.39e38 rounds to Inf likewise; fp32 denormal operands lose their low planes.
//
// PREC = 1 (opt-in, vnp_gemm_ex): TWO fp16 planes and THREE products (l h), (h l), (h h) on v_mfma_f32_16x16x32_f16.  fp16 keeps 11
// significant bits, so x s = h + l holds to 22-23 bits: with h = RN16(x s) the remainder x s - h is an exact fp32 number of at most 12
// significant bits and l = RN16(x s - h) drops at most its last one: |x s - h - l| <= 2^-23 |x s| (zero for three operands in four);
// the dropped product obeys |l l'| <= 2^-22 |x y|.  fp16 has 5 exponent bits, so each operand is scaled by a power of two
// s = 2^(14 - floor(log2 max|x|)) (exact; max |x| over the whole operand comes in as a device word, VnpGemmArgs::amax_a / amax_b) and
// the accumulators are multiplied by 1 / (s_a s_b) (exact) before the epilogue: the largest element lands in [2^14, 2^15), elements
// down to 2^-16 of it keep the full 22-23 bits, smaller ones an ABSOLUTE accuracy of 2^-39 max|x| through fp16's subnormals (the MFMA
// honours them: profiles/r02_mfma_f16_subnormals.txt) — so the result is accurate NORMWISE (relative to |A row| |B row|), not
// element by element like the three-plane form.
// The matrix pipe aligns every product to the largest exponent among the addends (C included) and cuts it 2^-26 below
// (scripts/mfma_round_probe.hip, profiles/r02_mfma_rounding.txt).  A bf16 product has 16 bits and never reaches the cut; an fp16
// product has 22 and, accumulated onto a running sum, loses low bits with a negative bias that grows along the chain (measured: 2-4x
// an sgemm's error at K = 2048).  So the three products of a K-tile start from ZERO accumulators and their sum joins the running sum
// through a VALU add (round to nearest): measured against fp64 the result is then at or below torch's CPU sgemm on the same operands
// in the rms normwise measure for K >= 64 (tests/test_gpu_gemm.py::test_fp16x2_is_sgemm_grade).
// Cost: half the MFMA work, two planes through LDS, 6 conversion ops per pair instead of 11, plus 96 VALU adds per K-tile: 12.9 -> 10.7 ms
// over the six big GEMMs of the step (9.2 with one accumulation chain, which is not accurate enough).  NOT the default: operand
// precision is narrowed (22-23 bits), and the two absmax passes per GEMM give the gain back unless producers emit the maxima.
//
// Structure = vnp_gemm_kernel's: persistent 256-thread workgroups (4 waves, 2(M) x 2(N)), 192x128x32 or 128x128x32
// tiles, global -> registers -> LDS with the next K-tile in flight during the MFMAs.  The split happens once per loaded
// element on the way into LDS (11 VALU ops per pair of elements); LDS holds the three bf16 planes of both operands as
// [row][32 k] rows of 64 bytes, 16-byte slots XOR-swizzled so that the ds_read_b128 fragment reads of 16 consecutive
// rows and the ds_write_b64 of the staging pass are both conflict-free (vnp_split_row_off).
// Operands that are not K-contiguous in memory (the weight-gradient form A[K,M], B[K,N]) are loaded as 4(k) x 4(m)
// register blocks and transposed in registers, so LDS always holds K-contiguous rows; their staging pass writes rows
// 4 apart, so that layout also permutes the rows (row -> (row & 3) * R/4 + row / 4) to keep both sides conflict-free.
//
// Measured on MI355X (random operands, in-kernel clock 1.75-1.95 GHz: the chip is power-limited in this loop): 170-205
// TFLOP/s fp32-equivalent on the step's shapes vs 115-135 for vnp_gemm_kernel.  MFMA shape: 16x16x32 runs ~8 % faster than
// 32x32x16 at the same tile per wave (it holds a higher clock), so each wave's 96x64 (64x64) tile is 6x4 (4x4) MFMA tiles.
#pragma once
#include "gemm_kernel.h"

typedef __attribute__((ext_vector_type(8))) __bf16 vnp_bf8;
typedef __attribute__((ext_vector_type(4))) unsigned vnp_u4;
typedef __attribute__((ext_vector_type(2))) unsigned vnp_u2;

typedef __attribute__((ext_vector_type(2))) float vnp_f2;
typedef __attribute__((ext_vector_type(2))) __bf16 vnp_bf2;
typedef __attribute__((ext_vector_type(8))) _Float16 vnp_h8;
typedef __attribute__((ext_vector_type(2))) _Float16 vnp_h2;

// (x0, x1) -> packed fp16 pair, round-to-nearest-even (x0 in the low half-word); |x| > 65504 becomes Inf, |x| < 2^-24 flushes
// through the subnormals: |x - fp16(x)| <= max(2^-11 |x|, 2^-25)
__device__ __forceinline__ unsigned vnp_pack_f16(float x0, float x1)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(vnp_f2{x0, x1}, vnp_h2));
}

// (x0, x1) -> packed bf16 pairs H, M, L with x = h + m + l exactly (x0 in the low half-word: the lower k); each level
// rounds to nearest-even and the remainders x - h, (x - h) - m are exact fp32 subtractions.  11 VALU ops per pair.
// (Measured alternatives, no gain: forming the remainders with v_dot2c_f32_bf16 (7 ops; its (-1, 0) selector must live in a
// register — as an immediate the assembler emits the 32-bit inline constant -1.0, which the packed operand reads as
// (0, -1)); splitting the prefetched tile between the two halves of the MFMA block instead of after it (spills).)
__device__ __forceinline__ void vnp_split_pair(float x0, float x1, unsigned& H, unsigned& M, unsigned& L)
{
    H = __builtin_bit_cast(unsigned, __builtin_convertvector(vnp_f2{x0, x1}, vnp_bf2));        // v_cvt_pk_bf16_f32 (RNE)
    const float r0 = x0 - __uint_as_float(H << 16);
    const float r1 = x1 - __uint_as_float(H & 0xffff0000u);
    M = __builtin_bit_cast(unsigned, __builtin_convertvector(vnp_f2{r0, r1}, vnp_bf2));
    const float s0 = r0 - __uint_as_float(M << 16);
    const float s1 = r1 - __uint_as_float(M & 0xffff0000u);
    L = __builtin_bit_cast(unsigned, __builtin_convertvector(vnp_f2{s0, s1}, vnp_bf2));
}

// (x0, x1) * s -> packed fp16 pairs H, L with x s = h + l to 2^-23 relative (see PREC = 1 above); s is a power of two
__device__ __forceinline__ void vnp_split_pair_h2(float x0, float x1, float s, unsigned& H, unsigned& L)
{
    const float y0 = x0 * s, y1 = x1 * s;
    const vnp_h2 h = __builtin_convertvector(vnp_f2{y0, y1}, vnp_h2);                           // v_cvt_pk_f16_f32 (RNE)
    H = __builtin_bit_cast(unsigned, h);
    L = vnp_pack_f16(y0 - (float)h[0], y1 - (float)h[1]);
}

// operand scale of the PREC = 1 split from the bit pattern of max |x|: s = 2^(14 - floor(log2 max)), inv = 1 / s; both stay normal
// fp32 numbers for every input word (zero, subnormal, Inf and NaN words included)
__device__ __forceinline__ void vnp_h2_scale(unsigned amax_bits, float& s, float& inv)
{
    int es = 268 - (int)((amax_bits >> 23) & 0xffu);
    es = es < 1 ? 1 : es > 253 ? 253 : es;
    s = __uint_as_float((unsigned)es << 23);
    inv = __uint_as_float((unsigned)(254 - es) << 23);
}

// byte offset of (row, 16-byte slot q) inside one plane of R rows.  One LDS cycle of a ds_read_b128 serves 16 lanes, and they cover all
// 64 banks when their 16-byte slot numbers 4 * (physical row & 3) + (q ^ g) are distinct.  Those 16 lanes are NOT consecutive ones: the
// hardware's groups are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63} (MI355X_MICROARCH.md,
// LDS), i.e. for a fragment read (lane l: row l & 15, slot l >> 4) rows 0-3 and 12-15 at slot q together with rows 4-11 at slot q ^ 1.
//   KC layout: physical row = row, g = (-(row >> 2)) & 3 = 0 3 2 1 for the row quads 0 1 2 3: the four quads of a group land on
//              q ^ 0, (q ^ 1) ^ 3, (q ^ 1) ^ 2, q ^ 1 — distinct.  (g = (row >> 2) & 3, the rounds 1-2 choice, is conflict-free for 16
//              CONSECUTIVE lanes and two-way conflicted on the real groups: SQ_LDS_BANK_CONFLICT was a third of SQ_LDS_IDX_ACTIVE,
//              profiles/r03_gemm_pmc.txt.)  Writers (ds_write_b64, 16 consecutive lanes): two whole rows, any g.
//   T  layout: physical row = (row & 3) * R/4 + (row >> 2), g = (row & 3) ^ ((row >> 3) & 3): the row quad picks the bank quarter and
//              the slot varies inside a quad, so the fragment reads (each row of a 16-row tile once) are conflict-free for any such g.
//              Writers: a ds_write_b64 is served in groups of 16 CONTIGUOUS lanes over 32 banks (128 bytes), so the 16 lanes of a group
//              must hit 16 distinct 8-byte units mod 128: with the thread map of vnp_split_m4 / vnp_split_kq a group holds 8
//              consecutive m4 (rows 4 m4 + e: two 64-byte row parities x four slots through g's (row >> 3) term) x both 8-byte halves
//              of a slot (kq & 1).  Rounds 1-3 had 16 consecutive m4 per group with ONE half and g from (row >> 4): every store
//              two-way conflicted (SQ_LDS_BANK_CONFLICT a third of SQ_LDS_IDX_ACTIVE on the weight-gradient kernels).
template <int R, bool KC>
__device__ __forceinline__ int vnp_split_row_off(int row, int q)
{
    if (KC) return row * 64 + ((q ^ ((0 - (row >> 2)) & 3)) << 4);
    return ((row & 3) * (R / 4) + (row >> 2)) * 64 + ((q ^ (row & 3) ^ ((row >> 3) & 3)) << 4);
}

// thread -> tile element maps.  KC (K-contiguous operand): float4 j of thread tid is row tid/8 + 32 j, k = 4 (tid % 8).
// !KC (row-contiguous operand, R = 128 rows): float4 j is k = 4 kq + j, rows 4 m4 .. 4 m4 + 3, where a wave covers
// 16 m4 x 4 kq (256-byte runs in memory; every group of 16 contiguous lanes 16 distinct 8-byte LDS units per ds_write_b64).
// tid bit 0: kq bit 0 (which 8-byte half of a 16-byte slot); bits 1-4: m4 bits 0-3; bit 5: kq bit 1; bit 6: m4 bit 4; bit 7: kq bit 2
__device__ __forceinline__ int vnp_split_m4(int tid) { return ((tid >> 1) & 15) + 16 * ((tid >> 6) & 1); }
__device__ __forceinline__ int vnp_split_kq(int tid) { return (tid & 1) + 2 * ((tid >> 5) & 1) + 4 * (tid >> 7); }

template <int R, bool KC>
__device__ __forceinline__ void vnp_split_offsets(int& voff0, int& step, int ld, int tid)
{
    static_assert(KC || R == 128, "row-contiguous operands use 128-row tiles");
    if (KC) { voff0 = (tid >> 3) * ld + (tid & 7) * 4; step = 32 * ld; }
    else    { voff0 = vnp_split_kq(tid) * 4 * ld + vnp_split_m4(tid) * 4; step = ld; }
}

template <int R, bool KC, int NV>
__device__ __forceinline__ void vnp_split_load_generic(vnp_f4 (&reg)[NV], const float* __restrict__ G, int ld,
                                                       int r0, int rmax, int k0, int kmax, bool vec_ok, int tid)
{
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        int r, k;
        if (KC) { r = (tid >> 3) + 32 * j; k = (tid & 7) * 4; }
        else    { k = vnp_split_kq(tid) * 4 + j;  r = vnp_split_m4(tid) * 4; }
        const int gr = r0 + r, gk = k0 + k;
        vnp_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (gr < rmax && gk < kmax) {
            if (KC) {
                const float* src = G + (size_t)gr * ld + gk;
                if (vec_ok && gk + 3 < kmax) v = *(const vnp_f4*)src;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gk + e < kmax) v[e] = src[e];
                }
            } else {
                const float* src = G + (size_t)gk * ld + gr;
                if (vec_ok && gr + 3 < rmax) v = *(const vnp_f4*)src;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gr + e < rmax) v[e] = src[e];
                }
            }
        }
        reg[j] = v;
    }
}

// fp32 registers -> packed bf16 plane registers: pl[u][plane] holds 4 consecutive k of one row (u = float4 index j for a
// K-contiguous operand, row-in-block e for a row-contiguous one, whose 4(k) x 4(row) block is transposed here)
template <bool KC, int NV, int NPL = 3, bool F16 = false>
__device__ __forceinline__ void vnp_split_convert(const vnp_f4 (&reg)[NV], vnp_u2 (&pl)[NV][3], float scale = 1.f)
{
    static_assert(KC || NV == 4, "row-contiguous operands: 4 k-rows x 4 rows per thread");
    static_assert(!F16 || NPL == 2 || (KC && NPL == 1), "fp16: one plane (screening pass, K-contiguous operands) or two (PREC = 1)");
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if (F16 && NPL == 1) {                      // one fp16 plane: the single-product screening pass
            pl[u][0] = vnp_u2{vnp_pack_f16(reg[u][0], reg[u][1]), vnp_pack_f16(reg[u][2], reg[u][3])};
            continue;
        }
        if (F16) {                                  // two fp16 planes of the scaled operand (PREC = 1)
            unsigned h0, l0, h1, l1;
            if (KC) {
                vnp_split_pair_h2(reg[u][0], reg[u][1], scale, h0, l0);
                vnp_split_pair_h2(reg[u][2], reg[u][3], scale, h1, l1);
            } else {
                vnp_split_pair_h2(reg[0][u], reg[1][u], scale, h0, l0);
                vnp_split_pair_h2(reg[2][u], reg[3][u], scale, h1, l1);
            }
            pl[u][0] = vnp_u2{h0, h1};
            pl[u][1] = vnp_u2{l0, l1};
            continue;
        }
        unsigned h0, m0, l0, h1, m1, l1;
        if (KC) {
            vnp_split_pair(reg[u][0], reg[u][1], h0, m0, l0);
            vnp_split_pair(reg[u][2], reg[u][3], h1, m1, l1);
        } else {
            vnp_split_pair(reg[0][u], reg[1][u], h0, m0, l0);
            vnp_split_pair(reg[2][u], reg[3][u], h1, m1, l1);
        }
        pl[u][0] = vnp_u2{h0, h1};
        if (NPL >= 2) pl[u][1] = vnp_u2{m0, m1};
        if (NPL == 3) pl[u][2] = vnp_u2{l0, l1};          // NPL = 2 (screening pass): the l plane is dead code and is never formed
    }
}

// one UNIT of vnp_split_convert's three-plane form: elements (2 h, 2 h + 1) of row u >> 1 (h = u & 1) -> word h of that row's h / m / l
// plane registers; the units 0 .. 2 NV - 1 together are vnp_split_convert<KC, NV, NPL> (vnp_gemm_split_il_kernel spreads them over its MFMAs)
template <bool KC, int NV, int NPL>
__device__ __forceinline__ void vnp_split_unit(const vnp_f4 (&reg)[NV], vnp_u2 (&pl)[NV][3], int u)
{
    static_assert(KC || NV == 4, "row-contiguous operands: 4 k-rows x 4 rows per thread");
    const int r = u >> 1, h = u & 1;
    unsigned H, M, L;
    if (KC) vnp_split_pair(reg[r][2 * h], reg[r][2 * h + 1], H, M, L);
    else    vnp_split_pair(reg[2 * h][r], reg[2 * h + 1][r], H, M, L);
    pl[r][0][h] = H;
    if (NPL >= 2) pl[r][1][h] = M;
    if (NPL == 3) pl[r][2][h] = L;
}

// plane registers -> LDS.  plane stride = R * 64 bytes.
template <int R, bool KC, int NV, int NPL = 3>
__device__ __forceinline__ void vnp_split_write(const vnp_u2 (&pl)[NV][3], char* S, int tid)
{
    if (KC) {
        const int r = tid >> 3, kc = tid & 7;
        const int base = vnp_split_row_off<R, true>(r, kc >> 1) + (kc & 1) * 8;      // rows r + 32 j share the swizzle
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) *(vnp_u2*)(S + base + j * 32 * 64 + q * R * 64) = pl[j][q];
    } else {
        const int kq = vnp_split_kq(tid), r4 = vnp_split_m4(tid) * 4;       // k = 4 kq .. 4 kq + 3 -> slot kq >> 1, half kq & 1
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            char* d = S + vnp_split_row_off<R, false>(r4 + e, kq >> 1) + (kq & 1) * 8;
#pragma unroll
            for (int q = 0; q < NPL; ++q) *(vnp_u2*)(d + q * R * 64) = pl[e][q];
        }
    }
}

// ---- epilogue of the screening passes (EPI = 3, 4), shared by vnp_gemm_split_kernel, its fp16 pass and vnp_gemm_split_pp_kernel
template <int TM, int TN>
__device__ __forceinline__ void vnp_split_screen_epilogue(const VnpGemmArgs& p, vnp_f4 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int lane)
{
    const int l16 = lane & 15, lq = lane >> 4;
    // Screening pass of the exact screened arg-max pool.  acc holds the 3-product values x~, d~ in the layout of EPI = 1.
    // For every (point, channel): s~ = <x~, d~> and a rigorous bound err >= |s6 - s~| on its distance to the value s6 the
    // 6-product kernels compute for the same pair:  |x6_t - x~_t| <= E |W_c| |f_(n,t)| =: ex_t (Cauchy-Schwarz on the
    // dropped products and on both accumulations' fp32 rounding; E = p.screen_eps, norms = upper bounds), same for d, so
    //   |s6 - s~| <= sum_t (|x~_t| ed_t + |d~_t| ex_t + ex_t ed_t) + 2^-21 sum_t |x~_t d~_t|   (last term: the two fmaf chains).
    // Per 32-point block: the largest lower bound, the two largest upper bounds with their points, and the third-largest
    // upper bound (three or more candidates in one block make the whole block a candidate).  A non-finite bound poisons the slot with +Inf, which makes the selection pass raise the fallback flag.
    const int pt0 = m0 / 3 + wm * 32;
    const int grp = pt0 / 32;
    const int loc0 = pt0 % p.pool_gs;
    const int Cch = p.N >> 1;
    const int rbase = m0 + wm * 96;
    float nf[2][4][3];
#pragma unroll
    for (int i3 = 0; i3 < 2; ++i3)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 3; ++t) nf[i3][e][t] = p.row_norm[rbase + (16 * i3 + 4 * lq + e) * 3 + t];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int colx = n0 + wn * 64 + jj * 16 + l16;
        const float cnx = p.col_norm[colx], cnd = p.col_norm[colx + 32];
        const float Ex = p.screen_eps * cnx, Ed = p.screen_eps * cnd;
        // fp16 pass only: operands below fp16's normal range lose absolute, not relative, accuracy: + A (|W_c| + |f|) per output
        const float Ax = p.screen_abs * cnx, Ad = p.screen_abs * cnd, An = p.screen_abs;
        float lmax = -INFINITY, u1 = -INFINITY, u2 = -INFINITY, u3 = -INFINITY;     // the three largest upper bounds of the block
        int i1 = 0, i2 = 0;                                                        // and the points of the first two
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
                float err = 0.f, mag = 0.f;
                {
                    const float ex0 = fmaf(Ex, nf[i3][e][0], fmaf(An, nf[i3][e][0], Ax)), ed0 = fmaf(Ed, nf[i3][e][0], fmaf(An, nf[i3][e][0], Ad));
                    const float ex1 = fmaf(Ex, nf[i3][e][1], fmaf(An, nf[i3][e][1], Ax)), ed1 = fmaf(Ed, nf[i3][e][1], fmaf(An, nf[i3][e][1], Ad));
                    const float ex2 = fmaf(Ex, nf[i3][e][2], fmaf(An, nf[i3][e][2], Ax)), ed2 = fmaf(Ed, nf[i3][e][2], fmaf(An, nf[i3][e][2], Ad));
                    err = fmaf(fabsf(x0), ed0, fmaf(fabsf(d0), ex0, ex0 * ed0));
                    err += fmaf(fabsf(x1), ed1, fmaf(fabsf(d1), ex1, ex1 * ed1));
                    err += fmaf(fabsf(x2), ed2, fmaf(fabsf(d2), ex2, ex2 * ed2));
                    mag = fabsf(x0 * d0) + fabsf(x1 * d1) + fabsf(x2 * d2);
                }
                err = fmaf(mag, 4.8e-7f, err) * 1.001f + 1e-37f;        // + the fmaf chains' rounding; inflated for its own rounding
                bad = bad || !(err < INFINITY) || !(fabsf(st) < INFINITY);
                const float lo = st - err, up = st + err;
                lmax = fmaxf(lmax, lo);
                insert(up, 16 * i3 + 4 * lq + e);
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
            const int ch = (n0 >> 1) + wn * 32 + jj * 16 + l16;
            const size_t slot = (size_t)grp * Cch + ch;
            float4 o4;
            o4.x = bad ? INFINITY : lmax; o4.y = bad ? INFINITY : u1; o4.z = bad ? INFINITY : u2; o4.w = bad ? INFINITY : u3;
            ((float4*)p.pool_val)[slot] = o4;
            p.pool_idx[slot] = (loc0 + i1) | ((loc0 + i2) << 16);          // pool_gs <= 2048: 16 bits each
        }
    }
}

// ---- epilogues of the six-product kernels (vnp_gemm_split_kernel and vnp_gemm_split_pp_kernel): EPI = 0 plain store (+ bias, accumulate,
// split-K slabs), 1 / 5 Vector-Neuron arg-max pool, 2 store + BatchNorm statistics, 3 screening bounds.  acc is the wave's TM x TN grid of
// 16x16 MFMA tiles; (wm, wn) the wave's position in the wave grid of the BM x BN block tile at (m0, n0): two waves along M, BN / (16 TN)
// along N (the pool / screening epilogues: 2 x 2 only).
template <int BM, int BN, bool FAST, int VAR, int EPI, int TM, int TN>
__device__ __forceinline__ void vnp_split_epilogue(const VnpGemmArgs& p, vnp_f4 (&acc)[TM][TN], int m0, int n0, int split, int batch,
                                                   int wm, int wn, int lane)
{
    const int l16 = lane & 15, lq = lane >> 4;
    if (EPI == 1 || EPI == 5) {
        // acc[3 i3 + t][j][e]: component t of wave-local point 16 i3 + 4 lq + e; j = 0, 1: x channels, j = 2, 3: d channels
        const int pt0 = m0 / 3 + wm * 32;                         // first point of this wave (global point index)
        const int grp = pt0 / 32;                                  // partial slot: one per 32 points
        const int loc0 = pt0 % p.pool_gs;                          // its index inside the pooling group
        const int Cch = p.N >> 1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            float bv = 0.f, bx0 = 0.f, bx1 = 0.f, bx2 = 0.f;
            int bi = 0;
#pragma unroll
            for (int i3 = 0; i3 < 2; ++i3)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = acc[3 * i3][jj][e], x1 = acc[3 * i3 + 1][jj][e], x2 = acc[3 * i3 + 2][jj][e];
                    const float d0 = acc[3 * i3][jj + 2][e], d1 = acc[3 * i3 + 1][jj + 2][e], d2 = acc[3 * i3 + 2][jj + 2][e];
                    const float dot = fmaf(x2, d2, fmaf(x1, d1, x0 * d0));
                    if ((i3 == 0 && e == 0) || dot > bv) { bv = dot; bi = 16 * i3 + 4 * lq + e; bx0 = x0; bx1 = x1; bx2 = x2; }
                    if (EPI == 5) {
                        // x = W f is needed by another consumer as well: store it (rows de-permuted as in EPI = 2: MFMA row-tile
                        // 3 i3 + t, row r  <->  component t of wave-local point 16 i3 + r); d is still never written
                        float* dst = p.C + (size_t)(m0 + wm * 96 + (16 * i3 + 4 * lq + e) * 3) * p.ldc + (n0 >> 1) + wn * 32 + jj * 16 + l16;
                        __builtin_nontemporal_store(x0, dst);
                        __builtin_nontemporal_store(x1, dst + p.ldc);
                        __builtin_nontemporal_store(x2, dst + 2 * p.ldc);
                    }
                }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {                    // the other lane groups hold the other points
                const float ov = __shfl_xor(bv, o), o0 = __shfl_xor(bx0, o), o1 = __shfl_xor(bx1, o), o2 = __shfl_xor(bx2, o);
                const int oi = __shfl_xor(bi, o);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; bx0 = o0; bx1 = o1; bx2 = o2; }
            }
            if (lq == 0) {
                const int ch = (n0 >> 1) + wn * 32 + jj * 16 + l16;
                const size_t slot = (size_t)grp * Cch + ch;
                p.pool_val[slot] = bv;
                p.pool_idx[slot] = loc0 + bi;
                p.pool_x[3 * slot + 0] = bx0; p.pool_x[3 * slot + 1] = bx1; p.pool_x[3 * slot + 2] = bx2;
            }
        }
    } else if (EPI == 3) {
        vnp_split_screen_epilogue<TM, TN>(p, acc, m0, n0, wm, wn, lane);
    } else if (EPI == 2) {
        // C store under the row permutation of EPI = 1 (MFMA row-tile 3 i3 + t, row r  <->  component t of wave-local point
        // 16 i3 + r) plus the BatchNorm statistics of the p columns: every lane holds whole 3-vectors, so n = |x| + eps is
        // formed in registers; Welford over the lane's 8 points, Chan merges across the 4 lane groups -> 32 points per wave.
        float* Cz = p.C;
        int g0 = 0, r0 = 0;
        const bool has_bias = p.bias != nullptr;
        const bool bias_fast = has_bias && p.rows_per_group >= BM;
        if (has_bias) { g0 = m0 / p.rows_per_group; r0 = m0 - g0 * p.rows_per_group; }
        float wmean[TN], wm2[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { wmean[j] = 0.f; wm2[j] = 0.f; }
#pragma unroll
        for (int i3 = 0; i3 < 2; ++i3)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int lr = wm * 96 + (16 * i3 + 4 * lq + e) * 3;         // tile-local row of component 0
                const float* b0 = nullptr; const float* b1 = nullptr; const float* b2 = nullptr;
                if (has_bias) {
                    const int grp = bias_fast ? g0 + ((r0 + lr) >= p.rows_per_group ? 1 : 0) : (m0 + lr) / p.rows_per_group;
                    b0 = p.bias + (size_t)(grp * 3) * p.ldbias; b1 = b0 + p.ldbias; b2 = b1 + p.ldbias;
                }
                float* d0 = Cz + (size_t)(m0 + lr) * p.ldc;
                float xs[3][TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * (TN * 16) + j * 16 + l16;
                    float x0 = acc[3 * i3][j][e], x1 = acc[3 * i3 + 1][j][e], x2 = acc[3 * i3 + 2][j][e];
                    if (has_bias) { x0 += b0[col]; x1 += b1[col]; x2 += b2[col]; }
                    xs[0][j] = x0; xs[1][j] = x1; xs[2][j] = x2;
                    // statistics only (the stored x is untouched): hardware sqrt (1 ulp) and the reciprocal of the running count — a
                    // compile-time constant 1 / 1 .. 1 / 8 after unrolling — instead of IEEE sqrtf and a division per point and column:
                    // the epilogue was ~700 VALU instructions per lane and tile, on the critical path of the one workgroup per CU
                    const float nrm = __builtin_amdgcn_sqrtf(x0 * x0 + x1 * x1 + x2 * x2) + VNP_EPS;
                    const float delta = nrm - wmean[j];
                    wmean[j] += delta * (1.0f / (float)(4 * i3 + e + 1));
                    wm2[j] += delta * (nrm - wmean[j]);
                }
                // stores row by row, the TN segments of a row back to back (as the plain epilogue does): with the three rows of a point
                // interleaved per segment the two halves of a 128-byte line arrived three stores apart, and the launch wrote 1.23x its
                // output through the fabric (profiles/r03_bench_summary.json: 3.97 GB for 3.22)
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        __builtin_nontemporal_store(xs[t][j], d0 + t * p.ldc + n0 + wn * (TN * 16) + j * 16 + l16);
            }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            // Chan merge with the partner lane group, twice (8 -> 16 -> 32 points); both sides hold equal counts
            float fm = wmean[j], fM = wm2[j];
            double cnt = 8.0;
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const double ma = (double)fm, Ma = (double)fM;
                const double mb = (double)__shfl_xor(fm, o), Mb = (double)__shfl_xor(fM, o);
                const double delta = mb - ma;
                fm = (float)(ma + delta * 0.5);
                fM = (float)(Ma + Mb + delta * delta * cnt * 0.5);
                cnt *= 2.0;
            }
            const int col = n0 + wn * (TN * 16) + j * 16 + l16;
            if (lq == 0 && col < p.stats_cols)
                p.stats_part[(size_t)(m0 / 96 + wm) * p.stats_cols + col] = make_float2(fm, fM);
        }
    } else
    // epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
    {
        float* Cz = p.C + (size_t)split * p.slab + (size_t)batch * p.bsc;
        const bool full = FAST || ((m0 + BM <= p.M) && (n0 + BN <= p.N));
        int g0 = 0, r0 = 0;
        const bool has_bias = !(VAR & 8) && p.bias;
        const bool bias_fast = has_bias && p.rows_per_group >= BM;
        if (has_bias) { g0 = m0 / p.rows_per_group; r0 = m0 - g0 * p.rows_per_group; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int lr = wm * (BM / 2) + i * 16 + 4 * lq + e;
                const int row = m0 + lr;
                if (!full && row >= p.M) continue;
                const float* brow = nullptr;
                if (has_bias) {
                    int grp;
                    if (bias_fast) grp = g0 + ((r0 + lr) >= p.rows_per_group ? 1 : 0);
                    else grp = row / p.rows_per_group;
                    brow = p.bias + (size_t)(grp * 3 + row % 3) * p.ldbias;
                }
                float* drow = Cz + (size_t)row * p.ldc;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * (TN * 16) + j * 16 + l16;
                    if (!full && col >= p.N) continue;
                    float v = acc[i][j][e];
                    if (brow) v += brow[col];
                    if (p.accumulate) v += (p.resid ? p.resid + (size_t)row * p.ldr : (const float*)drow)[col];
                    if ((VAR & 16) && v != 12345.678f) continue;          // diagnostic build: what the output stores cost
                    // activation-sized outputs are streamed (non-temporal: they do not fit any cache and would only evict the
                    // A / B panels); split-K slabs are re-read at once by the slab reduction and stay cached
                    if (p.nsplit == 1) __builtin_nontemporal_store(v, &drow[col]);
                    else drow[col] = v;
                }
            }
        }
    }
}

// VAR & 8: diagnostic build (clock stamps, scripts/gemm_split_bench.hip); the library instantiates VAR = 0 only.
// EPI = 1 (192x128 NT FAST only): Vector-Neuron arg-max pool fused into the epilogue.  A's rows are (point, t) triples and B's
// rows are interleaved per 32 channels as [32 rows of W | 32 rows of W'] (x = W f, d = W' f), so a wave's 64 columns are
// 32 channels of x and the same 32 channels of d.  The A tile is loaded under a row permutation that puts component t of
// point (16 i3 + r) into MFMA row-tile 3 i3 + t, row r: every lane then holds whole 3-vectors of x and d for 8 points x 2
// channels, forms <x, d> in registers and reduces the arg-max over the wave's 32 points.  Neither x nor d is ever written
// (EPI = 5: the same, but x IS stored to C [M, N/2] for the tensor's other consumer).
template <int BM, int BN, bool AKC, bool BKC, bool FAST, int VAR = 0, int EPI = 0, int PREC = 0>
__global__ void __launch_bounds__(256, 2) vnp_gemm_split_kernel(const VnpGemmArgs p)
{
    static_assert(PREC == 0 || (FAST && (EPI == 0 || EPI == 2)), "PREC = 1: aligned shapes, plain or BatchNorm-statistics epilogue");
    constexpr int BK = 32;
    constexpr int TM = BM / 32, TN = BN / 32;          // 16x16 MFMA tiles per wave: (BM/2)/16 x (BN/2)/16
    constexpr int AV = BM * BK / 4 / 256, BV = BN * BK / 4 / 256;
    constexpr int APL = BM * 64, BPL = BN * 64;                 // bytes per plane
    // EPI = 4 stages two 32-wide half-tiles of ONE fp16 plane (41 KB; three workgroups per CU would fit, but at 168 VGPRs the loop spills:
    // 7.7 ms instead of 2.0); everything else up to three planes
    constexpr int ASZ = EPI == 4 ? 2 * APL + 128 : 3 * APL, BSZ = EPI == 4 ? 2 * BPL + 128 : 3 * BPL;
    __shared__ __attribute__((aligned(16))) char smem[ASZ + BSZ];
    char* As = smem;
    char* Bs = smem + ASZ;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int nwg = nbm * nbn;
    const int total = nwg * p.nsplit * p.nbatch;
    const int nk = (p.K + BK - 1) / BK;

    const bool va = FAST || (((p.lda & 3) == 0) && ((((uintptr_t)p.A) & 15) == 0) && ((p.bsa & 3) == 0));
    const bool vb = FAST || (((p.ldb & 3) == 0) && ((((uintptr_t)p.B) & 15) == 0) && ((p.bsb & 3) == 0));
    int oa, sa, ob, sb;
    vnp_split_offsets<BM, AKC>(oa, sa, p.lda, tid);
    vnp_split_offsets<BN, BKC>(ob, sb, p.ldb, tid);
    static_assert(EPI == 0 || (BM == 192 && BN == 128 && AKC && BKC && FAST), "the VN epilogues need the aligned 192x128 NT kernel");
    // EPI = 3: bf16 screening pass, products (m h), (h m), (h h): the l plane is never formed, staged or read.
    // EPI = 4: fp16 screening pass: ONE plane (fp16, 11 significant bits) and ONE product per operand pair.
    constexpr bool F16 = EPI == 4 || PREC == 1;
    constexpr int NPL = (EPI == 3 || PREC == 1) ? 2 : EPI == 4 ? 1 : 3;          // planes staged through LDS
    constexpr int T0 = (EPI == 3 || PREC == 1) ? 3 : EPI == 4 ? 5 : 0;           // first of the six products issued
    float sca = 1.f, scb = 1.f, isa = 1.f, isb = 1.f;              // PREC = 1: operand scales and their inverses (powers of two)
    if (PREC == 1) { vnp_h2_scale(*p.amax_a, sca, isa); vnp_h2_scale(*p.amax_b, scb, isb); }
    if (p.pred && *p.pred == 0) return;            // predicated launch (wave-uniform)
    int perm_off[AV];                       // EPI >= 1: element offset of the global row that LDS row tid/8 + 32 j holds
    if (EPI >= 1) {
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            const int rho = (tid >> 3) + 32 * j, half = rho / 96, within = rho % 96, i = within >> 4, r16 = within & 15;
            perm_off[j] = (half * 96 + ((i / 3) * 16 + r16) * 3 + i % 3) * p.lda + (tid & 7) * 4;
        }
    }

    int m0, n0, kt0, kt1, split, batch;
    auto set_item = [&](int w) {                            // same XCD-aware walk as vnp_gemm_kernel
        int sbi, wg;
        if (p.xcd_split) {                                  // (K-slice, tile) -> XCD = slice % 8, see VnpGemmArgs::xcd_split
            const int j = w >> 3;
            wg = j % nwg; sbi = (j / nwg) * 8 + (w & 7);
        } else {
            sbi = w / nwg;
            const int t = w - sbi * nwg;
            const int q = nwg >> 3, r = nwg & 7, x = t & 7;
            wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3);
        }
        // column tiles in groups of cg: (group, row panel, column in group), column fastest
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
        const float* pA = p.A + (size_t)fbatch * p.bsa;
        const float* pB = p.B + (size_t)fbatch * p.bsb;
        const int k0 = kt * BK;
        const bool full = FAST || (k0 + BK <= p.K);
        if (EPI >= 1) {
            const float* blk = pA + (size_t)fm0 * p.lda + k0;
#pragma unroll
            for (int j = 0; j < AV; ++j) ra[j] = *(const vnp_f4*)(blk + perm_off[j]);
        } else if (FAST || (va && full && fm0 + BM <= p.M))
            vnp_load_tile_fast<AV>(ra, AKC ? pA + (size_t)fm0 * p.lda + k0 : pA + fm0 + (size_t)k0 * p.lda, oa, sa);
        else vnp_split_load_generic<BM, AKC, AV>(ra, pA, p.lda, fm0, p.M, k0, p.K, va, tid);
        if (FAST || (vb && full && fn0 + BN <= p.N))
            vnp_load_tile_fast<BV>(rb, BKC ? pB + (size_t)fn0 * p.ldb + k0 : pB + fn0 + (size_t)k0 * p.ldb, ob, sb);
        else vnp_split_load_generic<BN, BKC, BV>(rb, pB, p.ldb, fn0, p.N, k0, p.K, vb, tid);
    };
    vnp_u2 pa[AV][3], pb[BV][3];
    auto convert = [&]() {
        vnp_split_convert<AKC, AV, NPL, F16>(ra, pa, sca);
        vnp_split_convert<BKC, BV, NPL, F16>(rb, pb, scb);
    };
    auto write = [&]() {
        vnp_split_write<BM, AKC, AV, NPL>(pa, As, tid);
        vnp_split_write<BN, BKC, BV, NPL>(pb, Bs, tid);
    };

    // fragment read offsets (v_mfma_f32_16x16x32_bf16 operand: lane l holds row l & 15, k = 8 (l >> 4) .. + 7 = 16-byte slot l >> 4)
    const int l16 = lane & 15, lq = lane >> 4;          // 16x16x32 operand: row l16, k = 8 lq .. 8 lq + 7 (slot lq)
    int a_rd[TM], b_rd[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_rd[i] = vnp_split_row_off<BM, AKC>(wm * (BM / 2) + i * 16 + l16, lq);
#pragma unroll
    for (int j = 0; j < TN; ++j) b_rd[j] = vnp_split_row_off<BN, BKC>(wn * (BN / 2) + j * 16 + l16, lq);

    auto screen_epilogue = [&](vnp_f4 (&acc)[TM][TN], int m0, int n0) { vnp_split_screen_epilogue<TM, TN>(p, acc, m0, n0, wm, wn, lane); };

    if (EPI == 4) {
        // ---- the one-fp16-product screening pass has its own K loop.  Its operands ARE fp16 already (p.A / p.B point to [rows][K] halves,
        // lda / ldb in halves: pool_gemm.hip converts them once, in the pass that takes the row norms), so there is nothing to do between the
        // load and LDS — and with 24 MFMAs per 32-wide K-tile the pass is bound by the operand bytes a CU pulls from L2 (both panels are
        // re-read for every tile: ~45 GB/s per CU against the 66-73 the L2 serves), not by the matrix pipe.  So: half the bytes (fp16
        // images instead of fp32), K walked in steps of 64 — a thread loads 16 bytes and eight neighbouring lanes cover one whole
        // 128-byte line of a row — with the second 32-wide half staged in the LDS region of the (unused) second plane, 128 bytes
        // further so that the ds_write_b128 of a 16-lane group (two rows x two halves) still covers all 64 banks: half the barriers
        // and twice the MFMAs behind every load.  Measured on maxpool2's shape (196608 x 4096 x 1024): 2.91 ms (fp32 operands converted
        // in the loop) -> 2.24 (fp16 images) -> 2.07 (64-wide steps) -> see DESIGN.md; an LDS-DMA ring of three 32-wide tiles
        // (global_load_lds_dwordx4 two tiles ahead, one raw barrier per tile, source-side swizzle) measured 2.37: its pieces are
        // 64-byte row halves and it syncs twice as often.
        const unsigned short* A16 = (const unsigned short*)p.A;
        const unsigned short* B16 = (const unsigned short*)p.B;
        constexpr int BK2 = 2 * BK, AH = BM / 32, BH = BN / 32;          // 16-byte loads per thread and step: row tid/8 + 32 j, slot tid & 7
        constexpr int APL2 = APL + 64, BPL2 = BPL + 64;                  // where the second half-tile lives (64 bytes on: see screen_wide_kernel.h)
        const int nkt = p.K / BK2;                                       // (the host guarantees K % 64 == 0)
        int offA[AH], offB[BH];
#pragma unroll
        for (int j = 0; j < AH; ++j) {
            const int rho = (tid >> 3) + 32 * j, half = rho / 96, within = rho % 96, i = within >> 4, r16 = within & 15;
            offA[j] = (half * 96 + ((i / 3) * 16 + r16) * 3 + i % 3) * p.lda + (tid & 7) * 8;
        }
#pragma unroll
        for (int j = 0; j < BH; ++j) offB[j] = ((tid >> 3) + 32 * j) * p.ldb + (tid & 7) * 8;
        const int sub = (tid >> 2) & 1, q = tid & 3;                     // which 32-wide half, and the 16-byte slot inside it
        vnp_u4 ha[AH], hb[BH];
        auto fetch4 = [&](int fm0, int fn0, int kt) {
            const unsigned short* srcA = A16 + (size_t)fm0 * p.lda + kt * BK2;
            const unsigned short* srcB = B16 + (size_t)fn0 * p.ldb + kt * BK2;
#pragma unroll
            for (int j = 0; j < AH; ++j) ha[j] = *(const vnp_u4*)(srcA + offA[j]);
#pragma unroll
            for (int j = 0; j < BH; ++j) hb[j] = *(const vnp_u4*)(srcB + offB[j]);
        };
        auto write4 = [&]() {
#pragma unroll
            for (int j = 0; j < AH; ++j) *(vnp_u4*)(As + sub * APL2 + vnp_split_row_off<BM, true>((tid >> 3) + 32 * j, q)) = ha[j];
#pragma unroll
            for (int j = 0; j < BH; ++j) *(vnp_u4*)(Bs + sub * BPL2 + vnp_split_row_off<BN, true>((tid >> 3) + 32 * j, q)) = hb[j];
        };
        int w = blockIdx.x;
        if (w >= total) return;
        set_item(w);
        fetch4(m0, n0, 0);
        write4();
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
            const int cm0 = m0, cn0 = n0;
            for (int kt = 0; kt < nkt; ++kt) {
                if (kt + 1 < nkt) fetch4(cm0, cn0, kt + 1);
                else if (more) { set_item(wnext); fetch4(m0, n0, 0); }          // the next item's first step, behind this item's last MFMAs
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
                __syncthreads();
                if (kt + 1 < nkt) { write4(); __syncthreads(); }
            }
            screen_epilogue(acc, cm0, cn0);
            if (!more) break;
            w = wnext;
            write4();                // m0 / n0 already are the next item's (set above)
            __syncthreads();
        }
        return;
    }

    int w = blockIdx.x;
    if (w >= total) return;
    // diagnostic build only (VAR & 8, scripts/gemm_split_bench.hip): shader-clock and 100 MHz-clock stamps around the block's life
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
                    for (int pl = 0; pl < NPL; ++pl)
                        b[j][pl] = __builtin_bit_cast(vnp_bf8, *(const vnp_u4*)(Bs + pl * BPL + b_rd[j]));
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
                constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    vnp_bf8 a[3];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        a[pl] = __builtin_bit_cast(vnp_bf8, *(const vnp_u4*)(As + pl * APL + a_rd[i]));
                    if (PREC == 1) {
                        // products (l h), (h l), (h h) of this K-tile from ZERO accumulators; their sum joins acc through a VALU add (see PREC = 1 above)
                        vnp_f4 tq[TN];
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            tq[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vnp_h8, a[1]), __builtin_bit_cast(vnp_h8, b[j][0]),
                                                                           vnp_f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            tq[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vnp_h8, a[0]), __builtin_bit_cast(vnp_h8, b[j][1]), tq[j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            tq[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vnp_h8, a[0]), __builtin_bit_cast(vnp_h8, b[j][0]), tq[j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] += tq[j];
                        continue;
                    }
#pragma unroll
                    for (int t = T0; t < 6; ++t)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if (F16)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vnp_h8, a[0]), __builtin_bit_cast(vnp_h8, b[j][0]),
                                                                                   acc[i][j], 0, 0, 0);
                            else
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[t]], b[j][PB[t]], acc[i][j], 0, 0, 0);
                        }
                }
            }
            __syncthreads();
            if (kt + 1 < kt1) {
                convert();
                write();
                __syncthreads();
            }
        }

        if (PREC == 1) {                      // undo the operand scales (exact: powers of two)
            const float inv = isa * isb;
            const bool one = inv >= 1e-30f && inv <= 1e30f;          // the product itself may leave fp32's normal range
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][e] = one ? acc[i][j][e] * inv : (acc[i][j][e] * isa) * isb;
        }
        vnp_split_epilogue<BM, BN, FAST, VAR, EPI, TM, TN>(p, acc, m0, n0, split, batch, wm, wn, lane);

        if (!more) {
            if ((VAR & 8) && tid == 0) {
                unsigned long long* dbg = (unsigned long long*)p.bias;          // diagnostic build: bias carries the stamp buffer
                dbg[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_clk;
                dbg[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t_real;
            }
            break;
        }
        w = wnext;
        set_item(w);
        convert();
        write();                 // the next item's first K-tile (fetched before the last MFMAs) -> LDS
        __syncthreads();
    }
}
