// fp32-MFMA GEMM kernel template shared by gemm.hip (the library) and scripts/gemm_bench.hip (tuning harness).
#pragma once
#include "vnp_common.h"

struct VnpGemmArgs {
    const float* A; const float* B; float* C;
    const float* bias;                   // optional: C[r][c] += bias[(r / rows_per_group) * 3 + r % 3][c]
    int M, N, K;
    int lda, ldb, ldc, ldbias;
    int rows_per_group;
    int accumulate;                      // C += result instead of C = result
    const float* resid; int ldr;         // with accumulate: the addend is resid[row][col] (row stride ldr) instead of C's old value — a
                                         // residual connection taken in the epilogue (one batch; split-K: applied by the slab reduction)
    int tiles_per_split;                 // K-tiles handled by one blockIdx.y
    long long slab;                      // elements between the output slabs of consecutive splits
    int nsplit, nbatch;                  // work items = tiles x nsplit x nbatch, walked persistently
    long long bsa, bsb, bsc;             // batch strides (elements) of A, B, C (batched GEMM: attention heads)
    // VN max-pool epilogue of vnp_gemm_split_kernel<.., EPI = 1> (C is not written): per 32-point wave block and channel,
    // the best <x, d>, its point index inside the group of pool_gs points, and the winning x vector
    float* pool_val; int* pool_idx; float* pool_x; int pool_gs;
    // EPI = 3 (screening pass of the exact screened arg-max pool, pool_gemm.hip): three of the six bf16 products only; per 32-point
    // wave block and channel one float4 {max lower bound, largest upper bound, second-largest upper bound, its local index}
    // goes to pool_val (as float4*).  row_norm[M]: upper bounds of |A row|_2; col_norm[N]: of |B row|_2; screen_eps: the error
    // constant E with |exact - screened| <= E |A row| |B row| per output element.
    const float* row_norm; const float* col_norm; float screen_eps; float screen_abs;
    // when non-null the kernel returns at once unless *pred != 0 (device-side predicated launch: the exact 6-product fallback of
    // the screened pool is enqueued unconditionally and runs only if the screening pass raised its flag)
    const int* pred;
    // EPI = 2: C is stored as usual AND the BatchNorm statistics of the first stats_cols columns are taken from the
    // accumulators: Welford (mean, M2) of n = |x| + eps over each wave's 32 points -> stats_part[M/96][stats_cols]
    float2* stats_part; int stats_cols;
    int col_group;                       // split kernel: walk the column tiles in groups of this many (0 = all): keeps the B
                                         // panels of a group L2-resident while the row panels stream past
    // PREC = 1 (two fp16 planes, three products): device words holding the bit pattern of max |A|, max |B| over the whole operand
    // (vnp_absmax, or emitted by the kernel that produced the operand); the kernel derives its power-of-two operand scales from them
    const unsigned* amax_a; const unsigned* amax_b;
    // split kernel, split-K with few output tiles (weight gradients): all tiles of one K-slice go to ONE XCD (workgroup w runs on
    // XCD w % 8), so that the slice's A / B rows are fetched into that XCD's L2 once and shared by its tiles, instead of every XCD
    // streaming its own column blocks of the whole K range (nsplit % 8 == 0, one batch)
    int xcd_split;
};

template <int R, int BK, bool KC, int NV>
__device__ __forceinline__ void vnp_load_tile(vnp_f4 (&reg)[NV], const float* __restrict__ G, int ld,
                                              int r0, int rmax, int k0, int kmax, bool vec_ok, int tid)
{
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = tid + 256 * j;
        int r, k;
        if (KC) { r = f / (BK / 4); k = (f % (BK / 4)) * 4; }
        else    { k = f / (R / 4);  r = (f % (R / 4)) * 4; }
        const int gr = r0 + r, gk = k0 + k;
        vnp_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (KC) {
            if (gr < rmax && gk < kmax) {
                const float* src = G + (size_t)gr * ld + gk;
                if (vec_ok && gk + 3 < kmax) v = *(const vnp_f4*)src;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gk + e < kmax) v[e] = src[e];
                }
            }
        } else {
            if (gk < kmax && gr < rmax) {
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


// Fast path of vnp_load_tile: the whole tile is in range and 16-byte aligned.  `blk` is the wave-uniform address of
// the tile's first element for this K-tile; thread offsets are voff0 + j*step (f = tid + 256 j is linear in j for both
// tile shapes: KC -> row f/8, !KC with R = 128 -> k-row f/32), so no per-load address arithmetic is left in the loop.
template <int NV>
__device__ __forceinline__ void vnp_load_tile_fast(vnp_f4 (&reg)[NV], const float* __restrict__ blk, int voff0, int step)
{
#pragma unroll
    for (int j = 0; j < NV; ++j) reg[j] = *(const vnp_f4*)(blk + voff0 + j * step);
}

template <int R, int BK, bool KC>
__device__ __forceinline__ void vnp_tile_offsets(int& voff0, int& step, int ld, int tid)
{
    static_assert(KC || R == 128, "M/N-contiguous operands use 128-wide tiles");
    if (KC) { voff0 = (tid / (BK / 4)) * ld + (tid % (BK / 4)) * 4; step = (256 / (BK / 4)) * ld; }
    else    { voff0 = (tid / (R / 4)) * ld + (tid % (R / 4)) * 4;   step = (256 / (R / 4)) * ld; }
}

template <int R, int BK, bool KC, int NV, int LDS_LD>
__device__ __forceinline__ void vnp_store_tile(const vnp_f4 (&reg)[NV], float* S, int tid)
{
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = tid + 256 * j;
        int r, k;
        if (KC) { r = f / (BK / 4); k = (f % (BK / 4)) * 4; *(vnp_f4*)&S[r * LDS_LD + k] = reg[j]; }
        else    { k = f / (R / 4);  r = (f % (R / 4)) * 4; *(vnp_f4*)&S[k * LDS_LD + r] = reg[j]; }
    }
}

// Persistent kernel: gridDim.x <= (resident blocks per CU) x 256 workgroups walk the work items (output tile x K-split
// x batch) with a fixed stride.  While a tile's last K-step runs its MFMAs and its epilogue stores drain, the first
// K-tile of the block's NEXT work item is already in flight (global -> registers), so block start-up latency and the
// epilogue are paid once per workgroup instead of once per tile.
// FAST: every tile is full and 16-byte aligned (M % BM == 0, N % BN == 0, K % BK == 0, ld % 4 == 0): no bounds logic at
// all in the loop or the epilogue; the generic instantiation handles ragged shapes.
template <int BM, int BN, int BK, bool AKC, bool BKC, bool FAST>
__global__ void __launch_bounds__(256, 2) vnp_gemm_kernel(const VnpGemmArgs p)
{
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int LDAS = AKC ? BK + 4 : BM;       // +4 floats: conflict-free ds_read_b128 (stride 36 = 4*9, 9 odd)
    constexpr int LDBS = BKC ? BK + 4 : BN;
    constexpr int ASZ = AKC ? BM * LDAS : BK * LDAS;
    constexpr int BSZ = BKC ? BN * LDBS : BK * LDBS;
    constexpr int AV = BM * BK / 4 / 256, BV = BN * BK / 4 / 256;
    __shared__ __attribute__((aligned(16))) float smem[ASZ + BSZ];
    float* As = smem;
    float* Bs = smem + ASZ;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int nwg = nbm * nbn;                            // output tiles
    const int total = nwg * p.nsplit * p.nbatch;          // work items
    const int nk = (p.K + BK - 1) / BK;

    const bool va = FAST || (((p.lda & 3) == 0) && ((((uintptr_t)p.A) & 15) == 0) && ((p.bsa & 3) == 0));
    const bool vb = FAST || (((p.ldb & 3) == 0) && ((((uintptr_t)p.B) & 15) == 0) && ((p.bsb & 3) == 0));
    int oa, sa, ob, sb;
    vnp_tile_offsets<BM, BK, AKC>(oa, sa, p.lda, tid);
    vnp_tile_offsets<BN, BK, BKC>(ob, sb, p.ldb, tid);

    // work item w -> (m0, n0, K-tile range, split, batch).  XCD-aware order: workgroups b and b+8 share an XCD (round-robin
    // dispatch); XCD x walks the contiguous run of output tiles [x*nwg/8, (x+1)*nwg/8) (column tile fastest) so that
    // co-resident blocks share A/B panels in its L2.
    int m0, n0, kt0, kt1, split, batch;
    auto set_item = [&](int w) {
        const int sbi = w / nwg, t = w - sbi * nwg;
        const int q = nwg >> 3, r = nwg & 7, x = t & 7;
        const int wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3);
        m0 = (wg / nbn) * BM; n0 = (wg % nbn) * BN;
        split = sbi % p.nsplit; batch = sbi / p.nsplit;
        kt0 = split * p.tiles_per_split;
        kt1 = min(nk, kt0 + p.tiles_per_split);
    };

    vnp_f4 ra[AV], rb[BV];
    auto fetch = [&](int fm0, int fn0, int fbatch, int kt) {
        const float* pA = p.A + (size_t)fbatch * p.bsa;
        const float* pB = p.B + (size_t)fbatch * p.bsb;
        const int k0 = kt * BK;
        if (FAST) {
            vnp_load_tile_fast<AV>(ra, AKC ? pA + (size_t)fm0 * p.lda + k0 : pA + fm0 + (size_t)k0 * p.lda, oa, sa);
            vnp_load_tile_fast<BV>(rb, BKC ? pB + (size_t)fn0 * p.ldb + k0 : pB + fn0 + (size_t)k0 * p.ldb, ob, sb);
        } else {
            const bool full = k0 + BK <= p.K;
            if (va && full && fm0 + BM <= p.M)
                vnp_load_tile_fast<AV>(ra, AKC ? pA + (size_t)fm0 * p.lda + k0 : pA + fm0 + (size_t)k0 * p.lda, oa, sa);
            else vnp_load_tile<BM, BK, AKC, AV>(ra, pA, p.lda, fm0, p.M, k0, p.K, va, tid);
            if (vb && full && fn0 + BN <= p.N)
                vnp_load_tile_fast<BV>(rb, BKC ? pB + (size_t)fn0 * p.ldb + k0 : pB + fn0 + (size_t)k0 * p.ldb, ob, sb);
            else vnp_load_tile<BN, BK, BKC, BV>(rb, pB, p.ldb, fn0, p.N, k0, p.K, vb, tid);
        }
    };
    auto stage = [&]() {
        vnp_store_tile<BM, BK, AKC, AV, LDAS>(ra, As, tid);
        vnp_store_tile<BN, BK, BKC, BV, LDBS>(rb, Bs, tid);
    };

    int w = blockIdx.x;
    if (w >= total) return;
    set_item(w);
    fetch(m0, n0, batch, kt0);
    stage();
    __syncthreads();

    while (true) {
        vnp_f16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        const int wnext = w + gridDim.x;                     // this block's next work item
        const bool more = wnext < total;

        for (int kt = kt0; kt < kt1; ++kt) {
            if (kt + 1 < kt1) fetch(m0, n0, batch, kt + 1);
            else if (more) {
                // cross-item prefetch: the next item's first K-tile loads overlap the last MFMAs and the epilogue
                const int cm0 = m0, cn0 = n0, ck0 = kt0, ck1 = kt1, cs = split, cb = batch;
                set_item(wnext);
                fetch(m0, n0, batch, kt0);
                m0 = cm0; n0 = cn0; kt0 = ck0; kt1 = ck1; split = cs; batch = cb;
            }
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) {
                float a[TM][4], b[TN][4];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * (BM / 2) + i * 32 + li;
                    if (AKC) {
                        const vnp_f4 v = *(const vnp_f4*)&As[row * LDAS + 8 * q + 4 * lh];
                        a[i][0] = v[0]; a[i][1] = v[1]; a[i][2] = v[2]; a[i][3] = v[3];
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) a[i][s] = As[(8 * q + 4 * lh + s) * LDAS + row];
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = wn * (BN / 2) + j * 32 + li;
                    if (BKC) {
                        const vnp_f4 v = *(const vnp_f4*)&Bs[col * LDBS + 8 * q + 4 * lh];
                        b[j][0] = v[0]; b[j][1] = v[1]; b[j][2] = v[2]; b[j][3] = v[3];
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) b[j][s] = Bs[(8 * q + 4 * lh + s) * LDBS + col];
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (kt + 1 < kt1) {
                stage();
                __syncthreads();
            }
        }

        // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        {
            float* Cz = p.C + (size_t)split * p.slab + (size_t)batch * p.bsc;
            const bool full = FAST || ((m0 + BM <= p.M) && (n0 + BN <= p.N));
            // per-sample VN bias row = (row / rows_per_group)*3 + row%3 without a division per element: the tile's first
            // row fixes the group; a tile spans at most one group boundary when rows_per_group >= BM
            int g0 = 0, r0 = 0;
            const bool bias_fast = p.bias && p.rows_per_group >= BM;
            if (p.bias) { g0 = m0 / p.rows_per_group; r0 = m0 - g0 * p.rows_per_group; }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int lr = wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const int row = m0 + lr;
                    if (!full && row >= p.M) continue;
                    const float* brow = nullptr;
                    if (p.bias) {
                        int grp;
                        if (bias_fast) grp = g0 + ((r0 + lr) >= p.rows_per_group ? 1 : 0);
                        else grp = row / p.rows_per_group;
                        brow = p.bias + (size_t)(grp * 3 + row % 3) * p.ldbias;
                    }
                    float* drow = Cz + (size_t)row * p.ldc;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = n0 + wn * (BN / 2) + j * 32 + li;
                        if (!full && col >= p.N) continue;
                        float v = acc[i][j][e];
                        if (brow) v += brow[col];
                        if (p.accumulate) v += (p.resid ? p.resid + (size_t)row * p.ldr : (const float*)drow)[col];
                        drow[col] = v;
                    }
                }
            }
        }

        if (!more) break;
        w = wnext;
        set_item(w);
        stage();                 // the next item's first K-tile (fetched before the last MFMAs) -> LDS
        __syncthreads();
    }
}
