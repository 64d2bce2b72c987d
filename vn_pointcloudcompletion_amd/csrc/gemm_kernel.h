// fp32-MFMA GEMM kernel template shared by gemm.hip (the library) and scripts/gemm_bench.hip (tuning harness).
#pragma once
#include "vnp_common.h"

struct VnpGemmArgs {
    const float* A; const float* B; float* C;
    const float* A2; const float* B2;   // optional second product accumulated into the same tile (K2 > 0)
    const float* bias;                   // optional: C[r][c] += bias[(r / rows_per_group) * 3 + r % 3][c]
    int M, N, K, K2;
    int lda, ldb, ldc, lda2, ldb2, ldbias;
    int rows_per_group;
    int accumulate;                      // C += result instead of C = result
    int tiles_per_split;                 // K-tiles handled by one blockIdx.y
    long long slab;                      // elements between the output slabs of consecutive splits
    int nsplit;                          // gridDim.y = nsplit * nbatch: blockIdx.y = batch * nsplit + split
    long long bsa, bsb, bsc;             // batch strides (elements) of A, B, C (batched GEMM: attention heads)
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

template <int BM, int BN, int BK, bool AKC, bool BKC, bool DBUF = false, bool PRIO = false>
__global__ void __launch_bounds__(256, 2) vnp_gemm_kernel(const VnpGemmArgs p)
{
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int LDAS = AKC ? BK + 4 : BM;       // +4 floats: conflict-free ds_read_b128 (stride 36 = 4*9, 9 odd)
    constexpr int LDBS = BKC ? BK + 4 : BN;
    constexpr int ASZ = AKC ? BM * LDAS : BK * LDAS;
    constexpr int BSZ = BKC ? BN * LDBS : BK * LDBS;
    constexpr int AV = BM * BK / 4 / 256, BV = BN * BK / 4 / 256;
    __shared__ __attribute__((aligned(16))) float smem[(ASZ + BSZ) * (DBUF ? 2 : 1)];
    float* As = smem;
    float* Bs = smem + ASZ;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give every XCD a
    // contiguous run of tiles (column tile fastest) -> the co-resident blocks of one L2 share A/B panels.
    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int nwg = nbm * nbn;
    int wg;
    {
        const int id = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = id & 7;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    }
    const int m0 = (wg / nbn) * BM, n0 = (wg % nbn) * BN;

    const int nk1 = (p.K + BK - 1) / BK;
    const int nk2 = p.K2 > 0 ? (p.K2 + BK - 1) / BK : 0;
    const int nk = nk1 + nk2;
    const int split = blockIdx.y % p.nsplit, batch = blockIdx.y / p.nsplit;
    const int kt0 = split * p.tiles_per_split;
    const int kt1 = min(nk, kt0 + p.tiles_per_split);

    const bool va1 = ((p.lda & 3) == 0) && ((((uintptr_t)p.A) & 15) == 0);
    const bool vb1 = ((p.ldb & 3) == 0) && ((((uintptr_t)p.B) & 15) == 0);
    const bool va2 = nk2 && ((p.lda2 & 3) == 0) && ((((uintptr_t)p.A2) & 15) == 0);
    const bool vb2 = nk2 && ((p.ldb2 & 3) == 0) && ((((uintptr_t)p.B2) & 15) == 0);

    vnp_f4 ra[AV], rb[BV];
    vnp_f16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fast path: tile rows fully in range, operands 16-byte aligned with ld % 4 == 0 (K-tail tiles fall back)
    const bool fa1 = va1 && (m0 + BM <= p.M), fb1 = vb1 && (n0 + BN <= p.N);
    const bool fa2 = va2 && (m0 + BM <= p.M), fb2 = vb2 && (n0 + BN <= p.N);
    int oa1, sa1, ob1, sb1, oa2 = 0, sa2 = 0, ob2 = 0, sb2 = 0;
    vnp_tile_offsets<BM, BK, AKC>(oa1, sa1, p.lda, tid);
    vnp_tile_offsets<BN, BK, BKC>(ob1, sb1, p.ldb, tid);
    if (nk2) {
        vnp_tile_offsets<BM, BK, AKC>(oa2, sa2, p.lda2, tid);
        vnp_tile_offsets<BN, BK, BKC>(ob2, sb2, p.ldb2, tid);
    }
    const float* pA = p.A + (size_t)batch * p.bsa;
    const float* pB = p.B + (size_t)batch * p.bsb;
    const float* bA1 = AKC ? pA + (size_t)m0 * p.lda : pA + m0;
    const float* bB1 = BKC ? pB + (size_t)n0 * p.ldb : pB + n0;
    const float* bA2 = nk2 ? (AKC ? p.A2 + (size_t)m0 * p.lda2 : p.A2 + m0) : nullptr;
    const float* bB2 = nk2 ? (BKC ? p.B2 + (size_t)n0 * p.ldb2 : p.B2 + n0) : nullptr;

    auto fetch = [&](int kt) {
        if (kt < nk1) {
            const int k0 = kt * BK;
            const bool full = k0 + BK <= p.K;
            if (fa1 && full) vnp_load_tile_fast<AV>(ra, AKC ? bA1 + k0 : bA1 + (size_t)k0 * p.lda, oa1, sa1);
            else vnp_load_tile<BM, BK, AKC, AV>(ra, pA, p.lda, m0, p.M, k0, p.K, va1, tid);
            if (fb1 && full) vnp_load_tile_fast<BV>(rb, BKC ? bB1 + k0 : bB1 + (size_t)k0 * p.ldb, ob1, sb1);
            else vnp_load_tile<BN, BK, BKC, BV>(rb, pB, p.ldb, n0, p.N, k0, p.K, vb1, tid);
        } else {
            const int k0 = (kt - nk1) * BK;
            const bool full = k0 + BK <= p.K2;
            if (fa2 && full) vnp_load_tile_fast<AV>(ra, AKC ? bA2 + k0 : bA2 + (size_t)k0 * p.lda2, oa2, sa2);
            else vnp_load_tile<BM, BK, AKC, AV>(ra, p.A2, p.lda2, m0, p.M, k0, p.K2, va2, tid);
            if (fb2 && full) vnp_load_tile_fast<BV>(rb, BKC ? bB2 + k0 : bB2 + (size_t)k0 * p.ldb2, ob2, sb2);
            else vnp_load_tile<BN, BK, BKC, BV>(rb, p.B2, p.ldb2, n0, p.N, k0, p.K2, vb2, tid);
        }
    };
    auto stage = [&]() {
        vnp_store_tile<BM, BK, AKC, AV, LDAS>(ra, As, tid);
        vnp_store_tile<BN, BK, BKC, BV, LDBS>(rb, Bs, tid);
    };

    auto compute = [&](const float* As_, const float* Bs_) {
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            float a[TM][4], b[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / 2) + i * 32 + li;
                if (AKC) {
                    const vnp_f4 v = *(const vnp_f4*)&As_[row * LDAS + 8 * q + 4 * lh];
                    a[i][0] = v[0]; a[i][1] = v[1]; a[i][2] = v[2]; a[i][3] = v[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) a[i][s] = As_[(8 * q + 4 * lh + s) * LDAS + row];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * (BN / 2) + j * 32 + li;
                if (BKC) {
                    const vnp_f4 v = *(const vnp_f4*)&Bs_[col * LDBS + 8 * q + 4 * lh];
                    b[j][0] = v[0]; b[j][1] = v[1]; b[j][2] = v[2]; b[j][3] = v[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) b[j][s] = Bs_[(8 * q + 4 * lh + s) * LDBS + col];
                }
            }
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
        }
    };

    if (kt0 < kt1) {
        fetch(kt0);
        stage();
        __syncthreads();
    }
    if (DBUF) {
        // two LDS buffers: the next tile is written while the other buffer may still be read by slower waves;
        // one barrier per K-tile
        int cur = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            if (kt + 1 < kt1) fetch(kt + 1);
            compute(As + cur * (ASZ + BSZ), Bs + cur * (ASZ + BSZ));
            if (kt + 1 < kt1) {
                vnp_store_tile<BM, BK, AKC, AV, LDAS>(ra, As + (cur ^ 1) * (ASZ + BSZ), tid);
                vnp_store_tile<BN, BK, BKC, BV, LDBS>(rb, Bs + (cur ^ 1) * (ASZ + BSZ), tid);
            }
            __syncthreads();
            cur ^= 1;
        }
    } else {
        for (int kt = kt0; kt < kt1; ++kt) {
            if (kt + 1 < kt1) fetch(kt + 1);
            compute(As, Bs);
            __syncthreads();
            if (kt + 1 < kt1) {
                stage();
                __syncthreads();
            }
        }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* Cz = p.C + (size_t)split * p.slab + (size_t)batch * p.bsc;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (row < p.M && col < p.N) {
                    float v = acc[i][j][e];
                    if (p.bias) v += p.bias[(size_t)((row / p.rows_per_group) * 3 + row % 3) * p.ldbias + col];
                    float* dst = Cz + (size_t)row * p.ldc + col;
                    if (p.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

