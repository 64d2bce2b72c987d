// VNLinear followed by VNMaxPool without the linear output (models/pcn.py:169-174: first_conv[1] -> maxpool1, second_conv[1] ->
// maxpool2): the arg-max pool in the epilogue of the six-product GEMM, and the exact SCREENED search that replaces most of that product
// (one fp16 screening pass brackets every score, the surviving candidates are evaluated with the six-product sequence: same bits).
#include "gemm_host.h"
#include "screen_wide_kernel.h"
#include "screen_xwide_kernel.h"

// ---- VNLinear + VNMaxPool in one pass (models/pcn.py:173-174: second_conv[1] -> maxpool2) -------------------------------
// Stage 2 of the fused pool: reduce the per-32-point partials of a group (ascending order keeps the first maximum).
__global__ void vnp_pool_fused_final_kernel(const float* __restrict__ pval, const int* __restrict__ pidx, const float* __restrict__ px,
                                            int nparts, int C, int* __restrict__ idx, float* __restrict__ out, int ldo,
                                            const int* __restrict__ pred)
{
    if (pred && *pred == 0) return;                       // predicated launch (see VnpGemmArgs::pred)
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const long long g = blockIdx.y;
    size_t slot = (size_t)g * nparts * C + c;
    float best = pval[slot];
    size_t bs = slot;
    for (int k = 1; k < nparts; ++k) {
        slot += C;
        const float v = pval[slot];
        if (v > best) { best = v; bs = slot; }
    }
    idx[(size_t)g * C + c] = pidx[bs];
#pragma unroll
    for (int t = 0; t < 3; ++t) out[((size_t)g * 3 + t) * ldo + c] = px[3 * bs + t];
}

extern "C" long long vnp_linear_pool_ws_bytes(long long G, int gs, int C)
{
    if (G <= 0 || gs <= 0 || C <= 0 || gs % 32) return 0;
    return G * (gs / 32) * (long long)C * 20;
}

// out[g*3+t][c] = (W f)[(g*gs + n*)*3 + t][c] with n* = argmax_n <W f, W' f>(n, c) over the gs points of group g, idx[g][c] = n*.
// x [G*gs*3, K] rows layout; Wcat [2C, K]: per 32 channels, 32 rows of W then 32 rows of W' (W' = Wd W for VNMaxPool's
// direction map); ws: vnp_linear_pool_ws_bytes(G, gs, C).  Requirements (else VNP_EINVAL, callers fall back to
// vnp_gemm + vnp_vn_pool_fwd): split GEMM mode, (G*gs*3) % 192 == 0, gs % 32 == 0, C % 64 == 0, K % 32 == 0, 16-byte aligned.
extern "C" int vnp_linear_pool_fwd(hipStream_t st, const float* x, int ldx, const float* Wcat, int ldw, long long G, int gs, int C,
                                   int K, int* idx, float* out, int ldo, void* ws)
{
    if (!x || !Wcat || !idx || !out || !ws || G <= 0 || gs <= 0 || C <= 0 || K <= 0) return VNP_EINVAL;
    const long long M = G * gs * 3;
    if (vnp_gemm_get_mode() != 0 || M % 192 || gs % 32 || C % 64 || K % 32 || ldx % 4 || ldw % 4 || M > 2147483647LL ||
        (uintptr_t)x % 16 || (uintptr_t)Wcat % 16 || G > 65535)
        return VNP_EINVAL;
    const long long nslot = G * (gs / 32) * (long long)C;
    VnpGemmArgs p{};
    p.A = x; p.B = Wcat; p.C = nullptr; p.bias = nullptr;
    p.M = (int)M; p.N = 2 * C; p.K = K;
    p.lda = ldx; p.ldb = ldw; p.ldc = 0; p.rows_per_group = 1;
    p.tiles_per_split = K / 32; p.nsplit = 1; p.nbatch = 1;
    p.pool_val = (float*)ws;
    p.pool_idx = (int*)((char*)ws + nslot * 4);
    p.pool_x = (float*)((char*)ws + nslot * 8);
    p.pool_gs = gs;
    p.col_group = 8;
    const long long items = (M / 192) * (2LL * C / 128);
    if (items > 2147483647LL) return VNP_EINVAL;
    const long long cap = (long long)vnp_num_cus() * 2;
    dim3 grid((unsigned)(items < cap ? items : cap)), block(256);
    vnp_prof_begin(st);
    VNP_KLAUNCH((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 1>), grid, block, 0, st, p);
    vnp_prof_end(st, 0);
    hipLaunchKernelGGL(vnp_pool_fused_final_kernel, dim3(vnp_cdiv(C, 256), (unsigned)G), dim3(256), 0, st,
                       (const float*)p.pool_val, (const int*)p.pool_idx, (const float*)p.pool_x, gs / 32, C, idx, out, ldo,
                       (const int*)nullptr);
    return vnp_last_error();
}

// VNLinear -> VNMaxPool where the linear output has ANOTHER consumer too (encoder first_conv[1] -> maxpool1 and second_conv[0]): as
// vnp_linear_pool_fwd, and y = x W^T [G*gs*3, C] is stored as well (EPI = 5); the direction d = W' x still never exists in memory and
// the separate direction GEMM + pool pass over y and d are gone.  ws: vnp_linear_pool_ws_bytes(G, gs, C).
extern "C" int vnp_linear_pool_store_fwd(hipStream_t st, const float* x, int ldx, const float* Wcat, int ldw, long long G, int gs, int C,
                                         int K, float* y, int ldy, int* idx, float* out, int ldo, void* ws)
{
    if (!x || !Wcat || !y || !idx || !out || !ws || G <= 0 || gs <= 0 || C <= 0 || K <= 0) return VNP_EINVAL;
    const long long M = G * gs * 3;
    if (vnp_gemm_get_mode() != 0 || M % 192 || gs % 32 || C % 64 || K % 32 || ldx % 4 || ldw % 4 || ldy < C || M > 2147483647LL ||
        (uintptr_t)x % 16 || (uintptr_t)Wcat % 16 || G > 65535)
        return VNP_EINVAL;
    const long long nslot = G * (gs / 32) * (long long)C;
    VnpGemmArgs p{};
    p.A = x; p.B = Wcat; p.C = y; p.bias = nullptr;
    p.M = (int)M; p.N = 2 * C; p.K = K;
    p.lda = ldx; p.ldb = ldw; p.ldc = ldy; p.rows_per_group = 1;
    p.tiles_per_split = K / 32; p.nsplit = 1; p.nbatch = 1;
    p.pool_val = (float*)ws;
    p.pool_idx = (int*)((char*)ws + nslot * 4);
    p.pool_x = (float*)((char*)ws + nslot * 8);
    p.pool_gs = gs;
    p.col_group = 8;
    const long long items = (M / 192) * (2LL * C / 128);
    if (items > 2147483647LL) return VNP_EINVAL;
    const long long cap = (long long)vnp_num_cus() * 2;
    dim3 grid((unsigned)(items < cap ? items : cap)), block(256);
    vnp_prof_begin(st);
    VNP_KLAUNCH((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 5>), grid, block, 0, st, p);
    vnp_prof_end(st, 0);
    hipLaunchKernelGGL(vnp_pool_fused_final_kernel, dim3(vnp_cdiv(C, 256), (unsigned)G), dim3(256), 0, st,
                       (const float*)p.pool_val, (const int*)p.pool_idx, (const float*)p.pool_x, gs / 32, C, idx, out, ldo,
                       (const int*)nullptr);
    return vnp_last_error();
}

// ---- the same pool with an exact SCREENED arg-max ---------------------------------------------------------------------------
// The 6-product GEMM above spends 1.65 TFLOP (config 2) to pick ONE point per (sample, channel).  Here a 3-product pass (half
// the MFMA work, no l plane) brackets every score s6(n, c) = <x6, d6> the 6-product kernels would produce between rigorous
// bounds (EPI = 3 epilogue); a point can only be the arg-max if its upper bound reaches the largest lower bound of its group,
// which leaves ~1-2 candidates per (sample, channel); the candidates — (point, channel) PAIRS, not tiles — are then evaluated
// exactly: 16 pairs per wave as the DIAGONAL of a 16x16 MFMA tile whose A rows are the pairs' feature rows and whose B rows are
// the pairs' weight rows, with the 6-product sequence and K order of vnp_gemm_split_kernel, so every x6 / d6 / score has the bits
// the full kernels produce (an MFMA output element depends only on its own A row, B row and accumulator).  The arg-max over the
// candidates (ties to the lowest point) is therefore the arg-max of the full computation: indices and pooled vectors are
// bit-identical to vnp_linear_pool_fwd and to vnp_gemm + vnp_vn_pool_fwd (tests/test_gpu_layers.py).  If the candidate list
// overflows or any bound is not finite (NaN / Inf inputs), a device-side flag is raised and the full kernel — enqueued
// unconditionally behind it as a predicated launch — recomputes everything; without the flag it exits at once (~10 us).

// upper bound of |row|_2 for rows of K floats (one wave per row); inflated past its own fp32 rounding
// h16 (optional): the row rounded to fp16 (round to nearest even; |x| > 65504 -> Inf), [rows][K] halves — the operand image the
// one-product screening pass reads
__global__ void __launch_bounds__(256) vnp_row_norm_kernel(const float* __restrict__ x, int ld, long long rows, int K, float* __restrict__ out,
                                                          unsigned short* __restrict__ h16)
{
    const long long r = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* src = x + (size_t)r * ld;
    float s = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const vnp_f4 v = __builtin_nontemporal_load((const vnp_f4*)(src + k));
        s = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], s))));
        if (h16) *(vnp_u2*)(h16 + (size_t)r * K + k) = vnp_u2{vnp_pack_f16(v[0], v[1]), vnp_pack_f16(v[2], v[3])};
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[r] = sqrtf(s) * (1.0f + 2.0f * (float)K * 5.97e-8f) * 1.0001f;
}

struct VnpScreenHead { int count; int flag; int cap; int pad; };     // head of the workspace (device memory)
#define VNP_POOL_SCREEN_FP16 1
#define VNP_POOL_SCREEN_NO_XWIDE 2
#define VNP_POOL_SCREEN_NARROW 4

__global__ void vnp_pool_screen_init_kernel(VnpScreenHead* head, int cap) { head->count = 0; head->flag = 0; head->cap = cap; head->pad = 0; }

// per (group, channel): L = largest lower bound; every block whose largest upper bound reaches L contributes its top point, its
// second point when that one's upper bound reaches L too — or all 32 of its points when even the third-largest does.  A workgroup = 64 channels x 4 quarters of the
// group's 32-point blocks; each thread keeps its quarter's slots in registers (one pass over memory), the four quarters meet in
// LDS, and the workgroup appends ALL its pairs with one atomicAdd, so the candidates of neighbouring channels are contiguous in
// the list (coalesced reads in the exact and final passes).  grid (C/64, G), 256 threads; nparts <= 4 * SEL_MAXQ.
#define SEL_MAXQ 16
__global__ void __launch_bounds__(256) vnp_pool_screen_select_kernel(const float4* __restrict__ slots, const int* __restrict__ slot_idx,
                                                                     int nparts, int C, int gs,
                                                                     VnpScreenHead* __restrict__ head, int2* __restrict__ seg, int2* __restrict__ list)
{
    __shared__ float sL[4][64];
    __shared__ int sbad[4][64], scnt[4][64], sbase;
    const int cx = threadIdx.x & 63, qy = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // a wave = one point lane: scalar
    const int c = blockIdx.x * 64 + cx;
    const long long g = blockIdx.y;
    const int per = (nparts + 3) / 4, k0 = qy * per, k1 = min(nparts, k0 + per);
    const float4* sl = slots + (size_t)g * nparts * C + c;
    float4 v[SEL_MAXQ];
    float L = -INFINITY;
    int bad = 0;
#pragma unroll
    for (int j = 0; j < SEL_MAXQ; ++j) {
        if (k0 + j < k1) {
            v[j] = sl[(size_t)(k0 + j) * C];
            L = fmaxf(L, v[j].x);
            bad |= !(v[j].x < INFINITY) || !(v[j].y < INFINITY) || !(v[j].z < INFINITY) || !(v[j].w < INFINITY);
        }
    }
    sL[qy][cx] = L; sbad[qy][cx] = bad;
    __syncthreads();
    L = fmaxf(fmaxf(sL[0][cx], sL[1][cx]), fmaxf(sL[2][cx], sL[3][cx]));
    bad = sbad[0][cx] | sbad[1][cx] | sbad[2][cx] | sbad[3][cx];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < SEL_MAXQ; ++j)
        if (k0 + j < k1 && v[j].y >= L) cnt += (v[j].w >= L) ? 32 : (v[j].z >= L) ? 2 : 1;
    scnt[qy][cx] = cnt;
    __syncthreads();
    // exclusive prefix over (channel, quarter) in channel-major order: thread (cx, qy) starts after all pairs of channels < cx and
    // of quarters < qy of its own channel
    int before = 0, total = 0, mine_total = 0;
    for (int cc = 0; cc < 64; ++cc) {
        const int t = scnt[0][cc] + scnt[1][cc] + scnt[2][cc] + scnt[3][cc];
        if (cc < cx) before += t;
        if (cc == cx) mine_total = t;
        total += t;
    }
    for (int q = 0; q < qy; ++q) before += scnt[q][cx];
    int anybad = bad;
    anybad = __syncthreads_or(anybad || mine_total <= 0);
    if (threadIdx.x == 0) {
        int base = -1;
        if (anybad) atomicOr(&head->flag, 1);
        else {
            base = atomicAdd(&head->count, total);
            if (base + total > head->cap) { atomicOr(&head->flag, 1); base = -1; }
        }
        sbase = base;
    }
    __syncthreads();
    const int base = sbase;
    if (qy == 0) seg[(size_t)g * C + c] = base < 0 ? make_int2(0, 0) : make_int2(base + before, mine_total);
    if (base < 0) return;
    int w = base + before;
    const int pbase = (int)(g * gs);
#pragma unroll
    for (int j = 0; j < SEL_MAXQ; ++j) {
        if (k0 + j < k1 && v[j].y >= L) {
            if (v[j].w >= L) { for (int i = 0; i < 32; ++i) list[w++] = make_int2(pbase + (k0 + j) * 32 + i, c); }
            else {
                const int packed = slot_idx[((size_t)g * nparts + k0 + j) * C + c];
                list[w++] = make_int2(pbase + (packed & 0xffff), c);
                if (v[j].z >= L) list[w++] = make_int2(pbase + (packed >> 16), c);
            }
        }
    }
}

// ---- exact evaluation of the candidate pairs ---------------------------------------------------------------------------------------
// The pairs are counting-sorted by BIN = 5 consecutive points (hist -> scan -> scatter).  One wave evaluates one TILE = up to 16 pairs of
// one bin: the A operand of its MFMAs holds the bin's 15 feature rows (5 points x 3 components, row 3 s + t), the B operands the W and W'
// rows of the tile's 16 pairs, so  C[3 s + t][j] = <f[(5 bin + s, t)], W_(c_j)>  and pair j reads its x6 / d6 from the three rows of its
// own point — 48 of the 256 tile entries are useful instead of the 16 of a pure diagonal arrangement, and a bin's feature rows are
// converted once per tile instead of once per pair.  Product sequence and K order are those of vnp_gemm_split_kernel (K-tiles ascending;
// per K-tile (l h)(h l)(m m)(m h)(h m)(h h); A = features, B = weights), so x6, d6 and the score (the pool kernels' fmaf form) have the bits
// the full kernels produce: an MFMA output element depends only on its own A row, B row and accumulator.
#define SCR_BIN 5

__global__ void vnp_pool_screen_hist_kernel(const VnpScreenHead* __restrict__ head, const int2* __restrict__ list, int* __restrict__ hist)
{
    const int count = head->flag ? 0 : head->count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) atomicAdd(&hist[list[i].x / SCR_BIN], 1);
}

// hist[n] (counts) -> start[n + 1] (exclusive prefix of the counts), cursor[n] (a copy for the scatter pass), tstart[n + 1] (exclusive
// prefix of the tiles ceil(count / 16)); one workgroup of 1024 threads, contiguous chunks + an LDS scan of the chunk totals
__global__ void __launch_bounds__(1024) vnp_pool_screen_scan_kernel(const int* __restrict__ hist, int n, int* __restrict__ start,
                                                                    int* __restrict__ cursor, int* __restrict__ tstart)
{
    __shared__ int tot[1024], ttot[1024];
    const int t = threadIdx.x, per = (n + 1023) / 1024, lo = min(n, t * per), hi = min(n, lo + per);
    int s = 0, ts = 0;
    for (int i = lo; i < hi; ++i) { const int v = hist[i]; s += v; ts += (v + 15) >> 4; }
    tot[t] = s; ttot[t] = ts;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? tot[t - o] : 0, tv = t >= o ? ttot[t - o] : 0;
        __syncthreads();
        tot[t] += v; ttot[t] += tv;
        __syncthreads();
    }
    int run = tot[t] - s, trun = ttot[t] - ts;
    for (int i = lo; i < hi; ++i) {
        const int v = hist[i];
        start[i] = run; cursor[i] = run; tstart[i] = trun;
        run += v; trun += (v + 15) >> 4;
    }
    if (t == 1023) { start[n] = tot[1023]; tstart[n] = ttot[1023]; }
}

__global__ void vnp_pool_screen_scatter_kernel(const VnpScreenHead* __restrict__ head, const int2* __restrict__ list, int* __restrict__ cursor,
                                               int* __restrict__ order)
{
    const int count = head->flag ? 0 : head->count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) order[atomicAdd(&cursor[list[i].x / SCR_BIN], 1)] = i;
}

__global__ void __launch_bounds__(256) vnp_pool_screen_exact_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ Wcat, int ldw,
                                                                    int K, long long npts, const VnpScreenHead* __restrict__ head,
                                                                    const int2* __restrict__ list, const int* __restrict__ order,
                                                                    const int* __restrict__ start, const int* __restrict__ tstart, int nbins,
                                                                    float4* __restrict__ res)
{
    if (head->flag) return;
    const int lane = threadIdx.x & 63, l16 = lane & 15, lq = lane >> 4;
    const int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (tile >= tstart[nbins]) return;
    // the bin of this tile: largest b with tstart[b] <= tile (wave-uniform binary search)
    int lo = 0, hi = nbins;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tstart[mid] <= tile) lo = mid; else hi = mid; }
    const int bin = lo;
    const int first = start[bin] + 16 * (tile - tstart[bin]);
    const int npair = min(16, start[bin + 1] - first);
    const int mine = order[first + min(l16, npair - 1)];              // lanes past the tile's last pair repeat it
    const int2 pc = list[mine];
    // A row of this lane: component r % 3 of point 5 bin + r / 3 (r = lane & 15; row 15 is unused and repeats row 14)
    const int r = min(l16, 14);
    long long pa = (long long)bin * SCR_BIN + r / 3;
    if (pa >= npts) pa = npts - 1;
    const float* f0 = x + ((size_t)pa * 3 + r % 3) * ldx + lq * 8;
    const float* w0 = Wcat + (size_t)((pc.y >> 5) * 64 + (pc.y & 31)) * ldw + lq * 8;
    const float* w1 = w0 + (size_t)32 * ldw;
    vnp_f4 ax = {0.f, 0.f, 0.f, 0.f}, ad = {0.f, 0.f, 0.f, 0.f};
    auto planes = [](const vnp_f4 v0, const vnp_f4 v1, vnp_bf8 (&pl)[3]) {
        unsigned h[4], m[4], l[4];
        vnp_split_pair(v0[0], v0[1], h[0], m[0], l[0]);
        vnp_split_pair(v0[2], v0[3], h[1], m[1], l[1]);
        vnp_split_pair(v1[0], v1[1], h[2], m[2], l[2]);
        vnp_split_pair(v1[2], v1[3], h[3], m[3], l[3]);
        pl[0] = __builtin_bit_cast(vnp_bf8, vnp_u4{h[0], h[1], h[2], h[3]});
        pl[1] = __builtin_bit_cast(vnp_bf8, vnp_u4{m[0], m[1], m[2], m[3]});
        pl[2] = __builtin_bit_cast(vnp_bf8, vnp_u4{l[0], l[1], l[2], l[3]});
    };
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    const float* src[3] = {f0, w0, w1};
    vnp_f4 cur[3][2], nxt[3][2];
#pragma unroll
    for (int o = 0; o < 3; ++o) { cur[o][0] = *(const vnp_f4*)src[o]; cur[o][1] = *(const vnp_f4*)(src[o] + 4); }
    for (int k0 = 0; k0 < K; k0 += 32) {
        if (k0 + 32 < K) {                          // next K-step's operands in flight during this step's MFMAs
#pragma unroll
            for (int o = 0; o < 3; ++o) { nxt[o][0] = *(const vnp_f4*)(src[o] + k0 + 32); nxt[o][1] = *(const vnp_f4*)(src[o] + k0 + 36); }
        }
        vnp_bf8 a[3], bw[3], bd[3];
        planes(cur[0][0], cur[0][1], a);
        planes(cur[1][0], cur[1][1], bw);
        planes(cur[2][0], cur[2][1], bd);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            ax = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q]], bw[PB[q]], ax, 0, 0, 0);
            ad = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q]], bd[PB[q]], ad, 0, 0, 0);
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) { cur[o][0] = nxt[o][0]; cur[o][1] = nxt[o][1]; }
    }
    // C layout: lane (lq, col j = lane & 15) holds rows 4 lq + e of column j.  Pair j needs rows 3 s + t (s = its point's slot in the bin):
    // row R sits in lane (R >> 2) * 16 + j, register R & 3 — fetched with one shuffle per register and selected.
    const int sl = pc.x - bin * SCR_BIN;                                          // 0 .. 4
    float xv[3], dv[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int R = 3 * sl + t, srcl = (R >> 2) * 16 + l16, e = R & 3;
        const float x0 = __shfl(ax[0], srcl), x1 = __shfl(ax[1], srcl), x2 = __shfl(ax[2], srcl), x3 = __shfl(ax[3], srcl);
        const float d0 = __shfl(ad[0], srcl), d1 = __shfl(ad[1], srcl), d2 = __shfl(ad[2], srcl), d3 = __shfl(ad[3], srcl);
        xv[t] = e == 0 ? x0 : e == 1 ? x1 : e == 2 ? x2 : x3;
        dv[t] = e == 0 ? d0 : e == 1 ? d1 : e == 2 ? d2 : d3;
    }
    if (lq == 0 && l16 < npair) {
        float4 o;
        o.x = xv[0]; o.y = xv[1]; o.z = xv[2];
        o.w = fmaf(xv[2], dv[2], fmaf(xv[1], dv[1], xv[0] * dv[0]));      // the pool kernels' score form
        res[mine] = o;
    }
}

// per (group, channel): arg-max over its candidates' exact scores (strictly greater wins, ties to the lowest point: the first maximum
// of an ascending scan, as vnp_vn_pool_fwd), index and pooled vector out.  Eight lanes share a pair (a pair has 1-3 candidates as a rule
// but a whole 32-point block now and then: one thread per pair made the block wait for its longest loop).  Skipped when the flag is up
// (the fallback writes the outputs).  grid (ceil(C/32), G), 256 threads = 32 channels x 8 lanes
__global__ void __launch_bounds__(256) vnp_pool_screen_final_kernel(const VnpScreenHead* __restrict__ head, const int2* __restrict__ seg,
                                                                    const int2* __restrict__ list, const float4* __restrict__ res, int C, int gs,
                                                                    int* __restrict__ idx, float* __restrict__ out, int ldo)
{
    if (head->flag) return;
    const int sub = threadIdx.x & 63;                    // a wave per (group, channel) segment: a segment of a near-tied channel can hold
    const int c = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // hundreds of candidates, and eight lanes walking it set the kernel's time
    const long long g = blockIdx.y;
    const bool live = c < C;
    const int2 sg = live ? seg[(size_t)g * C + c] : make_int2(0, 0);
    float best = 0.f; int bp = 0x7fffffff; float bx = 0.f, by = 0.f, bz = 0.f;
    for (int i = sg.x + sub; i < sg.x + sg.y; i += 64) {
        const float4 r = res[i];
        const int pt = list[i].x;
        if (bp == 0x7fffffff || r.w > best || (r.w == best && pt < bp)) { best = r.w; bp = pt; bx = r.x; by = r.y; bz = r.z; }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float ob = __shfl_xor(best, o), ox = __shfl_xor(bx, o), oy = __shfl_xor(by, o), oz = __shfl_xor(bz, o);
        const int op = __shfl_xor(bp, o);
        if (op != 0x7fffffff && (bp == 0x7fffffff || ob > best || (ob == best && op < bp))) { best = ob; bp = op; bx = ox; by = oy; bz = oz; }
    }
    if (live && sub == 0) {
        idx[(size_t)g * C + c] = bp - (int)(g * gs);
        out[((size_t)g * 3 + 0) * ldo + c] = bx;
        out[((size_t)g * 3 + 1) * ldo + c] = by;
        out[((size_t)g * 3 + 2) * ldo + c] = bz;
    }
}

static inline long long vnp_align256(long long v) { return (v + 255) / 256 * 256; }

// candidate capacity: 32 per (group, channel) on average (measured means: 1.7 / 3.6 (bf16x3 / fp16 pass) on uniform random clouds with
// random-init weights, 8 / 21 on surface-like clouds, where neighbouring points have nearly equal scores)
static inline long long vnp_screen_cap(long long G, int C) { long long c = 32 * G * C; return c > 2000000000LL ? 2000000000LL : c; }

extern "C" long long vnp_linear_pool_screened_ws_bytes(long long G, int gs, int C, int K)
{
    if (G <= 0 || gs <= 0 || C <= 0 || K <= 0 || gs % 32) return 0;
    const long long nslot = G * (gs / 32) * (long long)C, cap = vnp_screen_cap(G, C);
    return 256 + vnp_align256(G * gs * 3 * (long long)K * 2) + vnp_align256(2LL * C * K * 2) + vnp_align256(G * gs * 3 * 4) + vnp_align256(2LL * C * 4) + vnp_align256(nslot * 16) + vnp_align256(nslot * 4) + vnp_align256(G * C * 8) +
           vnp_align256(cap * 8) + vnp_align256(cap * 16) + 4 * vnp_align256((G * gs / SCR_BIN + 2) * 4) + vnp_align256(cap * 4) +
           vnp_align256(vnp_linear_pool_ws_bytes(G, gs, C));
}

// Same contract, arguments and results (bit for bit) as vnp_linear_pool_fwd; ws: vnp_linear_pool_screened_ws_bytes(G, gs, C, K) bytes.
// After the call ws begins with {int candidates, int fallback_flag, int capacity} (diagnostics; device memory).
extern "C" int vnp_linear_pool_screened_fwd(hipStream_t st, const float* x, int ldx, const float* Wcat, int ldw, long long G, int gs, int C,
                                            int K, int* idx, float* out, int ldo, void* ws, int flags)
{
    if (!x || !Wcat || !idx || !out || !ws || G <= 0 || gs <= 0 || C <= 0 || K <= 0 || (flags & ~7)) return VNP_EINVAL;
    const long long M = G * gs * 3;
    if (vnp_gemm_get_mode() != 0 || M % 192 || gs % 32 || C % 64 || K % 32 || ldx % 4 || ldw % 4 || M > 2147483647LL ||
        (uintptr_t)x % 16 || (uintptr_t)Wcat % 16 || G > 65535 || G * gs > 2147483647LL || gs / 32 > 4 * SEL_MAXQ)
        return VNP_EINVAL;
    const long long nslot = G * (gs / 32) * (long long)C, cap = vnp_screen_cap(G, C);
    char* w = (char*)ws;
    VnpScreenHead* head = (VnpScreenHead*)w;       w += 256;
    unsigned short* x16 = (unsigned short*)w;      w += vnp_align256(M * (long long)K * 2);          // fp16 images of x and Wcat
    unsigned short* w16 = (unsigned short*)w;      w += vnp_align256(2LL * C * K * 2);
    float* rnorm = (float*)w;                      w += vnp_align256(M * 4);
    float* cnorm = (float*)w;                      w += vnp_align256(2LL * C * 4);
    float4* slots = (float4*)w;                    w += vnp_align256(nslot * 16);
    int* slot_idx = (int*)w;                       w += vnp_align256(nslot * 4);
    int2* seg = (int2*)w;                          w += vnp_align256(G * C * 8);
    int2* list = (int2*)w;                         w += vnp_align256(cap * 8);
    float4* res = (float4*)w;                      w += vnp_align256(cap * 16);
    const int nbins = (int)((G * gs + SCR_BIN - 1) / SCR_BIN);
    const long long binb = vnp_align256((G * gs / SCR_BIN + 2) * 4);
    int* hist = (int*)w;                           w += binb;
    int* start = (int*)w;                          w += binb;
    int* cursor = (int*)w;                         w += binb;
    int* tstart = (int*)w;                         w += binb;
    int* order = (int*)w;                          w += vnp_align256(cap * 4);
    void* ws_full = (void*)w;
    hipLaunchKernelGGL(vnp_pool_screen_init_kernel, dim3(1), dim3(1), 0, st, head, (int)cap);
    const bool f16pass = (flags & VNP_POOL_SCREEN_FP16) != 0 && K % 64 == 0;          // the fp16 pass walks K in steps of 64
    hipLaunchKernelGGL(vnp_row_norm_kernel, dim3((unsigned)vnp_cdiv(M, 4)), dim3(256), 0, st, x, ldx, M, K, rnorm, f16pass ? x16 : nullptr);
    hipLaunchKernelGGL(vnp_row_norm_kernel, dim3((unsigned)vnp_cdiv(2LL * C, 4)), dim3(256), 0, st, Wcat, ldw, 2LL * C, K, cnorm,
                       f16pass ? w16 : nullptr);
    VnpGemmArgs p{};
    p.A = x; p.B = Wcat; p.C = nullptr; p.bias = nullptr;
    p.M = (int)M; p.N = 2 * C; p.K = K;
    p.lda = ldx; p.ldb = ldw; p.ldc = 0; p.rows_per_group = 1;
    p.tiles_per_split = K / 32; p.nsplit = 1; p.nbatch = 1;
    p.pool_val = (float*)slots; p.pool_idx = slot_idx; p.pool_gs = gs; p.col_group = 8;
    p.row_norm = rnorm; p.col_norm = cnorm;
    const long long items = (M / 192) * (2LL * C / 128);
    if (items > 2147483647LL) return VNP_EINVAL;
    const long long capwg = (long long)vnp_num_cus() * 2;
    dim3 grid((unsigned)(items < capwg ? items : capwg)), block(256);
    vnp_prof_begin(st);
    if (f16pass) {
        // ONE fp16 product per operand pair (11 significant bits against bf16's 8): |w f - w~ f~| <= (2^-10 + 2^-22) |w||f| +
        // 2^-25 (|w| + |f|) (the second term: operands under fp16's normal range; an operand over 65504 becomes Inf and raises the
        // fallback flag through a non-finite bound), plus the worst-case fp32 rounding of both accumulations (K and 6K terms)
        p.screen_eps = 9.78e-4f + 7.2f * (float)K * 5.9605e-8f;
        p.screen_abs = 3.0e-8f * sqrtf((float)K) * 1.01f;
        p.A = (const float*)x16; p.B = (const float*)w16; p.lda = K; p.ldb = K;          // the pass reads the fp16 images (halves)
        const bool narrow = (flags & VNP_POOL_SCREEN_NARROW) != 0;
        if ((2 * C) % 256 == 0 && M % 384 == 0 && !narrow && !(flags & VNP_POOL_SCREEN_NO_XWIDE)) {
            // 384 x 256 tiles, operands by LDS-DMA straight into the two 80 KB stages (screen_xwide_kernel.h): 0.71x the bytes into the
            // CU and 0.7x the fragment reads per MFMA of the 192 x 256 pass, no staging registers, no ds_write pass; same slots.
            // maxpool2 (196608 x 4096 x 1024): 1.85-1.95 -> 1.56-1.61 ms (profiles/r04_screen_ablation.txt)
            const long long iw = (M / 384) * (2LL * C / 256);
            const long long cw = vnp_num_cus();
            p.col_group = 16;                       // measured 1 / 2 / 4 / 16 column tiles per XCD run: 1.68 / 1.61 / 1.61 / 1.57 ms
            constexpr int xwide_lds = 2 * (2 * 384 * 64 + 2 * 256 * 64);          // two stages of a 384 x 256 x 64 fp16 tile: all 160 KB
            static std::atomic<unsigned long long> xwide_ok{0};
            hipError_t ex = vnp_allow_dynamic_lds(xwide_ok, (const void*)vnp_screen_f16_xwide_kernel<0>, xwide_lds);
            if (ex != hipSuccess) return (int)ex;
            VNP_KLAUNCH(vnp_screen_f16_xwide_kernel<0>, dim3((unsigned)(iw < cw ? iw : cw)), dim3(512), xwide_lds, st, p);
        } else if ((2 * C) % 256 == 0 && !narrow) {
            // 192 x 256 tiles on eight waves, one workgroup per CU: 0.70x the operand bytes per MFMA (screen_wide_kernel.h); same slots
            const long long iw = (M / 192) * (2LL * C / 256);
            const long long cw = vnp_num_cus();
            // tile walk of an XCD: two column tiles (2 x 512 KB of W images) stay in its L2 while the A panels stream through; with
            // eight (4 MB = the whole L2) the panels evicted them: 1.99 -> 1.89 ms (measured 1 / 2 / 4 / 8 / 16: 1.90 1.89 1.95 1.99 2.00)
            p.col_group = 2;
            constexpr int wide_lds = 2 * (2 * (192 * 64 + 128) + 2 * (256 * 64 + 128));          // two stages of a 192 x 256 x 64 fp16 tile
            static std::atomic<unsigned long long> wide_ok{0};
            hipError_t ew = vnp_allow_dynamic_lds(wide_ok, (const void*)vnp_screen_f16_wide_kernel, wide_lds);
            if (ew != hipSuccess) return (int)ew;
            VNP_KLAUNCH(vnp_screen_f16_wide_kernel, dim3((unsigned)(iw < cw ? iw : cw)), dim3(512), wide_lds, st, p);
        } else
            VNP_KLAUNCH((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 4>), grid, block, 0, st, p);
    } else {
        // THREE bf16 products: dropped (h l)+(l h)+(m m)+... <= 3.05 * 2^-18 of |w||f| per term, plus the worst-case fp32 rounding of
        // BOTH accumulations in any summation order (3K and 6K terms, unit roundoff 2^-24, terms <= 1.01 |w||f|)
        p.screen_eps = 1.17e-5f + 9.2f * (float)K * 5.9605e-8f;
        p.screen_abs = 0.f;
        VNP_KLAUNCH((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 3>), grid, block, 0, st, p);
    }
    vnp_prof_end(st, f16pass ? 3 : 4);
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)nbins * 4, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(vnp_pool_screen_select_kernel, dim3(C / 64, (unsigned)G), dim3(256), 0, st,
                       (const float4*)slots, (const int*)slot_idx, gs / 32, C, gs, head, seg, list);
    hipLaunchKernelGGL(vnp_pool_screen_hist_kernel, dim3(1024), dim3(256), 0, st, (const VnpScreenHead*)head, (const int2*)list, hist);
    hipLaunchKernelGGL(vnp_pool_screen_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)hist, nbins, start, cursor, tstart);
    hipLaunchKernelGGL(vnp_pool_screen_scatter_kernel, dim3(1024), dim3(256), 0, st, (const VnpScreenHead*)head, (const int2*)list, cursor, order);
    // at most cap / 16 + nbins tiles (every bin's last tile may be partial); waves past the real tile count exit at once
    hipLaunchKernelGGL(vnp_pool_screen_exact_kernel, dim3((unsigned)vnp_cdiv(cap / 16 + nbins, 4)), dim3(256), 0, st, x, ldx, Wcat, ldw, K,
                       G * gs, (const VnpScreenHead*)head, (const int2*)list, (const int*)order, (const int*)start, (const int*)tstart, nbins, res);
    hipLaunchKernelGGL(vnp_pool_screen_final_kernel, dim3(vnp_cdiv(C, 4), (unsigned)G), dim3(256), 0, st,
                       (const VnpScreenHead*)head, (const int2*)seg, (const int2*)list, (const float4*)res, C, gs, idx, out, ldo);
    // predicated fallback: the full 6-product pool, runs only if the flag was raised
    VnpGemmArgs q{};
    q.A = x; q.B = Wcat; q.M = (int)M; q.N = 2 * C; q.K = K; q.lda = ldx; q.ldb = ldw; q.rows_per_group = 1;
    q.tiles_per_split = K / 32; q.nsplit = 1; q.nbatch = 1; q.col_group = 8; q.pool_gs = gs;
    q.pool_val = (float*)ws_full;
    q.pool_idx = (int*)((char*)ws_full + nslot * 4);
    q.pool_x = (float*)((char*)ws_full + nslot * 8);
    q.pred = &head->flag;
    hipLaunchKernelGGL((vnp_gemm_split_kernel<192, 128, true, true, true, 0, 1>), grid, block, 0, st, q);
    hipLaunchKernelGGL(vnp_pool_fused_final_kernel, dim3(vnp_cdiv(C, 256), (unsigned)G), dim3(256), 0, st,
                       (const float*)q.pool_val, (const int*)q.pool_idx, (const float*)q.pool_x, gs / 32, C, idx, out, ldo,
                       (const int*)&head->flag);
    return vnp_last_error();
}
