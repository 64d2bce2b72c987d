// Graph construction and gather kernels for the VN-DGCNN / VN-PoinTr encoders: replacements for the two CUDA-only
// third-party packages the reference imports (sources not vendored):
//   KNN_CUDA 0.2            knn(ref, query) -> k nearest indices      (models/dgcnn.py:11,236,257-259)
//   pointnet2_ops           furthest_point_sample, gather_operation   (models/dgcnn.py:210-218, models/pcn.py:6-9)
// Conventions chosen here (the packages' tie behaviour is not pinned by any reference test): squared L2, neighbours
// sorted by (distance, index) ascending, the query itself included when it is in the reference set; FPS starts at
// index 0 and resolves ties to the lowest index.  Point clouds / features are point-major: [B, N, D], D contiguous
// (coordinates [B,N,3]; VN features = the rows layout viewed as [B*N, 3*C]).
#include "vnp_common.h"

// ------------------------------------------------------------------------------------------------ kNN
// One thread per query, k selection passes over the reference cloud: pass p takes the smallest (d, idx) pair that is
// lexicographically greater than the pair chosen in pass p-1.  No per-thread arrays, deterministic order, exact ties.
// The reference tile is staged in LDS ([KNN_TILE][D] floats); the query vector is re-read from LDS as well.
#define KNN_THREADS 128
template <int DT>     // DT = 3: coordinates (registers); DT = 0: generic D through LDS
__global__ void __launch_bounds__(KNN_THREADS) vnp_knn_kernel(const float* __restrict__ ref, int Nr, const float* __restrict__ qry, int Nq,
                                                            int D, int k, int tile, int* __restrict__ idx_out, float* __restrict__ dist_out)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* rt = sm;                                   // [tile][D]
    float* qs = sm + (size_t)tile * D;                // [KNN_THREADS][D] (generic path only)
    const int b = blockIdx.y;
    const int q = blockIdx.x * KNN_THREADS + threadIdx.x;
    const bool live = q < Nq;
    const float* rb = ref + (size_t)b * Nr * D;
    const float* qb = qry + ((size_t)b * Nq + (live ? q : Nq - 1)) * D;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
    if (DT == 3) { q0 = qb[0]; q1 = qb[1]; q2 = qb[2]; }
    else for (int d = 0; d < D; ++d) qs[threadIdx.x * D + d] = qb[d];
    float last_d = -1.f; int last_i = -1;
    for (int p = 0; p < k; ++p) {
        float best = INFINITY; int bi = 0x7fffffff;
        for (int r0 = 0; r0 < Nr; r0 += tile) {
            const int cnt = min(tile, Nr - r0);
            __syncthreads();
            for (int i = threadIdx.x; i < cnt * D; i += KNN_THREADS) rt[i] = rb[(size_t)r0 * D + i];
            __syncthreads();
            for (int i = 0; i < cnt; ++i) {
                float dd;
                if (DT == 3) {
                    const float dx = rt[i * 3] - q0, dy = rt[i * 3 + 1] - q1, dz = rt[i * 3 + 2] - q2;
                    dd = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                } else {
                    dd = 0.f;
                    for (int d = 0; d < D; ++d) { const float t = rt[i * D + d] - qs[threadIdx.x * D + d]; dd = fmaf(t, t, dd); }
                }
                const int gi = r0 + i;
                const bool after = dd > last_d || (dd == last_d && gi > last_i);
                if (after && (dd < best || (dd == best && gi < bi))) { best = dd; bi = gi; }
            }
        }
        last_d = best; last_i = bi;
        if (live) {
            idx_out[((size_t)b * Nq + q) * k + p] = bi == 0x7fffffff ? 0 : bi;
            if (dist_out) dist_out[((size_t)b * Nq + q) * k + p] = best;
        }
    }
}

// Feature-space graphs (D = 3*C up to a few hundred; VN-DGCNN / VN-PoinTr build them in every edge-conv layer): the
// one-thread-per-query kernel above recomputes every distance in each of its k passes with the query vector in LDS, which
// at D = 192, N = 2048 costs 60 ms per call.  Instead: (1) the distance matrix is formed once, tiled like an SGEMM on the
// VALU — a 256-thread block owns a 64 x 64 tile, each thread a 4 x 4 block, both operands staged d-major in LDS so a thread
// reads them as two float4 — with the SAME arithmetic per entry (dd = fmaf(r - q, r - q, dd), d ascending), (2) one wave per
// query row then does the k selection passes over the 8-64 KB row in LDS.  Same indices and distances, bit for bit.
#define KD_T 64
#define KD_DC 32
__global__ void __launch_bounds__(256) vnp_knn_dist_kernel(const float* __restrict__ ref, int Nr, const float* __restrict__ qry, int Nq,
                                                          int D, float* __restrict__ dist)
{
    __shared__ __attribute__((aligned(16))) float qs[KD_DC][KD_T + 4], rs[KD_DC][KD_T + 4];
    const int b = blockIdx.z, q0 = blockIdx.y * KD_T, r0 = blockIdx.x * KD_T;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const float* qb = qry + (size_t)b * Nq * D;
    const float* rb = ref + (size_t)b * Nr * D;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int row = tid >> 2, dc = (tid & 3) * 8;
    for (int d0 = 0; d0 < D; d0 += KD_DC) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = d0 + dc + e;
            // rows / dims outside the problem are zero on BOTH sides: they add fmaf(0, 0, dd) = dd, i.e. nothing
            qs[dc + e][row] = (q0 + row < Nq && d < D) ? qb[(size_t)(q0 + row) * D + d] : 0.f;
            rs[dc + e][row] = (r0 + row < Nr && d < D) ? rb[(size_t)(r0 + row) * D + d] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int dd = 0; dd < KD_DC; ++dd) {
            const float4 qv = *(const float4*)&qs[dd][4 * ty];
            const float4 rv = *(const float4*)&rs[dd][4 * tx];
            const float q4[4] = {qv.x, qv.y, qv.z, qv.w}, r4[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float t = r4[j] - q4[i]; acc[i][j] = fmaf(t, t, acc[i][j]); }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = q0 + 4 * ty + i;
        if (q >= Nq) continue;
        float* drow = dist + ((size_t)b * Nq + q) * Nr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + 4 * tx + j;
            if (r < Nr) drow[r] = acc[i][j];
        }
    }
}

// one wave per query row of the distance matrix: k passes, pass p takes the smallest (d, idx) pair lexicographically greater
// than the pair of pass p-1 (exact ties, ascending (distance, index) order)
__global__ void __launch_bounds__(64) vnp_knn_select_kernel(const float* __restrict__ dist, int Nr, int k, int* __restrict__ idx_out,
                                                           float* __restrict__ dist_out)
{
    extern __shared__ float srow[];
    const size_t row = blockIdx.x;
    const int lane = threadIdx.x;
    const float* d = dist + row * Nr;
    for (int i = lane; i < Nr; i += 64) srow[i] = d[i];
    __syncthreads();
    float last_d = -1.f; int last_i = -1;
    for (int p = 0; p < k; ++p) {
        float best = INFINITY; int bi = 0x7fffffff;
        for (int i = lane; i < Nr; i += 64) {
            const float dd = srow[i];
            const bool after = dd > last_d || (dd == last_d && i > last_i);
            if (after && (dd < best || (dd == best && i < bi))) { best = dd; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
            if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        last_d = best; last_i = bi;
        if (lane == 0) {
            idx_out[row * k + p] = bi == 0x7fffffff ? 0 : bi;
            if (dist_out) dist_out[row * k + p] = best;
        }
    }
}

// The same selection for rows of up to 64 * NPL values with the scan taken out of the k passes: every lane sorts its NPL
// values once by (distance, index) — a bitonic network on registers, all indices static — and parks the sorted column in LDS;
// a pass is then one LDS read of the lane's current head, a wave arg-min, and a pointer bump in the winning lane.
// ~NPL log^2 NPL / 4 compare-exchanges + k * (1 read + 6 shuffle steps) instead of k * NPL dependent compare-selects.
template <int NPL>
__global__ void __launch_bounds__(64) vnp_knn_select_sorted_kernel(const float* __restrict__ dist, int Nr, int k, int* __restrict__ idx_out,
                                                                  float* __restrict__ dist_out)
{
    __shared__ float sd[NPL + 1][64];
    __shared__ int si[NPL + 1][64];
    const size_t row = blockIdx.x;
    const int lane = threadIdx.x;
    const float* d = dist + row * Nr;
    float v[NPL];
    int id[NPL];
#pragma unroll
    for (int u = 0; u < NPL; ++u) { const int i = lane + 64 * u; id[u] = i; v[u] = i < Nr ? d[i] : INFINITY; }
    // bitonic sort ascending by (v, id); NaN distances compare false everywhere and stay where they are, like in the scan kernel
#pragma unroll
    for (int kk = 2; kk <= NPL; kk <<= 1)
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int u = 0; u < NPL; ++u) {
                const int w = u ^ j;
                if (w > u) {
                    const bool up = (u & kk) == 0;
                    const bool gt = v[u] > v[w] || (v[u] == v[w] && id[u] > id[w]);
                    if (gt == up) { const float tv = v[u]; v[u] = v[w]; v[w] = tv; const int ti = id[u]; id[u] = id[w]; id[w] = ti; }
                }
            }
#pragma unroll
    for (int u = 0; u < NPL; ++u) { sd[u][lane] = v[u]; si[u][lane] = id[u] < Nr ? id[u] : 0x7fffffff; }
    sd[NPL][lane] = INFINITY; si[NPL][lane] = 0x7fffffff;          // sentinel: an exhausted lane
    __syncthreads();
    int head = 0;
    for (int p = 0; p < k; ++p) {
        float best = sd[head][lane]; int bi = si[head][lane];
        if (bi == 0x7fffffff) best = INFINITY;
        float wv = best; int wi = bi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(wv, o); const int oi = __shfl_xor(wi, o);
            if (ov < wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
        }
        if (wi == bi && bi != 0x7fffffff) ++head;                  // this lane's head was taken
        if (lane == 0) {
            idx_out[row * k + p] = wi == 0x7fffffff ? 0 : wi;
            if (dist_out) dist_out[row * k + p] = wv;
        }
    }
}

template <int NPL>
__global__ void vnp_knn_select_threshold_kernel(const float* __restrict__ dist, int Nr, int k, int* __restrict__ idx_out, float* __restrict__ dist_out);

// Same contract as vnp_knn with a caller-provided distance matrix: ws of B*Nq*Nr floats.  Any D; Nr <= 16384.
extern "C" int vnp_knn_matrix(hipStream_t st, int B, int D, const float* ref, int Nr, const float* qry, int Nq, int k,
                              int* idx, float* dist, float* ws)
{
    if (!ref || !qry || !idx || !ws || B <= 0 || D <= 0 || Nr <= 0 || Nq <= 0 || k <= 0 || k > Nr || B > 65535 || Nr > 16384) return VNP_EINVAL;
    if (vnp_cdiv(Nq, KD_T) > 65535 || (long long)B * Nq > 2147483647LL) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_knn_dist_kernel, dim3(vnp_cdiv(Nr, KD_T), vnp_cdiv(Nq, KD_T), B), dim3(256), 0, st, ref, Nr, qry, Nq, D, ws);
    static std::atomic<unsigned long long> lds_ok{0};
    hipError_t e = vnp_allow_dynamic_lds(lds_ok, (const void*)vnp_knn_select_kernel, 16384 * 4);
    if (e != hipSuccess) return (int)e;
    const dim3 rows((unsigned)((long long)B * Nq));
    // threshold selection (vnp_knc_select, defined with the coordinate kernel below): k <= 64
    if (k <= 64 && Nr <= 4096) {
        if (Nr <= 512)       hipLaunchKernelGGL(vnp_knn_select_threshold_kernel<8>, rows, dim3(64), 0, st, (const float*)ws, Nr, k, idx, dist);
        else if (Nr <= 1024) hipLaunchKernelGGL(vnp_knn_select_threshold_kernel<16>, rows, dim3(64), 0, st, (const float*)ws, Nr, k, idx, dist);
        else if (Nr <= 2048) hipLaunchKernelGGL(vnp_knn_select_threshold_kernel<32>, rows, dim3(64), 0, st, (const float*)ws, Nr, k, idx, dist);
        else                 hipLaunchKernelGGL(vnp_knn_select_threshold_kernel<64>, rows, dim3(64), 0, st, (const float*)ws, Nr, k, idx, dist);
        return vnp_last_error();
    }
    if (Nr <= 512)       hipLaunchKernelGGL(vnp_knn_select_sorted_kernel<8>, rows, dim3(64), 0, st, (const float*)ws, Nr, k, idx, dist);
    else if (Nr <= 1024) hipLaunchKernelGGL(vnp_knn_select_sorted_kernel<16>, rows, dim3(64), 0, st, (const float*)ws, Nr, k, idx, dist);
    else if (Nr <= 2048) hipLaunchKernelGGL(vnp_knn_select_sorted_kernel<32>, rows, dim3(64), 0, st, (const float*)ws, Nr, k, idx, dist);
    else hipLaunchKernelGGL(vnp_knn_select_kernel, rows, dim3(64), (size_t)Nr * 4, st, (const float*)ws, Nr, k, idx, dist);
    return vnp_last_error();
}

// Coordinate graphs (D = 3, the kNN of every VN-DGCNN layer and of the VN-PoinTr grouper's first stage): one wave per query, the
// sample's reference cloud staged once per workgroup in LDS, NO distance matrix.  Lane l holds the distances to references
// l, l + 64, ... in registers (same fmaf form as vnp_knn_kernel<3>: identical bits).  Selection by a threshold instead of k scans or
// a full sort: the k-th smallest of the 64 LANE MINIMA (one bitonic sort across the wave) is an upper bound T of the k-th smallest
// distance (the k smallest lane minima are k distinct elements <= T); the candidates d <= T — about 1.15 k of them on spread-out
// clouds — are compacted into a wave-private LDS list, sorted by (distance, index) across the wave and the first k written out: exact
// ties included, the same (distance, index) order as the other kernels.  More than 64 candidates (many exact ties) or fewer than k
// (NaN coordinates) take k selection passes over the registers, the semantics of vnp_knn_select_kernel.
#define KNC_WAVES 4
#define KNC_QPW 16           // queries per wave and workgroup at most: the 24 KB staging of a 2048-point cloud is shared by 64 queries
                             // (fewer when the graph is small: a wave walks its queries one after the other, and 128-point graphs left
                             // the chip to 64 workgroups of 16 dependent rounds each)
// one compare-exchange step of a bitonic network across the wave on (value, index) pairs: lane keeps the smaller pair iff keep_min
__device__ __forceinline__ void vnp_knc_cswap(float& d, int& i, int j, bool keep_min)
{
    const float od = __shfl_xor(d, j); const int oi = __shfl_xor(i, j);
    const bool other_less = od < d || (od == d && oi < i);
    if (other_less == keep_min) { d = od; i = oi; }
}
// Threshold selection of the k smallest of a wave's 64 * NPL register-resident values v[u] (element index lane + 64 u; INFINITY past Nr),
// m = the lane's minimum: see vnp_knn_coords_kernel.  wd / wi: 64-entry wave-private LDS lists.  Writes io[0..k) (and dio) in
// (value, index) ascending order.
template <int NPL>
__device__ __forceinline__ void vnp_knc_select(const float (&v)[NPL], float m, int Nr, int k, int lane, float* wd, int* wi, int* io, float* dio)
{
    // T = the k-th smallest of the 64 lane minima (k <= 64), by a bit-wise search instead of a sort (round 5): the minima are non-negative
    // floats (or +inf), so their bit patterns order like the values, and T's bits are the largest pattern P with fewer than k minima
    // strictly below it — 31 steps of one wave-wide compare + a scalar population count, no cross-lane permute (the 21-step bitonic
    // network this replaces was a chain of 42 dependent LDS-crossbar permutes per query)
    const unsigned mb = __float_as_uint(m);
    unsigned tb = 0u;
#pragma unroll
    for (int bit = 30; bit >= 0; --bit) {
        const unsigned cand = tb | (1u << bit);
        if (__popcll(__ballot(mb < cand)) < k) tb = cand;
    }
    const float T = __uint_as_float(tb);
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < NPL; ++u) cnt += (v[u] <= T) ? 1 : 0;
    int incl = cnt;                                        // inclusive prefix sum over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    const int total = __shfl(incl, 63);
        if (total >= k && total <= 64) {
        int pos = incl - cnt;
#pragma unroll
        for (int u = 0; u < NPL; ++u)
            if (v[u] <= T) { wd[pos] = v[u]; wi[pos] = lane + 64 * u; ++pos; }
        __builtin_amdgcn_wave_barrier();
        float cd = lane < total ? wd[lane] : INFINITY;
        int ci = lane < total ? wi[lane] : 0x7fffffff;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 2; kk <= 64; kk <<= 1)
#pragma unroll
            for (int j = kk >> 1; j > 0; j >>= 1)
                vnp_knc_cswap(cd, ci, j, ((lane & j) == 0) == ((lane & kk) == 0));
        if (lane < k) {
            io[lane] = ci;
            if (dio) dio[lane] = cd;
        }
    } else {
        float last_d = -1.f; int last_i = -1;
        for (int p = 0; p < k; ++p) {
            float best = INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int u = 0; u < NPL; ++u) {
                const int i = lane + 64 * u;
                const float dd = v[u];
                const bool after = dd > last_d || (dd == last_d && i > last_i);
                if (i < Nr && after && (dd < best || (dd == best && i < bi))) { best = dd; bi = i; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
                if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            last_d = best; last_i = bi;
            if (lane == 0) {
                io[p] = bi == 0x7fffffff ? 0 : bi;
                if (dio) dio[p] = best;
            }
        }
    }
}

template <int NPL>
__global__ void __launch_bounds__(64 * KNC_WAVES) vnp_knn_coords_kernel(const float* __restrict__ ref, int Nr, const float* __restrict__ qry, int Nq,
                                                                       int k, int qpw, int* __restrict__ idx_out, float* __restrict__ dist_out)
{
    extern __shared__ __attribute__((aligned(16))) float knc_sm[];
    float* rs = knc_sm;                                        // [Nr][3]
    float* ld = knc_sm + (size_t)NPL * 64 * 3;                 // [KNC_WAVES][64] candidate distances
    int* li = (int*)(ld + KNC_WAVES * 64);                     // [KNC_WAVES][64] candidate indices
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* rb = ref + (size_t)b * Nr * 3;
    for (int i = threadIdx.x; i < Nr * 3; i += 64 * KNC_WAVES) rs[i] = rb[i];
    __syncthreads();
    float* wd = ld + wave * 64;
    int* wi = li + wave * 64;
    for (int it = 0; it < qpw; ++it) {
        const int q = (blockIdx.x * qpw + it) * KNC_WAVES + wave;
        if (q >= Nq) break;                                    // wave-uniform; no workgroup barrier below
        const float* qp = qry + ((size_t)b * Nq + q) * 3;
        const float q0 = qp[0], q1 = qp[1], q2 = qp[2];
        float v[NPL];
        float m = INFINITY;
#pragma unroll
        for (int u = 0; u < NPL; ++u) {
            const int i = lane + 64 * u;
            float dd = INFINITY;
            if (i < Nr) {
                const float dx = rs[i * 3] - q0, dy = rs[i * 3 + 1] - q1, dz = rs[i * 3 + 2] - q2;
                dd = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            }
            v[u] = dd;
            m = fminf(m, dd);
        }
        vnp_knc_select<NPL>(v, m, Nr, k, lane, wd, wi, idx_out + ((size_t)b * Nq + q) * k, dist_out ? dist_out + ((size_t)b * Nq + q) * k : nullptr);
    }
}

// the same selection for rows of a precomputed distance matrix (feature-space graphs, any D): one wave per row
template <int NPL>
__global__ void __launch_bounds__(64) vnp_knn_select_threshold_kernel(const float* __restrict__ dist, int Nr, int k, int* __restrict__ idx_out,
                                                                     float* __restrict__ dist_out)
{
    __shared__ float wd[64];
    __shared__ int wi[64];
    const size_t row = blockIdx.x;
    const int lane = threadIdx.x;
    const float* d = dist + row * Nr;
    float v[NPL];
    float m = INFINITY;
#pragma unroll
    for (int u = 0; u < NPL; ++u) { const int i = lane + 64 * u; v[u] = i < Nr ? d[i] : INFINITY; m = fminf(m, v[u]); }
    vnp_knc_select<NPL>(v, m, Nr, k, lane, wd, wi, idx_out + row * k, dist_out ? dist_out + row * k : nullptr);
}

template <int NPL>
static int vnp_knn_coords_launch(hipStream_t st, int B, const float* ref, int Nr, const float* qry, int Nq, int k, int* idx, float* dist)
{
    static std::atomic<unsigned long long> ok{0};
    const int lds = NPL * 64 * 3 * 4 + KNC_WAVES * 64 * 8;
    hipError_t e = vnp_allow_dynamic_lds(ok, (const void*)vnp_knn_coords_kernel<NPL>, lds);
    if (e != hipSuccess) return (int)e;
    // queries per wave: as many as keep ~1024 workgroups on the chip, at most KNC_QPW
    long long qpw = (long long)B * Nq / (KNC_WAVES * 1024);
    qpw = qpw < 1 ? 1 : qpw > KNC_QPW ? KNC_QPW : qpw;
    hipLaunchKernelGGL((vnp_knn_coords_kernel<NPL>), dim3(vnp_cdiv(Nq, KNC_WAVES * (int)qpw), B), dim3(64 * KNC_WAVES), lds, st,
                       ref, Nr, qry, Nq, k, (int)qpw, idx, dist);
    return vnp_last_error();
}

// idx [B][Nq][k] (indices local to the sample), dist (optional) squared distances.  k <= Nr.
extern "C" int vnp_knn(hipStream_t st, int B, int D, const float* ref, int Nr, const float* qry, int Nq, int k,
                       int* idx, float* dist)
{
    if (!ref || !qry || !idx || B <= 0 || D <= 0 || Nr <= 0 || Nq <= 0 || k <= 0 || k > Nr || B > 65535) return VNP_EINVAL;
    if (D == 3 && Nr <= 4096 && k <= 64) {
        // one wave per query on register-resident distances, threshold selection (vnp_knn_coords_kernel)
        if (Nr <= 512)  return vnp_knn_coords_launch<8>(st, B, ref, Nr, qry, Nq, k, idx, dist);
        if (Nr <= 1024) return vnp_knn_coords_launch<16>(st, B, ref, Nr, qry, Nq, k, idx, dist);
        if (Nr <= 2048) return vnp_knn_coords_launch<32>(st, B, ref, Nr, qry, Nq, k, idx, dist);
        return vnp_knn_coords_launch<64>(st, B, ref, Nr, qry, Nq, k, idx, dist);
    }
    if (D == 3) {
        const int tile = 1024;
        hipLaunchKernelGGL((vnp_knn_kernel<3>), dim3(vnp_cdiv(Nq, KNN_THREADS), B), dim3(KNN_THREADS), (size_t)tile * 3 * 4, st,
                           ref, Nr, qry, Nq, D, k, tile, idx, dist);
    } else {
        if (D > 320) return VNP_EINVAL;                      // [128][D] query copy + tile must fit the 160 KB LDS
        int tile = (160 * 1024 / 4 - KNN_THREADS * D) / D;
        if (tile > 256) tile = 256;
        if (tile < 1) return VNP_EINVAL;
        const size_t lds = ((size_t)tile * D + (size_t)KNN_THREADS * D) * 4;
        hipError_t e = hipFuncSetAttribute((const void*)vnp_knn_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((vnp_knn_kernel<0>), dim3(vnp_cdiv(Nq, KNN_THREADS), B), dim3(KNN_THREADS), lds, st,
                           ref, Nr, qry, Nq, D, k, tile, idx, dist);
    }
    return vnp_last_error();
}

// ------------------------------------------------------------------------------------------------ FPS
// One 1024-thread block per sample; each thread keeps the running min-distance of up to FPS_PER_THREAD points in
// registers; every round: update against the last pick, block arg-max (wave shuffles + one LDS exchange).
#define FPS_THREADS 1024
#define FPS_MAX_PER_THREAD 16   // N <= 16384
// FPS_PER_THREAD = ceil(N / 1024) rounded up to 2, 4, 8 or 16: the per-round update loop is fully unrolled, so a 2048-point
// cloud must not pay for 16 slots
// SKIP: the convention of the package the reference calls (pointnet2_ops furthest_point_sampling_kernel: `if (mag <= 1e-3)
// continue;`): a candidate with |p|^2 <= 1e-3 is neither selected nor has its running distance updated; point 0 is still the
// first sample, and if every candidate is skipped the pick is index 0.
template <int FPS_PER_THREAD, bool SKIP>
__global__ void __launch_bounds__(FPS_THREADS) vnp_fps_kernel(const float* __restrict__ xyz, int N, int M, int* __restrict__ out)
{
    __shared__ float sv[FPS_THREADS / 64];
    __shared__ int si[FPS_THREADS / 64];
    __shared__ int pick;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = xyz + (size_t)b * N * 3;
    float px[FPS_PER_THREAD], py[FPS_PER_THREAD], pz[FPS_PER_THREAD], md[FPS_PER_THREAD];
#pragma unroll
    for (int u = 0; u < FPS_PER_THREAD; ++u) {
        const int i = tid + u * FPS_THREADS;
        if (i < N) { px[u] = p[i * 3]; py[u] = p[i * 3 + 1]; pz[u] = p[i * 3 + 2]; md[u] = 1e10f; }
        else { px[u] = py[u] = pz[u] = 0.f; md[u] = -1.f; }
        if (SKIP && i < N && px[u] * px[u] + py[u] * py[u] + pz[u] * pz[u] <= 1e-3f) md[u] = -1.f;     // never a candidate
    }
    int cur = 0;
    if (tid == 0) out[(size_t)b * M] = 0;
    for (int r = 1; r < M; ++r) {
        const float cx = p[cur * 3], cy = p[cur * 3 + 1], cz = p[cur * 3 + 2];
        float best = -1.f; int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < FPS_PER_THREAD; ++u) {
            const int i = tid + u * FPS_THREADS;
            if (i < N) {
                const float dx = px[u] - cx, dy = py[u] - cy, dz = pz[u] - cz;
                const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                md[u] = fminf(md[u], d);                          // a skipped point stays at -1 (d >= 0)
                if (md[u] > best) { best = md[u]; bi = i; }       // ascending i within a thread: first max kept
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
        __syncthreads();
        if (tid < 64) {
            float v = tid < FPS_THREADS / 64 ? sv[tid] : -2.f; int i = tid < FPS_THREADS / 64 ? si[tid] : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(v, o); const int oi = __shfl_xor(i, o);
                if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
            }
            if (tid == 0) { if (i == 0x7fffffff) i = 0; pick = i; out[(size_t)b * M + r] = i; }
        }
        __syncthreads();
        cur = pick;
    }
}

// Clouds of up to 4096 points (every FPS call of the networks): FOUR waves per sample instead of sixteen, the cloud's coordinates in LDS
// (the last pick's coordinates are an LDS broadcast read, not a dependent global load) and ONE barrier per round: a wave publishes its
// (best, index) in the slot set of the round's parity and every thread reduces the four slots itself, so the next round can start
// without a second barrier (a wave cannot get more than one round ahead of the slowest one, which is what the two slot sets cover).
// Same arithmetic, same tie rule (largest running distance, then smallest index), same picks as the kernel above.
#define FPS_SMALL_THREADS 256
#define FPS_SMALL_MAXN 4096
template <int PPT, bool SKIP, int THREADS = FPS_SMALL_THREADS>
__global__ void __launch_bounds__(THREADS) vnp_fps_small_kernel(const float* __restrict__ xyz, int N, int M, int* __restrict__ out)
{
    __shared__ float sx[FPS_SMALL_MAXN], sy[FPS_SMALL_MAXN], sz[FPS_SMALL_MAXN];
    __shared__ float sv[2][THREADS / 64];
    __shared__ int si[2][THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = xyz + (size_t)b * N * 3;
    float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const int i = tid + u * THREADS;
        if (i < N) { px[u] = p[i * 3]; py[u] = p[i * 3 + 1]; pz[u] = p[i * 3 + 2]; md[u] = 1e10f; sx[i] = px[u]; sy[i] = py[u]; sz[i] = pz[u]; }
        else { px[u] = py[u] = pz[u] = 0.f; md[u] = -1.f; }
        if (SKIP && i < N && px[u] * px[u] + py[u] * py[u] + pz[u] * pz[u] <= 1e-3f) md[u] = -1.f;     // never a candidate
    }
    int cur = 0;
    if (tid == 0) out[(size_t)b * M] = 0;
    __syncthreads();
    for (int r = 1; r < M; ++r) {
        const float cx = sx[cur], cy = sy[cur], cz = sz[cur];
        float best = -1.f; int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            // no bounds test: a slot past N holds md = -1 and coordinates 0, so it stays at -1 (d >= 0) and never beats `best`
            const int i = tid + u * THREADS;
            const float dx = px[u] - cx, dy = py[u] - cy, dz = pz[u] - cz;
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            md[u] = fminf(md[u], d);                              // a skipped point stays at -1 (d >= 0)
            if (md[u] > best) { best = md[u]; bi = i; }           // ascending i within a thread: first max kept
        }
        // wave arg-max on DPP: the largest running distance, then the smallest index among the lanes that hold it
        const float wbest = vnp_wave_max_f32(best);
        const int wbi = vnp_wave_min_i32(best == wbest ? bi : 0x7fffffff);
        float v = wbest; int i = wbi;
        if (THREADS > 64) {                                       // several waves: their winners meet in LDS (one barrier per round)
            const int par = r & 1;
            if ((tid & 63) == 0) { sv[par][tid >> 6] = wbest; si[par][tid >> 6] = wbi; }
            __syncthreads();
            v = sv[par][0]; i = si[par][0];
#pragma unroll
            for (int w = 1; w < THREADS / 64; ++w) {
                const float ov = sv[par][w]; const int oi = si[par][w];
                if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
            }
        }
        if (i == 0x7fffffff) i = 0;
        if (tid == 0) out[(size_t)b * M + r] = i;
        cur = i;
    }
}

template <bool SKIP>
static int vnp_fps_launch(hipStream_t st, int B, int N, const float* xyz, int M, int* idx)
{
    if (N <= 512) {
        // one wave per cloud, up to 8 points per lane: no barrier and no LDS exchange per round (measured inside config 3's step: 512 -> 128
        // picks 77 -> 50 us; at 2048 points the 32 points per lane cost more than the exchange saves: 550 against 459 us, so from there on four waves)
        if (N <= 128) hipLaunchKernelGGL((vnp_fps_small_kernel<2, SKIP, 64>), dim3(B), dim3(64), 0, st, xyz, N, M, idx);
        else          hipLaunchKernelGGL((vnp_fps_small_kernel<8, SKIP, 64>), dim3(B), dim3(64), 0, st, xyz, N, M, idx);
        return vnp_last_error();
    }
    if (N <= FPS_SMALL_MAXN) {
        if (N <= 2 * FPS_SMALL_THREADS)      hipLaunchKernelGGL((vnp_fps_small_kernel<2, SKIP>), dim3(B), dim3(FPS_SMALL_THREADS), 0, st, xyz, N, M, idx);
        else if (N <= 4 * FPS_SMALL_THREADS) hipLaunchKernelGGL((vnp_fps_small_kernel<4, SKIP>), dim3(B), dim3(FPS_SMALL_THREADS), 0, st, xyz, N, M, idx);
        else if (N <= 8 * FPS_SMALL_THREADS) hipLaunchKernelGGL((vnp_fps_small_kernel<8, SKIP>), dim3(B), dim3(FPS_SMALL_THREADS), 0, st, xyz, N, M, idx);
        else                                 hipLaunchKernelGGL((vnp_fps_small_kernel<16, SKIP>), dim3(B), dim3(FPS_SMALL_THREADS), 0, st, xyz, N, M, idx);
        return vnp_last_error();
    }
    if (N <= 2 * FPS_THREADS)      hipLaunchKernelGGL((vnp_fps_kernel<2, SKIP>), dim3(B), dim3(FPS_THREADS), 0, st, xyz, N, M, idx);
    else if (N <= 4 * FPS_THREADS) hipLaunchKernelGGL((vnp_fps_kernel<4, SKIP>), dim3(B), dim3(FPS_THREADS), 0, st, xyz, N, M, idx);
    else if (N <= 8 * FPS_THREADS) hipLaunchKernelGGL((vnp_fps_kernel<8, SKIP>), dim3(B), dim3(FPS_THREADS), 0, st, xyz, N, M, idx);
    else                           hipLaunchKernelGGL((vnp_fps_kernel<16, SKIP>), dim3(B), dim3(FPS_THREADS), 0, st, xyz, N, M, idx);
    return vnp_last_error();
}

// furthest_point_sample(xyz[B,N,3], M) -> idx [B][M]; N <= 16384.  Plain FPS: every point is a candidate.
extern "C" int vnp_fps(hipStream_t st, int B, int N, const float* xyz, int M, int* idx)
{
    if (!xyz || !idx || B <= 0 || N <= 0 || M <= 0 || M > N || N > FPS_THREADS * FPS_MAX_PER_THREAD) return VNP_EINVAL;
    return vnp_fps_launch<false>(st, B, N, xyz, M, idx);
}

// flags bit 0 (VNP_FPS_SKIP_NEAR_ORIGIN): pointnet2_ops' convention — candidates with |p|^2 <= 1e-3 are never selected
extern "C" int vnp_fps_ex(hipStream_t st, int B, int N, const float* xyz, int M, int* idx, int flags)
{
    if (!xyz || !idx || B <= 0 || N <= 0 || M <= 0 || M > N || N > FPS_THREADS * FPS_MAX_PER_THREAD || (flags & ~1)) return VNP_EINVAL;
    return (flags & 1) ? vnp_fps_launch<true>(st, B, N, xyz, M, idx) : vnp_fps_launch<false>(st, B, N, xyz, M, idx);
}

// ------------------------------------------------------------------------------------------------ gathers
// rows layout: out point i (3 rows) <- source point src[i];  grid (ceil(W/256), Pout)
__global__ void vnp_gather_points_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ src, int W,
                                         float* __restrict__ out, int ldo)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    const size_t i = blockIdx.y, s = (size_t)src[i] * 3;
#pragma unroll
    for (int t = 0; t < 3; ++t) out[(i * 3 + t) * ldo + c] = x[(s + t) * ldx + c];
}

// gx[s] = sum of g[i] over the i with src[i] == s, as a GATHER over the reverse adjacency (start, rev from vnp_edge_reverse: the i of a
// source ascending): summed in list order in double, one plain store per element, zeros for a source nobody points at — no atomics, no
// zero fill, the same bits on every run (rounds 1-3: float atomics in arrival order).       grid (ceil(W/256), Pin)
__global__ void vnp_gather_sum_rev_kernel(const float* __restrict__ g, int ldg, const int* __restrict__ start, const int* __restrict__ rev,
                                          int W, float* __restrict__ gx, int ldgx)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    const size_t s = blockIdx.y;
    const int b0 = start[s], b1 = start[s + 1];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int e = b0; e < b1; ++e) {
        const float* r = g + (size_t)rev[e] * 3 * ldg + c;
        a0 += (double)r[0]; a1 += (double)r[ldg]; a2 += (double)r[2 * (size_t)ldg];
    }
    float* w = gx + s * 3 * ldgx + c;
    w[0] = (float)a0; w[ldgx] = (float)a1; w[2 * (size_t)ldgx] = (float)a2;
}

// Edge tensor of a kNN graph with the EdgeConv weights already applied (see ops.EdgeGatherAddFn):
//   out[(p*k + j)*3 + t][c] = u[nbr[p*k+j]*3 + t][c] + v[p*3 + t][c]        grid (ceil(W/256), P*k)
__global__ void vnp_edge_gather_add_kernel(const float* __restrict__ u, int ldu, const float* __restrict__ v, int ldv,
                                           const int* __restrict__ nbr, int k, int W, float* __restrict__ out, int ldo)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    const size_t e = blockIdx.y, p = e / k, s = (size_t)nbr[e];
#pragma unroll
    for (int t = 0; t < 3; ++t) out[(e * 3 + t) * ldo + c] = u[(s * 3 + t) * ldu + c] + v[(p * 3 + t) * ldv + c];
}

// Sampled coordinates and global point ids of an FPS pick in one launch (instead of arange / mul / add / expand / gather):
//   out[b][i][:] = xyz[b][idx[b][i]][:],  gid[b * M + i] = b * N + idx[b][i]  (gid may be null).   grid ceil(B*M / 256)
__global__ void __launch_bounds__(256) vnp_gather_coords_kernel(const float* __restrict__ xyz, int N, const int* __restrict__ idx, int M, long long total,
                                                                float* __restrict__ out, int* __restrict__ gid)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long long b = e / M;
    const long long s = b * N + idx[e];
    const float* src = xyz + s * 3;
    out[e * 3 + 0] = src[0]; out[e * 3 + 1] = src[1]; out[e * 3 + 2] = src[2];
    if (gid) gid[e] = (int)s;
}

extern "C" int vnp_gather_coords(hipStream_t st, int B, int N, const float* xyz, const int* idx, int M, float* out, int* gid)
{
    if (!xyz || !idx || !out || B <= 0 || N <= 0 || M <= 0 || (long long)B * N > 2147483647LL) return VNP_EINVAL;
    const long long total = (long long)B * M;
    hipLaunchKernelGGL(vnp_gather_coords_kernel, dim3((unsigned)vnp_cdiv(total, 256)), dim3(256), 0, st, xyz, N, idx, M, total, out, gid);
    return vnp_last_error();
}

// [Wu; Wv] [4C'][C] of an EdgeConv layer from its VNLinearLeakyReLU weights Wf, Wd [C'][2C] (input channels: C of feature - x, then C of x;
// models/dgcnn.py:250-278): W cat(f_j - x, x) = W1 f_j + (W2 - W1) x, so Wu = [Wf1; Wd1] acts on the neighbour and Wv = [Wf2 - Wf1; Wd2 - Wd1]
// on the centre.  dir = 0: out <- (Wf, Wd); dir = 1: the adjoint, (gWf, gWd) <- g [4C'][C]: d/dW1 = gu - gv, d/dW2 = gv.  One launch each
// (five ATen launches forward and four backward per layer before).
__global__ void __launch_bounds__(256) vnp_edge_weights_kernel(float* __restrict__ Wf, float* __restrict__ Wd, float* __restrict__ out, int Cp, int C, int dir)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;             // over [2][Cp][C]: (which weight, output channel, column)
    if (e >= 2LL * Cp * C) return;
    const int c = (int)(e % C), o = (int)((e / C) % Cp), w = (int)(e / ((long long)C * Cp));
    float* W = w ? Wd : Wf;
    const size_t i1 = (size_t)o * 2 * C + c, i2 = i1 + C;                     // W1 = W[:, :C], W2 = W[:, C:]
    const size_t iu = ((size_t)w * Cp + o) * C + c, iv = iu + (size_t)2 * Cp * C;
    if (dir == 0) { const float a = W[i1]; out[iu] = a; out[iv] = W[i2] - a; }
    else          { const float gv = out[iv]; W[i1] = out[iu] - gv; W[i2] = gv; }
}

extern "C" int vnp_edge_weights(hipStream_t st, int dir, float* Wf, float* Wd, float* uv, int Cp, int C)
{
    if (!Wf || !Wd || !uv || Cp <= 0 || C <= 0 || (dir != 0 && dir != 1)) return VNP_EINVAL;
    hipLaunchKernelGGL(vnp_edge_weights_kernel, dim3((unsigned)vnp_cdiv(2LL * Cp * C, 256)), dim3(256), 0, st, Wf, Wd, uv, Cp, C, dir);
    return vnp_last_error();
}

// out [Pout*3][ldo] : point i copies point src[i] (global point ids) of x
extern "C" int vnp_gather_points(hipStream_t st, const float* x, int ldx, const int* src, long long Pout, int W, float* out, int ldo)
{
    if (!x || !src || !out || Pout <= 0 || W <= 0) return VNP_EINVAL;
    for (long long p0 = 0; p0 < Pout; p0 += 65535) {
        const int n = (int)((Pout - p0) < 65535 ? (Pout - p0) : 65535);
        hipLaunchKernelGGL(vnp_gather_points_kernel, dim3(vnp_cdiv(W, 256), n), dim3(256), 0, st, x, ldx, src + p0, W,
                           out + (size_t)p0 * 3 * ldo, ldo);
    }
    return vnp_last_error();
}

extern "C" int vnp_edge_reverse(hipStream_t st, const int* nbr, long long Pq, int k, long long Pu, int* start, int* rev, int* ws);

// gx [Pin*3][ldgx]: gx[s] = sum of g[i] over the i < Pout with src[i] == s (every row of gx is written).
// ws: 2 * (Pin + 1) + 2 * Pout ints (the reverse adjacency of src and its scratch).
extern "C" int vnp_scatter_add_points(hipStream_t st, const float* g, int ldg, const int* src, long long Pout, int W,
                                      float* gx, int ldgx, long long Pin, int* ws)
{
    if (!g || !src || !gx || !ws || Pout <= 0 || W <= 0 || Pin <= 0) return VNP_EINVAL;
    int* start = ws;
    int* rev = ws + (Pin + 1);
    int* scratch = rev + Pout;
    int rc = vnp_edge_reverse(st, src, Pout, 1, Pin, start, rev, scratch);
    if (rc) return rc;
    for (long long p0 = 0; p0 < Pin; p0 += 65535) {
        const int n = (int)((Pin - p0) < 65535 ? (Pin - p0) : 65535);
        hipLaunchKernelGGL(vnp_gather_sum_rev_kernel, dim3(vnp_cdiv(W, 256), n), dim3(256), 0, st, g, ldg, (const int*)start + p0,
                           (const int*)rev, W, gx + (size_t)p0 * 3 * ldgx, ldgx);
    }
    return vnp_last_error();
}

extern "C" int vnp_edge_gather_add(hipStream_t st, const float* u, int ldu, const float* v, int ldv, const int* nbr,
                                   long long P, int k, int W, float* out, int ldo)
{
    if (!u || !v || !nbr || !out || P <= 0 || k <= 0 || W <= 0) return VNP_EINVAL;
    const long long E = P * k;
    const long long step = (65535 / k) * (long long)k;          // whole points per launch so that p = e / k stays valid
    for (long long e0 = 0; e0 < E; e0 += step) {
        const int n = (int)((E - e0) < step ? (E - e0) : step);
        hipLaunchKernelGGL(vnp_edge_gather_add_kernel, dim3(vnp_cdiv(W, 256), n), dim3(256), 0, st, u, ldu,
                           v + (size_t)(e0 / k) * 3 * ldv, ldv, nbr + e0, k, W, out + (size_t)e0 * 3 * ldo, ldo);
    }
    return vnp_last_error();
}
