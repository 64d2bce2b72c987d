// Chamfer distance for 3-D point clouds: replaces the reference's NmDistanceKernel / NmDistanceGradKernel
// (extensions/chamfer_distance/chamfer3D.cu:12-134, 155-174).
//
// Forward, one direction (queries q[B,n,3] against references r[B,m,3]): squared distance to, and index of, the
// nearest reference; direct fp32 (r-q)^2 form written as fmaf(z,z,fmaf(y,y,x*x)); the FIRST minimum wins.
// The reference launches 32x16 blocks that each walk the whole reference cloud.  Here the work is cut in both
// dimensions: a block owns QPT*256 queries (held in registers) and ONE slice of the reference cloud, staged once
// into LDS as padded float4 (every lane reads the same address -> LDS broadcast, no bank conflicts), and merges
// its result into a packed 64-bit word per query with atomicMin:  (float bits of d) << 32 | index.  d >= 0, so the
// unsigned order of the bits is the order of the floats, and among equal distances the smaller index wins —
// exactly the reference's strict '<' scan.  A second tiny kernel unpacks (dist, idx).
// This is VALU-bound (~9.5 lane-op-equivalents per pair), not HBM-bound: ~12 B per point are read once per block.
#include "vnp_common.h"
#include <type_traits>

#define CH_THREADS 256
#define CH_QPT 4                 // queries per thread
#define CH_TILE 1024             // reference points staged per LDS tile (16 KB)

// Measured notes on the inner loop (kept for the next reader): (1) packing the arithmetic (v_pk_add/mul/fma_f32) halves the
// instruction count but not the time when the pair was two REFERENCE points (the extra v_mov splats made it 8 % slower); pairing two
// QUERIES needs no splats and is what the loop below does (round 2); (2) tracking only chunk minima (v_min3) and re-scanning
// a chunk when some lane improved was slower too: with 256 queries per wave some lane improves in 40-60 % of 8-reference
// chunks.  The plain form below — 3 sub, 1 mul, 2 fma, 1 cmp, 2 select per pair, 4 queries per thread sharing each LDS
// broadcast read — runs 18.3 G pairs/step in ~3.1 ms = ~70 % of the VALU issue bound.
__global__ void __launch_bounds__(CH_THREADS) vnp_chamfer_nn_kernel(const float* __restrict__ q, int n, const float* __restrict__ r, int m,
                                                                   int slice, unsigned long long* __restrict__ packed)
{
    __shared__ float4 tile[CH_TILE];
    const int b = blockIdx.z;
    const float* qb = q + (size_t)b * n * 3;
    const float* rb = r + (size_t)b * m * 3;
    const int k_begin = blockIdx.y * slice;
    const int k_end = min(m, k_begin + slice);

    float qx[CH_QPT], qy[CH_QPT], qz[CH_QPT], best[CH_QPT];
    int bi[CH_QPT];
    const int j0 = blockIdx.x * (CH_THREADS * CH_QPT) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < CH_QPT; ++u) {
        const int j = min(j0 + u * CH_THREADS, n - 1);      // clamp: out-of-range lanes redo the last point, never stored
        qx[u] = qb[j * 3 + 0]; qy[u] = qb[j * 3 + 1]; qz[u] = qb[j * 3 + 2];
        best[u] = INFINITY; bi[u] = 0;
    }
    for (int k0 = k_begin; k0 < k_end; k0 += CH_TILE) {
        const int cnt = min(CH_TILE, k_end - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += CH_THREADS) {
            const float* s = rb + (size_t)(k0 + i) * 3;
            tile[i] = make_float4(s[0], s[1], s[2], 0.f);
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < cnt; ++i) {
            const float4 t = tile[i];
            static_assert(CH_QPT % 2 == 0, "queries are evaluated in pairs");
#pragma unroll
            for (int u = 0; u < CH_QPT; u += 2) {
                // two queries per packed instruction (the operands of a pair are loop-invariant register pairs: no splats); each
                // distance is the same fmaf chain as before
                const vnp_f2 dx = t.x - vnp_f2{qx[u], qx[u + 1]}, dy = t.y - vnp_f2{qy[u], qy[u + 1]}, dz = t.z - vnp_f2{qz[u], qz[u + 1]};
                const vnp_f2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                if (d[0] < best[u]) { best[u] = d[0]; bi[u] = k0 + i; }
                if (d[1] < best[u + 1]) { best[u + 1] = d[1]; bi[u + 1] = k0 + i; }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < CH_QPT; ++u) {
        const int j = j0 + u * CH_THREADS;
        if (j < n && k_begin < k_end) {
            const unsigned long long v = ((unsigned long long)__float_as_uint(best[u]) << 32) | (unsigned)bi[u];
            atomicMin(&packed[(size_t)b * n + j], v);
        }
    }
}

__global__ void vnp_chamfer_unpack_kernel(const unsigned long long* __restrict__ packed, long long total,
                                          float* __restrict__ dist, int* __restrict__ idx)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const unsigned long long v = packed[i];
    dist[i] = __uint_as_float((unsigned)(v >> 32));
    idx[i] = (int)(unsigned)(v & 0xffffffffu);
}

// One direction of NmDistanceGradKernel (chamfer3D.cu:155-174): with v = 2*grad_dist[j]*(x1[j] - x2[idx[j]]),
//   own[j] (+)= v   and   other[idx[j]] -= v.
// Point j belongs to exactly one thread, so its own term is a plain store (FIRST pass: `own` need not be zeroed) or a plain
// read-modify-write (second pass, after the first pass's scatter into the same array has completed at the kernel boundary);
// only the scatter into the other cloud needs atomics.
template <bool FIRST>
__global__ void vnp_chamfer_grad_kernel(int n, const float* __restrict__ xyz1, int m, const float* __restrict__ xyz2,
                                        const float* __restrict__ grad_dist, const int* __restrict__ idx,
                                        float* __restrict__ own, float* __restrict__ other)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const size_t p = ((size_t)b * n + j) * 3;
    const int j2 = idx[(size_t)b * n + j];
    const size_t qq = ((size_t)b * m + j2) * 3;
    const float g = grad_dist[(size_t)b * n + j] * 2.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const float v = g * (xyz1[p + t] - xyz2[qq + t]);
        own[p + t] = FIRST ? v : own[p + t] + v;
        atomicAdd(&other[qq + t], -v);
    }
}

// The scatter half of the backward without global atomics and INDEPENDENT OF ARRIVAL ORDER.  Workgroup (part, b, direction) owns
// CHG_TARGETS consecutive points of the cloud that RECEIVES the scatter, scans the nearest-neighbour indices of all points of the other
// cloud and adds  -2 g_j (q_j - r_idx[j])  for those that land in its range into LDS accumulators, then adds the sums to the points' own
// terms, which vnp_chamfer_own_kernel has stored.  The accumulators are 64-bit INTEGERS: a first scan finds the largest |term| landing
// in the workgroup's range (integer atomic max on the float's bits), 2^e above it; the second scan adds each term as
// rint(term * 2^(40 - e)) with ds_add_u64.  Integer addition is associative, so the sum is the same whatever order the lanes arrive in:
// run-to-run identical bits (rounds 1-3 summed in double, exact only up to rounding ties; before that 1.5 M global float atomics, 0.67 ms
// per step).  A term carries 40 bits below the largest one — an absolute error of 2^-41 of it per term, four orders below the fp32
// rounding of the result — and <= 32768 terms stay below 2^56.
#define CHG_TARGETS 2048
__global__ void __launch_bounds__(1024) vnp_chamfer_scatter_kernel(int n, const float* __restrict__ xyz1, int m, const float* __restrict__ xyz2,
                                                                   const float* __restrict__ gd1, const int* __restrict__ idx1,
                                                                   const float* __restrict__ gd2, const int* __restrict__ idx2,
                                                                   float* __restrict__ g1, float* __restrict__ g2)
{
    __shared__ unsigned long long acc[CHG_TARGETS * 3];
    __shared__ unsigned maxbits;
    const int b = blockIdx.y, dir = blockIdx.z;
    // dir 0: queries = cloud 1 (n points, idx1 into cloud 2), receiver = cloud 2;  dir 1: the converse
    const int nq = dir == 0 ? n : m, nr = dir == 0 ? m : n;
    const int t0 = blockIdx.x * CHG_TARGETS;
    if (t0 >= nr) return;
    const float* q = (dir == 0 ? xyz1 : xyz2) + (size_t)b * nq * 3;
    const float* r = (dir == 0 ? xyz2 : xyz1) + (size_t)b * nr * 3;
    const float* gd = (dir == 0 ? gd1 : gd2) + (size_t)b * nq;
    const int* ix = (dir == 0 ? idx1 : idx2) + (size_t)b * nq;
    float* out = (dir == 0 ? g2 : g1) + (size_t)b * nr * 3;
    for (int i = threadIdx.x; i < CHG_TARGETS * 3; i += 1024) acc[i] = 0ull;
    if (threadIdx.x == 0) maxbits = 0u;
    __syncthreads();
    unsigned mb = 0u;
    for (int j = threadIdx.x; j < nq; j += 1024) {
        const int t = ix[j] - t0;
        if ((unsigned)t >= (unsigned)CHG_TARGETS) continue;
        const float g = gd[j] * 2.f;
        const float* rp = r + (size_t)(t0 + t) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned bits = __float_as_uint(g * (q[(size_t)j * 3 + c] - rp[c])) & 0x7fffffffu;
            mb = bits > mb ? bits : mb;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned ob = (unsigned)__shfl_xor((int)mb, o); mb = ob > mb ? ob : mb; }
    if ((threadIdx.x & 63) == 0 && mb) atomicMax(&maxbits, mb);
    __syncthreads();
    const unsigned mx = maxbits;
    const int lim = min(CHG_TARGETS, nr - t0) * 3;
    if (mx >= 0x7f800000u) {
        // a non-finite term (Inf / NaN gradient or coordinate) lands in this range: sum in double as rounds 1-3 did, so that it poisons
        // exactly the receivers a float sum would (determinism is moot for such a step)
        double* dacc = (double*)acc;                             // (zero bits are 0.0)
        for (int j = threadIdx.x; j < nq; j += 1024) {
            const int t = ix[j] - t0;
            if ((unsigned)t >= (unsigned)CHG_TARGETS) continue;
            const float g = gd[j] * 2.f;
            const float* rp = r + (size_t)(t0 + t) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) atomicAdd(&dacc[t * 3 + c], -(double)(g * (q[(size_t)j * 3 + c] - rp[c])));
        }
        __syncthreads();
        for (int i = threadIdx.x; i < lim; i += 1024) out[(size_t)t0 * 3 + i] += (float)dacc[i];
        return;
    }
    // every term is below 2^(ef - 126), ef the largest biased exponent: scaled by 2^sh it is below 2^40
    const int ef = (int)(mx >> 23);
    const int sh = (166 - ef) > 127 ? 127 : 166 - ef;            // (>= -88; the clamp only matters for sums of denormals)
    const float scale = ldexpf(1.f, sh);
    for (int j = threadIdx.x; j < nq; j += 1024) {
        const int t = ix[j] - t0;
        if ((unsigned)t >= (unsigned)CHG_TARGETS) continue;
        const float g = gd[j] * 2.f;
        const float* rp = r + (size_t)(t0 + t) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = g * (q[(size_t)j * 3 + c] - rp[c]);          // the same fp32 value the point's own term uses
            atomicAdd(&acc[t * 3 + c], (unsigned long long)(long long)rintf(-v * scale));          // two's complement: wrap-around addition is exact
        }
    }
    __syncthreads();
    const double down = ldexp(1.0, -sh);
    for (int i = threadIdx.x; i < lim; i += 1024) out[(size_t)t0 * 3 + i] += (float)((double)(long long)acc[i] * down);
}

// own terms of both directions: g1[j] = 2 gd1[j] (x1[j] - x2[idx1[j]]),  g2[j] = 2 gd2[j] (x2[j] - x1[idx2[j]])   (plain stores)
__global__ void vnp_chamfer_own_kernel(int n, const float* __restrict__ xyz1, int m, const float* __restrict__ xyz2,
                                       const float* __restrict__ gd1, const int* __restrict__ idx1,
                                       const float* __restrict__ gd2, const int* __restrict__ idx2,
                                       float* __restrict__ g1, float* __restrict__ g2)
{
    const int b = blockIdx.y, dir = blockIdx.z;
    const int nq = dir == 0 ? n : m, nr = dir == 0 ? m : n;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nq) return;
    const float* q = (dir == 0 ? xyz1 : xyz2) + ((size_t)b * nq + j) * 3;
    const int t = (dir == 0 ? idx1 : idx2)[(size_t)b * nq + j];
    const float* r = (dir == 0 ? xyz2 : xyz1) + ((size_t)b * nr + t) * 3;
    const float g = (dir == 0 ? gd1 : gd2)[(size_t)b * nq + j] * 2.f;
    float* o = (dir == 0 ? g1 : g2) + ((size_t)b * nq + j) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = g * (q[c] - r[c]);
}

static int vnp_chamfer_one_direction(hipStream_t st, int B, int n, const float* q, int m, const float* r,
                                     float* dist, int* idx, unsigned long long* packed)
{
    hipError_t e = hipMemsetAsync(packed, 0xff, (size_t)B * n * 8, st);
    if (e != hipSuccess) return (int)e;
    const int qblocks = vnp_cdiv(n, CH_THREADS * CH_QPT);
    // slice the reference cloud so that the launch has >= ~2048 blocks, in whole LDS tiles
    int nslice = vnp_cdiv(2048, qblocks * B);
    const int max_slices = vnp_cdiv(m, CH_TILE);
    if (nslice > max_slices) nslice = max_slices;
    if (nslice < 1) nslice = 1;
    int slice = vnp_cdiv(vnp_cdiv(m, nslice), CH_TILE) * CH_TILE;
    nslice = vnp_cdiv(m, slice);
    hipLaunchKernelGGL(vnp_chamfer_nn_kernel, dim3(qblocks, nslice, B), dim3(CH_THREADS), 0, st, q, n, r, m, slice, packed);
    const long long total = (long long)B * n;
    hipLaunchKernelGGL(vnp_chamfer_unpack_kernel, dim3(vnp_cdiv(total, 256)), dim3(256), 0, st,
                       (const unsigned long long*)packed, total, dist, idx);
    return vnp_last_error();
}

// ------------------------------------------------------------------------------------------------------------------
// Pruned exact search (dense clouds, n, m <= 32768): same outputs, bit for bit, as the brute-force kernel above.
//
// Each cloud is binned along a Morton curve once per call (vnp_chamfer_bin_kernel: one workgroup per sample, a counting
// sort on 15-bit keys — 5 bits per axis of the cloud's bounding box — with the 32768 counters in LDS) and cut into chunks
// of 64 consecutive points with an axis-aligned bounding box each.  A wave then owns 64*QPT consecutive sorted queries (a
// compact group) and walks the <= 256 reference chunks: a chunk is visited only if, for at least one of the wave's queries,
// the point-to-box lower bound does not exceed that query's current best (one ballot per chunk: a wave-uniform decision,
// no divergence).  Order: the chunk closest to the group's box first, then the chunks overlapping that box, then the rest —
// which prunes as well as a full sort by distance (simulated: 9-17 % of the chunks visited for uniform, surface-like and
// blob-vs-spread clouds).  When nothing can be skipped the cost is the brute-force cost.
// Exactness: distances use the same fmaf(dz,dz,fmaf(dy,dy,dx*dx)) on the same operands; fp32 subtraction is monotone, so
// every |r - q| component is >= the corresponding box gap as computed, and the skip test keeps a 1e-6 relative margin over
// the different association of the two sums; because chunks are visited out of index order the running minimum is the
// 64-bit key (distance bits << 32 | original index), whose unsigned order is "smaller distance, then smaller index" — the
// reference's strict '<' scan in index order.  The order of points inside a bin depends on LDS atomics and may change from
// run to run; the outputs do not.  A cloud with a non-finite coordinate disables skipping.
// A visited chunk is staged in LDS (1 KB) and read back as broadcast ds_read_b128, as in the brute-force kernel.
#define CHP_CHUNK 64              // measured: 32-point chunks are 3x slower (twice the box tests, 32 KB of boxes per wave in LDS)
#define CHP_QPT 1
#define CHP_MAXN 32768
#define CHP_BINS 32768

__device__ __forceinline__ unsigned vnp_morton_spread5(unsigned v)
{
    v &= 0x1fu;
    v = (v | (v << 8)) & 0x100fu;
    v = (v | (v << 4)) & 0x10c3u;
    v = (v | (v << 2)) & 0x1249u;
    return v;
}

// sorted[b][i] = (x, y, z, original index as int bits); boxes[b][c][0/1] = (min, max) of chunk c; flag[0] |= non-finite
// blockIdx.y selects the cloud: both clouds of a Chamfer call are binned by ONE launch (a workgroup per sample and cloud; the kernel is
// latency-bound — 32 workgroups of a 256-CU chip — so two launches of B workgroups cost twice one launch of 2 B)
__global__ void __launch_bounds__(1024) vnp_chamfer_bin_kernel(const float* __restrict__ xyz_a, int n_a, float4* sorted_a, float4* __restrict__ boxes_a,
                                                               const float* __restrict__ xyz_b, int n_b, float4* sorted_b, float4* __restrict__ boxes_b,
                                                               int* __restrict__ flag)
{
    extern __shared__ unsigned cnt[];                          // CHP_BINS counters, then running offsets
    __shared__ float red[6][16];
    __shared__ unsigned wsum[16];
    const bool second = blockIdx.y != 0;
    const float* __restrict__ xyz = second ? xyz_b : xyz_a;
    const int n = second ? n_b : n_a;
    float4* sorted = second ? sorted_b : sorted_a;
    float4* __restrict__ boxes = second ? boxes_b : boxes_a;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pts = xyz + (size_t)b * n * 3;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    for (int i = tid; i < n; i += 1024) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float v = pts[i * 3 + t];
            bad |= !(fabsf(v) <= 3.0e38f);
            lo[t] = fminf(lo[t], v); hi[t] = fmaxf(hi[t], v);
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[t] = fminf(lo[t], __shfl_xor(lo[t], o));
            hi[t] = fmaxf(hi[t], __shfl_xor(hi[t], o));
        }
        if (lane == 0) { red[t][wave] = lo[t]; red[3 + t][wave] = hi[t]; }
    }
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
    for (int i = tid; i < CHP_BINS; i += 1024) cnt[i] = 0;
    __syncthreads();
    float scale[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float l = red[t][0], h = red[3 + t][0];
        for (int w = 1; w < 16; ++w) { l = fminf(l, red[t][w]); h = fmaxf(h, red[3 + t][w]); }
        lo[t] = l;
        scale[t] = (h > l) ? 31.0f / (h - l) : 0.f;
    }
    unsigned short key[CHP_MAXN / 1024];
#pragma unroll
    for (int r = 0; r < CHP_MAXN / 1024; ++r) {
        const int i = tid + 1024 * r;
        key[r] = 0;
        if (i < n) {
            unsigned q[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float f = (pts[i * 3 + t] - lo[t]) * scale[t];
                f = fminf(fmaxf(f, 0.f), 31.f);                  // NaN -> 0 through fmaxf
                q[t] = (unsigned)f;
            }
            key[r] = (unsigned short)(vnp_morton_spread5(q[0]) | (vnp_morton_spread5(q[1]) << 1) | (vnp_morton_spread5(q[2]) << 2));
            atomicAdd(&cnt[key[r]], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the 32768 counters: 32 consecutive bins per thread
    unsigned local[32], tot = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) { local[k] = cnt[tid * 32 + k]; tot += local[k]; }
    unsigned inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned base = inc - tot;
    for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
    for (int k = 0; k < 32; ++k) { cnt[tid * 32 + k] = base; base += local[k]; }
    __syncthreads();
    float4* out = sorted + (size_t)b * n;
#pragma unroll
    for (int r = 0; r < CHP_MAXN / 1024; ++r) {
        const int i = tid + 1024 * r;
        if (i < n) {
            const unsigned slot = atomicAdd(&cnt[key[r]], 1u);
            out[slot] = make_float4(pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2], __int_as_float(i));
        }
    }
    __threadfence_block();
    __syncthreads();                                           // `out` of this workgroup is complete and visible to it
    const int nchunk = (n + CHP_CHUNK - 1) / CHP_CHUNK;
    for (int c = wave; c < nchunk; c += 16) {
        const int i = c * CHP_CHUNK + lane;
        float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (i < n) {
            const float4 v = out[i];
            l[0] = h[0] = v.x; l[1] = h[1] = v.y; l[2] = h[2] = v.z;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                l[t] = fminf(l[t], __shfl_xor(l[t], o));
                h[t] = fmaxf(h[t], __shfl_xor(h[t], o));
            }
        if (lane == 0) {
            boxes[((size_t)b * nchunk + c) * 2 + 0] = make_float4(l[0], l[1], l[2], 0.f);
            boxes[((size_t)b * nchunk + c) * 2 + 1] = make_float4(h[0], h[1], h[2], 0.f);
        }
    }
}

// LDS hand-off between the lanes of ONE wave: a wave's DS instructions execute in order, so a read issued after a write sees it; the wait
// retires the outstanding ones and the asm statement keeps the compiler from moving LDS accesses across it.  No workgroup barrier.
__device__ __forceinline__ void vnp_wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

// NG: chunk groups of 64 (one ballot word each) the target cloud may have: 4 (up to 16384 points: 8 KB of boxes in LDS) or 8 (up to
// 32768: 16 KB).  WPB: query waves per workgroup.  The waves of a workgroup scan the same target cloud and share ONE LDS copy of its chunk
// boxes (loaded together, one barrier); from there on every wave is on its own — its 64 queries, its KB tiles of 1 KB, its own sequence
// of chunks, no barrier (a wave's LDS operations execute in order).
// Round 5, what bounds the scan (profiles/r05_chamfer_pmc.txt, scripts/experiments/chamfer_pmc.py / chamfer_hist.py; B = 32, 16384 x 16384):
// 118 M VALU wave-instructions per launch, VALU busy 0.62 at ~2 resident waves per SIMD; a wave scans 25 chunks on average (uniform
// clouds: p99 46, max 54), i.e. 9 VALU instructions per (query, target) pair x 1590 pairs per query = the time.  Measured in one process
// against the round-4 kernel (one wave per workgroup, one chunk per visit), same outputs bit for bit: (WPB, KB) = (4, 4) -1.5 / -4 /
// -2.5 % per call on uniform / near-copy / blob-vs-spread clouds, (4, 1) 0 / -0.5 / -6 %, (4, 2) and (2, 4) slower.  Dropped: the
// minimum as a 32-bit (distance, index) pair with a tie flag and an exact second scan (VALU instructions -16 %, time +8...+27 %: the
// flag's compare-to-SGPR + s_or per target stalls the in-order issue); chunk boxes through scalar loads (every skip test waits for a
// memory round trip: +17 %).
template <int QPT, int NG, int WPB, int KB>
__global__ void __launch_bounds__(64 * WPB) vnp_chamfer_pruned_kernel(const float4* __restrict__ qs, int n, const float4* __restrict__ rs,
                                                                int m, const float4* __restrict__ rbox, const int* __restrict__ flag,
                                                                float* __restrict__ dist, int* __restrict__ idx,
                                                                unsigned long long* __restrict__ visited, unsigned* __restrict__ wave_chunks)
{
    static_assert(NG * 64 * CHP_CHUNK <= CHP_MAXN, "chunk groups beyond the binning kernel's range");
    __shared__ float4 box[2 * 64 * NG];
    // the chunk being scanned, two points interleaved per 32 bytes {x0 x1 y0 y1 | z0 z1 i0 i1}: a lane evaluates TWO targets per step
    // with packed fp32 arithmetic (v_pk_add / v_pk_mul / v_pk_fma: 6 instructions for the two distances instead of 12), each distance
    // by the same fmaf chain as before
    __shared__ __attribute__((aligned(16))) float tiles[WPB][KB * CHP_CHUNK * 4];
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* tile = tiles[wave];
    const int qgroup = blockIdx.x * WPB + wave;                    // this wave's group of 64 QPT queries; groups past the cloud's end idle
    const bool active = qgroup * (64 * QPT) < n;
    const int base = active ? qgroup * (64 * QPT) : 0;
    const float4* qb = qs + (size_t)b * n;
    const float4* rb = rs + (size_t)b * m;
    const int nchunk = (m + CHP_CHUNK - 1) / CHP_CHUNK;
    const float4* bx = rbox + (size_t)b * nchunk * 2;
    const bool prune = flag[0] == 0;

    float qx[QPT], qy[QPT], qz[QPT];
    int qo[QPT];
    unsigned long long best[QPT];
    float ql[3] = {INFINITY, INFINITY, INFINITY}, qh[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const int j = min(base + lane + 64 * u, n - 1);        // clamp: out-of-range lanes redo the last point, never stored
        const float4 q = qb[j];
        qx[u] = q.x; qy[u] = q.y; qz[u] = q.z; qo[u] = __float_as_int(q.w);
        best[u] = 0x7f800000ull << 32;                         // +inf, index 0: what an all-NaN scan leaves behind
        ql[0] = fminf(ql[0], q.x); ql[1] = fminf(ql[1], q.y); ql[2] = fminf(ql[2], q.z);
        qh[0] = fmaxf(qh[0], q.x); qh[1] = fmaxf(qh[1], q.y); qh[2] = fmaxf(qh[2], q.z);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ql[t] = fminf(ql[t], __shfl_xor(ql[t], o));
            qh[t] = fmaxf(qh[t], __shfl_xor(qh[t], o));
        }
    // chunk boxes -> LDS (all waves together), then the group-box bound of chunk lane + 64 g classifies it: overlapping (0) / apart (> 0)
    for (int c = threadIdx.x; c < nchunk; c += 64 * WPB) { box[2 * c] = bx[2 * c]; box[2 * c + 1] = bx[2 * c + 1]; }
    __syncthreads();
    if (!active) return;
    unsigned long long near_mask[NG], far_mask[NG];
    float vfar[NG];                                               // this lane's chunk of group g: squared distance between the two boxes
    float lmin = INFINITY;
    int cmin = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = lane + 64 * g;
        float v = INFINITY;
        if (c < nchunk) {
            const float4 l = box[2 * c], h = box[2 * c + 1];
            const float gx = fmaxf(0.f, fmaxf(l.x - qh[0], ql[0] - h.x));
            const float gy = fmaxf(0.f, fmaxf(l.y - qh[1], ql[1] - h.y));
            const float gz = fmaxf(0.f, fmaxf(l.z - qh[2], ql[2] - h.z));
            v = gx * gx + gy * gy + gz * gz;
            if (!(v >= 0.f)) v = 0.f;                            // NaN boxes: treated as overlapping
        }
        near_mask[g] = __ballot(c < nchunk && v == 0.f);
        far_mask[g] = __ballot(c < nchunk && v > 0.f);
        vfar[g] = v * (1.0f - 1e-6f);
        if (v < lmin) { lmin = v; cmin = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(lmin, o);
        const int oc = __shfl_xor(cmin, o);
        if (ov < lmin || (ov == lmin && oc < cmin)) { lmin = ov; cmin = oc; }
    }
    const int c0 = __builtin_amdgcn_readfirstlane(cmin);

    int npairs = 0, nvisit = 0;                                   // (query, target) pairs / chunks this wave evaluates: the kernel's executed work
    // Up to KB needed chunks are fetched TOGETHER (their loads in flight at once, KB KB of LDS per wave) and then scanned one after the
    // other: a wave pays the global-load latency of its chunks once per batch instead of once per chunk (round 5: per-segment stamps put
    // two thirds of a visit into waiting for its 1 KB).  Chunks of a batch are chosen with the minima as they were BEFORE the batch, so a
    // batch may scan a chunk that its own earlier chunks would have ruled out — more pairs, never a different result.
    auto visit_batch = [&](const int (&cs)[KB], int nb) {
        float4 v[KB];
        int cnt[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const int r0 = cs[j] * CHP_CHUNK;
            cnt[j] = j < nb ? min(CHP_CHUNK, m - r0) : 0;
            v[j] = make_float4(NAN, NAN, NAN, 0.f);               // slots past the chunk's end: NaN distances never win
            if (lane < cnt[j]) v[j] = rb[r0 + lane];
            npairs += cnt[j];
        }
        nvisit += nb;
        vnp_wave_lds_sync();                                      // the tiles are this wave's own: order their reuse, no workgroup barrier
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            float* dst = tile + j * (CHP_CHUNK * 4) + (lane >> 1) * 8 + (lane & 1);
            if (j < nb) { dst[0] = v[j].x; dst[2] = v[j].y; dst[4] = v[j].z; dst[6] = v[j].w; }
        }
        vnp_wave_lds_sync();
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const float* tj = tile + j * (CHP_CHUNK * 4);
#pragma unroll 4
            for (int i = 0; i < cnt[j]; i += 2) {
                const vnp_f4 ta = *(const vnp_f4*)(tj + i * 4);        // same address in every lane: LDS broadcast
                const vnp_f4 tb = *(const vnp_f4*)(tj + i * 4 + 4);
#pragma unroll
                for (int u = 0; u < QPT; ++u) {
                    const vnp_f2 dx = vnp_f2{ta[0], ta[1]} - qx[u], dy = vnp_f2{ta[2], ta[3]} - qy[u], dz = vnp_f2{tb[0], tb[1]} - qz[u];
                    const vnp_f2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                    const unsigned long long k0 = ((unsigned long long)__float_as_uint(d[0]) << 32) | __float_as_uint(tb[2]);
                    const unsigned long long k1 = ((unsigned long long)__float_as_uint(d[1]) << 32) | __float_as_uint(tb[3]);
                    if (k0 < best[u]) best[u] = k0;                   // NaN distances (bits above +inf) never win
                    if (k1 < best[u]) best[u] = k1;
                }
            }
        }
    };
    // does any query of the wave still need chunk c?  point-to-box bound of each query against its own best
    auto needed = [&](int c) -> bool {
        const float4 l = box[2 * c], h = box[2 * c + 1];
        bool need = false;
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            const float gx = fmaxf(0.f, fmaxf(l.x - qx[u], qx[u] - h.x));
            const float gy = fmaxf(0.f, fmaxf(l.y - qy[u], qy[u] - h.y));
            const float gz = fmaxf(0.f, fmaxf(l.z - qz[u], qz[u] - h.z));
            const float lbq = (gx * gx + gy * gy + gz * gz) * (1.0f - 1e-6f);
            need |= !(lbq > __uint_as_float((unsigned)(best[u] >> 32)));       // NaN bound -> needed
        }
        return __any(need);
    };

    // the largest best distance among the wave's queries: a chunk whose BOX-TO-BOX distance from the wave's queries exceeds it cannot hold
    // a nearer point for any of them (a query lies inside the wave's box, so its own distance to the chunk's box is at least that; the
    // per-query test below is the same formula term by term, and fp subtraction, multiplication and addition are monotone) — one
    // compare + ballot settles 64 far chunks at once, and the ~25-instruction per-query test runs only for the few that survive
    auto wave_max_best = [&]() -> float {
        float mb = __uint_as_float((unsigned)(best[0] >> 32));
#pragma unroll
        for (int u = 1; u < QPT; ++u) mb = fmaxf(mb, __uint_as_float((unsigned)(best[u] >> 32)));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mb = fmaxf(mb, __shfl_xor(mb, o));
        return mb;                                                // (+inf while any query has seen nothing; NaN distances never enter best)
    };
    int cs[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) cs[j] = c0;
    visit_batch(cs, 1);                                           // the closest chunk alone: it sets the minima every later test prunes with
    int nb = 0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float maxbest = (pass == 1 && prune) ? wave_max_best() : INFINITY;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            unsigned long long mask = pass == 0 ? near_mask[g] : far_mask[g];
            if (pass == 1 && prune) mask &= __ballot(!(vfar[g] > maxbest));
            while (mask) {
                const int bit = __builtin_ctzll(mask);
                mask &= mask - 1;
                const int c = 64 * g + bit;
                if (c == c0) continue;
                if (prune && !needed(c)) continue;
#pragma unroll
                for (int j = 0; j < KB; ++j) if (j == nb) cs[j] = c;          // (nb is wave-uniform: scalar selects, no indexed registers)
                if (++nb < KB) continue;
                visit_batch(cs, KB);
                nb = 0;
                if (pass == 1 && prune) {
                    maxbest = wave_max_best();
                    mask &= __ballot(!(vfar[g] > maxbest));
                }
            }
            if (nb > 0) {                                         // a group ends: its last chunks, before the next group's far test reads the minima
                visit_batch(cs, nb);
                nb = 0;
                if (pass == 1 && prune) maxbest = wave_max_best();
            }
        }
    }
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        if (base + lane + 64 * u < n) {
            dist[(size_t)b * n + qo[u]] = __uint_as_float((unsigned)(best[u] >> 32));
            idx[(size_t)b * n + qo[u]] = (int)(unsigned)(best[u] & 0xffffffffu);
        }
    }
    if (lane == 0) {
        atomicAdd(visited, (unsigned long long)npairs * (unsigned long long)min(64 * QPT, n - base));
        if (wave_chunks) wave_chunks[(size_t)b * ((n + 64 * QPT - 1) / (64 * QPT)) + qgroup] = (unsigned)nvisit;
    }
}

// byte offset inside ws of the per-wave scanned-chunk counts (unsigned [B][waves of direction 1], then [B][waves of direction 2]; a wave
// = 64 consecutive queries of the Morton-sorted cloud), written by every call: diagnostics.
// What they showed (round 4, bench.py `chamfer_dense_loss_in_step`): on the training step's clouds a wave scans 14-18 of 256 chunks on
// average and 86-114 at worst.  Two restructurings aimed at that longest chain were measured and dropped: the next needed chunk loaded
// behind the scan of the current one (359 -> 395 us per direction: no gain), and four waves per query group, each owning every fourth
// chunk, sharing their minima through 64-bit LDS atomic minima (bit-identical results, 465 us: the per-wave set-up — 512 box bounds,
// the group reductions — is paid four times and outweighs the shorter chains).  The launch is not bound by its longest chain.
extern "C" long long vnp_chamfer_pruned_ws_wave_counts(int B, int n, int m)
{
    if (B <= 0 || n < 1 || m < 1 || n > CHP_MAXN || m > CHP_MAXN) return 0;
    const long long c1 = vnp_cdiv(n, CHP_CHUNK), c2 = vnp_cdiv(m, CHP_CHUNK);
    return 16LL * B * ((long long)n + m) + 32LL * B * (c1 + c2) + 256;
}

// bytes of scratch vnp_chamfer_fwd_pruned needs (0: sizes outside the pruned path's range)
extern "C" long long vnp_chamfer_pruned_ws_bytes(int B, int n, int m)
{
    if (B <= 0 || n < 1 || m < 1 || n > CHP_MAXN || m > CHP_MAXN) return 0;
    const long long c1 = vnp_cdiv(n, CHP_CHUNK), c2 = vnp_cdiv(m, CHP_CHUNK);
    return 16LL * B * ((long long)n + m) + 32LL * B * (c1 + c2) + 256 + 4LL * B * (vnp_cdiv(n, 64 * CHP_QPT) + vnp_cdiv(m, 64 * CHP_QPT));
}

// Same contract as vnp_chamfer_fwd, by the pruned search.  n, m <= 32768; ws of vnp_chamfer_pruned_ws_bytes(B, n, m) bytes.
// After the call the 64-bit word at ws + 8 holds the number of (query, target) pairs the two scans evaluated (a full scan: 2 B n m):
// diagnostics, read by bench.py to price the kernel on its executed work.
extern "C" int vnp_chamfer_fwd_pruned(hipStream_t st, int B, int n, const float* xyz1, int m, const float* xyz2,
                                      float* dist1, int* idx1, float* dist2, int* idx2, void* ws)
{
    if (B <= 0 || n <= 0 || m <= 0 || !xyz1 || !xyz2 || !dist1 || !idx1 || !dist2 || !idx2 || !ws) return VNP_EINVAL;
    if (B > 65535 || n > CHP_MAXN || m > CHP_MAXN) return VNP_EINVAL;
    const int c1 = vnp_cdiv(n, CHP_CHUNK), c2 = vnp_cdiv(m, CHP_CHUNK);
    char* w = (char*)ws;
    int* flag = (int*)w;                       w += 256;
    float4* s1 = (float4*)w;                   w += 16LL * B * n;
    float4* s2 = (float4*)w;                   w += 16LL * B * m;
    float4* b1 = (float4*)w;                   w += 32LL * B * c1;
    float4* b2 = (float4*)w;                   w += 32LL * B * c2;
    // diagnostics (bench.py's imbalance histogram): chunks scanned by each query wave, direction 1 then direction 2
    unsigned* wc1 = (unsigned*)w;              w += 4LL * B * vnp_cdiv(n, 64 * CHP_QPT);
    unsigned* wc2 = (unsigned*)w;
    hipError_t e = hipMemsetAsync(flag, 0, 256, st);
    if (e != hipSuccess) return (int)e;
    static std::atomic<unsigned long long> lds_ok{0};
    e = vnp_allow_dynamic_lds(lds_ok, (const void*)vnp_chamfer_bin_kernel, CHP_BINS * 4);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(vnp_chamfer_bin_kernel, dim3(B, 2), dim3(1024), (size_t)CHP_BINS * 4, st, xyz1, n, s1, b1, xyz2, m, s2, b2, flag);
    const int g1 = vnp_cdiv(n, 64 * CHP_QPT), g2 = vnp_cdiv(m, 64 * CHP_QPT);
#define VNP_CHP_LAUNCH(NGV, WPBV, KBV, GROUPS, ...) hipLaunchKernelGGL((vnp_chamfer_pruned_kernel<CHP_QPT, NGV, WPBV, KBV>), dim3(vnp_cdiv(GROUPS, WPBV), B), dim3(64 * WPBV), 0, st, __VA_ARGS__)
#define VNP_CHP_BOTH(WPBV, KBV) do { \
        if (c2 <= 256) VNP_CHP_LAUNCH(4, WPBV, KBV, g1, (const float4*)s1, n, (const float4*)s2, m, (const float4*)b2, (const int*)flag, dist1, idx1, (unsigned long long*)(flag + 2), wc1); \
        else VNP_CHP_LAUNCH(8, WPBV, KBV, g1, (const float4*)s1, n, (const float4*)s2, m, (const float4*)b2, (const int*)flag, dist1, idx1, (unsigned long long*)(flag + 2), wc1); \
        if (c1 <= 256) VNP_CHP_LAUNCH(4, WPBV, KBV, g2, (const float4*)s2, m, (const float4*)s1, n, (const float4*)b1, (const int*)flag, dist2, idx2, (unsigned long long*)(flag + 2), wc2); \
        else VNP_CHP_LAUNCH(8, WPBV, KBV, g2, (const float4*)s2, m, (const float4*)s1, n, (const float4*)b1, (const int*)flag, dist2, idx2, (unsigned long long*)(flag + 2), wc2); } while (0)
    VNP_CHP_BOTH(4, 4);
#undef VNP_CHP_BOTH
#undef VNP_CHP_LAUNCH
    return vnp_last_error();
}

// Replaces chamfer_3D.forward(xyz1, xyz2, dist1, dist2, idx1, idx2) (chamfer_cuda.cpp:17-19, chamfer3D.cu:136-154).
// ws: scratch of max(n, m) * B * 8 bytes.  Outputs need not be zeroed.
extern "C" int vnp_chamfer_fwd(hipStream_t st, int B, int n, const float* xyz1, int m, const float* xyz2,
                               float* dist1, int* idx1, float* dist2, int* idx2, void* ws)
{
    if (B <= 0 || n <= 0 || m <= 0 || !xyz1 || !xyz2 || !dist1 || !idx1 || !dist2 || !idx2 || !ws) return VNP_EINVAL;
    if (B > 65535) return VNP_EINVAL;
    int rc = vnp_chamfer_one_direction(st, B, n, xyz1, m, xyz2, dist1, idx1, (unsigned long long*)ws);
    if (rc) return rc;
    return vnp_chamfer_one_direction(st, B, m, xyz2, n, xyz1, dist2, idx2, (unsigned long long*)ws);
}

// Replaces chamfer_3D.backward (chamfer_cuda.cpp:22-26, chamfer3D.cu:176-195).  grad_xyz1/2 need not be zeroed.
extern "C" int vnp_chamfer_bwd(hipStream_t st, int B, int n, const float* xyz1, int m, const float* xyz2,
                               const float* grad_dist1, const int* idx1, const float* grad_dist2, const int* idx2,
                               float* grad_xyz1, float* grad_xyz2)
{
    if (B <= 0 || n <= 0 || m <= 0 || !xyz1 || !xyz2 || !grad_dist1 || !idx1 || !grad_dist2 || !idx2 || !grad_xyz1 || !grad_xyz2)
        return VNP_EINVAL;
    if (B > 65535) return VNP_EINVAL;
    if (n <= 32768 && m <= 32768) {
        // own terms (plain stores), then the scatter terms through LDS accumulators: no global atomics
        const int big = n > m ? n : m;
        hipLaunchKernelGGL(vnp_chamfer_own_kernel, dim3(vnp_cdiv(big, 256), B, 2), dim3(256), 0, st, n, xyz1, m, xyz2, grad_dist1, idx1,
                           grad_dist2, idx2, grad_xyz1, grad_xyz2);
        hipLaunchKernelGGL(vnp_chamfer_scatter_kernel, dim3(vnp_cdiv(big, CHG_TARGETS), B, 2), dim3(1024), 0, st, n, xyz1, m, xyz2,
                           grad_dist1, idx1, grad_dist2, idx2, grad_xyz1, grad_xyz2);
        return vnp_last_error();
    }
    hipError_t e = hipMemsetAsync(grad_xyz2, 0, (size_t)B * m * 12, st);
    if (e != hipSuccess) return (int)e;
    // very large clouds: pass 1 stores grad_xyz1 and scatters (global atomics) into the zeroed grad_xyz2; pass 2 adds its own terms to
    // grad_xyz2 and scatters into grad_xyz1
    hipLaunchKernelGGL(vnp_chamfer_grad_kernel<true>, dim3(vnp_cdiv(n, 256), B), dim3(256), 0, st, n, xyz1, m, xyz2, grad_dist1, idx1, grad_xyz1, grad_xyz2);
    hipLaunchKernelGGL(vnp_chamfer_grad_kernel<false>, dim3(vnp_cdiv(m, 256), B), dim3(256), 0, st, m, xyz2, n, xyz1, grad_dist2, idx2, grad_xyz2, grad_xyz1);
    return vnp_last_error();
}

// ---- density-aware Chamfer term (the reference's calc_dcd, metrics/loss.py:77-118: per sample a bincount of the arg-min indices,
// then mean over the queries of 1 - exp(-alpha d) * frac / (count[idx]^lambda + 1e-6)) -------------------------------------------------
// hits[b][t] = number of queries of sample b whose nearest target is t.  Integer atomics: the result does not depend on their order.
__global__ void __launch_bounds__(256) vnp_dcd_hist_kernel(const int* __restrict__ idx, int N, int T, int* __restrict__ hits)
{
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int t = idx[(size_t)b * N + i];
    if (t >= 0 && t < T) atomicAdd(&hits[(size_t)b * T + t], 1);
}

// One workgroup per sample: loss[b] = mean_i (1 - exp(-alpha d_i) w_i), w_i = frac / (hits[idx_i]^lambda + 1e-6) (fp32, the reference's
// operation order); gfac[b][i] = d loss[b] / d d_i = alpha exp(-alpha d_i) w_i / N (the weight is a constant of the graph: .detach()).
// The per-sample sum is taken in double, per thread in index order and across the workgroup in a fixed tree: deterministic.
__global__ void __launch_bounds__(1024) vnp_dcd_side_kernel(const float* __restrict__ dist, const int* __restrict__ idx, const int* __restrict__ hits,
                                                            int N, int T, float alpha, float lambda, float frac,
                                                            float* __restrict__ loss, float* __restrict__ gfac)
{
    __shared__ double red[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* d = dist + (size_t)b * N;
    const int* ix = idx + (size_t)b * N;
    const int* h = hits + (size_t)b * T;
    const float inv_n = 1.f / (float)N;
    double acc = 0.0;
    for (int i = tid; i < N; i += 1024) {
        const int t = ix[i];
        const float cnt = (t >= 0 && t < T) ? (float)h[t] : 0.f;
        const float w = 1.f / (powf(cnt, lambda) + 1e-6f) * frac;
        const float e = expf(-alpha * d[i]);
        acc += (double)(1.f - e * w);
        if (gfac) gfac[(size_t)b * N + i] = alpha * e * w * inv_n;
    }
    red[tid] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) loss[b] = (float)(red[0] / (double)N);
}

// ---- cd_loss_L1 (metrics/loss.py:20-31): (mean sqrt(dist1) + mean sqrt(dist2)) / 2 over ALL B*n / B*m elements, and its gradient with
// respect to the two distance arrays, as two launches + one instead of four ATen kernels per array forward and five backward.
// Sums in double in a fixed order (block partials, then one block over the partials): run-to-run identical.
#define VNP_CDL_BLOCKS 256
__global__ void __launch_bounds__(256) vnp_cd_l1_partial_kernel(const float* __restrict__ d1, long long n1, const float* __restrict__ d2, long long n2,
                                                               double* __restrict__ part)
{
    __shared__ double red[2][256];
    double s1 = 0.0, s2 = 0.0;
    const long long stride = (long long)gridDim.x * 256, first = (long long)blockIdx.x * 256 + threadIdx.x;
    for (long long i = first; i < n1; i += stride) s1 += (double)sqrtf(d1[i]);
    for (long long i = first; i < n2; i += stride) s2 += (double)sqrtf(d2[i]);
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[blockIdx.x] = red[0][0]; part[VNP_CDL_BLOCKS + blockIdx.x] = red[1][0]; }
}

__global__ void __launch_bounds__(256) vnp_cd_l1_final_kernel(const double* __restrict__ part, int nblk, long long n1, long long n2, float* __restrict__ out)
{
    __shared__ double red[2][256];
    red[0][threadIdx.x] = (int)threadIdx.x < nblk ? part[threadIdx.x] : 0.0;
    red[1][threadIdx.x] = (int)threadIdx.x < nblk ? part[VNP_CDL_BLOCKS + threadIdx.x] : 0.0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float m1 = (float)(red[0][0] / (double)n1), m2 = (float)(red[1][0] / (double)n2);
        out[0] = (m1 + m2) / 2.0f; out[1] = m1; out[2] = m2;
    }
}

// gd[i] = ((g / 2) / n) / (2 sqrt(d[i])): the chain of autograd's own formulas for (mean(sqrt(d1)) + mean(sqrt(d2))) / 2 in the same
// order (division by 2, mean, sqrt), so the same bits; d = 0 gives Inf here and NaN after the Chamfer backward, as in the reference
__global__ void __launch_bounds__(256) vnp_cd_l1_grad_kernel(const float* __restrict__ g, const float* __restrict__ d1, long long n1,
                                                            const float* __restrict__ d2, long long n2, float* __restrict__ gd1, float* __restrict__ gd2)
{
    const float gh = g[0] / 2.0f;
    const float g1 = gh / (float)n1, g2 = gh / (float)n2;
    const long long stride = (long long)gridDim.x * 256, first = (long long)blockIdx.x * 256 + threadIdx.x;
    for (long long i = first; i < n1; i += stride) gd1[i] = g1 / (2.0f * sqrtf(d1[i]));
    for (long long i = first; i < n2; i += stride) gd2[i] = g2 / (2.0f * sqrtf(d2[i]));
}

// out[3] = {loss, mean sqrt(dist1), mean sqrt(dist2)}; part: 2 * 256 doubles of scratch
extern "C" int vnp_cd_l1_loss(hipStream_t st, const float* dist1, long long n1, const float* dist2, long long n2, double* part, float* out)
{
    if (!dist1 || !dist2 || !part || !out || n1 <= 0 || n2 <= 0) return VNP_EINVAL;
    const long long nmax = n1 > n2 ? n1 : n2;
    const int nblk = (int)(vnp_cdiv(nmax, 1024) < VNP_CDL_BLOCKS ? vnp_cdiv(nmax, 1024) : VNP_CDL_BLOCKS);
    hipLaunchKernelGGL(vnp_cd_l1_partial_kernel, dim3(nblk), dim3(256), 0, st, dist1, n1, dist2, n2, part);
    hipLaunchKernelGGL(vnp_cd_l1_final_kernel, dim3(1), dim3(256), 0, st, (const double*)part, nblk, n1, n2, out);
    return vnp_last_error();
}

// g: device scalar, the cotangent of the loss
extern "C" int vnp_cd_l1_grad(hipStream_t st, const float* g, const float* dist1, long long n1, const float* dist2, long long n2,
                              float* gdist1, float* gdist2)
{
    if (!g || !dist1 || !dist2 || !gdist1 || !gdist2 || n1 <= 0 || n2 <= 0) return VNP_EINVAL;
    const long long nmax = n1 > n2 ? n1 : n2;
    const int nblk = (int)(vnp_cdiv(nmax, 1024) < 1024 ? vnp_cdiv(nmax, 1024) : 1024);
    hipLaunchKernelGGL(vnp_cd_l1_grad_kernel, dim3(nblk), dim3(256), 0, st, g, dist1, n1, dist2, n2, gdist1, gdist2);
    return vnp_last_error();
}

// One side of calc_dcd.  dist / idx [B, N]: one-sided squared distances and arg-min indices (into T targets) from vnp_chamfer_fwd;
// hits [B, T] ints: scratch, zeroed here; loss [B]; gfac [B, N] or NULL (no gradient wanted).
extern "C" int vnp_dcd_side(hipStream_t st, int B, int N, int T, const float* dist, const int* idx, float alpha, float lambda, float frac,
                            int* hits, float* loss, float* gfac)
{
    if (B <= 0 || N <= 0 || T <= 0 || !dist || !idx || !hits || !loss || B > 65535) return VNP_EINVAL;
    hipError_t e = hipMemsetAsync(hits, 0, (size_t)B * T * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(vnp_dcd_hist_kernel, dim3(vnp_cdiv(N, 256), B), dim3(256), 0, st, idx, N, T, hits);
    hipLaunchKernelGGL(vnp_dcd_side_kernel, dim3(B), dim3(1024), 0, st, dist, idx, (const int*)hits, N, T, alpha, lambda, frac, loss, gfac);
    return vnp_last_error();
}
