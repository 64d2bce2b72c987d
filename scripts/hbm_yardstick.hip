// HBM yardstick for the streaming Vector-Neuron passes: what does THIS chip deliver for the read/write mixes of the
// BatchNorm / leaky kernels, with nothing but the loads and stores?  Maps are 1.6 GB each (the decoder's [1572864, 256] fp32
// activations), streamed once, non-temporal, in the kernels' own access pattern (a wave = 64 consecutive channels of one row,
// 4 B per lane, 4 point-lanes per block, 1024-point chunks) and in the 16-B-per-lane pattern.
//   build: hipcc --offload-arch=gfx950 -O3 scripts/hbm_yardstick.hip -o scripts/hbm_yardstick.bin     run: scripts/hbm_yardstick.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NR, int NW>
__global__ void __launch_bounds__(256) rows4(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                             float* __restrict__ o0, float* __restrict__ o1, long long P, int C, float* sink)
{
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + cx;
    const long long p0 = (long long)blockIdx.y * 1024, p1 = p0 + 1024 < P ? p0 + 1024 : P;
    float acc = 0.f;
    for (long long pt = p0 + py; pt < p1; pt += 4) {
        float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const size_t off = (size_t)(pt * 3 + t) * C + ch;
            float x = __builtin_nontemporal_load(a + off);
            if (NR > 1) x += __builtin_nontemporal_load(b + off);
            if (NR > 2) x += __builtin_nontemporal_load(c + off);
            v[t] = x;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const size_t off = (size_t)(pt * 3 + t) * C + ch;
            if (NW > 0) __builtin_nontemporal_store(v[t], o0 + off);
            if (NW > 1) __builtin_nontemporal_store(v[t] * 2.f, o1 + off);
            if (NW == 0) acc += v[t];
        }
    }
    if (NW == 0 && acc == 12345.678f) *sink = acc;
}

template <int NR, int NW>
__global__ void __launch_bounds__(256) flat16(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c,
                                              f4* __restrict__ o0, f4* __restrict__ o1, long long n4, float* sink)
{
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f4 x = __builtin_nontemporal_load(a + i);
        if (NR > 1) x += __builtin_nontemporal_load(b + i);
        if (NR > 2) x += __builtin_nontemporal_load(c + i);
        if (NW > 0) __builtin_nontemporal_store(x, o0 + i);
        if (NW > 1) __builtin_nontemporal_store(x * 2.f, o1 + i);
        if (NW == 0) acc += x[0] + x[1] + x[2] + x[3];
    }
    if (NW == 0 && acc == 12345.678f) *sink = acc;
}

template <typename F>
static double time_ms(F launch, int iters = 10)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main()
{
    const long long P = 524288; const int C = 256;
    const size_t n = (size_t)P * 3 * C, bytes = n * 4;
    float* buf[5]; float* sink;
    for (auto& p : buf) { if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; } hipMemset(p, 0, bytes); }
    hipMalloc(&sink, 4);
    dim3 grid(C / 64, (unsigned)(P / 1024)), block(256);
    const double gb = bytes / 1e9;
    printf("map = %.2f GB; mixes in maps (R = read, W = written)\n", gb);
#define ROWS(NR, NW) { double ms = time_ms([&] { hipLaunchKernelGGL((rows4<NR, NW>), grid, block, 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], P, C, sink); }); \
        printf("rows/4B-per-lane  %dR %dW : %7.3f ms  %6.2f TB/s\n", NR, NW, ms, (NR + NW) * gb / ms); }
#define FLAT(NR, NW) { double ms = time_ms([&] { hipLaunchKernelGGL((flat16<NR, NW>), dim3(256 * 16), block, 0, 0, (const f4*)buf[0], (const f4*)buf[1], (const f4*)buf[2], (f4*)buf[3], (f4*)buf[4], (long long)(n / 4), sink); }); \
        printf("flat/16B-per-lane %dR %dW : %7.3f ms  %6.2f TB/s\n", NR, NW, ms, (NR + NW) * gb / ms); }
    ROWS(1, 0) ROWS(3, 0) ROWS(1, 1) ROWS(2, 1) ROWS(3, 2) ROWS(2, 2)
    FLAT(1, 0) FLAT(3, 0) FLAT(1, 1) FLAT(2, 1) FLAT(3, 2) FLAT(2, 2)
    return 0;
}
