// Dev probe: how v_mfma_f32_16x16x32_{f16,bf16} round  D = C + sum_k a_k b_k.  Every lane holds the same operands, so D[0] is read.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_round_probe.hip -o scripts/mfma_round_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(8))) __bf16 b8;
typedef __attribute__((ext_vector_type(4))) float f4;
// a[k], b[k] for k = 0..31 (lane l holds k = 8 (l >> 4) .. + 7), c
template <bool BF>
__global__ void k(const float* a, const float* b, float c, float* out)
{
    const int q = threadIdx.x >> 4;
    f4 acc = {c, c, c, c};
    if (BF) { b8 va, vb; for (int i = 0; i < 8; ++i) { va[i] = (__bf16)a[8 * q + i]; vb[i] = (__bf16)b[8 * q + i]; }
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc, 0, 0, 0); }
    else    { h8 va, vb; for (int i = 0; i < 8; ++i) { va[i] = (_Float16)a[8 * q + i]; vb[i] = (_Float16)b[8 * q + i]; }
              acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, vb, acc, 0, 0, 0); }
    if (threadIdx.x == 0) out[0] = acc[0];
}
static float *da, *db, *dout;
static float run(bool bf, const float* a, const float* b, float c)
{
    hipMemcpy(da, a, 128, hipMemcpyHostToDevice); hipMemcpy(db, b, 128, hipMemcpyHostToDevice);
    if (bf) hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, da, db, c, dout);
    else hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, da, db, c, dout);
    float r; hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost); return r;
}
int main()
{
    hipMalloc(&da, 128); hipMalloc(&db, 128); hipMalloc(&dout, 4);
    for (int bf = 0; bf < 2; ++bf) {
        printf("==== %s ====\n", bf ? "bf16" : "f16");
        // 1. C = +-1, n products of 2^-25 each: sum = n/4 ulp(1) ; RNE vs truncation, and the direction for negative C
        for (float c : {1.0f, -1.0f})
            for (int n = 1; n <= 8; ++n) {
                float a[32] = {0}, b[32] = {0};
                for (int i = 0; i < n; ++i) { a[i * 4 % 32 + i / 8] = 1.0f; b[i * 4 % 32 + i / 8] = ldexpf(1.f, -25); }
                const float r = run(bf, a, b, c);
                const double exact = (double)c + n * ldexp(1.0, -25);
                printf("C=%+.0f + %d x 2^-25: got %+.10f = C %+g ulp;  RNE would give %+g ulp (exact %+g ulp)\n", c, n, r,
                       (r - c) / ldexp(1.0, -23), ((float)exact - c) / ldexp(1.0, -23), (exact - c) / ldexp(1.0, -23));
            }
        // 2. C = 0: one product of 1 and n products of 2^-25 (same sums): is the product sum itself rounded correctly?
        for (int n = 1; n <= 8; ++n) {
            float a[32] = {0}, b[32] = {0};
            a[31] = 1.0f; b[31] = 1.0f;
            for (int i = 0; i < n; ++i) { a[i] = 1.0f; b[i] = ldexpf(1.f, -25); }
            const float r = run(bf, a, b, 0.0f);
            printf("C=0: 1 + %d x 2^-25 among the products: got 1 %+g ulp (exact %+g ulp)\n", n, (r - 1.0) / ldexp(1.0, -23), n / 4.0);
        }
        // 3. products with long significands: (1 + 2^-7)^2 = 1 + 2^-6 + 2^-14 (bf16-exact operands), (1 + 2^-10)^2 = 1 + 2^-9 + 2^-20 (f16)
        {
            float a[32] = {0}, b[32] = {0};
            const float x = bf ? 1.0f + ldexpf(1.f, -7) : 1.0f + ldexpf(1.f, -10);
            a[0] = x; b[0] = x;
            for (float c : {0.0f, 1024.0f, 32768.0f, -32768.0f, 1048576.0f, -1048576.0f}) {
                const float r = run(bf, a, b, c);
                const double exact = (double)c + (double)x * x;
                printf("C=%g + x^2 (x = %.10g): got %.10g, RNE %.10g, diff %+g ulp(result)\n", c, x, r, (float)exact,
                       (r - (float)exact) / (double)(nextafterf(fabsf((float)exact), INFINITY) - fabsf((float)exact)));
            }
        }
        // 4. many equal products against a large C: C = 2^24, 32 products of 0.25+... (sum = 8 + 32 * 2^-?)
        for (float p : {0.25f, 0.3125f, 0.4375f, 0.46875f}) {
            float a[32], b[32];
            for (int i = 0; i < 32; ++i) { a[i] = 1.0f; b[i] = p; }
            for (float c : {16777216.0f, -16777216.0f}) {
                const float r = run(bf, a, b, c);
                const double exact = (double)c + 32.0 * p;
                printf("C=%+.0f + 32 x %g (sum %g): got C %+g, RNE C %+g\n", c, p, 32.0 * p, (double)r - c, (double)(float)exact - c);
            }
        }
    }
    return 0;
}
