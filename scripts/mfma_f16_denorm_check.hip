#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(8))) __bf16 b8;
typedef __attribute__((ext_vector_type(4))) float f4;
__global__ void k(float a, float b, float* out) {
    h8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = (_Float16)a; vb[i] = (_Float16)b; }
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, vb, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)va[0]; }
}
int main() {
    float* d; hipMalloc(&d, 16);
    const float as[] = {1.0f, 9.5367431640625e-07f /*2^-20: fp16 subnormal*/, 5.9604644775390625e-08f /*2^-24: smallest subnormal*/, 6.103515625e-05f /*2^-14 min normal*/};
    for (float a : as) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, 1024.0f, d);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=%g as fp16=%g  mfma sum over k=32 of a*1024 = %g  (expected %g)\n", a, h[1], h[0], 32.0 * (double)h[1] * 1024.0);
    }
    return 0;
}
