// Fused Adam step over one flat fp32 parameter buffer (replaces torch.optim.Adam(lr, betas=(0.9, 0.999)) of
// train.py:70 for the data-parallel step: every parameter / gradient / moment of the model is a view into four
// flat buffers, so the whole optimizer step is ONE HBM-bound launch and the gradient all-reduce ONE RCCL call).
// Arithmetic follows torch's single-tensor Adam: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).
#include "vnp_common.h"

__global__ void __launch_bounds__(256) vnp_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                      float bc1, float bc2_sqrt, float weight_decay)
{
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        vnp_f4 pp = ((vnp_f4*)p)[i], gg = ((const vnp_f4*)g)[i], mm = ((vnp_f4*)m)[i], vv = ((vnp_f4*)v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ge = gg[e];
            if (weight_decay != 0.f) ge += weight_decay * pp[e];
            mm[e] = b1 * mm[e] + (1.f - b1) * ge;
            vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
            pp[e] -= (lr / bc1) * mm[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
        }
        ((vnp_f4*)p)[i] = pp; ((vnp_f4*)m)[i] = mm; ((vnp_f4*)v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        float ge = g[i];
        if (weight_decay != 0.f) ge += weight_decay * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * ge;
        const float vi = b2 * v[i] + (1.f - b2) * ge * ge;
        m[i] = mi; v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

// all four buffers must be 16-byte aligned
extern "C" int vnp_adam_step(hipStream_t st, float* p, const float* g, float* m, float* v, long long n,
                             float lr, float beta1, float beta2, float eps, int step, float weight_decay)
{
    if (!p || !g || !m || !v || n <= 0 || step <= 0) return VNP_EINVAL;
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return VNP_EINVAL;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2 = (float)(1.0 - pow((double)beta2, (double)step));
    long long blocks = ((n >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(vnp_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps, bc1, sqrtf(bc2), weight_decay);
    return vnp_last_error();
}

// ---- gather of per-parameter gradient tensors into the flat gradient buffer ------------------------------------------------------
// autograd hands every parameter its gradient as a tensor of its own (when .grad is empty it keeps the incoming tensor instead of
// adding it into a preset view: no per-parameter add launch); one launch then copies up to VNP_PACK_MAX of them to their slices of the
// flat buffer (a NULL source zero-fills the slice).  The table travels in the kernel arguments, so nothing on the host has to outlive
// the call.
#define VNP_PACK_MAX 48
struct VnpPackArgs {
    const float* src[VNP_PACK_MAX];
    long long off[VNP_PACK_MAX];
    long long n[VNP_PACK_MAX];
    float* dst;
};

__global__ void __launch_bounds__(256) vnp_pack_kernel(const VnpPackArgs a)
{
    const int t = blockIdx.y;
    const float* __restrict__ s = a.src[t];
    float* __restrict__ d = a.dst + a.off[t];
    const long long n = a.n[t];
    const long long stride = (long long)gridDim.x * 256, i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool vec = ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0;
    if (vec) {
        const long long n4 = n >> 2;
        for (long long i = i0; i < n4; i += stride) {
            vnp_f4 v = {0.f, 0.f, 0.f, 0.f};
            if (s) v = ((const vnp_f4*)s)[i];
            ((vnp_f4*)d)[i] = v;
        }
        for (long long i = (n4 << 2) + i0; i < n; i += stride) d[i] = s ? s[i] : 0.f;
    } else {
        for (long long i = i0; i < n; i += stride) d[i] = s ? s[i] : 0.f;
    }
}

// dst[dst_off[i] .. + numel[i]) = src[i][0 .. numel[i])  (src[i] == NULL: zeros), i < count.  src / dst_off / numel are HOST arrays.
extern "C" int vnp_pack_tensors(hipStream_t st, int count, const void* const* src, const long long* dst_off, const long long* numel, float* dst)
{
    if (count < 0 || (count > 0 && (!src || !dst_off || !numel || !dst))) return VNP_EINVAL;
    for (int base = 0; base < count; base += VNP_PACK_MAX) {
        VnpPackArgs a;
        const int c = count - base < VNP_PACK_MAX ? count - base : VNP_PACK_MAX;
        long long big = 0;
        for (int i = 0; i < VNP_PACK_MAX; ++i) {
            const int k = i < c ? base + i : base;               // unused slots repeat a valid entry (never launched: gridDim.y = c)
            if (i < c && (numel[k] < 0 || dst_off[k] < 0)) return VNP_EINVAL;
            a.src[i] = (const float*)src[k]; a.off[i] = dst_off[k]; a.n[i] = numel[k];
            if (i < c && numel[k] > big) big = numel[k];
        }
        a.dst = dst;
        if (big == 0) continue;
        long long bx = (big / 4 + 255) / 256;
        if (bx > 512) bx = 512;
        if (bx < 1) bx = 1;
        hipLaunchKernelGGL(vnp_pack_kernel, dim3((unsigned)bx, (unsigned)c), dim3(256), 0, st, a);
    }
    return vnp_last_error();
}
