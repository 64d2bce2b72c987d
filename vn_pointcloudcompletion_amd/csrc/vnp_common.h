// Shared helpers for the gfx950 HIP kernels behind the C-ABI in include/vnpcc.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define VNP_OK 0
#define VNP_EINVAL (-1)

#define VNP_EPS 1e-6f          // models/vn_layers.py:10
#define VNP_WAVE 64

static inline int vnp_last_error() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VNP_OK : (int)e;
}

static inline int vnp_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Raise a kernel's dynamic-LDS limit once per DEVICE (the attribute belongs to the current device's code object).  Thread-safe
// and re-entrant: the call is idempotent, `done` (one bit per device ordinal) only avoids repeating it on every launch.
static inline hipError_t vnp_allow_dynamic_lds(std::atomic<unsigned long long>& done, const void* kernel, int bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

typedef float vnp_f4 __attribute__((ext_vector_type(4)));
typedef float vnp_f2 __attribute__((ext_vector_type(2)));
typedef float vnp_f16 __attribute__((ext_vector_type(16)));

// Activation layout used by every VN kernel ("rows layout"):
//   a VN feature map with P points and C channels is a row-major float matrix [P*3, ld],
//   row = point*3 + t (t = Cartesian component), channels contiguous.
// The reference's [B, C, 3, N] tensors (models/vn_layers.py:19) map to it by
//   rows[(b*N + n)*3 + t][c] = x[b][c][t][n].

// ---- wave-wide reductions on DPP (data-parallel primitives: the operand of a VALU instruction comes from another lane at no extra
// latency), for chains where the ~12 dependent ds_bpermute round trips of a __shfl_xor butterfly are the critical path (FPS: one
// arg-max per round, 500+ rounds).  Four steps reduce inside each row of 16 lanes (two quad permutes, row rotations by 4 and 8), then
// row_bcast:15 / row_bcast:31 fold the rows: lane 63 holds the result, v_readlane makes it wave-uniform.
#ifdef __HIPCC__
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int vnp_dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ float vnp_wave_max_f32(float v)
{
#define VNP_STEP(CTRL, MASK) v = fmaxf(v, __int_as_float(vnp_dpp_i<CTRL, MASK>(__float_as_int(v))))
    VNP_STEP(0xB1, 0xF); VNP_STEP(0x4E, 0xF); VNP_STEP(0x124, 0xF); VNP_STEP(0x128, 0xF); VNP_STEP(0x142, 0xA); VNP_STEP(0x143, 0xC);
#undef VNP_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int vnp_wave_min_i32(int v)
{
#define VNP_STEP(CTRL, MASK) v = min(v, vnp_dpp_i<CTRL, MASK>(v))
    VNP_STEP(0xB1, 0xF); VNP_STEP(0x4E, 0xF); VNP_STEP(0x124, 0xF); VNP_STEP(0x128, 0xF); VNP_STEP(0x142, 0xA); VNP_STEP(0x143, 0xC);
#undef VNP_STEP
    return __builtin_amdgcn_readlane(v, 63);
}
#endif
