// Host-side pieces shared by the matrix-product files (gemm.hip: the products themselves; pool_gemm.hip: VNLinear -> VNMaxPool in the
// product's epilogue and its exact screened search): the kernel templates, the measurement hook, the CU count.
#pragma once
#include "gemm_split_kernel.h"
#include "gemm_split_wide_kernel.h"
#include <hip/hip_ext.h>

extern "C" int vnp_gemm_get_mode(void);

// Measurement hook (bench.py): the NEXT matrix-product launch made by the calling host thread records e0 / e1 around its ONE dominant
// kernel on the launch stream — not around the slab reduction, the pool's candidate passes or anything else the entry point enqueues —
// and stores what that kernel was in *kind (host memory):
//   0 six-product bf16 split kernel, 1 fp32-input MFMA kernel, 2 streaming kernel of a degenerate shape (no MFMA),
//   3 screening pass, one fp16 product, 4 screening pass, three bf16 products, 5 two fp16 planes / three products (vnp_gemm_ex).
// Thread-local and consumed by that launch; a launch without a pending request costs one branch.
struct VnpProfReq { hipEvent_t e0, e1; int* kind; };
extern thread_local VnpProfReq g_vnp_prof;             // defined in gemm.hip

// Every matrix-product entry point holds one of these: a request that the call did not consume (argument error, a path without a
// bracketed kernel) is dropped when the call returns instead of arming some later, unrelated launch with stale event handles.
struct VnpProfScope { ~VnpProfScope() { g_vnp_prof = {nullptr, nullptr, nullptr}; } };

// The bracketed kernel is launched through VNP_KLAUNCH: with a request pending it goes out by hipExtLaunchKernelGGL, which stamps e0 / e1
// from the dispatch packet's OWN start / end — no marker packets in the queue.  (Rounds 3-4 recorded two events around the launch: each
// is a barrier packet of its own, and the ~45 instrumented launches of a step cost the timed region 0.26-0.36 ms on config 2, 1.2-1.4 ms
// on config 5's 150 products: scripts/experiments/hook_overhead.py.)
#define VNP_KLAUNCH(kern, grid, block, lds, st, ...)                                                                              \
    do {                                                                                                                         \
        if (g_vnp_prof.e0) hipExtLaunchKernelGGL(kern, grid, block, lds, st, g_vnp_prof.e0, g_vnp_prof.e1, 0, __VA_ARGS__);       \
        else hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                                          \
    } while (0)
static inline void vnp_prof_begin(hipStream_t) {}
static inline void vnp_prof_end(hipStream_t, int kind)
{
    if (!g_vnp_prof.e0) return;
    if (g_vnp_prof.kind) *g_vnp_prof.kind = kind;
    g_vnp_prof = {nullptr, nullptr, nullptr};
}

static inline int vnp_num_cus()
{
    static std::atomic<int> cached{0};             // every GPU of a node is the same part: one value serves all devices
    int n = cached.load(std::memory_order_relaxed);
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cached.store(n, std::memory_order_relaxed);
    }
    return n;
}
