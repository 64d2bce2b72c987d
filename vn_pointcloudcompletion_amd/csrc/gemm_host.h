// Host-side pieces shared by the matrix-product files (gemm.hip: the products themselves; pool_gemm.hip: VNLinear -> VNMaxPool in the
// product's epilogue and its exact screened search): the kernel templates, the measurement hook, the CU count.
#pragma once
#include "gemm_split_kernel.h"
#include "gemm_split_wide_kernel.h"

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

static inline void vnp_prof_begin(hipStream_t st) { if (g_vnp_prof.e0) (void)hipEventRecord(g_vnp_prof.e0, st); }
static inline void vnp_prof_end(hipStream_t st, int kind)
{
    if (!g_vnp_prof.e0) return;
    (void)hipEventRecord(g_vnp_prof.e1, st);
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
