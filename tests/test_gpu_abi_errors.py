"""Error behaviour of the C ABI (include/vnpcc.h): every entry point validates its arguments on the host and returns
VNP_EINVAL (-1) instead of launching — the reference's extensions printf() and carry on (chamfer3D.cu:145-151).  Through
_lib.call that is a VnpError.  Called with real device buffers so that only the argument under test is wrong."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rc(name, *args):
    from vn_pointcloudcompletion_amd import _lib
    return getattr(_lib.lib(), name)(torch.cuda.current_stream().cuda_stream, *args)


def test_einval_instead_of_launching(lib_built):
    from vn_pointcloudcompletion_amd._lib import ptr
    f = lambda *s: torch.zeros(*s, device="cuda")
    i = lambda *s: torch.zeros(*s, device="cuda", dtype=torch.int32)
    a, b = f(2, 8, 3), f(2, 9, 3)
    d1, d2, i1, i2, ws = f(2, 8), f(2, 9), i(2, 8), i(2, 9), torch.zeros(64, device="cuda", dtype=torch.int64)
    # Chamfer: null pointer, non-positive sizes, batch beyond the grid limit
    assert _rc("vnp_chamfer_fwd", 2, 8, None, 9, ptr(b), ptr(d1), ptr(i1), ptr(d2), ptr(i2), ptr(ws)) == -1
    assert _rc("vnp_chamfer_fwd", 2, 0, ptr(a), 9, ptr(b), ptr(d1), ptr(i1), ptr(d2), ptr(i2), ptr(ws)) == -1
    assert _rc("vnp_chamfer_fwd", 70000, 8, ptr(a), 9, ptr(b), ptr(d1), ptr(i1), ptr(d2), ptr(i2), ptr(ws)) == -1
    assert _rc("vnp_chamfer_fwd_pruned", 2, 40000, ptr(a), 9, ptr(b), ptr(d1), ptr(i1), ptr(d2), ptr(i2), ptr(ws)) == -1   # > 32768 points
    from vn_pointcloudcompletion_amd import _lib
    assert _lib.lib().vnp_chamfer_pruned_ws_bytes(2, 40000, 9) == 0
    # kNN: k larger than the reference cloud; FPS: more samples than points / more points than the kernel holds
    idx = i(2, 8, 4)
    assert _rc("vnp_knn", 2, 3, ptr(b), 9, ptr(a), 8, 10, ptr(idx), None) == -1
    assert _rc("vnp_knn", 2, 400, ptr(b), 9, ptr(a), 8, 2, ptr(idx), None) == -1                # D beyond the per-query kernel's LDS
    assert _rc("vnp_fps", 2, 8, ptr(a), 9, ptr(idx)) == -1
    assert _rc("vnp_fps", 1, 20000, ptr(a), 4, ptr(idx)) == -1
    # GEMM: missing operand, bias without a group size, split-K without scratch, unknown mode
    A, B, C = f(8, 4), f(5, 4), f(8, 5)
    assert _rc("vnp_gemm", 0, 0, 8, 5, 4, None, 4, ptr(B), 4, ptr(C), 5, None, 0, 0, 0, 1, None) == -1
    assert _rc("vnp_gemm", 0, 0, 8, 5, 4, ptr(A), 4, ptr(B), 4, ptr(C), 5, ptr(C), 5, 0, 0, 1, None) == -1
    assert _rc("vnp_gemm", 0, 0, 8, 5, 64, ptr(A), 4, ptr(B), 4, ptr(C), 5, None, 0, 0, 0, 2, None) == -1
    assert _lib.lib().vnp_gemm_set_mode(7) == -1 and _lib.get_gemm_mode() in ("split", "fp32")
    # fused linear + pool: shapes outside its tiling are refused (callers fall back to vnp_gemm + vnp_vn_pool_fwd)
    assert _rc("vnp_linear_pool_fwd", ptr(A), 4, ptr(B), 4, 1, 30, 64, 32, ptr(idx), ptr(C), 64, ptr(ws)) == -1
    # round-4 entry points: a residual without a matrix or narrower than the output, a group that does not fit the sort's LDS or whose
    # combined keys would overflow, grouped permutes beyond one token's LDS budget, a sampling gather without a destination, a loss without
    # scratch, EdgeConv weights in an unknown direction, the rank-1 EdgeConv without its weight columns
    assert _rc("vnp_gemm_add", 0, 0, 8, 5, 4, ptr(A), 4, ptr(B), 4, ptr(C), 5, None, 5, 1, None) == -1
    assert _rc("vnp_gemm_add", 0, 0, 8, 5, 4, ptr(A), 4, ptr(B), 4, ptr(C), 5, ptr(C), 3, 1, None) == -1
    assert _rc("vnp_sort_groups", ptr(idx), 2, 5000, 8, ptr(idx), ptr(idx)) == -1
    assert _rc("vnp_sort_groups", ptr(idx), 2, 4096, 1 << 20, ptr(idx), ptr(idx)) == -1
    assert _rc("vnp_rows_to_heads_grouped", ptr(A), 6000, 1, 1, 2000, 4, 3, ptr(C)) == -1
    assert _rc("vnp_gather_coords", 2, 8, ptr(a), ptr(idx), 4, None, None) == -1
    assert _rc("vnp_cd_l1_loss", ptr(d1), 16, ptr(d2), 18, None, ptr(C)) == -1
    assert _rc("vnp_cd_l1_grad", None, ptr(d1), 16, ptr(d2), 18, ptr(d1), ptr(d2)) == -1
    assert _rc("vnp_edge_weights", 2, ptr(A), ptr(A), ptr(C), 2, 2) == -1
    assert _rc("vnp_edge_fused_r1", 1, ptr(a), ptr(a), None, None, ptr(idx), 8, 4, 4, ptr(d1), ptr(d1), ptr(d1), ptr(d1), 0.2, None, ptr(C), 4,
               None, 0, None, 0, None, 0) == -1
    assert _rc("vnp_vn_add_layernorm_fwd", ptr(A), 4, None, 4, None, 0, 2, 4, ptr(d1), ptr(d1), ptr(C), 4, ptr(C), 4, ptr(d2)) == -1
    # and the Python layer turns the status into an exception
    with pytest.raises(_lib.VnpError):
        _lib.call("vnp_fps", 2, 8, ptr(a), 9, ptr(idx))
    torch.cuda.synchronize()                                    # nothing was launched, nothing faulted


def test_pack_tensors_table_of_many_sources(lib_built):
    """vnp_pack_tensors straight through the C ABI: 131 sources (three launches of up to 48 table entries), sizes from 1 to ~1 M elements,
    NULL sources (zero fill), sources and slices that are not 16-byte aligned; the words between and around the slices stay untouched."""
    import ctypes
    import torch
    from vn_pointcloudcompletion_amd._lib import call
    g = torch.Generator().manual_seed(9)
    sizes = [int(s) for s in torch.randint(1, 5000, (128,), generator=g)] + [1 << 20, 3, 1]
    offs, off = [], 5                                                  # slices start at odd offsets, one guard word between them
    for n in sizes:
        offs.append(off)
        off += n + 1
    flat = torch.full((off + 7,), 123.0, device="cuda")
    srcs = []
    for i, n in enumerate(sizes):
        if i % 7 == 3:
            srcs.append(None)
        else:
            base = torch.randn(n + 3, generator=g).cuda()
            srcs.append(base[i % 3:i % 3 + n])                          # misaligned views
    n = len(sizes)
    a_src = (ctypes.c_void_p * n)(*[None if s is None else s.data_ptr() for s in srcs])
    a_off = (ctypes.c_longlong * n)(*offs)
    a_num = (ctypes.c_longlong * n)(*sizes)
    call("vnp_pack_tensors", n, ctypes.addressof(a_src), ctypes.addressof(a_off), ctypes.addressof(a_num), flat.data_ptr())
    want = torch.full((off + 7,), 123.0, device="cuda")
    for s, o, k in zip(srcs, offs, sizes):
        want[o:o + k] = 0.0 if s is None else s
    assert torch.equal(flat, want)
    call("vnp_pack_tensors", 0, None, None, None, None)                 # an empty table is not an error
