"""Loader for libvnpcc.so (the C-ABI of include/vnpcc.h) and thin ctypes helpers.

The library is built in-tree by ``build()`` (``hipcc --offload-arch=gfx950``) and loaded with ctypes.
There is NO fallback: if the library is missing or a kernel launch fails the product raises.
"""
from __future__ import annotations

import ctypes
import glob
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libvnpcc.so")
_LIB = None


class VnpError(RuntimeError):
    """A libvnpcc entry point returned non-zero (bad argument or HIP launch error)."""


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _source_hash():
    """SHA-256 over the names and contents of every kernel source and header: what decides whether the library is stale (file times do
    not survive a copy to another machine)."""
    import hashlib
    h = hashlib.sha256()
    for f in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))):
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


_HASH_PATH = os.path.join(CSRC, "build", "libvnpcc.srchash")


def _up_to_date():
    if not os.path.exists(LIB_PATH):
        return False
    try:
        with open(_HASH_PATH) as fh:
            return fh.read().strip() == _source_hash()
    except OSError:                               # a library built before the hash existed: fall back to file times
        deps = sources() + glob.glob(os.path.join(CSRC, "*.h"))
        return all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into csrc/libvnpcc.so (cross-compiles without a GPU).  An up-to-date library (source hash
    recorded beside it) is returned without touching the file system, so a read-only install works.  A stale one is rebuilt under a
    file lock: the ranks of a multi-GPU launch build it once, not N times into the same files.  When the rebuild is impossible (no
    compiler, read-only tree) and a library exists, that library is used with a warning: a usable kernel library must not abort a
    launch because its sources were touched."""
    import fcntl
    import warnings
    if not force and _up_to_date():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    try:
        if not os.path.exists(hipcc):
            raise OSError(f"compiler {hipcc} not found")
        os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
        lock = open(os.path.join(CSRC, "build", ".lock"), "w")
    except OSError as e:
        if os.path.exists(LIB_PATH) and not force:
            warnings.warn(f"libvnpcc.so is older than its sources and cannot be rebuilt here ({e}); using the existing library")
            return LIB_PATH
        raise
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _up_to_date():       # another rank built it while this one waited for the lock
                return LIB_PATH
            return _build_locked(hipcc, force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(hipcc, force, verbose):
    srcs = sources()
    src_hash = _source_hash()
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(CSRC, "build", os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if (not force and os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(s)
                and all(os.path.getmtime(o) >= os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h")))):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{out.decode()}")
    tmp = LIB_PATH + ".tmp"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs)
    os.replace(tmp, LIB_PATH)                     # a process that has the old file mapped keeps it; nobody sees a half-written one
    with open(_HASH_PATH + ".tmp", "w") as fh:
        fh.write(src_hash + "\n")
    os.replace(_HASH_PATH + ".tmp", _HASH_PATH)
    return LIB_PATH


_vp, _i, _ll, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float

# name -> argtypes (after the leading hipStream_t); mirrors include/vnpcc.h
_SIGNATURES = {
    "vnp_chamfer_fwd": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "vnp_chamfer_bwd": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "vnp_chamfer_fwd_pruned": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "vnp_dcd_side": [_i, _i, _i, _vp, _vp, _f, _f, _f, _vp, _vp, _vp],
    "vnp_cd_l1_loss": [_vp, _ll, _vp, _ll, _vp, _vp],
    "vnp_cd_l1_grad": [_vp, _vp, _ll, _vp, _ll, _vp, _vp],
    "vnp_emd_approxmatch": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "vnp_emd_matchcost": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "vnp_emd_matchcost_bwd": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "vnp_gemm": [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp],
    "vnp_gemm_vn_stats": [_i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp],
    "vnp_gemm_vn_stats_ex": [_i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "vnp_gemm_add": [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp],
    "vnp_gemm_ex": [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "vnp_absmax": [_vp, _ll, _ll, _i, _vp],
    "vnp_gemm_batched": [_i, _i, _i, _i, _i, _vp, _i, _ll, _vp, _i, _ll, _vp, _i, _ll, _i, _i],
    "vnp_vn_layernorm_fwd": [_vp, _i, _ll, _i, _vp, _vp, _vp, _i, _vp],
    "vnp_vn_layernorm_bwd": [_vp, _i, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "vnp_vn_add_layernorm_fwd": [_vp, _i, _vp, _i, _vp, _i, _ll, _i, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "vnp_vn_layernorm_bwd_add": [_vp, _i, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "vnp_rows_to_heads": [_vp, _i, _i, _i, _i, _i, _vp],
    "vnp_heads_to_rows": [_vp, _i, _i, _i, _i, _vp, _i],
    "vnp_rows_to_heads_grouped": [_vp, _i, _i, _i, _i, _i, _i, _vp],
    "vnp_heads_to_rows_grouped": [_vp, _i, _i, _i, _i, _i, _vp, _i],
    "vnp_rows_to_heads_pad": [_vp, _i, _i, _i, _i, _i, _vp, _i],
    "vnp_rows_to_heads_fm": [_vp, _i, _i, _i, _i, _i, _vp, _i],
    "vnp_heads_fm_to_rows": [_vp, _i, _i, _i, _i, _i, _vp, _i],
    "vnp_softmax_fwd": [_vp, _ll, _i, _f],
    "vnp_softmax_bwd": [_vp, _vp, _ll, _i, _f],
    "vnp_vn_norm_stats": [_vp, _i, _ll, _i, _vp, _i],
    "vnp_vn_bn_finalize": [_vp, _i, _i, _ll, _vp, _vp, _vp, _vp, _f, _f],
    "vnp_rstd_from_var": [_vp, _vp, _i, _f],
    "vnp_vn_bn_leaky_fwd": [_vp, _i, _vp, _i, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _f],
    "vnp_vn_bn_leaky_bwd1": [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i],
    "vnp_vn_bn_leaky_bwd_pass": [_i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i,
                                 _vp, _ll, _vp],
    "vnp_vn_bn_leaky_rowdot_fwd": [_vp, _i, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp],
    "vnp_vn_bn_bwd2": [_vp, _i, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _vp, _ll],
    "vnp_reduce_rows": [_vp, _i, _ll, _i, _ll, _i, _vp, _ll],
    "vnp_reduce_pairs": [_vp, _i, _i, _vp, _vp],
    "vnp_linear_pool_fwd": [_vp, _i, _vp, _i, _ll, _i, _i, _i, _vp, _vp, _i, _vp],
    "vnp_linear_pool_store_fwd": [_vp, _i, _vp, _i, _ll, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp],
    "vnp_linear_pool_screened_fwd": [_vp, _i, _vp, _i, _ll, _i, _i, _i, _vp, _vp, _i, _vp, _i],
    "vnp_vn_pool_fwd": [_vp, _i, _vp, _i, _ll, _i, _i, _vp, _vp, _i, _vp, _vp],
    "vnp_vn_pool_bwd": [_vp, _i, _vp, _ll, _i, _i, _vp, _i],
    "vnp_vn_pool_bwd_acc": [_vp, _i, _vp, _ll, _i, _i, _vp, _i],
    "vnp_sort_groups": [_vp, _ll, _i, _i, _vp, _vp],
    "vnp_pool_lin_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i],
    "vnp_vn_mean_pool_fwd": [_vp, _i, _ll, _i, _i, _vp, _i],
    "vnp_vn_mean_pool_bwd": [_vp, _i, _ll, _i, _i, _vp, _i],
    "vnp_fold_fwd": [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i],
    "vnp_fold_bwd": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp],
    "vnp_knn": [_i, _i, _vp, _i, _vp, _i, _i, _vp, _vp],
    "vnp_knn_matrix": [_i, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp],
    "vnp_fps": [_i, _i, _vp, _i, _vp],
    "vnp_fps_ex": [_i, _i, _vp, _i, _vp, _i],
    "vnp_gather_points": [_vp, _i, _vp, _ll, _i, _vp, _i],
    "vnp_gather_coords": [_i, _i, _vp, _vp, _i, _vp, _vp],
    "vnp_scatter_add_points": [_vp, _i, _vp, _ll, _i, _vp, _i, _ll, _vp],
    "vnp_vn_bn_finalize_chunked": [_vp, _i, _i, _i, _ll, _vp, _vp, _vp, _vp, _f, _f],
    "vnp_edge_fused": [_i, _vp, _i, _vp, _i, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp, _i, _vp, _ll, _vp, _i, _ll,
                       _vp, _i],
    "vnp_edge_reverse": [_vp, _ll, _i, _ll, _vp, _vp, _vp],
    "vnp_edge_fused_gather": [_vp, _i, _vp, _i, _vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _ll, _vp, _i],
    "vnp_edge_weights": [_i, _vp, _vp, _vp, _i, _i],
    "vnp_edge_fused_r1": [_i, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp, _i, _vp, _ll, _vp, _i],
    "vnp_edge_fused_gather_r1": [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _ll, _vp, _i],
    "vnp_edge_gather_add": [_vp, _i, _vp, _i, _vp, _ll, _i, _i, _vp, _i],
    "vnp_fold_fused": [_i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _ll,
                       _vp, _vp, _i],
    "vnp_bn1d_act": [_i, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _i, _vp, _i, _vp, _vp, _ll],
    "vnp_group_fold_fused": [_i, _vp, _i, _vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _ll, _vp, _vp, _vp, _i],
    "vnp_vn_dir_layer": [_i, _vp, _i, _vp, _vp, _i, _ll, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _ll, _vp, _i, _vp],
    "vnp_vn_dir_layer_ex": [_i, _vp, _i, _vp, _vp, _i, _ll, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _ll, _vp, _i, _vp, _i],
    "vnp_unit_dirs": [_vp, _ll, _vp],
    "vnp_unit_dirs_r": [_vp, _ll, _vp, _vp],
    "vnp_dir_contract": [_vp, _i, _vp, _vp, _i, _ll, _i, _i],
    "vnp_dir_pool_fwd": [_vp, _i, _vp, _ll, _i, _i, _vp, _vp, _i],
    "vnp_dir_pool_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _ll, _i, _i, _i, _vp, _i, _vp, _i],
    "vnp_rowdot": [_vp, _i, _vp, _ll, _i, _vp, _i, _vp],
    "vnp_group_sum3": [_vp, _ll, _i, _vp, _i],
    "vnp_colsum": [_vp, _ll, _ll, _i, _ll, _i, _vp, _vp],
    "vnp_outer": [_vp, _vp, _ll, _i, _vp, _i],
    "vnp_rows_to_ref": [_vp, _i, _i, _i, _i, _vp],
    "vnp_ref_to_rows": [_vp, _i, _i, _i, _vp, _i],
    "vnp_adam_step": [_vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _i, _f],
    "vnp_pack_tensors": [_i, _vp, _vp, _vp, _vp],
    "vnp_edge_gn_max": [_i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i],
    "vnp_edge_gn_finalize": [_vp, _i, _i, _i, _i, _i, _f, _vp, _vp],
    "vnp_edge_gn_gather": [_vp, _i, _vp, _i, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp, _vp, _i],
    "vnp_bias_act": [_vp, _i, _vp, _ll, _i, _i, _f, _vp, _i, _vp, _i, _vp, _i],
    "vnp_rows_max": [_vp, _i, _i, _i, _i, _vp, _vp],
    "vnp_rows_max_bwd": [_vp, _vp, _i, _i, _i, _vp],
}
_NO_STREAM = {"vnp_gemm_profile_next": [_vp, _vp, _vp], "vnp_gemm_profile_cancel": [], "vnp_chunk_points": [], "vnp_chunk_points_pc": [_ll, _i], "vnp_gemm_set_mode": [_i], "vnp_gemm_get_mode": [], "vnp_edge_block_queries": [], "vnp_chamfer_pruned_ws_bytes": [_i, _i, _i], "vnp_chamfer_pruned_ws_wave_counts": [_i, _i, _i],
              "vnp_linear_pool_ws_bytes": [_ll, _i, _i], "vnp_linear_pool_screened_ws_bytes": [_ll, _i, _i, _i],
              "vnp_gemm_vn_stats_ws_floats": [_ll, _i], "vnp_gemm_tile_rows": [_i, _i, _i, _i, _i, _i], "vnp_bn1d_chunk_rows": []}
_RESTYPE = {"vnp_chamfer_pruned_ws_bytes": _ll, "vnp_chamfer_pruned_ws_wave_counts": _ll, "vnp_linear_pool_ws_bytes": _ll, "vnp_linear_pool_screened_ws_bytes": _ll,
            "vnp_gemm_vn_stats_ws_floats": _ll}


def exported_symbols():
    """Every symbol include/vnpcc.h declares (used by the CPU-side ABI test)."""
    return sorted(list(_SIGNATURES) + list(_NO_STREAM))


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise VnpError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU / PyTorch fallback for the HIP kernels)")
        L = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = [_vp] + args
            fn.restype = _i
        for name, args in _NO_STREAM.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, _i)
        _LIB = L
        mode = os.environ.get("VNP_GEMM_MODE", "split").lower()
        if mode not in ("split", "fp32"):
            raise VnpError(f"VNP_GEMM_MODE must be 'split' or 'fp32', got {mode!r}")
        L.vnp_gemm_set_mode(1 if mode == "fp32" else 0)
    return _LIB


class KernelTimer:
    """Raw HIP events for vnp_gemm_profile_next (bench.py): the library records them around the ONE dominant kernel of the next
    matrix-product launch, so the time excludes slab reductions, the pool's candidate passes and launch gaps (torch.cuda.Event
    brackets on the Python side include them).  ``kind`` is filled in by the library (see include/vnpcc.h)."""
    _hip = None

    def __init__(self):
        if KernelTimer._hip is None:
            KernelTimer._hip = ctypes.CDLL("libamdhip64.so")
        h = KernelTimer._hip
        self.e0, self.e1 = ctypes.c_void_p(), ctypes.c_void_p()
        for e in (self.e0, self.e1):
            if h.hipEventCreate(ctypes.byref(e)) != 0:
                raise VnpError("hipEventCreate failed")
        self._kind = ctypes.c_int(-1)

    def arm(self):
        if lib().vnp_gemm_profile_next(self.e0, self.e1, ctypes.byref(self._kind)) != 0:
            raise VnpError("vnp_gemm_profile_next failed")

    @property
    def kind(self):
        return self._kind.value

    def seconds(self):
        """Kernel duration (call after the stream has been synchronised); None if no kernel consumed the request."""
        if self._kind.value < 0:
            return None
        ms = ctypes.c_float()
        if KernelTimer._hip.hipEventElapsedTime(ctypes.byref(ms), self.e0, self.e1) != 0:
            return None
        return ms.value * 1e-3

    def __del__(self):
        try:                                    # (at interpreter shutdown the class attribute may already be gone)
            if self._kind.value < 0 and _LIB is not None:
                _LIB.vnp_gemm_profile_cancel()   # never consumed: make sure the library does not keep these handles
            h = KernelTimer._hip
            for e in (self.e0, self.e1):
                if h is not None and e:
                    h.hipEventDestroy(e)
        except Exception:
            pass


def set_gemm_mode(mode: str) -> None:
    """'split' (default): fp32 operands split exactly into three bf16 terms, six bf16 MFMA products, fp32 accumulation;
    'fp32': fp32-input MFMA.  Same results to fp32 rounding (tests/test_gpu_gemm.py); process-wide."""
    if mode not in ("split", "fp32"):
        raise VnpError(f"gemm mode must be 'split' or 'fp32', got {mode!r}")
    if lib().vnp_gemm_set_mode(1 if mode == "fp32" else 0) != 0:
        raise VnpError("vnp_gemm_set_mode failed")


def get_gemm_mode() -> str:
    return "fp32" if lib().vnp_gemm_get_mode() == 1 else "split"


def chunk_points() -> int:
    return lib().vnp_chunk_points()


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


_FN = {}                                               # entry point name -> bound ctypes function
# the raw handle of torch's current stream without building a torch.cuda.Stream object per launch (the step of the transformer configs
# is ~800 launches, and the host falls behind the GPU in their encoder's forward): a private torch call, with the public one as fallback
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _current_stream():
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke a stream-taking entry point on torch's current HIP stream; raise on a non-zero status."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib(), name)
    rc = fn(_current_stream(), *args)
    if rc != 0:
        raise VnpError(f"{name} failed with status {rc}" + (" (invalid argument)" if rc == -1 else " (hipError_t)"))


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise VnpError("libvnpcc kernels need tensors on a HIP device (no CPU fallback); got a CPU tensor")
