"""Tensor-level wrappers and autograd Functions over libvnpcc (include/vnpcc.h).

Everything here works on the *rows layout*: a VN feature map with P points and C channels is a
2-D fp32 tensor ``[P*3, C]`` (row = point*3 + t, channels contiguous, ``stride(1) == 1``).
torch supplies device memory, the current HIP stream and the autograd graph; all arithmetic
is done by the HIP kernels.  No function in this file has a CPU / PyTorch fallback.
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import call, ptr

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# VNMaxPool direction d = Wd (W x): evaluate as (Wd W) x when the pooled tensor is the output of a VNLinear (see
# VNLinearMaxPoolFn).  Set VNP_POOL_AS_WRITTEN=1 to keep the reference's evaluation order.
import os as _os
POOL_DIRECTION_REASSOCIATE = _os.environ.get("VNP_POOL_AS_WRITTEN", "0") != "1"
# EdgeConv (configs 3, 5): recompute the edge pre-activations in every pass instead of materialising the k-times larger edge
# tensor (EdgeConvFn); VNP_EDGE_UNFUSED=1 keeps gather-add -> BatchNorm/leaky -> mean-pool as separate kernels
EDGE_FUSED = _os.environ.get("VNP_EDGE_UNFUSED", "0") != "1"
# VNLinearLeakyReLU: take the BatchNorm statistics of p from the GEMM's accumulators (vnp_gemm_vn_stats) instead of re-reading
# p; VNP_STATS_UNFUSED=1 keeps the separate vnp_vn_norm_stats pass
STATS_IN_EPILOGUE = _os.environ.get("VNP_STATS_UNFUSED", "0") != "1"
# VNLinear -> VNMaxPool: fuse the pool into the GEMM epilogue (vnp_linear_pool_fwd) when shapes allow; VNP_POOL_UNFUSED=1 keeps
# the two GEMMs + pool kernels
POOL_FUSED_EPILOGUE = _os.environ.get("VNP_POOL_UNFUSED", "0") != "1"
# ... and pick the arg-max with the exact SCREENED search (vnp_linear_pool_screened_fwd: a 3-product bracketing pass + exact
# evaluation of the surviving candidates; same bits, half the MFMA work); VNP_POOL_SCREEN=0 keeps the full 6-product epilogue
POOL_SCREENED = _os.environ.get("VNP_POOL_SCREEN", "1") != "0"
# bracketing pass of the screened pool: "fp16" (default) = one fp16 product per operand pair, "bf16x3" = three bf16 products
POOL_SCREEN_PASS = _os.environ.get("VNP_POOL_SCREEN_PASS", "fp16")
# BatchNorm + leaky backward in two read-heavy passes (sums, then final gradients) instead of bwd1 -> reduce -> bwd2, which writes
# dL/dq and reads it back (VNP_BN_BWD_TWO_PASS=0 keeps that form); same arithmetic term for term
BN_BWD_TWO_PASS = _os.environ.get("VNP_BN_BWD_TWO_PASS", "1") != "0"
# A VNLinear whose output feeds ONLY a VNMaxPool and (concatenated with the pooled feature) a VNLinearLeakyReLU — encoder
# first_conv[1] -> maxpool1 / second_conv[0], models/pcn.py:168-173 — is a linear map in front of linear maps: W2 (W1 a) is evaluated
# as (W2 W1) a, contracting over first_conv[1]'s 128 INPUT channels instead of its 512 outputs (4x fewer FLOPs in the layer's three
# big GEMMs), and the 512-channel tensor is never stored.  Same arithmetic up to fp32 re-association, like POOL_DIRECTION_REASSOCIATE;
# VNP_ENC_AS_WRITTEN=1 keeps the reference's evaluation order.
ENC_LINEAR_REASSOCIATE = _os.environ.get("VNP_ENC_AS_WRITTEN", "0") != "1"
# ... and, since every VN vector behind VN_PointNet's single-channel first layer is parallel to its point's direction, run that layer on
# per-point scalars (VNDirLinearLeakyFn: a third of the GEMM rows, no [P*3, 2C] pre-activation tensor); VNP_ENC_DIR_LAYER=0 keeps the rows form
ENC_DIR_LAYER = _os.environ.get("VNP_ENC_DIR_LAYER", "1") != "0"
# ... and the single-channel first layer itself on the per-point scalar |x| (scalar output); VNP_ENC_FIRST_LAYER_SCALAR=0 keeps it in the
# rows layout and takes the scalars by contraction with the unit directions
ENC_FIRST_LAYER_SCALAR = _os.environ.get("VNP_ENC_FIRST_LAYER_SCALAR", "1") != "0"
# VNLinear -> VNMaxPool with a small contraction (K < this): the full fused epilogue costs less than the screened search's fixed passes
POOL_SCREEN_MIN_K = 256
# Opt-in (VNP_GEMM_FP16X2=1): large aligned GEMMs (K >= 64) on two fp16 planes of the power-of-two scaled operands and three MFMA products
# (vnp_gemm_ex; normwise error against fp64 at or below an fp32 sgemm's) instead of three bf16 planes and six products.  Off by default:
# operand precision is narrowed to 22-23 bits, and the absmax pass over each operand costs what the GEMM saves.
GEMM_FP16X2 = _os.environ.get("VNP_GEMM_FP16X2", "0") == "1"
GEMM_FP16X2_MIN_FLOP = 2.0e10
# tests set this to a list to receive the screened pool's diagnostics tensor (int32 [3]: candidates, fallback flag, capacity)
POOL_SCREEN_STATS = None
# If the screened search had to fall back (candidate list overflow, non-finite bounds) the bracketing pass was wasted work.  The flag of
# every call is copied to pinned host memory WITHOUT a sync and looked at when the next call of the same shape comes round: after a
# fallback the next POOL_SCREEN_BACKOFF calls of that shape go straight to the full six-product epilogue, then the search is tried again.
POOL_SCREEN_BACKOFF = 64
_screen_state = {}        # shape key -> {"host": pinned int32[4], "event": Event or None, "skip": calls left on the full path}


def _screen_allowed(key):
    st = _screen_state.get(key)
    if st is None:
        return True
    if st["event"] is not None and st["event"].query():
        if int(st["host"][1]) != 0:
            st["skip"] = POOL_SCREEN_BACKOFF
            POOL_PATH_COUNTS["fallback_flags_seen"] += 1
        st["event"] = None
    if st["skip"] > 0:
        st["skip"] -= 1
        return False
    return True


def _screen_report(key, ws):
    st = _screen_state.get(key)
    if st is None:
        st = _screen_state[key] = {"host": torch.zeros(4, dtype=torch.int32).pin_memory(), "event": None, "skip": 0}
    if st["event"] is None:                      # one copy in flight per shape is enough
        st["host"].copy_(ws[:2].view(torch.int32), non_blocking=True)
        st["event"] = torch.cuda.Event()
        st["event"].record()

# bench.py sets this to a list to time every matrix-product launch: entries are (variant, flops, nsplit, (M, N, K), timer) with
# timer a _lib.KernelTimer — HIP events recorded by the library around the launch's ONE dominant kernel on its launch stream
GEMM_PROFILE = None
# calls of the VNLinear -> VNMaxPool pool by path since import (bench.py reports the timed steps' share): the exact screened search,
# the full six-product epilogue (search skipped: short contraction, back-off after a fallback), and device-side fallbacks seen
POOL_PATH_COUNTS = {"screened": 0, "full_epilogue": 0, "unfused": 0, "fallback_flags_seen": 0}


def _prof_arm():
    prof = GEMM_PROFILE
    if prof is None:
        return None
    t = _lib.KernelTimer()
    t.arm()
    return t
# tests set this to a dict to learn which code paths a forward pass took: path name -> launch count, and under "pool_idx" the
# arg-max index tensors of the VNMaxPool calls in call order (tests/test_gpu_fullres_parity.py)
TRACE = None


def _trace(name, idx=None):
    tr = TRACE
    if tr is not None:
        tr[name] = tr.get(name, 0) + 1
        if idx is not None:
            tr.setdefault("pool_idx", []).append(idx)


# ------------------------------------------------------------------------------------------ helpers
def _rows(t: torch.Tensor) -> torch.Tensor:
    """Return t as a 2-D fp32 device tensor with unit inner stride (copying only if needed)."""
    _lib.require_gpu(t)
    if t.dtype != torch.float32:
        raise _lib.VnpError(f"fp32 expected, got {t.dtype}")
    if t.dim() != 2:
        raise _lib.VnpError(f"2-D rows tensor expected, got shape {tuple(t.shape)}")
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


def _nchunk(P: int) -> int:
    c = _lib.chunk_points()
    return (P + c - 1) // c


def _chunk_pc(P: int, C: int):
    """(points per workgroup, number of chunks) of the per-channel BatchNorm / leaky passes over a [P, C] map (vnp_chunk_points_pc:
    1024, halved until the grid fills the chip)."""
    c = _lib.lib().vnp_chunk_points_pc(P, C)
    return c, (P + c - 1) // c


def set_amax(t, word):
    """Attach the device word holding the bits of max |t| (written by the kernel that produced ``t``) to the tensor; it is trusted
    for as long as the tensor's version counter stands.  Code that rewrites ``t`` through a raw pointer must call ``drop_amax``."""
    t._vnp_amax = (word, t._version)
    return t


def drop_amax(t):
    if hasattr(t, "_vnp_amax"):
        del t._vnp_amax


def amax_word(t):
    """int32 [1] device tensor: bit pattern of max |t| (2-D, unit inner stride) — the producer's word when there is one, else one
    vnp_absmax pass.  None when the layout is outside vnp_absmax's contract (callers then use the six-product GEMM)."""
    a = getattr(t, "_vnp_amax", None)
    if a is not None and a[1] == t._version:
        return a[0]
    rows, cols = t.shape
    ld = _ld(t)
    if cols % 4 or ld % 4 or t.data_ptr() % 16:
        return None
    w = torch.empty(1, device=t.device, dtype=torch.int32)
    call("vnp_absmax", ptr(t), ld, rows, cols, ptr(w))
    _trace("absmax_pass")
    return w


def _fp16x2_shape(op_a, M, N, K, A, B, nsplit):
    """The shapes vnp_gemm_ex runs on the two-plane path (gemm.hip: aligned tiles, K >= 64) that are worth two absmax launches."""
    bm = 192 if op_a == 0 else 128
    return (GEMM_FP16X2 and K >= 64 and K % 32 == 0 and M % bm == 0 and N % 128 == 0 and 2.0 * M * N * K >= GEMM_FP16X2_MIN_FLOP
            and _lib.get_gemm_mode() == "split" and _ld(A) % 4 == 0 and _ld(B) % 4 == 0 and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def gemm(A, B, *, op_a=0, op_b=0, out=None, bias=None, rows_per_group=0, accumulate=False, nsplit=None, amax_a=None, amax_b=None):
    """C = op(A) op(B) (+ per-sample bias).  See vnp_gemm / vnp_gemm_ex in include/vnpcc.h.  ``amax_a`` / ``amax_b``: the operands'
    ``amax_word`` when the caller already holds it (an operand used by two GEMMs)."""
    A, B = _rows(A), _rows(B)
    M, K = (A.shape[0], A.shape[1]) if op_a == 0 else (A.shape[1], A.shape[0])
    N, Kb = (B.shape[0], B.shape[1]) if op_b == 0 else (B.shape[1], B.shape[0])
    if K != Kb:
        raise _lib.VnpError(f"gemm: inner dimensions differ ({K} vs {Kb})")
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    elif out.shape != (M, N) or out.stride(1) != 1:
        raise _lib.VnpError("gemm: bad output tensor")
    if bias is not None:
        bias = _rows(bias)
    if nsplit is None:
        nsplit = 1
        if bias is None:
            bm = 192 if op_a == 0 else 128
            tiles = ((M + bm - 1) // bm) * ((N + 127) // 128)
            nk = (K + 31) // 32
            if tiles < 256 and nk >= 8:
                nsplit = max(1, min(nk // 4, 512 // tiles))       # 512 = the persistent grid (2 workgroups x 256 CUs): one item each
    ws = None
    if nsplit > 1:
        ws = torch.empty((nsplit, M, N), device=A.device, dtype=torch.float32)
    h2 = _fp16x2_shape(op_a, M, N, K, A, B, nsplit)
    if h2:
        amax_a = amax_a if amax_a is not None else amax_word(A)
        amax_b = amax_b if amax_b is not None else amax_word(B)
        h2 = amax_a is not None and amax_b is not None
    timer = _prof_arm()
    if h2:
        _trace("gemm_fp16x2")
        call("vnp_gemm_ex", op_a, op_b, M, N, K, ptr(A), _ld(A), ptr(B), _ld(B), ptr(out), _ld(out),
             ptr(bias), _ld(bias) if bias is not None else 0, rows_per_group, 1 if accumulate else 0, nsplit, ptr(ws), ptr(amax_a), ptr(amax_b))
    else:
        call("vnp_gemm", op_a, op_b, M, N, K, ptr(A), _ld(A), ptr(B), _ld(B), ptr(out), _ld(out),
             ptr(bias), _ld(bias) if bias is not None else 0, rows_per_group, 1 if accumulate else 0, nsplit, ptr(ws))
    if timer is not None:
        GEMM_PROFILE.append(("NT" if (op_a, op_b) == (0, 0) else "NN" if (op_a, op_b) == (0, 1) else "TN" if (op_a, op_b) == (1, 1) else "TT",
                             2.0 * M * N * K, nsplit, (M, N, K), timer))
    return out


def reduce_pairs(part, n, C):
    """part: contiguous [n][C][2] partial (sum1, sum2) pairs -> (S [C, 2], planar [2, C]): the same sums in the interleaved layout the
    BatchNorm backward kernels read and as two contiguous rows (the per-channel parameter gradients), one launch (vnp_reduce_pairs)."""
    dev = part.device
    if n > 1024:
        m = next((d for d in range(512, 15, -1) if n % d == 0), 0)
        if m:
            n1 = n // m
            tmp = torch.empty((n1, 2 * C), device=dev, dtype=torch.float32)
            call("vnp_reduce_rows", ptr(part), n1, m * 2 * C, m, 2 * C, 2 * C, ptr(tmp), 2 * C)
            part, n = tmp, n1
    S = torch.empty((C, 2), device=dev, dtype=torch.float32)
    planar = torch.empty((2, C), device=dev, dtype=torch.float32)
    call("vnp_reduce_pairs", ptr(part), n, C, ptr(S), ptr(planar))
    return S, planar


def reduce_rows(part, nb, n, W):
    """part: contiguous [nb][n][W] partials -> [nb][W] (sum over n; two passes when n is large so that no thread
    walks more than a few hundred partials)."""
    dev = part.device
    if n == 1:
        return part.reshape(nb, W)
    if n > 1024:
        # first level over chunks of m partials, m the largest divisor of n in [16, 512] (n = 1248 -> 416); a count without
        # such a divisor (a large prime) is reduced in one level, which is only slower
        m = next((d for d in range(512, 15, -1) if n % d == 0), 0)
        if m:
            n1 = n // m
            tmp = torch.empty((nb, n1, W), device=dev, dtype=torch.float32)
            call("vnp_reduce_rows", ptr(part), nb * n1, m * W, m, W, W, ptr(tmp), W)
            part, n = tmp, n1
    out = torch.empty((nb, W), device=dev, dtype=torch.float32)
    call("vnp_reduce_rows", ptr(part), nb, n * W, n, W, W, ptr(out), W)
    return out


def _per_sample_colsum(g, rows_per_group):
    """g [R, W] contiguous -> [R/rows_per_group*3, W]: sum over the points of each sample, per (t, column)."""
    R, W = g.shape
    assert g.is_contiguous()
    nb = R // rows_per_group
    npts = rows_per_group // 3
    nch = _nchunk(npts)
    part = torch.empty((nb, nch, 3 * W), device=g.device, dtype=torch.float32)
    call("vnp_colsum", ptr(g), rows_per_group * W, 3 * W, nb, npts, 3 * W, None, ptr(part))
    if nch == 1:                                  # one chunk per group: the partials are the result (16384 groups of 16 points in config 4)
        return part.view(nb * 3, W)
    out = torch.empty((nb * 3, W), device=g.device, dtype=torch.float32)
    call("vnp_reduce_rows", ptr(part), nb, nch * 3 * W, nch, 3 * W, 3 * W, ptr(out), 3 * W)
    return out


def gemm_pd_stats(x, Wcat, pd, bias, rows_per_group, C, want_stats):
    """pd[R, 2C] = x Wcat^T (+ per-sample bias).  When the shapes allow (aligned 192x128 tiles, whole 1024-point chunks) and
    ``want_stats``, the GEMM's epilogue also produces the per-chunk Welford partials of |p| (vnp_gemm_vn_stats) and they are
    returned ([P/1024, C, 2]); otherwise None and the caller runs vnp_vn_norm_stats."""
    R, K = x.shape
    ok = (want_stats and STATS_IN_EPILOGUE and _lib.get_gemm_mode() == "split" and R % 3072 == 0 and (2 * C) % 128 == 0 and K % 32 == 0
          and C % 64 == 0 and _ld(x) % 4 == 0 and x.data_ptr() % 16 == 0 and Wcat.is_contiguous() and Wcat.data_ptr() % 16 == 0
          and (bias is None or (rows_per_group > 0 and rows_per_group % 3 == 0)))
    if not ok:
        _trace("pd_gemm_plain")
        gemm(x, Wcat, out=pd, bias=bias, rows_per_group=rows_per_group, nsplit=1)
        return None
    _trace("pd_gemm_stats_epilogue")
    part = torch.empty((R // 3072, C, 2), device=x.device, dtype=torch.float32)
    ws = torch.empty(_lib.lib().vnp_gemm_vn_stats_ws_floats(R, C), device=x.device, dtype=torch.float32)
    am_x = am_w = None
    if _fp16x2_shape(0, R, 2 * C, K, x, Wcat, 1):
        am_x, am_w = amax_word(x), amax_word(Wcat)
    timer = _prof_arm()
    if am_x is not None and am_w is not None:
        _trace("gemm_fp16x2")
        call("vnp_gemm_vn_stats_ex", R, 2 * C, K, ptr(x), _ld(x), ptr(Wcat), K, ptr(pd), _ld(pd), ptr(bias),
             _ld(bias) if bias is not None else 0, rows_per_group, C, ptr(part), ptr(ws), ptr(am_x), ptr(am_w))
    else:
        call("vnp_gemm_vn_stats", R, 2 * C, K, ptr(x), _ld(x), ptr(Wcat), K, ptr(pd), _ld(pd), ptr(bias),
             _ld(bias) if bias is not None else 0, rows_per_group, C, ptr(part), ptr(ws))
    if timer is not None:
        GEMM_PROFILE.append(("NT", 2.0 * R * 2 * C * K, 1, (R, 2 * C, K), timer))
    return part


def _bn_statistics(p, P, C, part, running_mean, running_var):
    """Batch mean / rstd of |p| + eps per channel (and the running-statistics update).  ``part``: per-1024-point Welford partials
    already produced by the GEMM epilogue (vnp_gemm_vn_stats), else a statistics pass over p runs here."""
    dev = p.device
    if part is None:
        chunk, nch = _chunk_pc(P, C)
        part = torch.empty((nch, C, 2), device=dev, dtype=torch.float32)
        call("vnp_vn_norm_stats", ptr(p), _ld(p), P, C, ptr(part), chunk)
    else:
        chunk, nch = _lib.chunk_points(), part.shape[0]
    mean = torch.empty(C, device=dev, dtype=torch.float32)
    rstd = torch.empty(C, device=dev, dtype=torch.float32)
    call("vnp_vn_bn_finalize_chunked", ptr(part), nch, chunk, C, P, ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var),
         BN_MOMENTUM, BN_EPS)
    return mean, rstd


def _bn_leaky_forward(p, d, C, gamma, beta, running_mean, running_var, training, ns, use_bn=True, part=None):
    """p, d: [R, C] views (d may be None).  Returns out, mean, rstd.  ``part``: precomputed statistics partials."""
    R = p.shape[0]
    P = R // 3
    dev = p.device
    mean = rstd = None
    if use_bn:
        if training:
            mean, rstd = _bn_statistics(p, P, C, part, running_mean, running_var)
        else:
            mean = running_mean
            rstd = torch.empty(C, device=dev, dtype=torch.float32)
            call("vnp_rstd_from_var", ptr(running_var), ptr(rstd), C, BN_EPS)
    out = torch.empty((R, C), device=dev, dtype=torch.float32)
    call("vnp_vn_bn_leaky_fwd", ptr(p), _ld(p), ptr(d), _ld(d) if d is not None else 0, ptr(out), C, P, C,
         ptr(mean), ptr(rstd), ptr(gamma) if use_bn else None, ptr(beta) if use_bn else None, ns)
    return out, mean, rstd


def _bn_leaky_backward(go, p, d, C, gamma, beta, mean, rstd, training, ns, use_bn=True, go_row=None, go_col=None,
                       want_gw=False, colsum_rows_per_group=0):
    """Returns (gpd [R, 2C or C] with dL/dp in [:, :C] and dL/dd in [:, C:], ggamma, gbeta[, gw][, gb]).
    Rank-1 upstream gradient: go=None, go[r, c] = go_row[r] * go_col[c]; want_gw adds dL/d(go_col) = sum_r go_row[r]*out[r, :].
    ``colsum_rows_per_group`` > 0 appends gb [R/rows_per_group*3, 2C] = per-sample column sums of gpd per component (the gradient of a
    per-sample bias), taken inside the final-gradient pass when the sample is a whole number of chunks (else from a pass over gpd)."""
    R = p.shape[0]
    P = R // 3
    dev = p.device
    has_d = d is not None
    gpd = torch.empty((R, 2 * C if has_d else C), device=dev, dtype=torch.float32)
    ldg = gpd.shape[1]
    gq = gpd
    gd = gpd[:, C:] if has_d else None
    chunk, nch = _chunk_pc(P, C)
    part = torch.empty((nch, C, 2), device=dev, dtype=torch.float32) if use_bn else None
    nch_pad = (nch + 255) // 256 * 256 if nch > 1024 else nch
    gw_part = None
    if want_gw:
        gw_part = (torch.zeros if nch_pad != nch else torch.empty)((nch_pad, C), device=dev, dtype=torch.float32)
    if go is not None:
        go = _rows(go)
    ggamma = gbeta = None
    rpg = colsum_rows_per_group
    fuse_colsum = bool(rpg) and use_bn and has_d and BN_BWD_TWO_PASS and rpg % 3 == 0 and (rpg // 3) % chunk == 0 and R % rpg == 0
    cs_part = torch.empty((nch, 3, 2 * C), device=dev, dtype=torch.float32) if fuse_colsum else None
    # two passes: the rank-1 (decoder tail) form saves a map (6 instead of 7); the matrix form is a wash on its own (measured) but lets
    # the final-gradient pass also produce the per-sample bias gradient instead of a separate 1.6 GB read (encoder second_conv[0])
    if use_bn and BN_BWD_TWO_PASS and (go is None or fuse_colsum):
        # pass 1: the per-channel sums (read-only);  pass 2: final dL/dp | dL/dd with phase 2 applied on the fly — dL/dq is never
        # written and re-read (vnp_vn_bn_leaky_bwd_pass)
        args = (ptr(go), _ld(go) if go is not None else 0, ptr(go_row), ptr(go_col), ptr(p), _ld(p), ptr(d), _ld(d) if has_d else 0,
                ptr(gq), ldg, ptr(gd), ldg, P, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ns)
        call("vnp_vn_bn_leaky_bwd_pass", 1, *args, ptr(part), ptr(gw_part), chunk, None, P, None)
        S, planar = reduce_pairs(part, nch, C)
        call("vnp_vn_bn_leaky_bwd_pass", 2, *args, None, None, chunk, ptr(S) if training else None, P, ptr(cs_part))
        gbeta, ggamma = planar[0], planar[1]
    else:
        call("vnp_vn_bn_leaky_bwd1", ptr(go), _ld(go) if go is not None else 0, ptr(go_row), ptr(go_col),
             ptr(p), _ld(p), ptr(d), _ld(d) if has_d else 0, ptr(gq), ldg, ptr(gd), ldg, P, C,
             ptr(mean) if use_bn else None, ptr(rstd), ptr(gamma) if use_bn else None, ptr(beta) if use_bn else None,
             ns, ptr(part), ptr(gw_part), chunk)
        if use_bn:
            S, planar = reduce_pairs(part, nch, C)
            call("vnp_vn_bn_bwd2", ptr(gq), ldg, ptr(p), _ld(p), P, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
                 ptr(S) if training else None, P)
            gbeta, ggamma = planar[0], planar[1]
    res = [gpd, ggamma, gbeta]
    if want_gw:
        res.append(reduce_rows(gw_part, 1, nch_pad, C))
    if rpg:
        if fuse_colsum:
            nb, cps = R // rpg, (rpg // 3) // chunk
            res.append(reduce_rows(cs_part.view(nb, cps, 6 * C), nb, cps, 6 * C).view(nb * 3, 2 * C))
        else:
            res.append(_per_sample_colsum(gpd, rpg))
    return tuple(res)


WGRAD_SLICE_TILES = 64          # K-tiles (of 32 rows) per split-K slice of a weight gradient that is multiplied on (see _wgrad)
WGRAD_SLAB_BYTES = 128 << 20    # ... as long as the slabs stay under this


def _wgrad(gy, x, fine=False):
    """dW [Cout, Cin] = gy[R, Cout]^T x[R, Cin]  (TN GEMM, split over the R rows; the slices are summed in double,
    vnp_reduce_slabs_kernel).  ``fine``: the layer was folded onto its producer (WeightProductFn), so dW is multiplied on into a
    product that cancels (measured at config 2: the factor's relative error is amplified ~300x); a slice then accumulates at most
    WGRAD_SLICE_TILES * 32 rows in fp32 on the matrix pipe, which keeps the gradient's rounding error at
    ~sqrt(2048) 2^-24 / sqrt(slices) of its terms however many rows it sums."""
    nsplit = None
    if fine:
        M, N, nk = gy.shape[1], x.shape[1], (gy.shape[0] + 31) // 32
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        auto = max(1, min(nk // 4, 512 // tiles)) if (tiles < 256 and nk >= 8) else 1
        nsplit = max(auto, min((nk + WGRAD_SLICE_TILES - 1) // WGRAD_SLICE_TILES, max(1, WGRAD_SLAB_BYTES // (M * N * 4))))
    return gemm(gy, x, op_a=1, op_b=1, nsplit=nsplit)


DGRAD_NT_ROWS = 49152           # activations of at least this many rows: see _dgrad


def _dgrad(gy, W):
    """dX [R, Cin] = gy [R, Cout] W [Cout, Cin].  For the big activations W is transposed first (a few hundred KB) so that both operands
    are K-contiguous: the product then runs on the aligned NT kernels (the 192 x 256 tile of vnp_gemm_split_wide_kernel where Cin % 256
    == 0) instead of the kernel that transposes W's tile in registers at every K-step — the same LDS image, the same products in the
    same order, the same bits (tests/test_gpu_gemm.py), 3-10 % less time."""
    if gy.shape[0] >= DGRAD_NT_ROWS and W.shape[1] % 128 == 0 and W.shape[0] % 32 == 0:
        return gemm(gy, W.t().contiguous())
    return gemm(gy, W, op_b=1)


# ------------------------------------------------------------------------------- autograd Functions
class SplitColsFn(Function):
    """W[:, 0:n0], W[:, n0:n0+n1], ... as views, with ONE concatenation as the backward: autograd's own slicing gives every slice a
    zero-filled full-size gradient, a copy into it and an add per extra slice (a weight whose input channels are split between a
    per-sample part and a per-point part — models/pcn.py:172, 375-385 — cost 5-8 small launches per step that way)."""

    @staticmethod
    def forward(ctx, W, *sizes):
        if W.dim() != 2 or sum(sizes) != W.shape[1]:
            raise _lib.VnpError(f"split_cols: sizes {sizes} do not cover {tuple(W.shape)}")
        ctx.sizes, ctx.rows = sizes, W.shape[0]
        parts = torch.split(W, list(sizes), dim=1)
        # a slice the aligned GEMM kernels cannot take as it stands (row stride or start not a multiple of 16 bytes: the 385-column
        # weights of Attention_VN_FoldingNet's folding stages) is copied once — a few hundred KB — instead of sending every product with
        # it (49152 x 512 x 384 per stage, forward and backward) to the bounds-checked kernel: 6.2 ms of config 4's step
        if W.stride(0) % 4 != 0 or W.stride(1) != 1:
            parts = tuple(p.contiguous() for p in parts)
        else:
            parts = tuple(p if p.data_ptr() % 16 == 0 else p.contiguous() for p in parts)
        return parts

    @staticmethod
    def backward(ctx, *gs):
        ref = next(g for g in gs if g is not None)
        parts = [g if g is not None else ref.new_zeros((ctx.rows, n)) for g, n in zip(gs, ctx.sizes)]
        return (torch.cat(parts, dim=1),) + (None,) * len(ctx.sizes)


def split_cols(W, *sizes):
    return SplitColsFn.apply(W, *sizes)


class WeightProductFn(Function):
    """(Wf W1, Wd W1) for weights Wf, Wd [C, K1] and W1 [K1, K0]: the layer that follows a VNLinear, folded onto that VNLinear's input
    channels (ENC_LINEAR_REASSOCIATE).  One GEMM over [Wf; Wd] forward, two backward; all of them a few hundred MFLOP."""

    @staticmethod
    def forward(ctx, Wf, Wd, W1):
        Wcat = torch.cat([Wf, Wd], dim=0)
        W1 = _rows(W1)
        prod = gemm(Wcat, W1, op_b=1)                       # [2C, K0]
        ctx.save_for_backward(Wcat, W1)
        ctx.C = Wf.shape[0]
        return prod[:ctx.C], prod[ctx.C:]

    @staticmethod
    @once_differentiable
    def backward(ctx, gf, gd):
        Wcat, W1 = ctx.saved_tensors
        C = ctx.C
        if gf is None or gd is None:
            ref = gf if gf is not None else gd
            gf = gf if gf is not None else torch.zeros_like(ref)
            gd = gd if gd is not None else torch.zeros_like(ref)
        base, K0 = gf._base, W1.shape[1]
        if (base is not None and base is gd._base and base.shape == (2 * C, K0) and base.is_contiguous() and gf.shape == (C, K0)
                and gd.shape == (C, K0) and gf.data_ptr() == base.data_ptr() and gd.data_ptr() == base.data_ptr() + 4 * C * K0):
            g = base                        # the two halves of one [2C, K0] weight gradient (VNLinearLeakyFn.backward): no copy
        else:
            g = torch.cat([gf, gd], dim=0)
        g = _rows(g)
        gWcat = gemm(g, W1) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None      # [2C, K1] = g W1^T
        gW1 = gemm(Wcat, g, op_a=1, op_b=1) if ctx.needs_input_grad[2] else None                  # [K1, K0] = Wcat^T g
        return (gWcat[:C] if gWcat is not None else None), (gWcat[C:] if gWcat is not None else None), gW1


class ToRowsFn(Function):
    """reference layout [B, C, 3, N] -> rows [(B*N)*3, C]."""

    @staticmethod
    def forward(ctx, x):
        _lib.require_gpu(x)
        x = x.contiguous()
        B, C, _, N = x.shape
        ctx.dims = (B, C, N)
        rows = torch.empty((B * N * 3, C), device=x.device, dtype=torch.float32)
        call("vnp_ref_to_rows", ptr(x), B, C, N, ptr(rows), C)
        return rows

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        B, C, N = ctx.dims
        g = _rows(g)
        out = torch.empty((B, C, 3, N), device=g.device, dtype=torch.float32)
        call("vnp_rows_to_ref", ptr(g), _ld(g), B, C, N, ptr(out))
        return out


class FromRowsFn(Function):
    """rows [(B*N)*3, C] -> reference layout [B, C, 3, N]."""

    @staticmethod
    def forward(ctx, rows, B, N):
        rows = _rows(rows)
        C = rows.shape[1]
        ctx.dims = (B, C, N)
        out = torch.empty((B, C, 3, N), device=rows.device, dtype=torch.float32)
        call("vnp_rows_to_ref", ptr(rows), _ld(rows), B, C, N, ptr(out))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        B, C, N = ctx.dims
        g = g.contiguous()
        rows = torch.empty((B * N * 3, C), device=g.device, dtype=torch.float32)
        call("vnp_ref_to_rows", ptr(g), B, C, N, ptr(rows), C)
        return rows, None, None


class VNLinearFn(Function):
    """y = x W^T (+ per-sample bias)  —  VNLinear, models/vn_layers.py:12-22."""

    @staticmethod
    def forward(ctx, x, W, bias=None, rows_per_group=0):
        x = _rows(x)
        y = gemm(x, W, bias=bias, rows_per_group=rows_per_group)
        ctx.save_for_backward(x, W)
        ctx.rpg = rows_per_group
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        gy = _rows(gy)
        gx = _dgrad(gy, W) if ctx.needs_input_grad[0] else None
        gW = _wgrad(gy, x) if ctx.needs_input_grad[1] else None
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _per_sample_colsum(gy.contiguous(), ctx.rpg)
        return gx, gW, gb, None


class VNLinearLeakyFn(Function):
    """VNLinearLeakyReLU (models/vn_layers.py:46-74): p = W_f x -> BatchNorm over |p| -> leaky projection on d = W_d x.

    ``bias`` [B*3, 2C] (feat | dir halves) carries the de-duplicated broadcast input channels."""

    @staticmethod
    def forward(ctx, x, Wf, Wd, gamma, beta, running_mean, running_var, bias, rows_per_group, training, ns, folded=False):
        """``folded``: Wf, Wd are products with the producing VNLinear's weight (WeightProductFn): finer split-K for their gradient."""
        x = _rows(x)
        R = x.shape[0]
        C = Wf.shape[0]
        ctx.folded = bool(folded)
        pd = torch.empty((R, 2 * C), device=x.device, dtype=torch.float32)
        if bias is not None:
            bias = _rows(bias)
        # one GEMM for both halves: [Wf; Wd] is copied into a [2C, Cin] matrix (a few MB at most) so that x is streamed
        # once and the launch has twice the columns
        Wcat = torch.cat([Wf, Wd], dim=0)
        part = gemm_pd_stats(x, Wcat, pd, bias, rows_per_group, C, training)
        out, mean, rstd = _bn_leaky_forward(pd[:, :C], pd[:, C:], C, gamma, beta, running_mean, running_var, training, ns, part=part)
        ctx.save_for_backward(x, Wcat, gamma, beta, pd, mean, rstd)
        ctx.cfg = (C, rows_per_group, training, ns, bias is not None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        x, Wcat, gamma, beta, pd, mean, rstd = ctx.saved_tensors
        C, rpg, training, ns, has_bias = ctx.cfg
        want_gb = has_bias and ctx.needs_input_grad[7]
        out = _bn_leaky_backward(go, pd[:, :C], pd[:, C:], C, gamma, beta, mean, rstd, training, ns,
                                 colsum_rows_per_group=rpg if want_gb else 0)
        gpd, ggamma, gbeta = out[:3]
        gb = out[3] if want_gb else None
        gx = _dgrad(gpd, Wcat) if ctx.needs_input_grad[0] else None          # dX = [dP | dD] [Wf; Wd]
        if gx is not None:
            gx._vnp_private = True               # a fresh buffer nobody else holds (see _grad_is_private)
        gWf = gWd = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gW = _wgrad(gpd, x, fine=ctx.folded)                                    # [2C, Cin] in one split-K launch
            gWf, gWd = gW[:C], gW[C:]
        return gx, gWf, gWd, ggamma, gbeta, None, None, gb, None, None, None, None


def unit_dirs(xyz, return_norm=False):
    """[.., 3] points -> [P, 3] unit directions x / |x| (0 for the origin) and, with ``return_norm``, the norms |x| [P, 1]; no gradient
    (functions of the input cloud only)."""
    x = xyz.detach().reshape(-1, 3).contiguous().float()
    _lib.require_gpu(x)
    u = torch.empty_like(x)
    r = torch.empty((x.shape[0], 1), device=x.device, dtype=torch.float32) if return_norm else None
    call("vnp_unit_dirs_r", ptr(x), x.shape[0], ptr(u), ptr(r))
    return (u, r) if return_norm else u


class DirExpandFn(Function):
    """Per-point scalars h [P, C] of a map parallel to the unit directions u -> its rows u (x) h [P*3, C]; backward: the contraction
    sum_t u_t g_t (the adjoint; DirContractFn the other way round)."""

    @staticmethod
    def forward(ctx, h, u):
        h = _rows(h)
        P, C = h.shape
        rows = torch.empty((P * 3, C), device=h.device, dtype=torch.float32)
        call("vnp_dir_contract", ptr(rows), C, ptr(u), ptr(h), _ld(h), P, C, 1)
        ctx.save_for_backward(u)
        return rows

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        g = _rows(g)
        P, C = g.shape[0] // 3, g.shape[1]
        gh = torch.empty((P, C), device=g.device, dtype=torch.float32)
        call("vnp_dir_contract", ptr(g), _ld(g), ptr(u), ptr(gh), C, P, C, 0)
        return gh, None


class DirContractFn(Function):
    """rows [P*3, C] whose VN vectors are parallel to the point's unit direction u -> their scalars h [P, C] = sum_t u_t rows_t
    (rows = u (x) h); backward: the adjoint u_t gh.  See VNDirLinearLeakyFn."""

    @staticmethod
    def forward(ctx, rows, u):
        rows = _rows(rows)
        P, C = rows.shape[0] // 3, rows.shape[1]
        h = torch.empty((P, C), device=rows.device, dtype=torch.float32)
        call("vnp_dir_contract", ptr(rows), _ld(rows), ptr(u), ptr(h), C, P, C, 0)
        ctx.save_for_backward(u)
        return h

    @staticmethod
    @once_differentiable
    def backward(ctx, gh):
        (u,) = ctx.saved_tensors
        gh = _rows(gh)
        P, C = gh.shape
        g = torch.empty((P * 3, C), device=gh.device, dtype=torch.float32)
        call("vnp_dir_contract", ptr(g), C, ptr(u), ptr(gh), _ld(gh), P, C, 1)
        return g, None


class DirLinearMaxPoolFn(Function):
    """VNLinear followed by VNMaxPool (its only consumer) on a map u (x) h given as per-point scalars h [P, K] — maxpool1 behind
    VN_PointNet's single-channel first layer (models/pcn.py:168-169, models/vn_layers.py:153-167).  The pooled map is u (x) y with
    y = h W^T and its direction u (x) y' with y' = h (Wd W)^T, so a point's score <x, d> is y y': one GEMM on points over [W ; Wd W] and an
    arg-max over scalars (vnp_dir_pool_fwd) instead of a GEMM with a pooling epilogue over three rows per point.  Returns the pooled
    rows [G*3, C] and the indices.  Backward: one entry of dL/dy per (group, channel), gathered into dL/dh and dL/dW (vnp_dir_pool_bwd);
    Wd receives no gradient (the arg-max detaches it in the reference as well)."""

    @staticmethod
    def forward(ctx, h, u, W, Wd, G, gs):
        h = _rows(h)
        C, K = W.shape
        dev = h.device
        Wcat = torch.cat([_rows(W), gemm(Wd, W, op_b=1)], dim=0)              # [2C, K]
        y = gemm(h, Wcat, nsplit=1)                                            # [P, 2C]
        idx = torch.empty((G, C), device=dev, dtype=torch.int32)
        out = torch.empty((G * 3, C), device=dev, dtype=torch.float32)
        call("vnp_dir_pool_fwd", ptr(y), 2 * C, ptr(u), G, gs, C, ptr(idx), ptr(out), C)
        idx, out = _forced_pool(idx, out, lambda: (u.reshape(-1, 3, 1) * y[:, :C].reshape(-1, 1, C)).reshape(-1, C), G, gs)
        _trace("dir_pool", idx)
        ctx.save_for_backward(h, u, W, idx)
        ctx.cfg = (G, gs, C)
        ctx.mark_non_differentiable(idx)
        return out, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, gout, _gidx):
        h, u, W, idx = ctx.saved_tensors
        G, gs, C = ctx.cfg
        gout = _rows(gout)
        K = h.shape[1]
        dev = h.device
        gW = torch.empty((C, K), device=dev, dtype=torch.float32) if ctx.needs_input_grad[2] else None
        gh = order = pts = None
        if ctx.needs_input_grad[0]:
            gh = torch.empty((G * gs, K), device=dev, dtype=torch.float32)
            pts64, order64 = torch.sort(idx, dim=1, stable=True)        # index bookkeeping only
            pts, order = pts64.int().contiguous(), order64.int().contiguous()
        Wc = _rows(W)
        call("vnp_dir_pool_bwd", ptr(gout), _ld(gout), ptr(u), ptr(idx), ptr(order), ptr(pts), ptr(h), _ld(h), ptr(Wc), _ld(Wc),
             G, gs, C, K, ptr(gW), K, ptr(gh), K)
        return gh, None, gW, None, None, None


class VNDirLinearLeakyFn(Function):
    """VNLinearLeakyReLU (models/vn_layers.py:46-74) on a map whose VN vectors are all parallel to one unit direction u per point — the
    layer behind VN_PointNet's single-channel first layer (models/pcn.py:165-173): input h [P, K] (one scalar per point and channel, the
    map is u (x) h), output in the rows layout [P*3, C].  The GEMMs run on points, a third of the rows layout's FLOPs, and the layer's
    passes rebuild p = u v_p + bias, d = u v_d + bias from v = h [Wf; Wd]^T [P, 2C] in registers (vnp_vn_dir_layer): the [P*3, 2C]
    pre-activations (1.6 GB at config 2) and their gradient are never written."""

    @staticmethod
    def _chunk(P, C):
        """Points per workgroup of vnp_vn_dir_layer: 1024, halved (down to 64) until the grid has 4096 workgroups — the operands are
        one row per POINT, so a grid sized like the rows-layout passes' (>= 1024 workgroups) would half-fill the chip."""
        chunk, cols = 1024, (C + 63) // 64
        while chunk > 64 and ((P + chunk - 1) // chunk) * cols < 4096:
            chunk //= 2
        return chunk, (P + chunk - 1) // chunk

    @staticmethod
    def supported(P, C, points_per_group, has_bias):
        chunk, _ = VNDirLinearLeakyFn._chunk(P, C)
        return (not has_bias) or (points_per_group > 0 and points_per_group % chunk == 0 and P % points_per_group == 0)

    @staticmethod
    def forward(ctx, h, u, Wf, Wd, gamma, beta, running_mean, running_var, bias, points_per_group, training, ns, folded=False,
                scalar_out=False):
        """``scalar_out`` (needs ``bias is None``): the output is parallel to u as well — the single-channel first layer itself — and is
        returned as per-point scalars [P, C]; the upstream gradient then comes as [P, C] too."""
        if scalar_out and bias is not None:
            raise _lib.VnpError("VNDirLinearLeakyFn: scalar_out needs a layer without per-sample bias")
        sio = 1 if scalar_out else 0
        h = _rows(h)
        P, C = h.shape[0], Wf.shape[0]
        dev = h.device
        if bias is not None:
            bias = _rows(bias)
        Wcat = torch.cat([Wf, Wd], dim=0)
        v = gemm(h, Wcat, nsplit=1)                                   # [P, 2C]
        chunk, nch = VNDirLinearLeakyFn._chunk(P, C)
        common = (ptr(v), 2 * C, ptr(u), ptr(bias), _ld(bias) if bias is not None else 0, P, C, points_per_group, chunk)
        if training:
            part = torch.empty((nch, C, 2), device=dev, dtype=torch.float32)
            call("vnp_vn_dir_layer_ex", 0, *common, None, None, None, None, ns, None, 0, None, 0, ptr(part), None, 0, None, 0, None, sio)
            mean = torch.empty(C, device=dev, dtype=torch.float32)
            rstd = torch.empty(C, device=dev, dtype=torch.float32)
            call("vnp_vn_bn_finalize_chunked", ptr(part), nch, chunk, C, P, ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var),
                 BN_MOMENTUM, BN_EPS)
        else:
            mean = running_mean
            rstd = torch.empty(C, device=dev, dtype=torch.float32)
            call("vnp_rstd_from_var", ptr(running_var), ptr(rstd), C, BN_EPS)
        out = torch.empty((P if scalar_out else P * 3, C), device=dev, dtype=torch.float32)
        call("vnp_vn_dir_layer_ex", 1, *common, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ns, ptr(out), C, None, 0, None, None, 0,
             None, 0, None, sio)
        _trace("dir_layer_scalar_out" if scalar_out else "dir_layer")
        ctx.save_for_backward(h, u, Wcat, v, bias, gamma, beta, mean, rstd)
        ctx.cfg = (C, points_per_group, training, ns, bool(folded), chunk, nch, sio)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        h, u, Wcat, v, bias, gamma, beta, mean, rstd = ctx.saved_tensors
        C, ppg, training, ns, folded, chunk, nch, sio = ctx.cfg
        P = h.shape[0]
        dev = h.device
        go = _rows(go)
        common = (ptr(v), 2 * C, ptr(u), ptr(bias), _ld(bias) if bias is not None else 0, P, C, ppg, chunk)
        stats = (ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ns)
        part = torch.empty((nch, C, 2), device=dev, dtype=torch.float32)
        call("vnp_vn_dir_layer_ex", 2, *common, *stats, None, 0, ptr(go), _ld(go), ptr(part), None, 0, None, 0, None, sio)
        S, planar = reduce_pairs(part, nch, C)
        want_gb = bias is not None and ctx.needs_input_grad[8]
        gv = torch.empty((P, 2 * C), device=dev, dtype=torch.float32)
        cs_part = torch.empty((nch, 3, 2 * C), device=dev, dtype=torch.float32) if want_gb else None
        call("vnp_vn_dir_layer_ex", 3, *common, *stats, None, 0, ptr(go), _ld(go), None, ptr(S) if training else None, P,
             ptr(gv), 2 * C, ptr(cs_part), sio)
        gb = None
        if want_gb:
            nb, cps = P // ppg, ppg // chunk
            gb = reduce_rows(cs_part.view(nb, cps, 6 * C), nb, cps, 6 * C).view(nb * 3, 2 * C)
        gh = _dgrad(gv, Wcat) if ctx.needs_input_grad[0] else None
        gWf = gWd = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            gW = _wgrad(gv, h, fine=folded)
            gWf, gWd = gW[:C], gW[C:]
        return gh, None, gWf, gWd, planar[1], planar[0], None, None, gb, None, None, None, None, None


class VNLinearLeakyOutFn(Function):
    """Last two layers of VN_FoldingNet fused (models/pcn.py:341-346,387): VNLinearLeakyReLU(C_in -> C) followed by
    VNLinear(C -> 1) and the residual point_feat.  The [P*3, C] activation between them is never written (forward) nor
    its gradient (backward: rank-1 upstream gradient g_fine[r] * w_out[c] generated inside the BatchNorm/leaky backward).
    Returns fine [P*3]."""

    @staticmethod
    def forward(ctx, x, Wf, Wd, gamma, beta, running_mean, running_var, w_out, coarse, S, training, ns):
        x = _rows(x)
        R = x.shape[0]
        P = R // 3
        C = Wf.shape[0]
        dev = x.device
        Wcat = torch.cat([Wf, Wd], dim=0)
        pd = torch.empty((R, 2 * C), device=dev, dtype=torch.float32)
        part = gemm_pd_stats(x, Wcat, pd, None, 0, C, training)                # [R, 2C] (+ the statistics of p from the epilogue)
        p, d = pd[:, :C], pd[:, C:]
        if training:
            mean, rstd = _bn_statistics(p, P, C, part, running_mean, running_var)
        else:
            mean = running_mean
            rstd = torch.empty(C, device=dev, dtype=torch.float32)
            call("vnp_rstd_from_var", ptr(running_var), ptr(rstd), C, BN_EPS)
        w = w_out.reshape(-1).contiguous()
        coarse = coarse.contiguous() if coarse is not None else None          # None: no residual (the attention decoder's folding stages)
        fine = torch.empty(R, device=dev, dtype=torch.float32)
        _trace("decoder_tail_fused")
        call("vnp_vn_bn_leaky_rowdot_fwd", ptr(p), _ld(p), ptr(d), _ld(d), P, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ns,
             ptr(w), ptr(coarse), S if coarse is not None else 0, ptr(fine))
        ctx.save_for_backward(x, Wcat, gamma, beta, pd, mean, rstd, w)
        ctx.cfg = (C, S, tuple(coarse.shape) if coarse is not None else None, training, ns)
        return fine

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, Wcat, gamma, beta, pd, mean, rstd, w = ctx.saved_tensors
        C, S, cshape, training, ns = ctx.cfg
        g = g.contiguous()
        gpd, ggamma, gbeta, gw = _bn_leaky_backward(None, pd[:, :C], pd[:, C:], C, gamma, beta, mean, rstd, training, ns,
                                                    go_row=g, go_col=w, want_gw=True)
        gx = _dgrad(gpd, Wcat) if ctx.needs_input_grad[0] else None
        gW = _wgrad(gpd, x)
        gcoarse = None
        if cshape is not None and ctx.needs_input_grad[8]:
            gcoarse = torch.empty(cshape, device=g.device, dtype=torch.float32)
            call("vnp_group_sum3", ptr(g), g.numel() // 3 // S, S, ptr(gcoarse), 0)
        return gx, gW[:C], gW[C:], ggamma, gbeta, None, None, gw.reshape(1, C), gcoarse, None, None, None


class VNBnLeakyFn(Function):
    """BatchNorm-over-norms (+ optional leaky projection) on an existing [R, 2C] (p | d) or [R, C] (p) tensor.
    VNBatchNorm (models/vn_layers.py:107-127) and the tail of VNLinearLeakyReLU (:66-73)."""

    @staticmethod
    def forward(ctx, pd, has_d, gamma, beta, running_mean, running_var, training, ns):
        pd = _rows(pd)
        C = pd.shape[1] // 2 if has_d else pd.shape[1]
        p = pd[:, :C]
        d = pd[:, C:] if has_d else None
        out, mean, rstd = _bn_leaky_forward(p, d, C, gamma, beta, running_mean, running_var, training, ns)
        ctx.save_for_backward(pd, gamma, beta, mean, rstd)
        ctx.cfg = (C, has_d, training, ns)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        pd, gamma, beta, mean, rstd = ctx.saved_tensors
        C, has_d, training, ns = ctx.cfg
        gpd, ggamma, gbeta = _bn_leaky_backward(go, pd[:, :C], pd[:, C:] if has_d else None, C, gamma, beta, mean, rstd,
                                                training, ns)
        return gpd, None, ggamma, gbeta, None, None, None, None


class VNLeakyFn(Function):
    """Leaky projection without BatchNorm: out = leaky(x, d) (models/vn_layers.py:37-42); returns grads for x and d."""

    @staticmethod
    def forward(ctx, x, d, ns):
        x, d = _rows(x), _rows(d)
        C = x.shape[1]
        out, _, _ = _bn_leaky_forward(x, d, C, None, None, None, None, False, ns, use_bn=False)
        ctx.save_for_backward(x, d)
        ctx.ns = ns
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        x, d = ctx.saved_tensors
        C = x.shape[1]
        gpd, _, _ = _bn_leaky_backward(go, x, d, C, None, None, None, None, False, ctx.ns, use_bn=False)
        return gpd[:, :C], gpd[:, C:], None


class VNMaxPoolFn(Function):
    """VNMaxPool (models/vn_layers.py:153-167) over G groups of gs consecutive points.  W_d receives no gradient
    (the arg-max detaches it in the reference as well)."""

    @staticmethod
    def forward(ctx, x, Wd, G, gs, x_prev=None, W_prev=None):
        """``x_prev``, ``W_prev``: when x = x_prev W_prev^T was produced by a VNLinear, the direction d = Wd x can be evaluated
        as (Wd W_prev) x_prev (contraction over the producer's input channels; see POOL_DIRECTION_REASSOCIATE)."""
        x = _rows(x)
        C = x.shape[1]
        if x_prev is not None and POOL_DIRECTION_REASSOCIATE:
            d = gemm(_rows(x_prev), gemm(Wd, W_prev, op_b=1), nsplit=1)
        else:
            d = gemm(x, Wd, nsplit=1)
        nch = _nchunk(gs)
        ws_val = torch.empty((G, nch, C), device=x.device, dtype=torch.float32)
        ws_idx = torch.empty((G, nch, C), device=x.device, dtype=torch.int32)
        idx = torch.empty((G, C), device=x.device, dtype=torch.int32)
        out = torch.empty((G * 3, C), device=x.device, dtype=torch.float32)
        call("vnp_vn_pool_fwd", ptr(x), _ld(x), ptr(d), _ld(d), G, gs, C, ptr(idx), ptr(out), C, ptr(ws_val), ptr(ws_idx))
        _trace("pool_unfused", idx)
        ctx.save_for_backward(idx)
        ctx.cfg = (G, gs, C)
        ctx.mark_non_differentiable(idx)
        return out, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, gout, _gidx):
        (idx,) = ctx.saved_tensors
        G, gs, C = ctx.cfg
        gout = _rows(gout)
        gx = torch.empty((G * gs * 3, C), device=gout.device, dtype=torch.float32)
        call("vnp_vn_pool_bwd", ptr(gout), _ld(gout), ptr(idx), G, gs, C, ptr(gx), C)
        return gx, None, None, None, None, None


# TEST HOOK (tests/test_gpu_layers.py): a list of arg-max index tensors [G, C]; while it is set, every VNMaxPool-type forward below takes
# its next entry instead of its own pick (the pooled vectors are re-gathered with plain tensor ops) — "teacher forcing", so that two
# evaluation orders of the network can be compared output for output although a handful of rounding-level ties would otherwise be broken
# differently and move everything downstream.  Never set on the product path.
POOL_IDX_FORCE = None


def _forced_pool(idx, out, rows_of_y, G, gs):
    f = POOL_IDX_FORCE
    if f is None:
        return idx, out
    forced = f.pop(0).to(device=idx.device, dtype=torch.int32).contiguous()
    if forced.shape != idx.shape:
        raise _lib.VnpError(f"POOL_IDX_FORCE: index tensor {tuple(forced.shape)} offered to a pool of shape {tuple(idx.shape)}")
    if torch.equal(forced, idx):
        return idx, out
    y = rows_of_y()                                                               # [G*gs*3, C]
    C = idx.shape[1]
    pt = torch.arange(G, device=idx.device).view(G, 1) * gs + forced.long()        # [G, C] global point
    rows = (pt * 3).unsqueeze(1) + torch.arange(3, device=idx.device).view(1, 3, 1)
    cols = torch.arange(C, device=idx.device).view(1, 1, C).expand(G, 3, C)
    return forced, y[rows, cols].reshape(G * 3, C).contiguous()


def _grad_is_private(g):
    """True when an incoming gradient tensor may be accumulated into IN PLACE: it carries the mark VNLinearLeakyFn.backward puts on the
    input gradient it has just allocated.  Autograd does not promise that a node is the only user of the tensor it is handed (a
    consumer whose backward returns one tensor for several inputs, hooks, retain_grad); a gradient that autograd itself summed from
    several consumers, or that any other backward produced, arrives without the mark and is added out of place."""
    return getattr(g, "_vnp_private", False) is True


class VNMaxPoolTapFn(Function):
    """VNMaxPool of a tensor that has a SECOND consumer (encoder: first_conv[1]'s output feeds maxpool1 and, concatenated with the
    pooled feature, second_conv[0]; models/pcn.py:169-172).  Returns (x, pooled, idx): the caller hands the returned x to the other
    consumer, so that in backward this node receives that consumer's gradient and ADDS the pool's one-row-per-channel gradient
    into it in place (vnp_vn_pool_bwd_acc) — instead of materialising a dense, mostly-zero [P*3, C] gradient and letting autograd
    sum two 400 MB tensors (zero-fill + scatter + add: 0.3 ms per step of config 2)."""

    @staticmethod
    def forward(ctx, x, Wd, G, gs, x_prev=None, W_prev=None):
        xr = _rows(x)
        C = xr.shape[1]
        if x_prev is not None and POOL_DIRECTION_REASSOCIATE:
            d = gemm(_rows(x_prev), gemm(Wd, W_prev, op_b=1), nsplit=1)
        else:
            d = gemm(xr, Wd, nsplit=1)
        nch = _nchunk(gs)
        ws_val = torch.empty((G, nch, C), device=xr.device, dtype=torch.float32)
        ws_idx = torch.empty((G, nch, C), device=xr.device, dtype=torch.int32)
        idx = torch.empty((G, C), device=xr.device, dtype=torch.int32)
        out = torch.empty((G * 3, C), device=xr.device, dtype=torch.float32)
        call("vnp_vn_pool_fwd", ptr(xr), _ld(xr), ptr(d), _ld(d), G, gs, C, ptr(idx), ptr(out), C, ptr(ws_val), ptr(ws_idx))
        idx, out = _forced_pool(idx, out, lambda: xr, G, gs)
        _trace("pool_tap", idx)
        ctx.save_for_backward(idx)
        ctx.cfg = (G, gs, C)
        ctx.mark_non_differentiable(idx)
        return x, out, idx                      # x as-is: autograd hands back an alias whose gradient flows through this node

    @staticmethod
    @once_differentiable
    def backward(ctx, gx, gout, _gidx):
        (idx,) = ctx.saved_tensors
        G, gs, C = ctx.cfg
        gout = _rows(gout)
        if gx is None or gx.dim() != 2 or gx.stride(1) != 1 or gx.shape != (G * gs * 3, C) or not _grad_is_private(gx):
            dense = torch.empty((G * gs * 3, C), device=gout.device, dtype=torch.float32)
            call("vnp_vn_pool_bwd", ptr(gout), _ld(gout), ptr(idx), G, gs, C, ptr(dense), C)
            return (dense if gx is None else gx + dense), None, None, None, None, None
        # gx is the other consumer's freshly allocated gradient and this node its only user (_grad_is_private): accumulate in place
        call("vnp_vn_pool_bwd_acc", ptr(gout), _ld(gout), ptr(idx), G, gs, C, ptr(gx), _ld(gx))
        return gx, None, None, None, None, None


class VNLinearPoolTapFn(Function):
    """VNLinear followed by a VNMaxPool of its output AND by another consumer of that output (encoder first_conv[1] -> maxpool1 /
    second_conv[0], models/pcn.py:169-172): one GEMM over the interleaved [W ; Wd W] rows stores y = x W^T and pools in its epilogue
    (vnp_linear_pool_store_fwd) — no direction GEMM, no pass over y and d.  Returns (y, pooled, idx).  Backward: the pool's
    one-row-per-channel gradient is added into y's gradient from the other consumer (as VNMaxPoolTapFn), then the VNLinear backward."""

    @staticmethod
    def supported(x, W, G, gs):
        C, K = W.shape
        return (POOL_DIRECTION_REASSOCIATE and POOL_FUSED_EPILOGUE and _lib.get_gemm_mode() == "split" and (G * gs * 3) % 192 == 0
                and gs % 32 == 0 and C % 64 == 0 and K % 32 == 0 and G <= 65535 and x.dim() == 2 and x.stride(1) == 1
                and _ld(x) % 4 == 0 and x.data_ptr() % 16 == 0)

    @staticmethod
    def forward(ctx, x, W, Wd, G, gs):
        x = _rows(x)
        C, K = W.shape
        dev = x.device
        Wp = gemm(Wd, W, op_b=1)
        Wcat = torch.stack([_rows(W).view(C // 32, 32, K), Wp.view(C // 32, 32, K)], dim=1).reshape(2 * C, K)
        y = torch.empty((G * gs * 3, C), device=dev, dtype=torch.float32)
        idx = torch.empty((G, C), device=dev, dtype=torch.int32)
        out = torch.empty((G * 3, C), device=dev, dtype=torch.float32)
        ws = torch.empty((_lib.lib().vnp_linear_pool_ws_bytes(G, gs, C) + 7) // 8, device=dev, dtype=torch.int64)
        timer = _prof_arm()
        call("vnp_linear_pool_store_fwd", ptr(x), _ld(x), ptr(Wcat), K, G, gs, C, K, ptr(y), C, ptr(idx), ptr(out), C, ptr(ws))
        if timer is not None:
            GEMM_PROFILE.append(("NT", 2.0 * (G * gs * 3) * (2 * C) * K, 1, (G * gs * 3, 2 * C, K), timer))
        idx, out = _forced_pool(idx, out, lambda: y, G, gs)
        _trace("linear_pool_tap_fused", idx)
        ctx.save_for_backward(x, W, idx)
        ctx.cfg = (G, gs, C)
        ctx.mark_non_differentiable(idx)
        return y, out, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gout, _gidx):
        x, W, idx = ctx.saved_tensors
        G, gs, C = ctx.cfg
        if gout is not None:
            gout = _rows(gout)
            if gy is None or gy.dim() != 2 or gy.stride(1) != 1 or gy.shape != (G * gs * 3, C) or not _grad_is_private(gy):
                dense = torch.empty((G * gs * 3, C), device=x.device, dtype=torch.float32)
                call("vnp_vn_pool_bwd", ptr(gout), _ld(gout), ptr(idx), G, gs, C, ptr(dense), C)
                gy = dense if gy is None else gy + dense
            else:
                call("vnp_vn_pool_bwd_acc", ptr(gout), _ld(gout), ptr(idx), G, gs, C, ptr(gy), _ld(gy))
        gy = _rows(gy)
        gx = _dgrad(gy, W) if ctx.needs_input_grad[0] else None
        gW = _wgrad(gy, x) if ctx.needs_input_grad[1] else None
        return gx, gW, None, None, None


class VNLinearMaxPoolFn(Function):
    """VNLinear followed by VNMaxPool when the pool is the linear output's ONLY consumer (encoder second_conv[1] ->
    maxpool2, models/pcn.py:173-174).  Forward: y = x W^T, d = y Wd^T, arg-max pool (y and d are dropped afterwards).
    Backward: the gradient of y has one non-zero point per (sample, channel); dW and dx are gathered directly
    (vnp_pool_lin_bwd) instead of two dense GEMMs over a 1.6 GB mostly-zero matrix."""

    @staticmethod
    def forward(ctx, x, W, Wd, G, gs):
        x = _rows(x)
        C, K = W.shape
        idx = torch.empty((G, C), device=x.device, dtype=torch.int32)
        out = torch.empty((G * 3, C), device=x.device, dtype=torch.float32)
        fused = (POOL_DIRECTION_REASSOCIATE and POOL_FUSED_EPILOGUE and _lib.get_gemm_mode() == "split" and (G * gs * 3) % 192 == 0
                 and gs % 32 == 0 and C % 64 == 0 and K % 32 == 0 and G <= 65535 and _ld(x) % 4 == 0 and x.data_ptr() % 16 == 0)
        screened = False
        if fused:
            # one GEMM over the interleaved [W ; Wd W] rows whose epilogue does the arg-max pool: y and d are never written
            Wp = gemm(Wd, W, op_b=1)
            Wcat = torch.stack([_rows(W).view(C // 32, 32, K), Wp.view(C // 32, 32, K)], dim=1).reshape(2 * C, K)
            skey = (x.device.index, G, gs, C, K)
            # the selection pass keeps a group's 32-point blocks in registers (gs <= 2048); back off after a fallback (see above)
            # (a short contraction — first_conv[1] -> maxpool1, K = 128 — is cheaper on the full epilogue than the search's fixed passes)
            screened = (POOL_SCREENED and gs <= 2048 and (K >= POOL_SCREEN_MIN_K or POOL_SCREEN_STATS is not None)
                        and (POOL_SCREEN_STATS is not None or _screen_allowed(skey)))
            entry = "vnp_linear_pool_screened_fwd" if screened else "vnp_linear_pool_fwd"
            nb = _lib.lib().vnp_linear_pool_screened_ws_bytes(G, gs, C, K) if screened else _lib.lib().vnp_linear_pool_ws_bytes(G, gs, C)
            ws = torch.empty((nb + 7) // 8, device=x.device, dtype=torch.int64)
            timer = _prof_arm()
            POOL_PATH_COUNTS["screened" if screened else "full_epilogue"] += 1
            if screened:
                if POOL_SCREEN_PASS not in ("fp16", "bf16x3"):
                    raise _lib.VnpError(f"VNP_POOL_SCREEN_PASS must be 'fp16' or 'bf16x3', got {POOL_SCREEN_PASS!r}")
                call(entry, ptr(x), _ld(x), ptr(Wcat), K, G, gs, C, K, ptr(idx), ptr(out), C, ptr(ws), 1 if POOL_SCREEN_PASS == "fp16" else 0)
            else:
                call(entry, ptr(x), _ld(x), ptr(Wcat), K, G, gs, C, K, ptr(idx), ptr(out), C, ptr(ws))
            if timer is not None:
                # flops = 2 M (2C) K: what the full epilogue executes six-fold; the screening pass (the timed kernel when `screened`)
                # executes ONE fp16 product of it (three bf16 products with VNP_POOL_SCREEN_PASS=bf16x3)
                GEMM_PROFILE.append(("NT-pool-screened" if screened else "NT", 2.0 * (G * gs * 3) * (2 * C) * K, 1, (G * gs * 3, 2 * C, K), timer))
            if screened:
                if POOL_SCREEN_STATS is not None:
                    POOL_SCREEN_STATS.append(ws[:2].view(torch.int32)[:3].clone())
                else:
                    _screen_report(skey, ws)
        else:
            POOL_PATH_COUNTS["unfused"] += 1
            y = gemm(x, W, nsplit=1)
            if POOL_DIRECTION_REASSOCIATE:
                # d = Wd (W x) = (Wd W) x: the product of the two weight matrices is formed once per step (8.6 GFLOP for
                # 2048x2048x1024) and the per-point direction GEMM contracts over Cin instead of Cout (half the FLOPs here).
                # Same arithmetic up to fp32 re-association.
                d = gemm(x, gemm(Wd, W, op_b=1), nsplit=1)
            else:
                d = gemm(y, Wd, nsplit=1)
            nch = _nchunk(gs)
            ws_val = torch.empty((G, nch, C), device=x.device, dtype=torch.float32)
            ws_idx = torch.empty((G, nch, C), device=x.device, dtype=torch.int32)
            call("vnp_vn_pool_fwd", ptr(y), _ld(y), ptr(d), _ld(d), G, gs, C, ptr(idx), ptr(out), C, ptr(ws_val), ptr(ws_idx))
        idx, out = _forced_pool(idx, out, lambda: gemm(x, W, nsplit=1), G, gs)
        _trace(("linear_pool_screened" if screened else "linear_pool_fused_epilogue") if fused else "linear_pool_unfused", idx)
        ctx.save_for_backward(x, W, idx)
        ctx.cfg = (G, gs, C)
        ctx.mark_non_differentiable(idx)
        return out, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, gout, _gidx):
        x, W, idx = ctx.saved_tensors
        G, gs, C = ctx.cfg
        gout = _rows(gout)
        Ci = x.shape[1]
        dev = x.device
        gW = torch.empty((C, Ci), device=dev, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        gx = order = pts = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((G * gs * 3, Ci), device=dev, dtype=torch.float32)
            pts64, order64 = torch.sort(idx, dim=1, stable=True)        # index bookkeeping only
            pts, order = pts64.int().contiguous(), order64.int().contiguous()
        Wc = _rows(W)
        call("vnp_pool_lin_bwd", ptr(gout), _ld(gout), ptr(idx), ptr(order), ptr(pts), ptr(x), _ld(x), ptr(Wc), _ld(Wc),
             G, gs, C, Ci, ptr(gW), Ci, ptr(gx), Ci)
        return gx, gW, None, None, None


class VNMeanPoolFn(Function):
    """mean_pool (models/vn_layers.py:170-171) over groups of gs consecutive points."""

    @staticmethod
    def forward(ctx, x, G, gs):
        x = _rows(x)
        C = x.shape[1]
        out = torch.empty((G * 3, C), device=x.device, dtype=torch.float32)
        call("vnp_vn_mean_pool_fwd", ptr(x), _ld(x), G, gs, C, ptr(out), C)
        ctx.cfg = (G, gs, C)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        G, gs, C = ctx.cfg
        gout = _rows(gout)
        gx = torch.empty((G * gs * 3, C), device=gout.device, dtype=torch.float32)
        call("vnp_vn_mean_pool_bwd", ptr(gout), _ld(gout), G, gs, C, ptr(gx), C)
        return gx, None, None


class FoldFn(Function):
    """Rank-2 expansion of the first FoldingNet layer (models/pcn.py:375-387 with the broadcast channels de-duplicated):
    pd[((b*Mc+mc)*S+s)*3+t, j] = bias[b*3+t, j] + wseed[j]*seed[(b), s, t] + wcoarse[j]*coarse[b, mc, t]."""

    @staticmethod
    def forward(ctx, bias, wseed, wcoarse, seed, coarse):
        bias = _rows(bias)
        B, Mc, _ = coarse.shape
        coarse = coarse.contiguous()
        seed = seed.contiguous()
        S = seed.shape[-2]
        if seed.dim() != 3 or seed.shape[0] not in (1, B) or seed.shape[2] != 3:
            raise _lib.VnpError(f"fold: seed must be [1 or B, S, 3], got {tuple(seed.shape)}")
        batched = 1 if seed.shape[0] == B else 0
        W = bias.shape[1]
        wseed, wcoarse = wseed.contiguous(), wcoarse.contiguous()
        out = torch.empty((B * Mc * S * 3, W), device=bias.device, dtype=torch.float32)
        call("vnp_fold_fwd", ptr(bias), _ld(bias), ptr(wseed), ptr(wcoarse), ptr(seed), batched, ptr(coarse), B, Mc, S, W,
             ptr(out), W)
        _trace("fold_layer0_unfused")
        ctx.save_for_backward(wcoarse, seed, coarse)
        ctx.cfg = (B, Mc, S, W, batched)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        wcoarse, seed, coarse = ctx.saved_tensors
        B, Mc, S, W, batched = ctx.cfg
        g = _rows(g)
        dev = g.device
        part = torch.empty((B * Mc, 5, W), device=dev, dtype=torch.float32)
        call("vnp_fold_bwd", ptr(g), _ld(g), ptr(seed), batched, ptr(coarse), B, Mc, S, W, ptr(part))
        per_sample = reduce_rows(part, B, Mc, 5 * W)                    # [B][5W]: sums over the coarse points
        gbias = per_sample[:, :3 * W].reshape(B * 3, W)
        tot = reduce_rows(per_sample, 1, B, 5 * W)                      # [1][5W]
        gws, gwc = tot[0, 3 * W:4 * W], tot[0, 4 * W:]
        gcoarse = None
        if ctx.needs_input_grad[4]:
            R = B * Mc * S * 3
            tmp = torch.empty(R, device=dev, dtype=torch.float32)
            call("vnp_rowdot", ptr(g), _ld(g), ptr(wcoarse), R, W, None, 1, ptr(tmp))
            gcoarse = torch.empty((B, Mc, 3), device=dev, dtype=torch.float32)
            call("vnp_group_sum3", ptr(tmp), B * Mc, S, ptr(gcoarse), 0)
        return gbias, gws, gwc, None, gcoarse


class FoldBnLeakyFn(Function):
    """FoldingNet layer 0 end to end (models/pcn.py:375-387 + vn_layers.py:46-74 with the broadcast input de-duplicated):
    out = leaky(bn(p), d) with p | d = bias + wseed*seed + wcoarse*coarse recomputed inside every pass, so the 3.2 GB
    (p | d) tensor of FoldFn + VNBnLeakyFn is never written or read (vnp_fold_fused, modes 0-3)."""

    @staticmethod
    def forward(ctx, bias, wseed, wcoarse, seed, coarse, gamma, beta, running_mean, running_var, training, ns):
        bias = _rows(bias)
        B, Mc, _ = coarse.shape
        coarse, seed = coarse.contiguous(), seed.contiguous()
        if seed.dim() != 3 or seed.shape[0] not in (1, B) or seed.shape[2] != 3:
            raise _lib.VnpError(f"fold: seed must be [1 or B, S, 3], got {tuple(seed.shape)}")
        S = seed.shape[1]
        batched = 1 if seed.shape[0] == B else 0
        C = bias.shape[1] // 2
        wseed, wcoarse = wseed.contiguous(), wcoarse.contiguous()
        dev = bias.device
        P = B * Mc * S
        cpc = FoldBnLeakyFn.coarse_per_chunk(Mc, S)
        if not cpc:
            raise _lib.VnpError(f"FoldBnLeakyFn: grid^2 = {S} exceeds vnp_chunk_points()")
        nch = B * Mc // cpc
        args = (ptr(bias), _ld(bias), ptr(wseed), ptr(wcoarse), ptr(seed), batched, ptr(coarse), B, Mc, S, C)
        if training:
            part = torch.empty((nch, C, 2), device=dev, dtype=torch.float32)
            call("vnp_fold_fused", 0, *args, None, None, None, None, ns, None, 0, None, 0, ptr(part), None, 0, None, None, cpc)
            mean = torch.empty(C, device=dev, dtype=torch.float32)
            rstd = torch.empty(C, device=dev, dtype=torch.float32)
            call("vnp_vn_bn_finalize_chunked", ptr(part), nch, cpc * S, C, P, ptr(mean), ptr(rstd), ptr(running_mean),
                 ptr(running_var), BN_MOMENTUM, BN_EPS)
        else:
            mean = running_mean
            rstd = torch.empty(C, device=dev, dtype=torch.float32)
            call("vnp_rstd_from_var", ptr(running_var), ptr(rstd), C, BN_EPS)
        out = torch.empty((P * 3, C), device=dev, dtype=torch.float32)
        _trace("fold_layer0_fused")
        call("vnp_fold_fused", 1, *args, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ns, ptr(out), C, None, 0, None, None, 0,
             None, None, cpc)
        ctx.save_for_backward(bias, wseed, wcoarse, seed, coarse, gamma, beta, mean, rstd)
        ctx.cfg = (B, Mc, S, C, batched, training, ns, cpc)
        return out

    @staticmethod
    def coarse_per_chunk(Mc, S):
        """Coarse points per statistics chunk: the largest divisor of ``Mc`` whose S dense points each fit a chunk of
        vnp_chunk_points() points, so that a chunk never straddles two samples (0: no such divisor, S too large)."""
        cap = _lib.chunk_points() // S
        for d in range(min(cap, Mc), 0, -1):
            if Mc % d == 0:
                return d
        return 0

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        bias, wseed, wcoarse, seed, coarse, gamma, beta, mean, rstd = ctx.saved_tensors
        B, Mc, S, C, batched, training, ns, cpc = ctx.cfg
        go = _rows(go)
        dev = go.device
        P = B * Mc * S
        nch = B * Mc // cpc
        args = (ptr(bias), _ld(bias), ptr(wseed), ptr(wcoarse), ptr(seed), batched, ptr(coarse), B, Mc, S, C)
        bn = (ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ns)
        part = torch.empty((nch, C, 2), device=dev, dtype=torch.float32)
        call("vnp_fold_fused", 2, *args, *bn, None, 0, ptr(go), _ld(go), ptr(part), None, 0, None, None, cpc)
        Ssum, planar = reduce_pairs(part, nch, C)
        part5 = torch.empty((nch, 5, 2 * C), device=dev, dtype=torch.float32)
        gcoarse = torch.empty((B, Mc, 3), device=dev, dtype=torch.float32)
        call("vnp_fold_fused", 3, *args, *bn, None, 0, ptr(go), _ld(go), None, ptr(Ssum) if training else None, P,
             ptr(part5), ptr(gcoarse), cpc)
        W = 2 * C
        per_sample = reduce_rows(part5, B, Mc // cpc, 5 * W)                           # [B][5W]: a chunk lies inside one sample
        gbias = per_sample[:, :3 * W].reshape(B * 3, W)
        tot = reduce_rows(per_sample, 1, B, 5 * W)
        gws, gwc = tot[0, 3 * W:4 * W], tot[0, 4 * W:]
        return (gbias, gws, gwc, None, gcoarse if ctx.needs_input_grad[4] else None, planar[1], planar[0], None, None, None, None)


GROUP_FOLD = os.environ.get("VNP_NO_GROUP_FOLD", "0") != "1"      # A/B: the unfused form (outer product + BatchNorm / leaky passes over p | d)


def group_fold_ok(x_cols, S, C):
    """Can GroupFoldBnLeakyFn take a VNLinearLeakyReLU whose per-point input has ``x_cols`` channels, S points per group and C outputs?"""
    return GROUP_FOLD and x_cols == 1 and S % 2 == 0 and 2 <= S <= 16 and S <= _lib.chunk_points()


class GroupFoldBnLeakyFn(Function):
    """VNLinearLeakyReLU on cat([x, group features]) where x is ONE VN channel per point and the features are shared by the S points
    of a group (Attention_VN_FoldingNet's folding stages, models/pcn.py:486-488; vn_layers.py:46-74), end to end without the
    [P*3, 2C] pre-activation tensor: p | d = bias[group] + w x[point] is rebuilt inside each of the four passes (vnp_group_fold_fused).
    x [P*3, 1] rows, w [2C] = [Wf[:, x] | Wd[:, x]], bias [G*3, 2C] = group features times their weight columns (VNLinearFn).
    Replaces, per stage and step: an outer-product launch writing 1.6 GB, a statistics pass, the BatchNorm / leaky forward, two
    backward passes, a column-sum pass (bias gradient) and a matrix-vector pass (input gradient) over that tensor."""

    @staticmethod
    def forward(ctx, x, w, bias, S, gamma, beta, running_mean, running_var, training, ns):
        bias = _rows(bias)
        x = x.reshape(-1).contiguous()
        w = w.contiguous()
        C = bias.shape[1] // 2
        G = bias.shape[0] // 3
        P = G * S
        if x.numel() != P * 3 or w.numel() != 2 * C:
            raise _lib.VnpError(f"group fold: x {tuple(x.shape)}, w {tuple(w.shape)}, bias {tuple(bias.shape)}, S = {S} do not fit")
        dev = bias.device
        cpc = FoldBnLeakyFn.coarse_per_chunk(G, S)
        nch = G // cpc
        args = (ptr(bias), _ld(bias), ptr(w), ptr(x), G, S, C)
        if training:
            part = torch.empty((nch, C, 2), device=dev, dtype=torch.float32)
            call("vnp_group_fold_fused", 0, *args, None, None, None, None, ns, None, 0, None, 0, ptr(part), None, 0, None, None, None, cpc)
            mean = torch.empty(C, device=dev, dtype=torch.float32)
            rstd = torch.empty(C, device=dev, dtype=torch.float32)
            call("vnp_vn_bn_finalize_chunked", ptr(part), nch, cpc * S, C, P, ptr(mean), ptr(rstd), ptr(running_mean),
                 ptr(running_var), BN_MOMENTUM, BN_EPS)
        else:
            mean = running_mean
            rstd = torch.empty(C, device=dev, dtype=torch.float32)
            call("vnp_rstd_from_var", ptr(running_var), ptr(rstd), C, BN_EPS)
        out = torch.empty((P * 3, C), device=dev, dtype=torch.float32)
        _trace("group_fold_fused")
        call("vnp_group_fold_fused", 1, *args, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ns, ptr(out), C, None, 0, None, None, 0,
             None, None, None, cpc)
        ctx.save_for_backward(x, w, bias, gamma, beta, mean, rstd)
        ctx.cfg = (G, S, C, training, ns, cpc)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        x, w, bias, gamma, beta, mean, rstd = ctx.saved_tensors
        G, S, C, training, ns, cpc = ctx.cfg
        go = _rows(go)
        dev = go.device
        P = G * S
        nch = G // cpc
        args = (ptr(bias), _ld(bias), ptr(w), ptr(x), G, S, C)
        bn = (ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ns)
        part = torch.empty((nch, C, 2), device=dev, dtype=torch.float32)
        call("vnp_group_fold_fused", 2, *args, *bn, None, 0, ptr(go), _ld(go), ptr(part), None, 0, None, None, None, cpc)
        Ssum, planar = reduce_pairs(part, nch, C)
        part5 = torch.empty((nch, 5, 2 * C), device=dev, dtype=torch.float32)
        gbias = torch.empty((G * 3, 2 * C), device=dev, dtype=torch.float32)
        ncb = (C + 63) // 64
        gx_part = torch.empty((ncb, P * 3), device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        call("vnp_group_fold_fused", 3, *args, *bn, None, 0, ptr(go), _ld(go), None, ptr(Ssum) if training else None, P,
             ptr(part5), ptr(gbias), ptr(gx_part), cpc)
        gw = reduce_rows(part5[:, 3, :].contiguous(), 1, nch, 2 * C)[0]
        gx = reduce_rows(gx_part, 1, ncb, P * 3).view(P * 3, 1) if gx_part is not None else None
        return gx, gw, gbias, None, planar[1], planar[0], None, None, None, None


class LinOutFn(Function):
    """Final VNLinear(C -> 1) plus the residual point_feat (models/pcn.py:346,387):
    fine[(b*Md+m)*3+t] = sum_c x[.., c] w[c] + coarse[b, m // S, t]."""

    @staticmethod
    def forward(ctx, x, w, coarse, S):
        x = _rows(x)
        R, C = x.shape
        w = w.reshape(-1).contiguous()
        coarse = coarse.contiguous()
        out = torch.empty(R, device=x.device, dtype=torch.float32)
        call("vnp_rowdot", ptr(x), _ld(x), ptr(w), R, C, ptr(coarse), S, ptr(out))
        ctx.save_for_backward(x, w)
        ctx.cfg = (S, tuple(coarse.shape))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        S, cshape = ctx.cfg
        R, C = x.shape
        g = g.contiguous()
        dev = g.device
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((R, C), device=dev, dtype=torch.float32)
            call("vnp_outer", ptr(g), ptr(w), R, C, ptr(gx), C)
        nch = _nchunk(R)
        nch_pad = (nch + 255) // 256 * 256 if nch > 1024 else nch
        part = torch.zeros((nch_pad, C), device=dev, dtype=torch.float32) if nch_pad != nch else \
            torch.empty((nch, C), device=dev, dtype=torch.float32)
        call("vnp_colsum", ptr(x), 0, _ld(x), 1, R, C, ptr(g), ptr(part))
        gw = reduce_rows(part, 1, nch_pad, C)
        gcoarse = torch.empty(cshape, device=dev, dtype=torch.float32)
        call("vnp_group_sum3", ptr(g), R // 3 // S, S, ptr(gcoarse), 0)
        return gx, gw, gcoarse, None


# both clouds at least this large -> vnp_chamfer_fwd_pruned (below it the brute-force kernel's slicing fills the chip better);
# VNP_CHAMFER_BRUTE=1 forces the brute-force kernel everywhere
CHAMFER_PRUNED_MIN_POINTS = (1 << 30) if os.environ.get("VNP_CHAMFER_BRUTE") == "1" else 4096
# bench.py sets this to a list to receive, per pruned call, the device word with the number of (query, target) pairs the two scans
# evaluated (int64 [1]; a full scan evaluates 2 B n m)
CHAMFER_STATS = None


class ChamferFn(Function):
    """chamfer_3DFunction (extensions/chamfer_distance/chamfer_distance.py:29-71) on libvnpcc."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        _lib.require_gpu(xyz1, xyz2)
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        B, n, _ = xyz1.shape
        m = xyz2.shape[1]
        dev = xyz1.device
        dist1 = torch.empty((B, n), device=dev, dtype=torch.float32)
        dist2 = torch.empty((B, m), device=dev, dtype=torch.float32)
        idx1 = torch.empty((B, n), device=dev, dtype=torch.int32)
        idx2 = torch.empty((B, m), device=dev, dtype=torch.int32)
        nbytes = _lib.lib().vnp_chamfer_pruned_ws_bytes(B, n, m) if min(n, m) >= CHAMFER_PRUNED_MIN_POINTS else 0
        if nbytes > 0:
            # dense clouds: pruned exact search (same bits as the brute-force kernel; tests/test_gpu_chamfer.py)
            ws = torch.empty((nbytes + 7) // 8, device=dev, dtype=torch.int64)
            call("vnp_chamfer_fwd_pruned", B, n, ptr(xyz1), m, ptr(xyz2), ptr(dist1), ptr(idx1), ptr(dist2), ptr(idx2), ptr(ws))
            _trace("chamfer_pruned")
            if CHAMFER_STATS is not None:
                CHAMFER_STATS.append(ws[1:2].clone())
        else:
            _trace("chamfer_brute")
            ws = torch.empty(B * max(n, m), device=dev, dtype=torch.int64)
            call("vnp_chamfer_fwd", B, n, ptr(xyz1), m, ptr(xyz2), ptr(dist1), ptr(idx1), ptr(dist2), ptr(idx2), ptr(ws))
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    @once_differentiable
    def backward(ctx, g1, g2, _gi1, _gi2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        B, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1, g2 = g1.contiguous(), g2.contiguous()
        gx1 = torch.empty_like(xyz1)
        gx2 = torch.empty_like(xyz2)
        call("vnp_chamfer_bwd", B, n, ptr(xyz1), m, ptr(xyz2), ptr(g1), ptr(idx1), ptr(g2), ptr(idx2), ptr(gx1), ptr(gx2))
        return gx1, gx2


# ------------------------------------------------------------------------------- graph ops (configs 3, 5)
# coordinate graphs (D = 3) on the fused one-wave-per-query kernel; VNP_KNN_COORDS=0 keeps the distance-matrix path (A/B)
KNN_COORDS = os.environ.get("VNP_KNN_COORDS", "1") != "0"


def knn_indices(ref, query, k, return_dist=False):
    """ref [B,Nr,D], query [B,Nq,D] (point-major, fp32, contiguous) -> idx [B,Nq,k] int32 local indices, sorted by
    (squared distance, index).  Replaces KNN_CUDA (models/dgcnn.py:11,236)."""
    _lib.require_gpu(ref, query)
    ref, query = ref.contiguous().float(), query.contiguous().float()
    B, Nr, D = ref.shape
    Nq = query.shape[1]
    idx = torch.empty((B, Nq, k), device=ref.device, dtype=torch.int32)
    dist = torch.empty((B, Nq, k), device=ref.device, dtype=torch.float32) if return_dist else None
    if D == 3 and Nr <= 4096 and k <= 64 and KNN_COORDS:
        # coordinate graphs: one wave per query on register-resident distances with a threshold selection, no distance matrix
        # (vnp_knn_coords_kernel behind vnp_knn; same bits and order)
        call("vnp_knn", B, D, ptr(ref), Nr, ptr(query), Nq, k, ptr(idx), ptr(dist))
    elif Nr <= 16384 and Nq * Nr >= 4096 and B * Nq * Nr <= (1 << 29):
        # distance matrix once + per-row selection instead of k full passes per query (same bits as vnp_knn; ~50x faster for the
        # feature-space graphs at D = 192, ~3x for coordinates)
        ws = torch.empty(B * Nq * Nr, device=ref.device, dtype=torch.float32)
        call("vnp_knn_matrix", B, D, ptr(ref), Nr, ptr(query), Nq, k, ptr(idx), ptr(dist), ptr(ws))
    else:
        call("vnp_knn", B, D, ptr(ref), Nr, ptr(query), Nq, k, ptr(idx), ptr(dist))
    return (idx, dist) if return_dist else idx


# pointnet2_ops' furthest_point_sampling_kernel skips candidates with |p|^2 <= 1e-3 (`if (mag <= 1e-3) continue;`): on clouds
# centred at the origin ~0.3 points per 2048 fall inside that ball and would be picked by plain FPS but never by the package the
# reference calls.  Default: the package's convention (so reference checkpoints see the centres they were trained with);
# VNP_FPS_PLAIN=1 (or skip_near_origin=False) selects plain FPS.  PARITY UNPINNED either way (third-party source absent).
FPS_SKIP_NEAR_ORIGIN = os.environ.get("VNP_FPS_PLAIN", "0") != "1"


def fps_indices(xyz, m, skip_near_origin=None):
    """furthest_point_sample(xyz[B,N,3], m) -> idx [B,m] int32 (starts at 0).  Replaces pointnet2_ops (models/pcn.py:7)."""
    _lib.require_gpu(xyz)
    xyz = xyz.contiguous().float()
    B, N, _ = xyz.shape
    idx = torch.empty((B, m), device=xyz.device, dtype=torch.int32)
    skip = FPS_SKIP_NEAR_ORIGIN if skip_near_origin is None else bool(skip_near_origin)
    call("vnp_fps_ex", B, N, ptr(xyz), m, ptr(idx), 1 if skip else 0)
    return idx


class GatherPointsFn(Function):
    """rows layout: out point i <- x point src[i] (global point ids, int32 [Pout]).  gather_operation of pointnet2_ops
    (models/dgcnn.py:214-218) on VN features; backward = scatter-add."""

    @staticmethod
    def forward(ctx, x, src):
        x = _rows(x)
        src = src.reshape(-1).contiguous()
        P_out, W = src.numel(), x.shape[1]
        out = torch.empty((P_out * 3, W), device=x.device, dtype=torch.float32)
        call("vnp_gather_points", ptr(x), _ld(x), ptr(src), P_out, W, ptr(out), W)
        ctx.save_for_backward(src)
        ctx.cfg = (x.shape[0] // 3, W)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (src,) = ctx.saved_tensors
        P_in, W = ctx.cfg
        g = _rows(g)
        gx = torch.empty((P_in * 3, W), device=g.device, dtype=torch.float32)
        call("vnp_scatter_add_points", ptr(g), _ld(g), ptr(src), src.numel(), W, ptr(gx), W, P_in)
        return gx, None


class EdgeGatherAddFn(Function):
    """Edge tensor of vn_get_graph_feature (models/dgcnn.py:250-278) after the layer's linear maps:
    out[(p,j)] = u[nbr[p,j]] + v[p], with u = x W1^T and v = x (W2-W1)^T computed per POINT (k times fewer GEMM rows than
    applying W to cat(feature - x, x) per EDGE).  nbr: global point ids [P,k] int32."""

    @staticmethod
    def forward(ctx, u, v, nbr):
        u, v = _rows(u), _rows(v)
        nbr = nbr.contiguous()
        P, k = nbr.shape
        W = u.shape[1]
        out = torch.empty((P * k * 3, W), device=u.device, dtype=torch.float32)
        call("vnp_edge_gather_add", ptr(u), _ld(u), ptr(v), _ld(v), ptr(nbr), P, k, W, ptr(out), W)
        ctx.save_for_backward(nbr)
        ctx.cfg = (u.shape[0] // 3, P, k, W)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (nbr,) = ctx.saved_tensors
        Pu, P, k, W = ctx.cfg
        g = _rows(g)
        dev = g.device
        gu = torch.empty((Pu * 3, W), device=dev, dtype=torch.float32)
        call("vnp_scatter_add_points", ptr(g), _ld(g), ptr(nbr), P * k, W, ptr(gu), W, Pu)
        gv = torch.empty((P * 3, W), device=dev, dtype=torch.float32)
        call("vnp_vn_mean_pool_fwd", ptr(g), _ld(g), P, k, W, ptr(gv), W)          # mean over k ...
        gv.mul_(float(k))                                                          # ... times k = sum
        return gu, gv, None


# VNP_EDGE_ATOMICS=1: the neighbour-feature gradient of the fused EdgeConv by float atomics (the rounds 1-2 form) instead of the gather over
# the reverse adjacency (A/B measurement)
EDGE_GRAD_ATOMICS = os.environ.get("VNP_EDGE_ATOMICS", "0") == "1"


def edge_reverse(nbr, Pu):
    """Reverse adjacency (start [Pu + 1], rev [Pq * k]) of a neighbour table, cached on the tensor: layers that share a graph share it."""
    cached = getattr(nbr, "_vnp_rev", None)
    if cached is not None and cached[0].shape[0] == Pu + 1:
        return cached
    Pq, k = nbr.shape
    start = torch.empty(Pu + 1, device=nbr.device, dtype=torch.int32)
    rev = torch.empty(Pq * k, device=nbr.device, dtype=torch.int32)
    ws = torch.empty(Pu + 1, device=nbr.device, dtype=torch.int32)
    call("vnp_edge_reverse", ptr(nbr), Pq, k, Pu, ptr(start), ptr(rev), ptr(ws))
    nbr._vnp_rev = (start, rev)
    return start, rev


def _edge_conv_forward(u, v, nbr, gamma, beta, running_mean, running_var, training, ns):
    """The two forward passes of the fused EdgeConv on u [Pu*3, >= 2C], v [Pq*3, >= 2C] (row strides free): (out, mean, rstd, nblk)."""
    Pq, k = nbr.shape
    C = u.shape[1] // 2
    dev = u.device
    qb = _lib.lib().vnp_edge_block_queries()
    nblk = (Pq + qb - 1) // qb
    if training:
        part = torch.empty((nblk, C, 2), device=dev, dtype=torch.float32)
        call("vnp_edge_fused", 0, ptr(u), _ld(u), ptr(v), _ld(v), ptr(nbr), Pq, k, C, None, None, None, None, float(ns),
             ptr(part), None, 0, None, 0, None, 0, None, 0, 0, None, 0)
        mean = torch.empty(C, device=dev, dtype=torch.float32)
        rstd = torch.empty(C, device=dev, dtype=torch.float32)
        call("vnp_vn_bn_finalize_chunked", ptr(part), nblk, qb * k, C, Pq * k, ptr(mean), ptr(rstd), ptr(running_mean),
             ptr(running_var), BN_MOMENTUM, BN_EPS)
    else:
        mean = running_mean
        rstd = torch.empty(C, device=dev, dtype=torch.float32)
        call("vnp_rstd_from_var", ptr(running_var), ptr(rstd), C, BN_EPS)
    out = torch.empty((Pq * 3, C), device=dev, dtype=torch.float32)
    call("vnp_edge_fused", 1, ptr(u), _ld(u), ptr(v), _ld(v), ptr(nbr), Pq, k, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
         float(ns), None, ptr(out), C, None, 0, None, 0, None, 0, 0, None, 0)
    return out, mean, rstd, nblk


def _edge_conv_backward(go, u, v, nbr, gamma, beta, mean, rstd, training, ns, nblk, gu, gv):
    """The backward passes: fills gu [Pu*3, >= 2C] and gv [Pq*3, >= 2C] (row strides free: column halves of one buffer in the packed
    form) and returns the per-channel sums as planar [2, C] rows (dL/dbeta, dL/dgamma)."""
    Pq, k = nbr.shape
    C = u.shape[1] // 2
    Pu = u.shape[0] // 3
    part = torch.empty((nblk, C, 2), device=go.device, dtype=torch.float32)
    call("vnp_edge_fused", 2, ptr(u), _ld(u), ptr(v), _ld(v), ptr(nbr), Pq, k, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ns,
         ptr(part), None, 0, ptr(go), _ld(go), None, 0, None, 0, 0, None, 0)
    S, planar = reduce_pairs(part, nblk, C)
    sums = ptr(S) if training else None
    if EDGE_GRAD_ATOMICS:
        call("vnp_edge_fused", 3, ptr(u), _ld(u), ptr(v), _ld(v), ptr(nbr), Pq, k, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ns,
             None, None, 0, ptr(go), _ld(go), sums, Pq * k, ptr(gu), _ld(gu), Pu, ptr(gv), _ld(gv))
    else:
        # gv per query as before; gu per SOURCE point through the reverse adjacency (built once per graph: conv4 / conv5 share theirs)
        start, rev = edge_reverse(nbr, Pu)
        call("vnp_edge_fused", 3, ptr(u), _ld(u), ptr(v), _ld(v), ptr(nbr), Pq, k, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ns,
             None, None, 0, ptr(go), _ld(go), sums, Pq * k, None, 0, 0, ptr(gv), _ld(gv))
        call("vnp_edge_fused_gather", ptr(u), _ld(u), ptr(v), _ld(v), ptr(start), ptr(rev), Pu, k, C, ptr(mean), ptr(rstd), ptr(gamma),
             ptr(beta), ns, ptr(go), _ld(go), sums, Pq * k, ptr(gu), _ld(gu))
    return planar


class EdgeConvFn(Function):
    """VNLinearLeakyReLU on the kNN edge tensor followed by mean_pool over the k neighbours, with the edge tensor never
    materialised (vnp_edge_fused): u, v = the layer's linear maps applied per point ([P*3, 2C'], columns feat | dir),
    nbr [Pq, k] global neighbour ids.  Equivalent to EdgeGatherAddFn -> VNBnLeakyFn -> VNMeanPoolFn."""

    @staticmethod
    def forward(ctx, u, v, nbr, gamma, beta, running_mean, running_var, training, ns):
        u, v = _rows(u), _rows(v)
        nbr = nbr.contiguous()
        out, mean, rstd, nblk = _edge_conv_forward(u, v, nbr, gamma, beta, running_mean, running_var, training, ns)
        ctx.save_for_backward(u, v, nbr, gamma, beta, mean, rstd)
        ctx.cfg = (bool(training), float(ns), nblk)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        u, v, nbr, gamma, beta, mean, rstd = ctx.saved_tensors
        training, ns, nblk = ctx.cfg
        go = _rows(go)
        C = u.shape[1] // 2
        gu = torch.empty((u.shape[0], 2 * C), device=go.device, dtype=torch.float32)
        gv = torch.empty((v.shape[0], 2 * C), device=go.device, dtype=torch.float32)
        S = _edge_conv_backward(go, u, v, nbr, gamma, beta, mean, rstd, training, ns, nblk, gu, gv)
        return gu, gv, None, S[1], S[0], None, None, None, None


class EdgeConvPackedFn(Function):
    """EdgeConvFn for a graph on ONE point set with u and v side by side in one matrix uv [P*3, 4C'] = x [Wu; Wv]^T (one GEMM instead
    of two, forward and backward): u = uv[:, :2C'], v = uv[:, 2C':]; the gradient comes back as one [P*3, 4C'] matrix whose column halves
    the backward kernels fill in place (no zero-padded slice gradients, no add)."""

    @staticmethod
    def forward(ctx, uv, nbr, gamma, beta, running_mean, running_var, training, ns):
        uv = _rows(uv)
        nbr = nbr.contiguous()
        C2 = uv.shape[1] // 2
        out, mean, rstd, nblk = _edge_conv_forward(uv[:, :C2], uv[:, C2:], nbr, gamma, beta, running_mean, running_var, training, ns)
        ctx.save_for_backward(uv, nbr, gamma, beta, mean, rstd)
        ctx.cfg = (bool(training), float(ns), nblk)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        uv, nbr, gamma, beta, mean, rstd = ctx.saved_tensors
        training, ns, nblk = ctx.cfg
        go = _rows(go)
        C2 = uv.shape[1] // 2
        guv = torch.empty_like(uv)
        S = _edge_conv_backward(go, uv[:, :C2], uv[:, C2:], nbr, gamma, beta, mean, rstd, training, ns, nblk, guv[:, :C2], guv[:, C2:])
        return guv, None, S[1], S[0], None, None, None, None


class EdgeWeightsFn(Function):
    """[Wu; Wv] [4C', C] of an EdgeConv layer from its VNLinearLeakyReLU weights Wf, Wd [C', 2C] (input channels: C of feature - x, then C
    of x): W cat(f_j - x, x) = W1 f_j + (W2 - W1) x, so Wu = [Wf1; Wd1] acts on the neighbour and Wv = [Wf2 - Wf1; Wd2 - Wd1] on the centre.
    One Function instead of eight autograd nodes of slices, subtractions and concatenations per layer and step."""

    @staticmethod
    def forward(ctx, Wf, Wd):
        C = Wf.shape[1] // 2
        W = torch.stack([Wf, Wd])                                # [2, C', 2C]
        a, b = W[:, :, :C], W[:, :, C:]
        ctx.C = C
        return torch.cat([a.reshape(-1, C), (b - a).reshape(-1, C)], dim=0)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        C = ctx.C
        half = g.shape[0] // 2
        gu, gv = g[:half], g[half:]
        gW = torch.cat([(gu - gv).reshape(2, -1, C), gv.reshape(2, -1, C)], dim=2)      # [2, C', 2C]: d/dW1 = gu - gv, d/dW2 = gv
        return gW[0], gW[1]


# ------------------------------------------------------------------------------- transformer ops (configs 4, 5)
def gemm_batched(A, B, *, op_a=0, op_b=0, out=None):
    """A, B: 3-D contiguous [nbatch, ., .]; C_i = op(A_i) op(B_i)."""
    _lib.require_gpu(A, B)
    A, B = A.contiguous(), B.contiguous()
    nb = A.shape[0]
    M, K = (A.shape[1], A.shape[2]) if op_a == 0 else (A.shape[2], A.shape[1])
    N = B.shape[1] if op_b == 0 else B.shape[2]
    if out is None:
        out = torch.empty((nb, M, N), device=A.device, dtype=torch.float32)
    call("vnp_gemm_batched", op_a, op_b, M, N, K, ptr(A), A.shape[2], A.shape[1] * A.shape[2], ptr(B), B.shape[2],
         B.shape[1] * B.shape[2], ptr(out), N, M * N, nb, 0)
    return out


def _to_heads(rows, B, N, H):
    C = rows.shape[1]
    heads = torch.empty((B * H, N, (C // H) * 3), device=rows.device, dtype=torch.float32)
    call("vnp_rows_to_heads", ptr(rows), _ld(rows), B, N, C, H, ptr(heads))
    return heads


def _from_heads(heads, B, N, C, H, out=None):
    """heads [B*H, N, (C/H)*3] -> rows [B*N*3, C]; ``out``: a [B*N*3, C] view (any row stride) to fill instead of a new tensor."""
    rows = torch.empty((B * N * 3, C), device=heads.device, dtype=torch.float32) if out is None else out
    call("vnp_heads_to_rows", ptr(heads), B, N, C, H, ptr(rows), _ld(rows))
    return rows


ATTN_TOKEN_MAJOR = os.environ.get("VNP_ATTN_TOKEN_MAJOR", "0") == "1"      # A/B: the rounds 1-2 layout (head dimension on N / K of the products)


ATTN_FM_MIN_FLOP = 8e9          # per attention product (2 B H N^2 Dh): below this the products are launch-bound and the extra permutes cost more


def _attn_token_major(B, N, C, H):
    """The rounds 1-2 layout (head dimension on N / K, bounds-checked kernels where it is not a multiple of 128 / 32) for small attention
    problems — VN-PoinTr's 128 tokens: 1.2 GFLOP per product, measured 29.1 (this layout) vs 30.2 ms per step — and the padded
    token-major / feature-major layout for big ones (Attention_VN_FoldingNet's 1024 tokens: 39 GFLOP per product, 32.8 -> 31.8 ms)."""
    return ATTN_TOKEN_MAJOR or 2.0 * B * H * N * N * (C // H) * 3 < ATTN_FM_MIN_FLOP


def _attn_dims(C, H):
    """(Dh, ldh, fr): the head dimension 3 C / H, its K padding (multiple of 32) and its M padding (whole 96-row tiles)."""
    Dh = (C // H) * 3
    return Dh, (Dh + 31) // 32 * 32, (Dh + 95) // 96 * 96


def _heads_tm(rows, B, N, H, ldh):
    """rows [B*N*3, C] (any row stride) -> token-major heads [B*H, N, ldh], columns beyond 3 C / H zero."""
    C = rows.shape[1]
    heads = torch.empty((B * H, N, ldh), device=rows.device, dtype=torch.float32)
    call("vnp_rows_to_heads_pad", ptr(rows), _ld(rows), B, N, C, H, ptr(heads), ldh)
    return heads


def _heads_fm(rows, B, N, H, fr):
    """rows [B*N*3, C] (any row stride) -> feature-major heads [B*H, fr, N], rows beyond 3 C / H zero."""
    C = rows.shape[1]
    heads = torch.empty((B * H, fr, N), device=rows.device, dtype=torch.float32)
    call("vnp_rows_to_heads_fm", ptr(rows), _ld(rows), B, N, C, H, ptr(heads), fr)
    return heads


def _rows_from_fm(hf, B, N, C, H, out=None):
    rows = torch.empty((B * N * 3, C), device=hf.device, dtype=torch.float32) if out is None else out
    call("vnp_heads_fm_to_rows", ptr(hf), hf.shape[1], B, N, C, H, ptr(rows), _ld(rows))
    return rows


def _attention_forward(q, k, v, B, N, H, scale):
    """softmax(q k^T scale) v per (sample, head); q, k, v rows [B*N*3, C] (column slices allowed).  Returns (out rows, P [B*H, N, N]).
    The head dimension Dh = 3 C / H (144 in Attention_VN_FoldingNet, 192 in VN-PoinTr) is zero-padded to a multiple of 32 where it is
    the K of a product (q k^T) and laid out feature-major, padded to whole row tiles, where it would have been the N (out^T = v^T P^T):
    every product then runs on the aligned kernels (round 3; before: the bounds-checked ones, 5.4 ms of config 4's step at 70-100 TF/s)."""
    C = q.shape[1]
    _, ldh, fr = _attn_dims(C, H)
    P = gemm_batched(_heads_tm(q, B, N, H, ldh), _heads_tm(k, B, N, H, ldh))      # [B*H, N(q), N(k)] scores
    call("vnp_softmax_fwd", ptr(P), B * H * N, N, float(scale))
    o_fm = gemm_batched(_heads_fm(v, B, N, H, fr), P)                               # [B*H, fr, N(q)] = v^T P^T
    return _rows_from_fm(o_fm, B, N, C, H), P


def _attention_backward(go, q, k, v, P, B, N, H, scale, outs):
    """Fills outs = (dq, dk, dv) rows views [B*N*3, C] (any row stride).  P is consumed (it stays P; dP becomes dS in place)."""
    C = q.shape[1]
    _, ldh, fr = _attn_dims(C, H)
    dP = gemm_batched(_heads_tm(go, B, N, H, ldh), _heads_tm(v, B, N, H, ldh))    # dout v^T: [B*H, N(q), N(k)]
    dv_fm = gemm_batched(_heads_fm(go, B, N, H, fr), P, op_b=1)                     # dout^T P: [B*H, fr, N(k)]
    _rows_from_fm(dv_fm, B, N, C, H, out=outs[2])
    del dv_fm
    call("vnp_softmax_bwd", ptr(P), ptr(dP), B * H * N, N, scale)                  # dP <- dS
    dq_fm = gemm_batched(_heads_fm(k, B, N, H, fr), dP)                             # k^T dS^T: [B*H, fr, N(q)]
    _rows_from_fm(dq_fm, B, N, C, H, out=outs[0])
    del dq_fm
    dk_fm = gemm_batched(_heads_fm(q, B, N, H, fr), dP, op_b=1)                     # q^T dS: [B*H, fr, N(k)]
    _rows_from_fm(dk_fm, B, N, C, H, out=outs[1])


class VNAttentionFn(Function):
    """softmax(q k^T * scale) v over tokens whose features are the flattened (C/H, 3) vectors of each head
    (models/transformer.py:91-104).  q, k, v: rows [B*N*3, C].  The matrix products are batched vnp_gemm GEMMs (fp32 results; split-bf16 or
    fp32-input MFMA, see gemm.hip) on padded token-major / feature-major head tensors (_attention_forward)."""

    @staticmethod
    def forward(ctx, q, k, v, B, N, H, scale):
        q, k, v = _rows(q), _rows(k), _rows(v)
        C = q.shape[1]
        ctx.cfg = (B, N, C, H, float(scale))
        ctx.token_major = _attn_token_major(B, N, C, H)
        if ctx.token_major:
            qh, kh, vh = _to_heads(q, B, N, H), _to_heads(k, B, N, H), _to_heads(v, B, N, H)
            P = gemm_batched(qh, kh)                                      # [B*H, N, N] scores
            call("vnp_softmax_fwd", ptr(P), B * H * N, N, float(scale))
            oh = gemm_batched(P, vh, op_b=1)                               # [B*H, N, Dh]
            ctx.save_for_backward(qh, kh, vh, P)
            return _from_heads(oh, B, N, C, H)
        out, P = _attention_forward(q, k, v, B, N, H, scale)
        ctx.save_for_backward(q, k, v, P)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        B, N, C, H, scale = ctx.cfg
        if ctx.token_major:
            qh, kh, vh, P = ctx.saved_tensors
            goh = _to_heads(_rows(go), B, N, H)
            dP = gemm_batched(goh, vh)                                     # [B*H, N, N]
            dvh = gemm_batched(P, goh, op_a=1, op_b=1)                     # P^T goh
            call("vnp_softmax_bwd", ptr(P), ptr(dP), B * H * N, N, scale)  # dP <- dS
            dqh = gemm_batched(dP, kh, op_b=1)
            dkh = gemm_batched(dP, qh, op_a=1, op_b=1)
            return (_from_heads(dqh, B, N, C, H), _from_heads(dkh, B, N, C, H), _from_heads(dvh, B, N, C, H),
                    None, None, None, None)
        q, k, v, P = ctx.saved_tensors
        outs = tuple(torch.empty((B * N * 3, C), device=go.device, dtype=torch.float32) for _ in range(3))
        _attention_backward(_rows(go), q, k, v, P, B, N, H, scale, outs)
        return outs + (None, None, None, None)


class VNAttentionPackedFn(Function):
    """VNAttentionFn on q | k | v side by side in one matrix qkv [B*N*3, 3C] — the output of ONE projection GEMM over [Wq; Wk; Wv] instead
    of three (a third of the launches, three times the tiles per launch: the 12288-row projections of VN-PoinTr's blocks fill a quarter
    of the chip each) — with the gradient returned as one [B*N*3, 3C] matrix whose column thirds are filled in place."""

    @staticmethod
    def forward(ctx, qkv, B, N, H, scale):
        qkv = _rows(qkv)
        C = qkv.shape[1] // 3
        ctx.cfg = (B, N, C, H, float(scale))
        ctx.token_major = _attn_token_major(B, N, C, H)
        if ctx.token_major:
            qh, kh, vh = (_to_heads(qkv[:, i * C:(i + 1) * C], B, N, H) for i in range(3))
            P = gemm_batched(qh, kh)                                      # [B*H, N, N] scores
            call("vnp_softmax_fwd", ptr(P), B * H * N, N, float(scale))
            oh = gemm_batched(P, vh, op_b=1)                               # [B*H, N, Dh]
            ctx.save_for_backward(qh, kh, vh, P)
            return _from_heads(oh, B, N, C, H)
        out, P = _attention_forward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, N, H, scale)
        ctx.save_for_backward(qkv, P)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        B, N, C, H, scale = ctx.cfg
        g = torch.empty((B * N * 3, 3 * C), device=go.device, dtype=torch.float32)
        if ctx.token_major:
            qh, kh, vh, P = ctx.saved_tensors
            goh = _to_heads(_rows(go), B, N, H)
            dP = gemm_batched(goh, vh)                                     # [B*H, N, N]
            dvh = gemm_batched(P, goh, op_a=1, op_b=1)                     # P^T goh
            call("vnp_softmax_bwd", ptr(P), ptr(dP), B * H * N, N, scale)  # dP <- dS
            dqh = gemm_batched(dP, kh, op_b=1)
            dkh = gemm_batched(dP, qh, op_a=1, op_b=1)
            for i, h in enumerate((dqh, dkh, dvh)):
                _from_heads(h, B, N, C, H, out=g[:, i * C:(i + 1) * C])
            return g, None, None, None, None
        qkv, P = ctx.saved_tensors
        _attention_backward(_rows(go), qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], P, B, N, H, scale,
                            (g[:, :C], g[:, C:2 * C], g[:, 2 * C:]))
        return g, None, None, None, None


class VNLayerNormFn(Function):
    """VNLayerNorm (models/vn_layers.py:129-150) on the rows layout."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        x = _rows(x)
        R, C = x.shape
        P = R // 3
        out = torch.empty((R, C), device=x.device, dtype=torch.float32)
        stats = torch.empty((P, 2), device=x.device, dtype=torch.float32)
        call("vnp_vn_layernorm_fwd", ptr(x), _ld(x), P, C, ptr(gamma), ptr(beta), ptr(out), C, ptr(stats))
        ctx.save_for_backward(x, gamma, beta, stats)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        x, gamma, beta, stats = ctx.saved_tensors
        go = _rows(go)
        R, C = x.shape
        P = R // 3
        nblk = (P + 3) // 4
        nblk_pad = (nblk + 255) // 256 * 256 if nblk > 1024 else nblk
        gx = torch.empty((R, C), device=x.device, dtype=torch.float32)
        part = (torch.zeros if nblk_pad != nblk else torch.empty)((nblk_pad, 2 * C), device=x.device, dtype=torch.float32)
        call("vnp_vn_layernorm_bwd", ptr(go), _ld(go), ptr(x), _ld(x), P, C, ptr(gamma), ptr(beta), ptr(stats), ptr(gx), C, ptr(part))
        gb = reduce_rows(part, 1, nblk_pad, 2 * C)[0]
        return gx, gb[:C].contiguous(), gb[C:].contiguous()
