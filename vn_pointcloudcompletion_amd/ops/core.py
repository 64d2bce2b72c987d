"""Rows-layout helpers, the GEMM wrapper, the reductions and the BatchNorm / leaky passes every Function of the package builds on
(include/vnpcc.h: vnp_gemm*, vnp_reduce_*, vnp_vn_bn_*), plus the layout and weight-plumbing Functions."""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from .._lib import call, ptr
from ._state import HK, SW, POOL_PATH_COUNTS, prof_arm, trace

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


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
    trace("absmax_pass")
    return w


def _fp16x2_shape(op_a, M, N, K, A, B, nsplit):
    """The shapes vnp_gemm_ex runs on the two-plane path (gemm.hip: aligned tiles, K >= 64) that are worth two absmax launches."""
    bm = 192 if op_a == 0 else 128
    return (SW.GEMM_FP16X2 and K >= 64 and K % 32 == 0 and M % bm == 0 and N % 128 == 0 and 2.0 * M * N * K >= SW.GEMM_FP16X2_MIN_FLOP
            and _lib.get_gemm_mode() == "split" and _ld(A) % 4 == 0 and _ld(B) % 4 == 0 and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def gemm(A, B, *, op_a=0, op_b=0, out=None, bias=None, rows_per_group=0, accumulate=False, nsplit=None, amax_a=None, amax_b=None, resid=None):
    """C = op(A) op(B) (+ per-sample bias | + resid).  See vnp_gemm / vnp_gemm_ex / vnp_gemm_add in include/vnpcc.h.  ``amax_a`` /
    ``amax_b``: the operands' ``amax_word`` when the caller already holds it (an operand used by two GEMMs).  ``resid`` [M, N]: a
    residual connection added in the epilogue."""
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
    if resid is not None:
        resid = _rows(resid)
        if bias is not None or accumulate or resid.shape != (M, N):
            raise _lib.VnpError("gemm: resid needs an [M, N] addend and excludes bias / accumulate")
    if nsplit is None:
        nsplit = 1
        if bias is None:
            bm = 192 if op_a == 0 else 128
            tiles = ((M + bm - 1) // bm) * ((N + 127) // 128)
            nk = (K + 31) // 32
            if tiles < 64 and nk >= 8:
                nsplit = max(1, min(nk // 4, 512 // tiles))       # 512 = the persistent grid (2 workgroups x 256 CUs): one item each
            elif tiles < 256 and nk >= 16:
                # a quarter of the chip is busy already: split less (one item per CU, slices of >= 8 K-tiles) — measured on MI355X
                # (scripts/experiments/nsplit_sweep.py): 12288 x 128 x 384 31 -> 22 us unsplit, 12288 x 128 x 1152 (NN) 46 -> 40 us at 4
                nsplit = max(1, min(nk // 8, 256 // tiles))
    ws = None
    if nsplit > 1:
        ws = torch.empty((nsplit, M, N), device=A.device, dtype=torch.float32)
    h2 = resid is None and _fp16x2_shape(op_a, M, N, K, A, B, nsplit)
    if h2:
        amax_a = amax_a if amax_a is not None else amax_word(A)
        amax_b = amax_b if amax_b is not None else amax_word(B)
        h2 = amax_a is not None and amax_b is not None
    timer = prof_arm()
    if h2:
        trace("gemm_fp16x2")
        call("vnp_gemm_ex", op_a, op_b, M, N, K, ptr(A), _ld(A), ptr(B), _ld(B), ptr(out), _ld(out),
             ptr(bias), _ld(bias) if bias is not None else 0, rows_per_group, 1 if accumulate else 0, nsplit, ptr(ws), ptr(amax_a), ptr(amax_b))
    elif resid is not None:
        call("vnp_gemm_add", op_a, op_b, M, N, K, ptr(A), _ld(A), ptr(B), _ld(B), ptr(out), _ld(out), ptr(resid), _ld(resid), nsplit, ptr(ws))
    else:
        call("vnp_gemm", op_a, op_b, M, N, K, ptr(A), _ld(A), ptr(B), _ld(B), ptr(out), _ld(out),
             ptr(bias), _ld(bias) if bias is not None else 0, rows_per_group, 1 if accumulate else 0, nsplit, ptr(ws))
    if timer is not None:
        HK.gemm_profile.append(("NT" if (op_a, op_b) == (0, 0) else "NN" if (op_a, op_b) == (0, 1) else "TN" if (op_a, op_b) == (1, 1) else "TT",
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
    ok = (want_stats and SW.STATS_IN_EPILOGUE and _lib.get_gemm_mode() == "split" and R % 3072 == 0 and (2 * C) % 128 == 0 and K % 32 == 0
          and C % 64 == 0 and _ld(x) % 4 == 0 and x.data_ptr() % 16 == 0 and Wcat.is_contiguous() and Wcat.data_ptr() % 16 == 0
          and (bias is None or (rows_per_group > 0 and rows_per_group % 3 == 0)))
    if not ok:
        trace("pd_gemm_plain")
        gemm(x, Wcat, out=pd, bias=bias, rows_per_group=rows_per_group, nsplit=1)
        return None
    trace("pd_gemm_stats_epilogue")
    part = torch.empty((R // 3072, C, 2), device=x.device, dtype=torch.float32)
    ws = torch.empty(_lib.lib().vnp_gemm_vn_stats_ws_floats(R, C), device=x.device, dtype=torch.float32)
    am_x = am_w = None
    if _fp16x2_shape(0, R, 2 * C, K, x, Wcat, 1):
        am_x, am_w = amax_word(x), amax_word(Wcat)
    timer = prof_arm()
    if am_x is not None and am_w is not None:
        trace("gemm_fp16x2")
        call("vnp_gemm_vn_stats_ex", R, 2 * C, K, ptr(x), _ld(x), ptr(Wcat), K, ptr(pd), _ld(pd), ptr(bias),
             _ld(bias) if bias is not None else 0, rows_per_group, C, ptr(part), ptr(ws), ptr(am_x), ptr(am_w))
    else:
        call("vnp_gemm_vn_stats", R, 2 * C, K, ptr(x), _ld(x), ptr(Wcat), K, ptr(pd), _ld(pd), ptr(bias),
             _ld(bias) if bias is not None else 0, rows_per_group, C, ptr(part), ptr(ws))
    if timer is not None:
        HK.gemm_profile.append(("NT", 2.0 * R * 2 * C * K, 1, (R, 2 * C, K), timer))
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
    fuse_colsum = bool(rpg) and use_bn and has_d and SW.BN_BWD_TWO_PASS and rpg % 3 == 0 and (rpg // 3) % chunk == 0 and R % rpg == 0
    cs_part = torch.empty((nch, 3, 2 * C), device=dev, dtype=torch.float32) if fuse_colsum else None
    # two passes: the rank-1 (decoder tail) form saves a map (6 instead of 7); the matrix form is a wash on its own (measured) but lets
    # the final-gradient pass also produce the per-sample bias gradient instead of a separate 1.6 GB read (encoder second_conv[0])
    if use_bn and SW.BN_BWD_TWO_PASS and (go is None or fuse_colsum):
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




def _wgrad(gy, x, fine=False):
    """dW [Cout, Cin] = gy[R, Cout]^T x[R, Cin]  (TN GEMM, split over the R rows; the slices are summed in double,
    vnp_reduce_slabs_kernel).  ``fine``: the layer was folded onto its producer (WeightProductFn), so dW is multiplied on into a
    product that cancels (measured at config 2: the factor's relative error is amplified ~300x); a slice then accumulates at most
    SW.WGRAD_SLICE_TILES * 32 rows in fp32 on the matrix pipe, which keeps the gradient's rounding error at
    ~sqrt(2048) 2^-24 / sqrt(slices) of its terms however many rows it sums."""
    nsplit = None
    if fine:
        M, N, nk = gy.shape[1], x.shape[1], (gy.shape[0] + 31) // 32
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        auto = max(1, min(nk // 4, 512 // tiles)) if (tiles < 256 and nk >= 8) else 1
        nsplit = max(auto, min((nk + SW.WGRAD_SLICE_TILES - 1) // SW.WGRAD_SLICE_TILES, max(1, SW.WGRAD_SLAB_BYTES // (M * N * 4))))
    return gemm(gy, x, op_a=1, op_b=1, nsplit=nsplit)




def _dgrad(gy, W):
    """dX [R, Cin] = gy [R, Cout] W [Cout, Cin].  For the big activations W is transposed first (a few hundred KB) so that both operands
    are K-contiguous: the product then runs on the aligned NT kernels (the 192 x 256 tile of vnp_gemm_split_wide_kernel where Cin % 256
    == 0) instead of the kernel that transposes W's tile in registers at every K-step — the same LDS image, the same products in the
    same order, the same bits (tests/test_gpu_gemm.py), 3-10 % less time."""
    if gy.shape[0] >= SW.DGRAD_NT_ROWS and W.shape[1] % 128 == 0 and W.shape[0] % 32 == 0:
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


def cat_rows(*ws):
    """[W0; W1; ...] — as a VIEW when the matrices already sit one after the other in one buffer (the halves PackPdColsFn returns; a
    layer's weights inside optim.FlatBuffers, which honours ``pack_hint``), else one concatenation.  For use inside a Function's forward
    (no autograd through it)."""
    w0 = ws[0]
    ok = all(w.dim() == 2 and w.shape[1] == w0.shape[1] and w.is_contiguous() and w.dtype == w0.dtype and w.device == w0.device
             for w in ws)
    if ok:
        base, off = w0.untyped_storage().data_ptr(), w0.storage_offset()
        for w in ws:
            if w.untyped_storage().data_ptr() != base or w.storage_offset() != off:
                ok = False
                break
            off += w.numel()
    if ok:
        return torch.as_strided(w0, (sum(w.shape[0] for w in ws), w0.shape[1]), (w0.shape[1], 1), w0.storage_offset())
    return torch.cat(list(ws), dim=0)


def pack_hint(*params):
    """Ask optim.FlatBuffers to place these parameters one after the other (in this order) in its flat buffers, so that ``cat_rows`` of
    them is a view: a layer's [Wf; Wd], an attention's [Wq; Wk; Wv].  Only a hint: without FlatBuffers (or with a parameter frozen or
    of a size that needs padding) the concatenation is a copy, as before."""
    for a, b in zip(params[:-1], params[1:]):
        a._vnp_pack_next = b


class PackRowsFn(Function):
    """[W0; W1; ...] of parameters with autograd (an attention's projection weights for ONE product): a view when they are adjacent
    (``pack_hint`` + FlatBuffers), else a concatenation; the gradient comes back as the row blocks of one matrix."""

    @staticmethod
    def forward(ctx, *ws):
        ctx.rows = [w.shape[0] for w in ws]
        return cat_rows(*[w.detach() for w in ws])

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return tuple(torch.split(g, ctx.rows, dim=0))


class PackPdColsFn(Function):
    """Column ranges of a layer's two weights Wf, Wd [C, K], each as ONE [2C, n] matrix [Wf[:, a:b]; Wd[:, a:b]] (rows: feat then dir):
    a VNLinearLeakyReLU whose input is cat(per-sample channels, per-point channels) (models/pcn.py:172, 375-385) applies the first to the
    per-sample features in ONE product (the layer's per-sample bias [G*3, 2C], columns feat | dir) and the second to the points.
    ``halves[i]``: return range i as its two halves [C, n] (adjacent views of one buffer: cat_rows joins them again without a copy) for
    the Functions that take Wf and Wd separately.  One concatenation per range forward, two backward — against a copy per misaligned
    slice, two products instead of one, an addition of their input gradients and autograd's zero-padded slice gradients."""

    @staticmethod
    def forward(ctx, Wf, Wd, sizes, halves):
        if Wf.dim() != 2 or Wf.shape != Wd.shape or sum(sizes) != Wf.shape[1] or len(halves) != len(sizes):
            raise _lib.VnpError(f"pack_pd_cols: sizes {sizes} do not cover {tuple(Wf.shape)} / {tuple(Wd.shape)}")
        C = Wf.shape[0]
        ctx.cfg = (C, tuple(sizes), tuple(bool(h) for h in halves))
        outs, a = [], 0
        for n, hv in zip(sizes, halves):
            m = torch.cat([Wf[:, a:a + n], Wd[:, a:a + n]], dim=0)
            outs += [m[:C], m[C:]] if hv else [m]
            a += n
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        C, sizes, halves = ctx.cfg
        ref = next(g for g in gs if g is not None)
        pf, pd, i = [], [], 0
        for n, hv in zip(sizes, halves):
            if hv:
                gf, gd = gs[i], gs[i + 1]
                i += 2
            else:
                gf, gd = (gs[i][:C], gs[i][C:]) if gs[i] is not None else (None, None)
                i += 1
            pf.append(gf if gf is not None else ref.new_zeros((C, n)))
            pd.append(gd if gd is not None else ref.new_zeros((C, n)))
        return torch.cat(pf, dim=1), torch.cat(pd, dim=1), None, None


def pack_pd_cols(Wf, Wd, sizes, halves=None):
    return PackPdColsFn.apply(Wf, Wd, tuple(sizes), tuple(halves) if halves is not None else (False,) * len(sizes))


class WeightProductFn(Function):
    """(Wf W1, Wd W1) for weights Wf, Wd [C, K1] and W1 [K1, K0]: the layer that follows a VNLinear, folded onto that VNLinear's input
    channels (SW.ENC_LINEAR_REASSOCIATE).  One GEMM over [Wf; Wd] forward, two backward; all of them a few hundred MFLOP."""

    @staticmethod
    def forward(ctx, Wf, Wd, W1):
        Wcat = cat_rows(Wf, Wd)
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


