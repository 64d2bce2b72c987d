"""Switches and instrumentation hooks of the ops package — the only mutable module state next to the screened pool's back-off table.

``SW`` holds the choices between two PRODUCT paths (each path serves some shapes, so both stay compiled and tested).  Four of them can
be set from the environment because they change the evaluation order or the convention of the arithmetic (documented in DESIGN.md §4);
the others are attributes only: tests flip them to hold a fused kernel against the general path it replaces.

``HK`` holds what tests and ``bench.py`` attach to a run (path trace, teacher-forced discrete choices, kernel timers, diagnostics);
it is only ever set through the ``hooks(...)`` context manager, which restores the previous values on exit, and it is all-None on the
product path.
"""
from __future__ import annotations

import contextlib
import os

from .. import _lib


class _Switches:
    # --- settable from the environment (semantics of the arithmetic) ----------------------------------------------------------
    # VNMaxPool direction d = Wd (W x) evaluated as (Wd W) x when the pooled tensor is the output of a VNLinear;
    # VNP_POOL_AS_WRITTEN=1 keeps the reference's evaluation order
    POOL_DIRECTION_REASSOCIATE = os.environ.get("VNP_POOL_AS_WRITTEN", "0") != "1"
    # encoder second_conv[0] folded onto first_conv[1]'s 128 input channels, W2 (W1 a) = (W2 W1) a (models/pcn.py:168-173);
    # VNP_ENC_AS_WRITTEN=1 keeps the reference's evaluation order
    ENC_LINEAR_REASSOCIATE = os.environ.get("VNP_ENC_AS_WRITTEN", "0") != "1"
    # exact screened arg-max for VNLinear -> VNMaxPool (same bits as the full six-product epilogue); VNP_POOL_SCREEN=0 opts out
    POOL_SCREENED = os.environ.get("VNP_POOL_SCREEN", "1") != "0"
    # pointnet2_ops' furthest_point_sampling_kernel skips candidates with |p|^2 <= 1e-3; VNP_FPS_PLAIN=1 selects plain FPS
    # (PARITY UNPINNED either way: third-party source absent)
    FPS_SKIP_NEAR_ORIGIN = os.environ.get("VNP_FPS_PLAIN", "0") != "1"
    # both clouds at least this large -> pruned exact Chamfer search; VNP_CHAMFER_BRUTE=1 forces the brute-force kernel everywhere
    CHAMFER_PRUNED_MIN_POINTS = (1 << 30) if os.environ.get("VNP_CHAMFER_BRUTE") == "1" else 4096

    # --- attributes only: fused kernel vs. the general path that serves the shapes it does not take -----------------------------
    EDGE_FUSED = True               # EdgeConv without its edge tensor (vnp_edge_fused) vs gather-add -> BatchNorm / leaky -> mean pool
    EDGE_RANK1 = True               # ... and, for a one-channel input (the coordinates), without its per-point matrices either (EdgeConvRank1Fn)
    STATS_IN_EPILOGUE = True        # BatchNorm statistics from the GEMM's accumulators vs a vnp_vn_norm_stats pass
    POOL_FUSED_EPILOGUE = True      # VNLinear -> VNMaxPool in the GEMM epilogue vs two GEMMs + pool kernels
    POOL_SCREEN_PASS = "fp16"       # bracketing pass of the screened pool: one fp16 product, or "bf16x3"
    POOL_SCREEN_TILE = "auto"       # its tile: "auto" (384 x 256 / 192 x 256 / 192 x 128 by shape), "wide" (no 384-row tile), "narrow"
    POOL_SCREEN_MIN_K = 256         # shorter contractions: the full epilogue costs less than the search's fixed passes
    POOL_SCREEN_BACKOFF = 64        # calls of a shape that skip the search after a device-side fallback
    BN_BWD_TWO_PASS = True          # BatchNorm + leaky backward as sums pass + final-gradient pass vs bwd1 -> reduce -> bwd2
    ENC_DIR_LAYER = True            # layers behind VN_PointNet's single-channel first layer on per-point scalars
    ENC_FIRST_LAYER_SCALAR = True   # ... and that first layer itself with a scalar output
    GROUP_FOLD = True               # folding stages' first layer without its pre-activation tensor (vnp_group_fold_fused)
    ATTN_TOKEN_MAJOR = False        # force the head-dimension-on-N/K layout that small attention problems take anyway
    ATTN_FM_MIN_FLOP = 8e9          # per attention product (2 B H N^2 Dh): below this the extra permutes cost more than they save
    GEMM_FP16X2 = False             # two fp16 planes / three products for large aligned GEMMs (vnp_gemm_ex): 22-23 operand bits, opt-in
    GEMM_FP16X2_MIN_FLOP = 2.0e10
    DGRAD_NT_ROWS = 49152           # activations of at least this many rows: input gradient in NT form on the wide tile
    WGRAD_SLICE_TILES = 64          # K-tiles (of 32 rows) per split-K slice of a weight gradient that is multiplied on
    WGRAD_SLAB_BYTES = 128 << 20    # ... as long as the slabs stay under this
    FPS_AHEAD = True                # an encoder's furthest-point samplings on a side HIP stream beside its kNN / EdgeConv launches (FpsPlan)
    REV_AHEAD = True                # a graph's reverse adjacency (needed by the backward pass only) built on the side stream during the forward


SW = _Switches()


class _Hooks:
    trace = None                # dict: code path -> launch count; "pool_idx" / "knn_idx" / "fps_idx": the discrete choices in call order
    pool_idx_force = None       # list of arg-max index tensors [G, C], consumed in call order by every VNMaxPool-type forward
    knn_force = None            # list of neighbour tables [B, Nq, k] (local ids), consumed in call order by knn_indices
    gemm_profile = None         # list receiving (variant, flops, nsplit, (M, N, K), _lib.KernelTimer) per matrix-product launch
    pool_screen_stats = None    # list receiving the screened pool's diagnostics (int32 [3]: candidates, fallback flag, capacity)
    chamfer_stats = None        # list receiving, per pruned call, the int64 word counting the pairs the scans evaluated


HK = _Hooks()


@contextlib.contextmanager
def hooks(**kw):
    """``with ops.hooks(trace={}, pool_idx_force=[...]) as hk:`` — attach instrumentation / teacher forcing for the duration of the
    block.  Never used on the product path."""
    for k in kw:
        if k.startswith("_") or not hasattr(_Hooks, k):
            raise TypeError(f"ops.hooks: unknown hook {k!r}")
    old = {k: getattr(HK, k) for k in kw}
    for k, v in kw.items():
        setattr(HK, k, v)
    try:
        yield HK
    finally:
        for k, v in old.items():
            setattr(HK, k, v)


@contextlib.contextmanager
def switches(**kw):
    """``with ops.switches(GROUP_FOLD=False):`` — flip product-path switches for the duration of the block (tests)."""
    for k in kw:
        if k.startswith("_") or not hasattr(_Switches, k):
            raise TypeError(f"ops.switches: unknown switch {k!r}")
    old = {k: getattr(SW, k) for k in kw}
    for k, v in kw.items():
        setattr(SW, k, v)
    try:
        yield SW
    finally:
        for k, v in old.items():
            setattr(SW, k, v)


def trace(name, idx=None, key="pool_idx"):
    tr = HK.trace
    if tr is not None:
        tr[name] = tr.get(name, 0) + 1
        if idx is not None:
            tr.setdefault(key, []).append(idx)


_TIMER_POOL = []


def reserve_timers(n):
    """Create ``n`` KernelTimers (two HIP events each) ahead of an instrumented region, so that arming one inside it is a list pop and one
    library call instead of two hipEventCreate round trips through ctypes per matrix-product launch (bench.py: before its timed steps)."""
    while len(_TIMER_POOL) < n:
        _TIMER_POOL.append(_lib.KernelTimer())


def prof_arm():
    """A KernelTimer armed for the next matrix-product launch when a profile list is attached, else None."""
    if HK.gemm_profile is None:
        return None
    t = _TIMER_POOL.pop() if _TIMER_POOL else _lib.KernelTimer()
    t.arm()
    return t


# calls of the VNLinear -> VNMaxPool pool by path since import (bench.py reports the timed steps' share): the exact screened search,
# the full six-product epilogue (search skipped: short contraction, back-off after a fallback), and device-side fallbacks seen
POOL_PATH_COUNTS = {"screened": 0, "full_epilogue": 0, "unfused": 0, "fallback_flags_seen": 0}
