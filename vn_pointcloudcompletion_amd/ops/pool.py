"""VNMaxPool in its unfused, tap, GEMM-epilogue and screened forms, and mean_pool (/root/reference/models/vn_layers.py:153-171)."""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from .._lib import call, ptr
from ._state import HK, SW, POOL_PATH_COUNTS, prof_arm, trace
from .core import (BN_EPS, BN_MOMENTUM, _bn_leaky_backward, _bn_leaky_forward, _bn_statistics, _chunk_pc, _dgrad, _ld, _nchunk,
                   _per_sample_colsum, _rows, _wgrad, cat_rows, gemm, gemm_pd_stats, reduce_pairs, reduce_rows)

# If the screened search had to fall back (candidate list overflow, non-finite bounds) the bracketing pass was wasted work.  The flag of
# every call is copied to pinned host memory WITHOUT a sync and looked at when the next call of the same shape comes round: after a
# fallback the next SW.POOL_SCREEN_BACKOFF calls of that shape go straight to the full six-product epilogue, then the search is tried again.
_screen_state = {}        # shape key -> {"host": pinned int32[4], "event": Event or None, "skip": calls left on the full path}


def _screen_allowed(key):
    st = _screen_state.get(key)
    if st is None:
        return True
    if st["event"] is not None and st["event"].query():
        if int(st["host"][1]) != 0:
            st["skip"] = SW.POOL_SCREEN_BACKOFF
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


class VNMaxPoolFn(Function):
    """VNMaxPool (models/vn_layers.py:153-167) over G groups of gs consecutive points.  W_d receives no gradient
    (the arg-max detaches it in the reference as well)."""

    @staticmethod
    def forward(ctx, x, Wd, G, gs, x_prev=None, W_prev=None):
        """``x_prev``, ``W_prev``: when x = x_prev W_prev^T was produced by a VNLinear, the direction d = Wd x can be evaluated
        as (Wd W_prev) x_prev (contraction over the producer's input channels; see SW.POOL_DIRECTION_REASSOCIATE)."""
        x = _rows(x)
        C = x.shape[1]
        if x_prev is not None and SW.POOL_DIRECTION_REASSOCIATE:
            d = gemm(_rows(x_prev), gemm(Wd, W_prev, op_b=1), nsplit=1)
        else:
            d = gemm(x, Wd, nsplit=1)
        nch = _nchunk(gs)
        ws_val = torch.empty((G, nch, C), device=x.device, dtype=torch.float32)
        ws_idx = torch.empty((G, nch, C), device=x.device, dtype=torch.int32)
        idx = torch.empty((G, C), device=x.device, dtype=torch.int32)
        out = torch.empty((G * 3, C), device=x.device, dtype=torch.float32)
        call("vnp_vn_pool_fwd", ptr(x), _ld(x), ptr(d), _ld(d), G, gs, C, ptr(idx), ptr(out), C, ptr(ws_val), ptr(ws_idx))
        trace("pool_unfused", idx)
        ctx.save_for_backward(idx)
        ctx.cfg = (G, gs, C)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)                  # no zero-filled gradient for the index output
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


def sort_groups(idx, gs):
    """(pts, order) [G, C] int32: every group's arg-max indices ascending and the channels they came from, ties in channel order —
    vnp_sort_groups; groups of more than 4096 channels: torch.sort(stable=True)."""
    G, C = idx.shape
    if C > 4096 or gs * (1 << max(C - 1, 1).bit_length()) >= 2 ** 31:
        pts64, order64 = torch.sort(idx, dim=1, stable=True)
        return pts64.int().contiguous(), order64.int().contiguous()
    idx = idx.contiguous()
    pts, order = torch.empty_like(idx), torch.empty_like(idx)
    call("vnp_sort_groups", ptr(idx), G, C, gs, ptr(pts), ptr(order))
    return pts, order


def _forced_pool(idx, out, rows_of_y, G, gs):
    f = HK.pool_idx_force
    if f is None:
        return idx, out
    forced = f.pop(0).to(device=idx.device, dtype=torch.int32).contiguous()
    if forced.shape != idx.shape:
        raise _lib.VnpError(f"HK.pool_idx_force: index tensor {tuple(forced.shape)} offered to a pool of shape {tuple(idx.shape)}")
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
        if x_prev is not None and SW.POOL_DIRECTION_REASSOCIATE:
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
        trace("pool_tap", idx)
        ctx.save_for_backward(idx)
        ctx.cfg = (G, gs, C)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)                  # no zero-filled gradient for the index output
        return x, out, idx                      # x as-is: autograd hands back an alias whose gradient flows through this node

    @staticmethod
    @once_differentiable
    def backward(ctx, gx, gout, _gidx):
        (idx,) = ctx.saved_tensors
        G, gs, C = ctx.cfg
        if gout is None:                                  # the pooled output was not used: only the pass-through carries a gradient
            return gx, None, None, None, None, None
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
        return (SW.POOL_DIRECTION_REASSOCIATE and SW.POOL_FUSED_EPILOGUE and _lib.get_gemm_mode() == "split" and (G * gs * 3) % 192 == 0
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
        timer = prof_arm()
        call("vnp_linear_pool_store_fwd", ptr(x), _ld(x), ptr(Wcat), K, G, gs, C, K, ptr(y), C, ptr(idx), ptr(out), C, ptr(ws))
        if timer is not None:
            HK.gemm_profile.append(("NT", 2.0 * (G * gs * 3) * (2 * C) * K, 1, (G * gs * 3, 2 * C, K), timer))
        idx, out = _forced_pool(idx, out, lambda: y, G, gs)
        trace("linear_pool_tap_fused", idx)
        ctx.save_for_backward(x, W, idx)
        ctx.cfg = (G, gs, C)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)                  # no zero-filled gradient for the index output
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
        fused = (SW.POOL_DIRECTION_REASSOCIATE and SW.POOL_FUSED_EPILOGUE and _lib.get_gemm_mode() == "split" and (G * gs * 3) % 192 == 0
                 and gs % 32 == 0 and C % 64 == 0 and K % 32 == 0 and G <= 65535 and _ld(x) % 4 == 0 and x.data_ptr() % 16 == 0)
        screened = False
        if fused:
            # one GEMM over the interleaved [W ; Wd W] rows whose epilogue does the arg-max pool: y and d are never written
            Wp = gemm(Wd, W, op_b=1)
            Wcat = torch.stack([_rows(W).view(C // 32, 32, K), Wp.view(C // 32, 32, K)], dim=1).reshape(2 * C, K)
            skey = (x.device.index, G, gs, C, K)
            # the selection pass keeps a group's 32-point blocks in registers (gs <= 2048); back off after a fallback (see above)
            # (a short contraction — first_conv[1] -> maxpool1, K = 128 — is cheaper on the full epilogue than the search's fixed passes)
            screened = (SW.POOL_SCREENED and gs <= 2048 and (K >= SW.POOL_SCREEN_MIN_K or HK.pool_screen_stats is not None)
                        and (HK.pool_screen_stats is not None or _screen_allowed(skey)))
            entry = "vnp_linear_pool_screened_fwd" if screened else "vnp_linear_pool_fwd"
            nb = _lib.lib().vnp_linear_pool_screened_ws_bytes(G, gs, C, K) if screened else _lib.lib().vnp_linear_pool_ws_bytes(G, gs, C)
            ws = torch.empty((nb + 7) // 8, device=x.device, dtype=torch.int64)
            timer = prof_arm()
            POOL_PATH_COUNTS["screened" if screened else "full_epilogue"] += 1
            if screened:
                if SW.POOL_SCREEN_PASS not in ("fp16", "bf16x3"):
                    raise _lib.VnpError(f"POOL_SCREEN_PASS must be 'fp16' or 'bf16x3', got {SW.POOL_SCREEN_PASS!r}")
                if SW.POOL_SCREEN_TILE not in ("auto", "wide", "narrow"):
                    raise _lib.VnpError(f"POOL_SCREEN_TILE must be 'auto', 'wide' or 'narrow', got {SW.POOL_SCREEN_TILE!r}")
                flags = (1 if SW.POOL_SCREEN_PASS == "fp16" else 0) | {"auto": 0, "wide": 2, "narrow": 4}[SW.POOL_SCREEN_TILE]
                call(entry, ptr(x), _ld(x), ptr(Wcat), K, G, gs, C, K, ptr(idx), ptr(out), C, ptr(ws), flags)
            else:
                call(entry, ptr(x), _ld(x), ptr(Wcat), K, G, gs, C, K, ptr(idx), ptr(out), C, ptr(ws))
            if timer is not None:
                # flops = 2 M (2C) K: what the full epilogue executes six-fold; the screening pass (the timed kernel when `screened`)
                # executes ONE fp16 product of it (three bf16 products with VNP_POOL_SCREEN_PASS=bf16x3)
                HK.gemm_profile.append(("NT-pool-screened" if screened else "NT", 2.0 * (G * gs * 3) * (2 * C) * K, 1, (G * gs * 3, 2 * C, K), timer))
            if screened:
                if HK.pool_screen_stats is not None:
                    HK.pool_screen_stats.append(ws[:2].view(torch.int32)[:3].clone())
                else:
                    _screen_report(skey, ws)
        else:
            POOL_PATH_COUNTS["unfused"] += 1
            y = gemm(x, W, nsplit=1)
            if SW.POOL_DIRECTION_REASSOCIATE:
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
        trace(("linear_pool_screened" if screened else "linear_pool_fused_epilogue") if fused else "linear_pool_unfused", idx)
        ctx.save_for_backward(x, W, idx)
        ctx.cfg = (G, gs, C)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)                  # no zero-filled gradient for the index output
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
            pts, order = sort_groups(idx, gs)                           # the channels of a group by the point they picked (stable)
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


