"""FoldingNet layer 0 (rank-2 expansion of the de-duplicated broadcast input), the folding stages of Attention_VN_FoldingNet and
the final VNLinear(C -> 1) + residual (/root/reference/models/pcn.py:335-346, 375-387, 479-492)."""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from .._lib import call, ptr
from ._state import HK, SW, POOL_PATH_COUNTS, prof_arm, trace
from .core import (BN_EPS, BN_MOMENTUM, _bn_leaky_backward, _bn_leaky_forward, _bn_statistics, _chunk_pc, _dgrad, _ld, _nchunk,
                   _per_sample_colsum, _rows, _wgrad, cat_rows, gemm, gemm_pd_stats, reduce_pairs, reduce_rows)


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
        trace("fold_layer0_unfused")
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
        trace("fold_layer0_fused")
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
        ncb = (C + 63) // 64
        gc_part = torch.empty((ncb, B * Mc * 3), device=dev, dtype=torch.float32)       # dL/dcoarse: one partial per 64-column block
        call("vnp_fold_fused", 3, *args, *bn, None, 0, ptr(go), _ld(go), None, ptr(Ssum) if training else None, P,
             ptr(part5), ptr(gc_part), cpc)
        gcoarse = reduce_rows(gc_part, 1, ncb, B * Mc * 3).view(B, Mc, 3) if ctx.needs_input_grad[4] else None
        W = 2 * C
        per_sample = reduce_rows(part5, B, Mc // cpc, 5 * W)                           # [B][5W]: a chunk lies inside one sample
        gbias = per_sample[:, :3 * W].reshape(B * 3, W)
        tot = reduce_rows(per_sample, 1, B, 5 * W)
        gws, gwc = tot[0, 3 * W:4 * W], tot[0, 4 * W:]
        return (gbias, gws, gwc, None, gcoarse, planar[1], planar[0], None, None, None, None)




def group_fold_ok(x_cols, S, C):
    """Can GroupFoldBnLeakyFn take a VNLinearLeakyReLU whose per-point input has ``x_cols`` channels, S points per group and C outputs?"""
    return SW.GROUP_FOLD and x_cols == 1 and S % 2 == 0 and 2 <= S <= 16 and S <= _lib.chunk_points()


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
        trace("group_fold_fused")
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


