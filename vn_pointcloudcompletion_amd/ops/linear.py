"""VNLinear, VNLinearLeakyReLU, VNBatchNorm, VNLeakyReLU and the fused decoder tail on the rows layout
(/root/reference/models/vn_layers.py:12-127, models/pcn.py:341-346)."""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from .._lib import call, ptr
from ._state import HK, SW, POOL_PATH_COUNTS, prof_arm, trace
from .core import (BN_EPS, BN_MOMENTUM, _bn_leaky_backward, _bn_leaky_forward, _bn_statistics, _chunk_pc, _dgrad, _ld, _nchunk,
                   _per_sample_colsum, _rows, _wgrad, cat_rows, gemm, gemm_pd_stats, reduce_pairs, reduce_rows)


class VNLinearFn(Function):
    """y = x W^T (+ per-sample bias | + residual)  —  VNLinear, models/vn_layers.py:12-22.  ``residual`` [R, Cout]: the tensor the
    layer's output is added to right after it (a transformer block's skip connection, models/pointr/vn_pointr.py:401-410), added in the
    GEMM's epilogue; its gradient is the output gradient itself."""

    @staticmethod
    def forward(ctx, x, W, bias=None, rows_per_group=0, residual=None):
        x = _rows(x)
        y = gemm(x, W, bias=bias, rows_per_group=rows_per_group, resid=residual)
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
        return gx, gW, gb, None, (gy if len(ctx.needs_input_grad) > 4 and ctx.needs_input_grad[4] else None)


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
        Wcat = cat_rows(Wf, Wd)
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
        Wcat = cat_rows(Wf, Wd)
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
        trace("decoder_tail_fused")
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




class VNLeakyDirFn(Function):
    """VNLeakyReLU (models/vn_layers.py:25-43) with its direction map inside: d = x Wd^T, out = leaky(x, d).  x reaches the output twice
    (directly and through d); its two gradient contributions are joined in the epilogue of the direction map's input-gradient product
    (vnp_gemm_add) instead of by an addition of autograd's."""

    @staticmethod
    def forward(ctx, x, Wd, ns):
        x = _rows(x)
        C = x.shape[1]
        if Wd.shape != (C, C):
            raise _lib.VnpError("VNLeakyDirFn: a [C, C] direction map expected")
        d = gemm(x, Wd)
        out, _, _ = _bn_leaky_forward(x, d, C, None, None, None, None, False, ns, use_bn=False)
        ctx.save_for_backward(x, d, Wd)
        ctx.ns = ns
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        x, d, Wd = ctx.saved_tensors
        C = x.shape[1]
        gpd, _, _ = _bn_leaky_backward(go, x, d, C, None, None, None, None, False, ctx.ns, use_bn=False)
        gxl, gd = gpd[:, :C], gpd[:, C:]
        gx = gemm(gd, Wd, op_b=1, resid=gxl) if ctx.needs_input_grad[0] else None        # dL/dx = dL/dx|leaky + gd Wd
        gW = _wgrad(gd, x) if ctx.needs_input_grad[1] else None
        return gx, gW, None
