"""Layers behind VN_PointNet's single-channel first layer on per-point scalars: every VN vector there is parallel to its point's
unit direction (/root/reference/models/pcn.py:165-173)."""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from .._lib import call, ptr
from ._state import HK, SW, POOL_PATH_COUNTS, prof_arm, trace
from .core import (BN_EPS, BN_MOMENTUM, _bn_leaky_backward, _bn_leaky_forward, _bn_statistics, _chunk_pc, _dgrad, _ld, _nchunk,
                   _per_sample_colsum, _rows, _wgrad, cat_rows, gemm, gemm_pd_stats, reduce_pairs, reduce_rows)
from .pool import _forced_pool, sort_groups


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
        trace("dir_pool", idx)
        ctx.save_for_backward(h, u, W, idx)
        ctx.cfg = (G, gs, C)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)                  # no zero-filled gradient for the index output
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
            pts, order = sort_groups(idx, gs)                           # the channels of a group by the point they picked (stable)
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
        Wcat = cat_rows(Wf, Wd)
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
        trace("dir_layer_scalar_out" if scalar_out else "dir_layer")
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


