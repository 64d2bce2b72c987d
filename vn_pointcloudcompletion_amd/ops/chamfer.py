"""chamfer_3DFunction on libvnpcc (/root/reference/extensions/chamfer_distance/chamfer_distance.py:29-71)."""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from .._lib import call, ptr
from ._state import HK, SW, POOL_PATH_COUNTS, prof_arm, trace


# both clouds at least this large -> vnp_chamfer_fwd_pruned (below it the brute-force kernel's slicing fills the chip better);
# VNP_CHAMFER_BRUTE=1 forces the brute-force kernel everywhere
# bench.py sets this to a list to receive, per pruned call, the device word with the number of (query, target) pairs the two scans
# evaluated (int64 [1]; a full scan evaluates 2 B n m)


def _chamfer_forward(xyz1, xyz2):
    """(dist1, dist2, idx1, idx2) of contiguous fp32 device clouds [B, n, 3], [B, m, 3]: the pruned exact search for dense clouds, the
    brute-force kernel otherwise (same bits; tests/test_gpu_chamfer.py)."""
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dev = xyz1.device
    dist1 = torch.empty((B, n), device=dev, dtype=torch.float32)
    dist2 = torch.empty((B, m), device=dev, dtype=torch.float32)
    idx1 = torch.empty((B, n), device=dev, dtype=torch.int32)
    idx2 = torch.empty((B, m), device=dev, dtype=torch.int32)
    nbytes = _lib.lib().vnp_chamfer_pruned_ws_bytes(B, n, m) if min(n, m) >= SW.CHAMFER_PRUNED_MIN_POINTS else 0
    if nbytes > 0:
        ws = torch.empty((nbytes + 7) // 8, device=dev, dtype=torch.int64)
        call("vnp_chamfer_fwd_pruned", B, n, ptr(xyz1), m, ptr(xyz2), ptr(dist1), ptr(idx1), ptr(dist2), ptr(idx2), ptr(ws))
        trace("chamfer_pruned")
        if HK.chamfer_stats is not None:
            HK.chamfer_stats.append(ws[1:2].clone())
            off = _lib.lib().vnp_chamfer_pruned_ws_wave_counts(B, n, m)
            nw = B * ((n + 63) // 64 + (m + 63) // 64)
            HK.chamfer_stats.append(ws.view(torch.int32)[off // 4: off // 4 + nw].clone())      # chunks scanned per query wave
    else:
        trace("chamfer_brute")
        ws = torch.empty(B * max(n, m), device=dev, dtype=torch.int64)
        call("vnp_chamfer_fwd", B, n, ptr(xyz1), m, ptr(xyz2), ptr(dist1), ptr(idx1), ptr(dist2), ptr(idx2), ptr(ws))
    return dist1, dist2, idx1, idx2


class ChamferFn(Function):
    """chamfer_3DFunction (extensions/chamfer_distance/chamfer_distance.py:29-71) on libvnpcc."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        _lib.require_gpu(xyz1, xyz2)
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        dist1, dist2, idx1, idx2 = _chamfer_forward(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        ctx.set_materialize_grads(False)                  # no zero-filled "gradients" for the two index outputs (two launches per call)
        return dist1, dist2, idx1, idx2

    @staticmethod
    @once_differentiable
    def backward(ctx, g1, g2, _gi1, _gi2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        if g1 is None and g2 is None:
            return None, None
        B, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1 = g1.contiguous() if g1 is not None else torch.zeros((B, n), device=xyz1.device, dtype=torch.float32)
        g2 = g2.contiguous() if g2 is not None else torch.zeros((B, m), device=xyz1.device, dtype=torch.float32)
        gx1 = torch.empty_like(xyz1)
        gx2 = torch.empty_like(xyz2)
        call("vnp_chamfer_bwd", B, n, ptr(xyz1), m, ptr(xyz2), ptr(g1), ptr(idx1), ptr(g2), ptr(idx2), ptr(gx1), ptr(gx2))
        return gx1, gx2


class ChamferL1LossFn(Function):
    """cd_loss_L1 (metrics/loss.py:20-31) = (mean sqrt(dist1) + mean sqrt(dist2)) / 2 of the Chamfer operator in one Function: the search
    as in ChamferFn, the loss by vnp_cd_l1_loss (two small launches, double sums in a fixed order), and backward the distance gradients
    ((g / 2) / n) / (2 sqrt(dist)) by vnp_cd_l1_grad straight into vnp_chamfer_bwd — instead of nine ATen launches per distance array and
    step.  Returns a 0-dim tensor."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        _lib.require_gpu(xyz1, xyz2)
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        dist1, dist2, idx1, idx2 = _chamfer_forward(xyz1, xyz2)
        part = torch.empty(512, device=xyz1.device, dtype=torch.float64)
        out = torch.empty(3, device=xyz1.device, dtype=torch.float32)
        call("vnp_cd_l1_loss", ptr(dist1), dist1.numel(), ptr(dist2), dist2.numel(), ptr(part), ptr(out))
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2, dist1, dist2)
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xyz1, xyz2, idx1, idx2, dist1, dist2 = ctx.saved_tensors
        B, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g = g.to(device=xyz1.device, dtype=torch.float32).contiguous()
        gd1, gd2 = torch.empty_like(dist1), torch.empty_like(dist2)
        call("vnp_cd_l1_grad", ptr(g), ptr(dist1), dist1.numel(), ptr(dist2), dist2.numel(), ptr(gd1), ptr(gd2))
        gx1 = torch.empty_like(xyz1)
        gx2 = torch.empty_like(xyz2)
        call("vnp_chamfer_bwd", B, n, ptr(xyz1), m, ptr(xyz2), ptr(gd1), ptr(idx1), ptr(gd2), ptr(idx2), ptr(gx1), ptr(gx2))
        return gx1, gx2
