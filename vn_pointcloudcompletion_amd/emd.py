"""Earth Mover's Distance operator surface of the reference (extensions/earth_movers_distance/emd.py:6-43) on libvnpcc."""
from __future__ import annotations

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import call, ptr


class EarthMoverDistanceFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        _lib.require_gpu(xyz1, xyz2)
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        B, n, c1 = xyz1.shape
        B2, m, c2 = xyz2.shape
        if B2 != B or c1 != 3 or c2 != 3:                       # CHECK_EQ of emd_kernel.cu:178-180
            raise _lib.VnpError(f"EMD: shapes {tuple(xyz1.shape)} / {tuple(xyz2.shape)}")
        dev = xyz1.device
        match = torch.empty((B, m, n), device=dev, dtype=torch.float32)
        temp = torch.empty((B, (n + m) * 2), device=dev, dtype=torch.float32)
        call("vnp_emd_approxmatch", B, n, m, ptr(xyz1), ptr(xyz2), ptr(match), ptr(temp))
        cost = torch.empty(B, device=dev, dtype=torch.float32)
        ws = torch.empty(((m + 63) // 64, B), device=dev, dtype=torch.float32)
        call("vnp_emd_matchcost", B, n, m, ptr(xyz1), ptr(xyz2), ptr(match), ptr(cost), ptr(ws))
        ctx.save_for_backward(xyz1, xyz2, match)
        return cost

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_cost):
        xyz1, xyz2, match = ctx.saved_tensors               # match is a constant in the backward (emd.py:17-22)
        B, n, _ = xyz1.shape
        m = xyz2.shape[1]
        grad_cost = grad_cost.contiguous().float()
        g1, g2 = torch.empty_like(xyz1), torch.empty_like(xyz2)
        call("vnp_emd_matchcost_bwd", B, n, m, ptr(grad_cost), ptr(xyz1), ptr(xyz2), ptr(match), ptr(g1), ptr(g2))
        return g1, g2


class EarthMoverDistance(nn.Module):
    def forward(self, xyz1, xyz2):
        if xyz1.dim() == 2:
            xyz1 = xyz1.unsqueeze(0)
        if xyz2.dim() == 2:
            xyz2 = xyz2.unsqueeze(0)
        return EarthMoverDistanceFunction.apply(xyz1, xyz2)
