"""Chamfer operator surface of the reference on libvnpcc:
``ChamferDistance()(a, b) -> (dist1, dist2)`` (extensions/chamfer_distance/chamfer_distance.py:74-84) and
``chamfer_3DDist()(a, b) -> (dist1, dist2, idx1, idx2)`` (extensions/ChamferDistancePytorch/chamfer3D/dist_chamfer_3D.py:67-74).
GPU tensors only, like the reference (chamfer_distance.py:28)."""
from __future__ import annotations

from torch import nn

from .ops import ChamferFn

chamfer_3DFunction = ChamferFn


class ChamferDistance(nn.Module):
    def forward(self, input1, input2):
        dist1, dist2, _, _ = ChamferFn.apply(input1, input2)
        return dist1, dist2


class chamfer_3DDist(nn.Module):
    def forward(self, input1, input2):
        return ChamferFn.apply(input1.contiguous(), input2.contiguous())
