"""Evaluation metrics of the reference (/root/reference/metrics/metric.py) on the libvnpcc Chamfer operator."""
from __future__ import annotations

import torch

from .chamfer_distance import ChamferDistance

CD = ChamferDistance()


def l2_cd(pcs1, pcs2):
    """metrics/metric.py:12-16 (sum over the batch)."""
    dist1, dist2 = CD(pcs1, pcs2)
    return torch.sum(torch.mean(dist1, dim=1) + torch.mean(dist2, dim=1))


def l1_cd(pcs1, pcs2):
    """metrics/metric.py:19-23 (sum over the batch)."""
    dist1, dist2 = CD(pcs1, pcs2)
    return torch.sum(torch.mean(torch.sqrt(dist1), 1) + torch.mean(torch.sqrt(dist2), 1)) / 2


def emd(pcs1, pcs2):
    """metrics/metric.py:26-28."""
    from .emd import EarthMoverDistance
    return torch.sum(EarthMoverDistance()(pcs1, pcs2))


def fscore(dist1, dist2, threshold=0.001):
    """extensions/ChamferDistancePytorch/fscore.py:3-16: F-score from squared Chamfer distances."""
    precision_1 = torch.mean((dist1 < threshold).float(), dim=1)
    precision_2 = torch.mean((dist2 < threshold).float(), dim=1)
    f = 2 * precision_1 * precision_2 / (precision_1 + precision_2)
    f[torch.isnan(f)] = 0
    return f, precision_1, precision_2


def f_score(pred, gt, th=0.01):
    """metrics/metric.py:31-48 computed on device from the Chamfer distances instead of open3d KD-trees:
    a point counts when its nearest-neighbour distance is < th (distances here are squared, hence th^2).
    Accepts [N,3] numpy arrays / tensors like the reference and returns a Python float."""
    p = torch.as_tensor(pred, dtype=torch.float32)
    g = torch.as_tensor(gt, dtype=torch.float32)
    dev = p.device if p.is_cuda else (g.device if g.is_cuda else torch.device("cuda"))
    d1, d2 = CD(p.to(dev).unsqueeze(0), g.to(dev).unsqueeze(0))
    precision = (d1 < th * th).float().mean().item()
    recall = (d2 < th * th).float().mean().item()
    return 2 * recall * precision / (recall + precision) if recall + precision else 0
