"""Training losses of the reference (/root/reference/metrics/loss.py) on the libvnpcc Chamfer / EMD operators."""
from __future__ import annotations

import torch

from .chamfer_distance import ChamferDistance, chamfer_3DDist

CD = ChamferDistance()


def cd_loss_L1(pcs1, pcs2):
    """metrics/loss.py:20-31: (mean sqrt(dist1) + mean sqrt(dist2)) / 2 over ALL B*n elements."""
    dist1, dist2 = CD(pcs1, pcs2)
    return (torch.mean(torch.sqrt(dist1)) + torch.mean(torch.sqrt(dist2))) / 2.0


def cd_loss_L2(pcs1, pcs2):
    """metrics/loss.py:34-43."""
    dist1, dist2 = CD(pcs1, pcs2)
    return torch.mean(dist1) + torch.mean(dist2)


def emd_loss(pcs1, pcs2):
    """metrics/loss.py:46-55."""
    from .emd import EarthMoverDistance
    return torch.mean(EarthMoverDistance()(pcs1, pcs2))


def calc_cd(output, gt, calc_f1=False, return_raw=False, normalize=False, separate=False):
    """metrics/loss.py:58-75."""
    dist1, dist2, idx1, idx2 = chamfer_3DDist()(gt, output)
    cd_p = (torch.sqrt(dist1).mean(1) + torch.sqrt(dist2).mean(1)) / 2
    cd_t = dist1.mean(1) + dist2.mean(1)
    if separate:
        res = [torch.cat([torch.sqrt(dist1).mean(1).unsqueeze(0), torch.sqrt(dist2).mean(1).unsqueeze(0)]),
               torch.cat([dist1.mean(1).unsqueeze(0), dist2.mean(1).unsqueeze(0)])]
    else:
        res = [cd_p, cd_t]
    if calc_f1:
        from .metric import fscore
        f1, _, _ = fscore(dist1, dist2, 0.0001)
        res.append(f1)
    if return_raw:
        res.extend([dist1, dist2, idx1, idx2])
    return res


def calc_dcd(x, gt, alpha=1000, n_lambda=1, return_raw=False, non_reg=False):
    """Density-aware Chamfer distance, metrics/loss.py:77-118."""
    x = x.float()
    gt = gt.float()
    batch_size, n_x, _ = x.shape
    _, n_gt, _ = gt.shape
    assert x.shape[0] == gt.shape[0]
    if non_reg:
        frac_12 = max(1, n_x / n_gt)
        frac_21 = max(1, n_gt / n_x)
    else:
        frac_12 = n_x / n_gt
        frac_21 = n_gt / n_x
    cd_p, cd_t, dist1, dist2, idx1, idx2 = calc_cd(x, gt, return_raw=True)
    exp_dist1, exp_dist2 = torch.exp(-dist1 * alpha), torch.exp(-dist2 * alpha)
    # per-sample histogram of the arg-min indices, batched (the reference loops over the batch in Python)
    i1, i2 = idx1.long(), idx2.long()
    count1 = torch.zeros((batch_size, n_x), device=x.device, dtype=torch.float32).scatter_add_(
        1, i1, torch.ones_like(i1, dtype=torch.float32))
    count2 = torch.zeros((batch_size, n_gt), device=x.device, dtype=torch.float32).scatter_add_(
        1, i2, torch.ones_like(i2, dtype=torch.float32))
    weight1 = (count1.gather(1, i1).detach() ** n_lambda + 1e-6) ** (-1) * frac_21
    weight2 = (count2.gather(1, i2).detach() ** n_lambda + 1e-6) ** (-1) * frac_12
    loss1 = (-exp_dist1 * weight1 + 1.).mean(1)
    loss2 = (-exp_dist2 * weight2 + 1.).mean(1)
    loss = (loss1 + loss2) / 2
    res = [loss, cd_p, cd_t]
    if return_raw:
        res.extend([dist1, dist2, idx1, idx2])
    return res
