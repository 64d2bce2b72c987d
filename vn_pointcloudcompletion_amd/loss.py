"""Training losses of the reference (/root/reference/metrics/loss.py) on the libvnpcc Chamfer / EMD operators."""
from __future__ import annotations

import torch

from .chamfer_distance import ChamferDistance, chamfer_3DDist

CD = ChamferDistance()


def cd_loss_L1(pcs1, pcs2):
    """metrics/loss.py:20-31: (mean sqrt(dist1) + mean sqrt(dist2)) / 2 over ALL B*n elements."""
    from .ops import ChamferL1LossFn
    return ChamferL1LossFn.apply(pcs1, pcs2)            # the search + the sqrt / mean / add / div chain and its backward in one Function


def cd_loss_L2(pcs1, pcs2):
    """metrics/loss.py:34-43."""
    dist1, dist2 = CD(pcs1, pcs2)
    return torch.mean(dist1) + torch.mean(dist2)


def emd_loss(pcs1, pcs2):
    """metrics/loss.py:46-55."""
    from .emd import EarthMoverDistance
    return torch.mean(EarthMoverDistance()(pcs1, pcs2))


def _sample_means(d_a, d_b):
    """Per-sample means of two one-sided squared distances and of their square roots."""
    return d_a.mean(dim=1), d_b.mean(dim=1), d_a.sqrt().mean(dim=1), d_b.sqrt().mean(dim=1)


def calc_cd(output, gt, calc_f1=False, return_raw=False, normalize=False, separate=False):
    """metrics/loss.py:58-75: per-sample Chamfer terms [cd_p (L1 form), cd_t (L2 form)] (+ F-score@1e-4, + raw outputs).
    The operator is called as (gt, output), so dist1/idx1 belong to the ground-truth points, as in the reference."""
    d_gt, d_out, i_gt, i_out = chamfer_3DDist()(gt, output)
    m_gt, m_out, r_gt, r_out = _sample_means(d_gt, d_out)
    res = [torch.stack([r_gt, r_out]), torch.stack([m_gt, m_out])] if separate else [(r_gt + r_out) / 2, m_gt + m_out]
    if calc_f1:
        from .metric import fscore
        res.append(fscore(d_gt, d_out, 0.0001)[0])
    if return_raw:
        res += [d_gt, d_out, i_gt, i_out]
    return res


class _DensitySideFn(torch.autograd.Function):
    """One direction of the density-aware term on the device (vnp_dcd_side): histogram of the arg-min indices, the weighted
    exponential and its per-sample mean in one call; the gradient reaches ``dist`` only (the weight is detached in the reference)."""

    @staticmethod
    def forward(ctx, dist, idx, n_targets, alpha, n_lambda, frac):
        from ._lib import call, ptr
        dist = dist.contiguous()
        idx = idx.to(torch.int32).contiguous()
        B, N = dist.shape
        hits = torch.empty((B, n_targets), device=dist.device, dtype=torch.int32)
        loss = torch.empty((B,), device=dist.device, dtype=torch.float32)
        gfac = torch.empty_like(dist) if ctx.needs_input_grad[0] else None
        call("vnp_dcd_side", B, N, n_targets, ptr(dist), ptr(idx), float(alpha), float(n_lambda), float(frac), ptr(hits), ptr(loss), ptr(gfac))
        if gfac is not None:
            ctx.save_for_backward(gfac)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (gfac,) = ctx.saved_tensors
        return gloss.unsqueeze(1) * gfac, None, None, None, None, None


def _density_side(dist, idx, n_targets, alpha, n_lambda, frac):
    """One direction of the density-aware term: every query is weighted by 1 / (number of queries that share its nearest
    neighbour)^lambda (a batched histogram of the arg-min indices; the reference loops over the batch with bincount)."""
    return _DensitySideFn.apply(dist.float(), idx, n_targets, alpha, n_lambda, frac)


def calc_dcd(x, gt, alpha=1000, n_lambda=1, return_raw=False, non_reg=False):
    """Density-aware Chamfer distance, metrics/loss.py:77-118.  Returns [dcd per sample, cd_p, cd_t] (+ raw outputs)."""
    x, gt = x.float(), gt.float()
    if x.shape[0] != gt.shape[0]:
        raise AssertionError("calc_dcd: batch sizes differ")
    n_x, n_gt = x.shape[1], gt.shape[1]
    ratio_x, ratio_gt = n_x / n_gt, n_gt / n_x
    if non_reg:
        ratio_x, ratio_gt = max(1, ratio_x), max(1, ratio_gt)
    cd_p, cd_t, d_gt, d_out, i_gt, i_out = calc_cd(x, gt, return_raw=True)
    # d_gt / i_gt: each ground-truth point's nearest prediction (indices into x); d_out / i_out: the converse
    side_gt = _density_side(d_gt, i_gt, n_x, alpha, n_lambda, ratio_gt)
    side_out = _density_side(d_out, i_out, n_gt, alpha, n_lambda, ratio_x)
    res = [(side_gt + side_out) / 2, cd_p, cd_t]
    if return_raw:
        res += [d_gt, d_out, i_gt, i_out]
    return res
