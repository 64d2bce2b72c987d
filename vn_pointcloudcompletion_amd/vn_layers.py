"""Vector-Neuron layer classes with the reference's names, constructor signatures and ``state_dict`` keys
(/root/reference/models/vn_layers.py), running on the HIP kernels of libvnpcc.

Every class has two entry points:
  * ``forward(x)`` takes / returns the reference layout ``[B, C, 3, N]`` or ``[B, C, 3, N, K]`` and is what a
    user of the reference calls;
  * ``forward_rows(rows, ...)`` takes / returns the internal rows layout ``[P*3, C]`` (see ops/__init__.py) and is what
    the networks in pcn.py chain, so that no layout conversion happens between layers.
Parameters are held by ``nn.Linear(bias=False)`` / ``nn.BatchNorm*`` members exactly as in the reference, which
keeps the ``state_dict`` keys and the default-init RNG stream identical (same seed -> same weights).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops

EPS = 1e-6


def to_rows(x):
    """[B, C, 3, N(, K)] -> (rows [(B*N*K)*3, C], meta)."""
    shape = tuple(x.shape)
    B, C = shape[0], shape[1]
    if len(shape) < 4 or shape[2] != 3:
        raise ValueError(f"VN tensor [B, C, 3, N, ...] expected, got {shape}")
    nflat = 1
    for s in shape[3:]:
        nflat *= s
    rows = ops.ToRowsFn.apply(x.reshape(B, C, 3, nflat).float())
    return rows, (B, shape[3:], nflat)


def from_rows(rows, meta):
    B, tail, nflat = meta
    y = ops.FromRowsFn.apply(rows, B, nflat)
    return y.reshape(B, rows.shape[1], 3, *tail)


class VNLinear(nn.Module):
    """models/vn_layers.py:12-22."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.map_to_feat = nn.Linear(in_channels, out_channels, bias=False)

    def forward_rows(self, rows, bias=None, rows_per_group=0, residual=None):
        """``residual``: a tensor of the output's shape added to it in the GEMM's epilogue (a block's skip connection)."""
        return ops.VNLinearFn.apply(rows, self.map_to_feat.weight, bias, rows_per_group, residual)

    def forward(self, x):
        rows, meta = to_rows(x)
        return from_rows(self.forward_rows(rows), meta)


class VNLeakyReLU(nn.Module):
    """models/vn_layers.py:25-43."""

    def __init__(self, in_channels, share_nonlinearity=False, negative_slope=0.2):
        super().__init__()
        self.map_to_dir = nn.Linear(in_channels, 1 if share_nonlinearity else in_channels, bias=False)
        self.share_nonlinearity = share_nonlinearity
        self.negative_slope = negative_slope

    def forward_rows(self, rows):
        if not self.share_nonlinearity:
            return ops.VNLeakyDirFn.apply(rows, self.map_to_dir.weight, self.negative_slope)
        d = ops.VNLinearFn.apply(rows, self.map_to_dir.weight, None, 0)
        d = d.expand(-1, rows.shape[1]).contiguous()
        return ops.VNLeakyFn.apply(rows, d, self.negative_slope)

    def forward(self, x):
        rows, meta = to_rows(x)
        return from_rows(self.forward_rows(rows), meta)


# BatchNorm's num_batches_tracked counters: a layer called on its own increments its counter at once (one small launch); inside
# PCNNet.forward the counters of all layers are collected and incremented together when the forward ends (VN-PoinTr: 22 launches -> 1).
_DEFERRED_TICKS = None


def begin_deferred_ticks():
    """Start collecting counters; returns the previous collector (None if none was active) for ``end_deferred_ticks``."""
    global _DEFERRED_TICKS
    prev = _DEFERRED_TICKS
    if prev is None:
        _DEFERRED_TICKS = []
    return prev


def end_deferred_ticks(prev):
    """Increment everything collected since the matching ``begin_deferred_ticks`` (the outermost pair does the work)."""
    global _DEFERRED_TICKS
    if prev is None and _DEFERRED_TICKS is not None:
        pending, _DEFERRED_TICKS = _DEFERRED_TICKS, None
        if pending:
            # a layer that ran twice in this forward is in the list twice: the multi-tensor add does not order its read-modify-writes on
            # one tensor, so duplicates are counted and each counter is touched once
            counts = {}
            for c in pending:
                counts.setdefault(id(c), [c, 0])[1] += 1
            with torch.no_grad():
                if all(n == 1 for _, n in counts.values()):
                    torch._foreach_add_(pending, 1)
                else:
                    for c, n in counts.values():
                        c.add_(n)


class VNBatchNorm(nn.Module):
    """models/vn_layers.py:107-127."""

    def __init__(self, num_features, dim):
        super().__init__()
        self.dim = dim
        if dim == 3 or dim == 4:
            self.bn = nn.BatchNorm1d(num_features)
        elif dim == 5:
            self.bn = nn.BatchNorm2d(num_features)

    def _tick(self):
        if self.training and self.bn.track_running_stats and self.bn.num_batches_tracked is not None:
            if _DEFERRED_TICKS is not None:
                _DEFERRED_TICKS.append(self.bn.num_batches_tracked)      # PCNNet.forward: one launch for all counters at its end
            else:
                self.bn.num_batches_tracked.add_(1)

    def forward_rows(self, rows):
        self._tick()
        return ops.VNBnLeakyFn.apply(rows, False, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var,
                                     self.training, 0.0)

    def forward(self, x):
        rows, meta = to_rows(x)
        return from_rows(self.forward_rows(rows), meta)


class VNLayerNorm(nn.Module):
    """models/vn_layers.py:129-150."""

    def __init__(self, num_features):
        super().__init__()
        self.layer_norm = nn.LayerNorm(num_features)

    def forward_rows(self, rows):
        return ops.VNLayerNormFn.apply(rows, self.layer_norm.weight, self.layer_norm.bias)

    def forward_rows_carry(self, rows, a1=None, a2=None):
        """(s, LayerNorm(s)) with s = rows + a1 + a2: for a tensor that is normalised AND carried on as a skip connection — feed the
        RETURNED s to the skip connection (ops.AddLayerNormFn: the adds ride in the LayerNorm pass, forward and backward)."""
        return ops.AddLayerNormFn.apply(rows, a1, a2, self.layer_norm.weight, self.layer_norm.bias)

    def forward(self, x):
        rows, meta = to_rows(x)
        return from_rows(self.forward_rows(rows), meta)


class VNLinearLeakyReLU(nn.Module):
    """models/vn_layers.py:46-74 (direction computed from the layer input)."""

    def __init__(self, in_channels, out_channels, dim=5, share_nonlinearity=False, negative_slope=0.2):
        super().__init__()
        self.dim = dim
        self.negative_slope = negative_slope
        self.share_nonlinearity = share_nonlinearity
        self.map_to_feat = nn.Linear(in_channels, out_channels, bias=False)
        self.batchnorm = VNBatchNorm(out_channels, dim=dim)
        self.map_to_dir = nn.Linear(in_channels, 1 if share_nonlinearity else out_channels, bias=False)
        if not share_nonlinearity:
            ops.pack_hint(self.map_to_feat.weight, self.map_to_dir.weight)      # [Wf; Wd] as one block of the optimizer's flat buffer

    def forward_rows(self, rows, global_rows=None, rows_per_group=0, global_first=True, pre_linear=None):
        """``global_rows`` [G*3, Cg]: per-group features standing for Cg of the input channels, i.e. the layer acts on
        cat([global.expand(n), x], dim=1) (``global_first``, models/pcn.py:172) or cat([x, global.expand(n)], dim=1)
        (models/pcn.py:486,488) without materialising the broadcast; a group is ``rows_per_group``/3 consecutive points.
        ``pre_linear`` [Cx, K0]: the per-point input x is the (never materialised) output ``rows @ pre_linear.T`` of a VNLinear; the
        layer's weights are folded onto that layer's K0 input channels (ops.SW.ENC_LINEAR_REASSOCIATE)."""
        bn = self.batchnorm.bn
        self.batchnorm._tick()
        Wf, Wd = self.map_to_feat.weight, self.map_to_dir.weight
        if self.share_nonlinearity:
            Wd = Wd.expand(Wf.shape[0], -1)              # one shared direction (vn_layers.py:56-57): same row for every channel
        bias = None
        if global_rows is not None:
            # the per-sample channels' columns of both weights as one [2C, cg] matrix: ONE product gives the bias (feat | dir); the
            # per-point columns come back as the two halves of one buffer (ops.PackPdColsFn)
            cg = global_rows.shape[1]
            cx = Wf.shape[1] - cg
            fold = (pre_linear is None and rows_per_group % 3 == 0 and rows.dim() == 2 and cx == 1
                    and ops.group_fold_ok(rows.shape[1], rows_per_group // 3, Wf.shape[0]))
            if global_first:
                Wg, *Wx = ops.pack_pd_cols(Wf, Wd, (cg, cx), (False, not fold))
            else:
                *Wx, Wg = ops.pack_pd_cols(Wf, Wd, (cx, cg), (not fold, False))
            bias = ops.VNLinearFn.apply(global_rows, Wg, None, 0)
            if fold:
                # ONE per-point VN channel beside per-group features (the attention decoder's folding stages): all passes rebuild
                # p | d = bias[group] + w x[point] in registers instead of streaming a [P*3, 2C] tensor six times
                return ops.GroupFoldBnLeakyFn.apply(rows, Wx[0].squeeze(1), bias, rows_per_group // 3, bn.weight, bn.bias,
                                                    bn.running_mean, bn.running_var, self.training, self.negative_slope)
            Wf, Wd = Wx
        if pre_linear is not None:
            ops.trace("linear_folded_onto_producer")
            Wf, Wd = ops.WeightProductFn.apply(Wf, Wd, pre_linear)
        return ops.VNLinearLeakyFn.apply(rows, Wf, Wd, bn.weight, bn.bias, bn.running_mean, bn.running_var, bias,
                                         rows_per_group, self.training, self.negative_slope, pre_linear is not None)

    def forward_dir(self, h, u, global_rows=None, points_per_group=0, pre_linear=None, scalar_out=False):
        """The layer on a map u (x) h whose VN vectors are parallel to one unit direction ``u`` [P, 3] per point, given as per-point
        scalars ``h`` [P, K] (ops.VNDirLinearLeakyFn); ``global_rows`` / ``pre_linear`` as in forward_rows (global channels first).
        Returns rows [P*3, C], or — ``scalar_out``, for a layer without ``global_rows``, whose output is parallel to u again — the
        per-point scalars [P, C] of its output."""
        bn = self.batchnorm.bn
        self.batchnorm._tick()
        Wf, Wd = self.map_to_feat.weight, self.map_to_dir.weight
        bias = None
        if global_rows is not None:
            cg = global_rows.shape[1]
            cx = Wf.shape[1] - cg
            Wg, Wf, Wd = ops.pack_pd_cols(Wf, Wd, (cg, cx), (False, True))
            bias = ops.VNLinearFn.apply(global_rows, Wg, None, 0)
        if pre_linear is not None:
            ops.trace("linear_folded_onto_producer")
            Wf, Wd = ops.WeightProductFn.apply(Wf, Wd, pre_linear)
        return ops.VNDirLinearLeakyFn.apply(h, u, Wf, Wd, bn.weight, bn.bias, bn.running_mean, bn.running_var, bias,
                                            points_per_group, self.training, self.negative_slope, pre_linear is not None, scalar_out)

    def forward(self, x):
        rows, meta = to_rows(x)
        return from_rows(self.forward_rows(rows), meta)


class VNLinearAndLeakyReLU(nn.Module):
    """models/vn_layers.py:77-104 (direction computed from the linear output)."""

    def __init__(self, in_channels, out_channels, dim=5, share_nonlinearity=False, use_batchnorm="norm", negative_slope=0.2):
        super().__init__()
        self.dim = dim
        self.share_nonlinearity = share_nonlinearity
        self.use_batchnorm = use_batchnorm
        self.negative_slope = negative_slope
        self.linear = VNLinear(in_channels, out_channels)
        self.leaky_relu = VNLeakyReLU(out_channels, share_nonlinearity=share_nonlinearity, negative_slope=negative_slope)
        if use_batchnorm != "none":
            self.batchnorm = VNBatchNorm(out_channels, dim=dim)

    def forward_rows(self, rows):
        y = self.linear.forward_rows(rows)
        if self.use_batchnorm != "none":
            y = self.batchnorm.forward_rows(y)
        return self.leaky_relu.forward_rows(y)

    def forward(self, x):
        rows, meta = to_rows(x)
        return from_rows(self.forward_rows(rows), meta)


class VNMaxPool(nn.Module):
    """models/vn_layers.py:153-167: arg-max over the LAST axis of the reference tensor."""

    def __init__(self, in_channels):
        super().__init__()
        self.map_to_dir = nn.Linear(in_channels, in_channels, bias=False)

    def forward_rows(self, rows, groups, group_size, return_index=False, producer=None):
        """``producer`` = (x_prev, VNLinear) when ``rows`` is that layer's output: lets the direction GEMM contract over the
        producer's input channels (ops.SW.POOL_DIRECTION_REASSOCIATE)."""
        xp, wp = (producer[0].detach(), producer[1].map_to_feat.weight.detach()) if producer is not None else (None, None)
        out, idx = ops.VNMaxPoolFn.apply(rows, self.map_to_dir.weight.detach(), groups, group_size, xp, wp)
        return (out, idx) if return_index else out

    def forward_rows_tap(self, rows, groups, group_size, producer=None):
        """Pool a tensor that has another consumer: returns (rows, pooled).  Feed the RETURNED rows to that consumer — the pool's
        gradient is then added into its gradient in place (ops.VNMaxPoolTapFn)."""
        xp, wp = (producer[0].detach(), producer[1].map_to_feat.weight.detach()) if producer is not None else (None, None)
        rows2, out, _ = ops.VNMaxPoolTapFn.apply(rows, self.map_to_dir.weight.detach(), groups, group_size, xp, wp)
        return rows2, out

    def forward(self, x):
        rows, (B, tail, nflat) = to_rows(x)
        gs = tail[-1]
        G = B * nflat // gs
        out = self.forward_rows(rows, G, gs)                     # [G*3, C]
        y = ops.FromRowsFn.apply(out, B, G // B)                  # [B, C, 3, G/B]
        return y.reshape(B, rows.shape[1], 3, *tail[:-1])


def mean_pool(x, dim=-1, keepdim=False):
    """models/vn_layers.py:170-171 (reference layout; used on the neighbour axis)."""
    if dim in (-1, x.dim() - 1) and not keepdim and x.dim() >= 5 and x.is_cuda:
        rows, (B, tail, nflat) = to_rows(x)
        gs = tail[-1]
        G = B * nflat // gs
        out = ops.VNMeanPoolFn.apply(rows, G, gs)
        y = ops.FromRowsFn.apply(out, B, G // B)
        return y.reshape(B, rows.shape[1], 3, *tail[:-1])
    return x.mean(dim=dim, keepdim=keepdim)


class VNStdFeature(nn.Module):
    """models/vn_layers.py:174-220: a learned, rotation-equivariant frame z0 and the invariant features x_std = x . z0.
    Not used by any PCNNet path; the two VN layers run on libvnpcc, the 3x3 frame algebra (a few
    floats per point) stays in torch."""

    def __init__(self, in_channels, dim=4, normalize_frame=False, share_nonlinearity=False, negative_slope=0.2):
        super().__init__()
        self.dim = dim
        self.normalize_frame = normalize_frame
        self.vn1 = VNLinearLeakyReLU(in_channels, in_channels // 2, dim=dim, share_nonlinearity=share_nonlinearity,
                                     negative_slope=negative_slope)
        self.vn2 = VNLinearLeakyReLU(in_channels // 2, in_channels // 4, dim=dim, share_nonlinearity=share_nonlinearity,
                                     negative_slope=negative_slope)
        self.vn_lin = nn.Linear(in_channels // 4, 2 if normalize_frame else 3, bias=False)

    def forward(self, x):
        rows, meta = to_rows(x)
        z = self.vn2.forward_rows(self.vn1.forward_rows(rows))
        z0 = from_rows(ops.VNLinearFn.apply(z, self.vn_lin.weight, None, 0), meta)          # [B, 2|3, 3, N, ...]
        if self.normalize_frame:
            v1 = z0[:, 0, :]
            u1 = v1 / (torch.sqrt((v1 * v1).sum(1, keepdims=True)) + EPS)
            v2 = z0[:, 1, :]
            v2 = v2 - (v2 * u1).sum(1, keepdims=True) * u1
            u2 = v2 / (torch.sqrt((v2 * v2).sum(1, keepdims=True)) + EPS)
            u3 = torch.cross(u1, u2, dim=1)
            z0 = torch.stack([u1, u2, u3], dim=1).transpose(1, 2)
        else:
            z0 = z0.transpose(1, 2)
        if self.dim == 4:
            x_std = torch.einsum("bijm,bjkm->bikm", x, z0)
        elif self.dim == 3:
            x_std = torch.einsum("bij,bjk->bik", x, z0)
        else:
            x_std = torch.einsum("bijmn,bjkmn->bikmn", x, z0)
        return x_std, z0
