"""The two non-equivariant members of the reference's plugin surface:

  * ``FoldingNet``   — dec_type "foldingnet"  (/root/reference/models/pcn.py:274-317)
  * ``DGCNN_fps``    — enc_type "dgcnn_fps"   (/root/reference/models/dgcnn.py:19-161)

Same constructor signatures, attribute names and ``state_dict`` keys as the reference (parameters live in the same
``nn.Conv1d`` / ``nn.Conv2d`` / ``nn.BatchNorm1d`` / ``nn.GroupNorm`` / ``nn.Linear`` members, created in the same order,
so the same seed gives the same weights and reference checkpoints load).

These are baselines outside the measured hot path.  Everything with real arithmetic in it runs on libvnpcc — every
1x1 convolution / linear layer is ``vnp_gemm`` (point-major rows ``[B*N, C]``), neighbour search is ``vnp_knn``, sampling is
``vnp_fps``, the loss is the HIP Chamfer, BatchNorm1d with the ReLU / LeakyReLU behind it is ``vnp_bn1d_act`` (round 3), the EdgeConv's
GroupNorm(4) + LeakyReLU + max over the k neighbours is ``vnp_edge_gn_max`` (round 4: four passes that recompute the edge values from the
per-point terms, the neighbour term's gradient gathered over the sorted reverse adjacency — no edge tensor, no atomics), biases and the
coarse head's ReLU are ``vnp_bias_act``, the global max is ``vnp_rows_max`` — and there is no CPU path (CPU tensors raise).  What is left
in ATen on the device is index plumbing and the K = 2 seed term of the decoder's first layer.  As in the VN networks, broadcast input channels are
never materialised: the decoder's first layer is a per-sample bias plus a rank-5 per-point term instead of a GEMM over a
``[B, 6149, 16384]`` tensor, and EdgeConv is ``W1 x_k[idx] + (W2 - W1) x_q`` (two per-point GEMMs and a gather) instead of a
GEMM over the k-times larger edge tensor.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib, ops


class _BiasActFn(torch.autograd.Function):
    """y = act(x + bias) on rows [P, C] (vnp_bias_act): nn.Linear / Conv1d biases and the ReLU behind them; act 0 none, 1 ReLU, 2 LeakyReLU."""

    @staticmethod
    def forward(ctx, x, bias, act, slope):
        from ._lib import call, ptr
        from .ops import _ld, _rows
        x = _rows(x)
        P, C = x.shape
        y = torch.empty((P, C), device=x.device, dtype=torch.float32)
        call("vnp_bias_act", ptr(x), _ld(x), ptr(bias), P, C, act, float(slope), ptr(y), C, None, 0, None, 0)
        ctx.save_for_backward(x, bias)
        ctx.cfg = (act, float(slope))
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        from ._lib import call, ptr
        from .ops import _ld, _rows, reduce_rows
        x, bias = ctx.saved_tensors
        act, slope = ctx.cfg
        gy = _rows(gy)
        P, C = x.shape
        if act == 0:
            gx = gy
        else:
            gx = torch.empty((P, C), device=x.device, dtype=torch.float32)
            call("vnp_bias_act", ptr(x), _ld(x), ptr(bias), P, C, act, slope, None, 0, ptr(gy), _ld(gy), ptr(gx), C)
        gb = None
        if bias is not None and ctx.needs_input_grad[1]:
            from .ops.core import _nchunk
            g = gx.contiguous()
            nch = _nchunk(P)
            part = torch.empty((nch, C), device=x.device, dtype=torch.float32)
            call("vnp_colsum", ptr(g), 0, C, 1, P, C, None, ptr(part))
            gb = reduce_rows(part, 1, nch, C)[0]
        return gx, gb, None, None


def _linear(rows, weight, bias=None, act=0, slope=0.0):
    """act(rows [P, Cin] x weight [Cout, Cin]^T + bias) — the product on vnp_gemm, bias and activation on vnp_bias_act — with autograd."""
    y = ops.VNLinearFn.apply(rows, weight, None, 0)
    return y if (bias is None and act == 0) else _BiasActFn.apply(y, bias, act, slope)


class _RowsMaxFn(torch.autograd.Function):
    """[B*n, C] rows -> [B, C]: max over the n rows of each sample (torch.max(dim), models/dgcnn.py:157) on vnp_rows_max."""

    @staticmethod
    def forward(ctx, x, B, n):
        from ._lib import call, ptr
        from .ops import _ld, _rows
        x = _rows(x)
        C = x.shape[1]
        out = torch.empty((B, C), device=x.device, dtype=torch.float32)
        amax = torch.empty((B, C), device=x.device, dtype=torch.int32)
        call("vnp_rows_max", ptr(x), _ld(x), B, n, C, ptr(out), ptr(amax))
        ctx.save_for_backward(amax)
        ctx.cfg = (B, n, C)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        from ._lib import call, ptr
        (amax,) = ctx.saved_tensors
        B, n, C = ctx.cfg
        go = go.contiguous()
        gx = torch.empty((B * n, C), device=go.device, dtype=torch.float32)
        call("vnp_rows_max_bwd", ptr(go), ptr(amax), B, n, C, ptr(gx))
        return gx, None, None


class _EdgeGnLeakyMaxFn(torch.autograd.Function):
    """GroupNorm(G) + LeakyReLU + max over the k neighbours of the edge values e(q, j) = u[nbr[q][j]] + v[q] (vnp_edge_gn_max): the tail of
    DGCNN_fps' EdgeConv blocks (models/dgcnn.py:36-54, 101-124) without the [B, C, Nq, k] edge tensor.  u [Pk, C], v [B*nq, C] rows,
    nbr [B*nq, k] int32 global row ids of u.  Returns [B*nq, C]."""

    @staticmethod
    def forward(ctx, u, v, nbr, gamma, beta, B, nq, G, eps, slope):
        from ._lib import call, ptr
        from .ops import _ld, _rows
        u, v = _rows(u), _rows(v)
        nbr = nbr.contiguous()
        C, k = u.shape[1], nbr.shape[1]
        dev = u.device
        nblk = B * ((nq + 63) // 64)
        part = torch.empty((nblk, C, 2), device=dev, dtype=torch.float32)
        common = (ptr(u), _ld(u), ptr(v), _ld(v), ptr(nbr), B, nq, k, C, G)
        call("vnp_edge_gn_max", 0, *common, None, None, None, None, float(slope), None, 0, None, None, 0, ptr(part), None, None, 0)
        mean = torch.empty((B, G), device=dev, dtype=torch.float32)
        rstd = torch.empty((B, G), device=dev, dtype=torch.float32)
        call("vnp_edge_gn_finalize", ptr(part), B, nq, k, C, G, float(eps), ptr(mean), ptr(rstd))
        out = torch.empty((B * nq, C), device=dev, dtype=torch.float32)
        amax = torch.empty((B * nq, C), device=dev, dtype=torch.uint8)
        call("vnp_edge_gn_max", 1, *common, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), float(slope), ptr(out), C, ptr(amax), None, 0, None,
             None, None, 0)
        ctx.save_for_backward(u, v, nbr, gamma, beta, mean, rstd, amax)
        ctx.cfg = (B, nq, G, float(slope))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        from ._lib import call, ptr
        from .ops import _ld, _rows, edge_reverse, reduce_rows
        u, v, nbr, gamma, beta, mean, rstd, amax = ctx.saved_tensors
        B, nq, G, slope = ctx.cfg
        go = _rows(go)
        C, k, Pk = u.shape[1], nbr.shape[1], u.shape[0]
        dev = u.device
        nblk_s = (nq + 63) // 64
        common = (ptr(u), _ld(u), ptr(v), _ld(v), ptr(nbr), B, nq, k, C, G)
        gn = (ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), slope)
        part = torch.empty((B * nblk_s, C, 4), device=dev, dtype=torch.float32)
        call("vnp_edge_gn_max", 2, *common, *gn, None, 0, ptr(amax), ptr(go), _ld(go), ptr(part), None, None, 0)
        sums = reduce_rows(part.view(B, nblk_s, 4 * C), B, nblk_s, 4 * C)                 # [B, C * 4]: per sample (the group terms)
        tot = reduce_rows(sums, 1, B, 4 * C)[0].view(C, 4)                                # over the batch: dL/dbeta, dL/dgamma
        gv = torch.empty((B * nq, C), device=dev, dtype=torch.float32)
        call("vnp_edge_gn_max", 3, *common, *gn, None, 0, ptr(amax), ptr(go), _ld(go), None, ptr(sums), ptr(gv), C)
        start, rev = edge_reverse(nbr, Pk)
        gu = torch.empty((Pk, C), device=dev, dtype=torch.float32)
        call("vnp_edge_gn_gather", ptr(u), _ld(u), ptr(v), _ld(v), ptr(nbr), ptr(start), ptr(rev), Pk, B, nq, k, C, G, *gn, ptr(amax),
             ptr(go), _ld(go), ptr(sums), ptr(gu), C)
        return gu, gv, None, tot[:, 1].contiguous(), tot[:, 0].contiguous(), None, None, None, None, None


class _BatchNorm1dActFn(torch.autograd.Function):
    """nn.BatchNorm1d over point-major rows [P, C] + the activation behind it on libvnpcc (vnp_bn1d_act): Welford statistics merged in
    double (running statistics updated as torch does), one forward pass, two backward passes with the pre-activation recomputed."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, act, slope):
        from ._lib import call, ptr
        from .ops import _ld, _rows, reduce_pairs
        x = _rows(x)
        P, C = x.shape
        dev = x.device
        if training:
            nch = (P + _lib.lib().vnp_bn1d_chunk_rows() - 1) // _lib.lib().vnp_bn1d_chunk_rows()
            part = torch.empty((nch, C, 2), device=dev, dtype=torch.float32)
            call("vnp_bn1d_act", 0, ptr(x), _ld(x), P, C, None, None, None, None, act, slope, None, 0, None, 0, ptr(part), None, 0)
            mean = torch.empty(C, device=dev, dtype=torch.float32)
            rstd = torch.empty(C, device=dev, dtype=torch.float32)
            call("vnp_vn_bn_finalize_chunked", ptr(part), nch, _lib.lib().vnp_bn1d_chunk_rows(), C, P, ptr(mean), ptr(rstd),
                 ptr(running_mean), ptr(running_var), float(momentum), float(eps))
        else:
            mean = running_mean
            rstd = torch.empty(C, device=dev, dtype=torch.float32)
            call("vnp_rstd_from_var", ptr(running_var), ptr(rstd), C, float(eps))
        y = torch.empty((P, C), device=dev, dtype=torch.float32)
        call("vnp_bn1d_act", 1, ptr(x), _ld(x), P, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act, slope, ptr(y), C, None, 0, None, None, 0)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (training, act, slope)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        from ._lib import call, ptr
        from .ops import _ld, _rows, reduce_pairs
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        training, act, slope = ctx.cfg
        gy = _rows(gy)
        P, C = x.shape
        dev = x.device
        nch = (P + _lib.lib().vnp_bn1d_chunk_rows() - 1) // _lib.lib().vnp_bn1d_chunk_rows()
        part = torch.empty((nch, C, 2), device=dev, dtype=torch.float32)
        bn = (ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act, slope)
        call("vnp_bn1d_act", 2, ptr(x), _ld(x), P, C, *bn, None, 0, ptr(gy), _ld(gy), ptr(part), None, 0)
        S, planar = reduce_pairs(part, nch, C)
        gx = torch.empty((P, C), device=dev, dtype=torch.float32)
        call("vnp_bn1d_act", 3, ptr(x), _ld(x), P, C, *bn, ptr(gx), C, ptr(gy), _ld(gy), None, ptr(S) if training else None, P)
        return gx, planar[1], planar[0], None, None, None, None, None, None, None


def _bn_rows(bn: nn.BatchNorm1d, rows, act=0, slope=0.0):
    """nn.BatchNorm1d of a [B, C, N] tensor, evaluated on its point-major rows [B*N, C] (same statistics), with the activation that
    follows it (0 none, 1 ReLU, 2 LeakyReLU(slope)) in the same kernels."""
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    training = bn.training or not bn.track_running_stats
    momentum = bn.momentum if bn.momentum is not None else 0.1
    return _BatchNorm1dActFn.apply(rows, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum, bn.eps, act, float(slope))


class FoldingNet(nn.Module):
    """models/pcn.py:274-317.  forward(coarse [B,Mc,3], feature_global [B,2048,3] | [B,6144], rot ignored) -> fine [B,Mc*S,3]."""

    def __init__(self, config, grid_size=4):
        super().__init__()
        self.grid_size = grid_size
        if config.num_coarse == 448:
            self.num_coarse = config.num_coarse // 2
            self.num_dense = 14336
            self.grid_size = 8
        else:
            self.num_coarse = config.num_coarse
            self.num_dense = 16384
            self.grid_size = 4
        self.final_conv = nn.Sequential(
            nn.Conv1d(2048 * 3 + 3 + 2, 512, 1),
            nn.BatchNorm1d(512),
            nn.ReLU(inplace=True),
            nn.Conv1d(512, 512, 1),
            nn.BatchNorm1d(512),
            nn.ReLU(inplace=True),
            nn.Conv1d(512, 3, 1),
        )
        g = self.grid_size
        lin = torch.linspace(-0.05, 0.05, steps=g, dtype=torch.float)
        a = lin.view(1, g).expand(g, g).reshape(1, -1)
        b = lin.view(g, 1).expand(g, g).reshape(1, -1)
        self.folding_seed = torch.cat([a, b], dim=0).view(1, 2, g ** 2)         # plain attribute, as in the reference

    def forward(self, coarse, feature_global, rot=None):
        B, Mc, _ = coarse.shape
        S = self.grid_size ** 2
        if Mc * S != self.num_dense:
            raise RuntimeError(f"FoldingNet: num_coarse*grid^2 = {Mc * S} != num_dense = {self.num_dense}")
        dev = coarse.device
        coarse = coarse.contiguous().float()
        g = feature_global.reshape(B, -1).float()
        c0, bn0, c1, bn1, c2 = self.final_conv[0], self.final_conv[1], self.final_conv[3], self.final_conv[4], self.final_conv[6]
        W0 = c0.weight[:, :, 0]
        L = g.shape[1]
        if W0.shape[1] != L + 5:
            raise RuntimeError(f"FoldingNet expects a {W0.shape[1] - 5}-value global feature, got {L}")
        # cat([global.expand(Md), seed, point_feat]) @ W0^T  =  per-sample bias + seed term (per grid cell) + coarse term (per centre)
        bias = _linear(g, W0[:, :L], c0.bias)                                                   # [B, 512]
        seed = self.folding_seed.to(dev)[0].t().contiguous()                                     # [S, 2]
        t_seed = seed @ W0[:, L:L + 2].t()                                                       # [S, 512]   (K = 2: ATen)
        t_coarse = _linear(coarse.reshape(B * Mc, 3), W0[:, L + 2:L + 5])                        # [B*Mc, 512]
        pre = (bias.view(B, 1, 1, -1) + t_coarse.view(B, Mc, 1, -1) + t_seed.view(1, 1, S, -1)).reshape(B * Mc * S, -1)
        h = _bn_rows(bn0, pre, act=1)
        h = _bn_rows(bn1, _linear(h, c1.weight[:, :, 0], c1.bias), act=1)
        point_feat = coarse.unsqueeze(2).expand(-1, -1, S, -1).reshape(B * Mc * S, 3)
        fine = _linear(h, c2.weight[:, :, 0], c2.bias) + point_feat
        return fine.reshape(B, Mc * S, 3)


class DGCNN_fps(nn.Module):
    """models/dgcnn.py:19-161.  forward(x [B,N,3]) -> (coarse [B,Mc,3] | (coarse, coarse_cat), feature_global [B,1024])."""

    def __init__(self, config, latent_dim=1024, grid_size=4, only_coarse=False, num_dense=16384):
        super().__init__()
        self.latent_dim = latent_dim
        self.num_dense = num_dense
        self.grid_size = grid_size
        self.num_coarse = config.num_coarse // 2 if config.num_coarse == 448 else config.num_coarse
        self.only_coarse = only_coarse
        self.k = 16
        self.input_trans = nn.Conv1d(3, 8, 1)

        def block(cin, cout):
            return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.GroupNorm(4, cout), nn.LeakyReLU(negative_slope=0.2))
        self.layer1 = block(16, 32)
        self.layer2 = block(64, 64)
        self.layer3 = block(128, 64)
        self.layer4 = block(128, 128)
        self.increase_dim = nn.Sequential(nn.Conv1d(128, 1024, 1), nn.BatchNorm1d(1024), nn.LeakyReLU(negative_slope=0.2),
                                          nn.Conv1d(1024, 1024, 1))
        self.coarse_pred = nn.Sequential(nn.Linear(1024, 1024), nn.ReLU(inplace=True), nn.Linear(1024, 3 * self.num_coarse))

    @staticmethod
    def fps_downsample(coor, x, num_group):
        """coor [B,N,3], x rows [B,N,C] -> the num_group furthest-point samples of both (vnp_fps: starts at point 0)."""
        idx = ops.fps_indices(coor, num_group).long()
        return (torch.gather(coor, 1, idx[..., None].expand(-1, -1, 3)),
                torch.gather(x, 1, idx[..., None].expand(-1, -1, x.shape[2])))

    def _edge_conv(self, layer, coor_q, x_q, coor_k, x_k):
        """get_graph_feature + Conv2d(1x1, no bias) + GroupNorm(4) + LeakyReLU + max over the k neighbours
        (models/dgcnn.py:101-124 and the layerN blocks).  coor [B,N,3]; x rows [B,N,C] -> [B,Nq,Cout]."""
        conv, gn = layer[0], layer[1]
        B, Nq, C = x_q.shape
        Nk = x_k.shape[1]
        W = conv.weight[:, :, 0, 0]
        Cout = W.shape[0]
        with torch.no_grad():
            idx = ops.knn_indices(coor_k, coor_q, self.k)                              # [B,Nq,k] int32, sorted by (distance, index)
            gidx = (idx + (torch.arange(B, device=idx.device, dtype=torch.int32) * Nk).view(B, 1, 1)).reshape(B * Nq, self.k)
        u = _linear(x_k.reshape(B * Nk, C), W[:, :C])                                   # neighbour term  W1 x_k
        v = _linear(x_q.reshape(B * Nq, C), W[:, C:] - W[:, :C])                        # centre term  (W2 - W1) x_q
        out = _EdgeGnLeakyMaxFn.apply(u, v, gidx, gn.weight, gn.bias, B, Nq, gn.num_groups, gn.eps, 0.2)
        return out.view(B, Nq, Cout)

    def forward(self, x):
        B, N, _ = x.shape
        coor = x.contiguous().float()
        f = _linear(coor.reshape(B * N, 3), self.input_trans.weight[:, :, 0], self.input_trans.bias).view(B, N, -1)
        f = self._edge_conv(self.layer1, coor, f, coor, f)
        coor_q, f_q = self.fps_downsample(coor, f, 512)
        f = self._edge_conv(self.layer2, coor_q, f_q, coor, f)
        coor = coor_q
        f = self._edge_conv(self.layer3, coor, f, coor, f)
        coor_q, f_q = self.fps_downsample(coor, f, 128)
        f = self._edge_conv(self.layer4, coor_q, f_q, coor, f)                           # [B,128,128]
        n4 = f.shape[1]
        c0, bn, c3 = self.increase_dim[0], self.increase_dim[1], self.increase_dim[3]
        h = _bn_rows(bn, _linear(f.reshape(B * n4, -1), c0.weight[:, :, 0], c0.bias), act=2, slope=0.2)
        h = _linear(h, c3.weight[:, :, 0], c3.bias)                                      # [B*n4, 1024]
        feature_global = _RowsMaxFn.apply(h, B, n4)                                      # [B,1024]
        l0, l2 = self.coarse_pred[0], self.coarse_pred[2]
        coarse = _linear(_linear(feature_global, l0.weight, l0.bias, act=1), l2.weight, l2.bias).reshape(-1, self.num_coarse, 3)
        if self.num_coarse == 224:
            idx = ops.fps_indices(x.contiguous().float(), 224).long()
            inp_sparse = torch.gather(x, 1, idx[..., None].expand(-1, -1, 3))
            return (coarse.contiguous(), torch.cat([coarse, inp_sparse], dim=1).contiguous()), feature_global
        return coarse.contiguous(), feature_global
