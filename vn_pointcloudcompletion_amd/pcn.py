"""VN-PointNet encoder and VN-FoldingNet decoder with the reference's class names, constructor signatures and
``state_dict`` keys (/root/reference/models/pcn.py:110-184, 319-389), chained on the rows layout so that the whole
forward/backward runs in libvnpcc kernels.

Two algebraic de-duplications remove the reference's broadcast tensors:
  * encoder ``second_conv[0]`` acts on cat([global.expand(N), feature]) (pcn.py:172): the global half becomes a
    per-sample bias  W[:, :512] g  and only the 512 per-point channels go through the big GEMM;
  * decoder ``final_conv[0]`` acts on cat([global.expand(Md), seed, point_feat]) (pcn.py:385, 12.9 GB at B=32):
    W[:, :latent] g is a per-sample bias and the two remaining channels a rank-2 per-point term (ops.FoldFn).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib, ops
from .vn_layers import VNLinear, VNLinearAndLeakyReLU, VNLinearLeakyReLU, VNMaxPool


def global_to_rows(feature_global):
    """reference [B, C, 3, 1] -> rows [B*3, C]."""
    B, C = feature_global.shape[0], feature_global.shape[1]
    return feature_global.reshape(B, C, 3).permute(0, 2, 1).reshape(B * 3, C).contiguous()


def rows_to_global(rows, B):
    """rows [B*3, C] -> reference [B, C, 3, 1]."""
    C = rows.shape[1]
    return rows.reshape(B, 3, C).permute(0, 2, 1).unsqueeze(-1)


def fps(pc, num):
    """models/pcn.py:6-9 — furthest point sampling of a cloud (used only when num_coarse == 448)."""
    from .sampling import furthest_point_sample, gather_points
    idx = furthest_point_sample(pc, num)
    return gather_points(pc, idx)


class VN_PointNet(nn.Module):
    """models/pcn.py:110-184."""

    def __init__(self, config, num_dense=16384, latent_dim=1024):
        super().__init__()
        self.num_dense = num_dense
        self.latent_dim = latent_dim
        self.num_coarse = config.num_coarse // 2 if config.num_coarse == 448 else config.num_coarse
        self.first_conv = nn.Sequential(VNLinearLeakyReLU(1, 128, dim=4), VNLinear(128, 512))
        self.maxpool1 = VNMaxPool(512)
        self.second_conv = nn.Sequential(VNLinearLeakyReLU(1024, 1024, dim=4), VNLinear(1024, self.latent_dim * 2))
        self.maxpool2 = VNMaxPool(self.latent_dim * 2)
        self.mlp = nn.Sequential(
            VNLinearAndLeakyReLU(self.latent_dim * 2, 1024 * 2, dim=4, use_batchnorm="none"),
            VNLinearAndLeakyReLU(1024 * 2, 1024, dim=4, use_batchnorm="none"),
            VNLinear(1024, self.num_coarse),
        )

    def forward(self, xyz):
        B, N, _ = xyz.shape
        W1 = self.first_conv[1].map_to_feat.weight
        l1, l2 = self.first_conv[0], self.second_conv[0]
        dir_ok = (ops.SW.ENC_LINEAR_REASSOCIATE and ops.SW.ENC_DIR_LAYER and not l1.share_nonlinearity and not l2.share_nonlinearity
                  and ops.VNDirLinearLeakyFn.supported(B * N, l2.map_to_feat.weight.shape[0], N, True))
        # first_conv[0] has ONE input channel, the point itself (x = u |x| with the unit direction u): each of its output vectors is a
        # multiple of u, a1 = u (x) h, and second_conv[0] — folded onto first_conv[1]'s 128 input channels, see below — runs on the
        # per-point scalars h and rebuilds its pre-activations u v + bias in registers (ops.VNDirLinearLeakyFn)
        if dir_ok and ops.SW.ENC_FIRST_LAYER_SCALAR and _lib.get_gemm_mode() == "split":
            # the first layer itself on the per-point scalar |x|, returning h (scalar_out).  (Default GEMM mode only: VNP_GEMM_MODE=fp32 is the conservative comparison path and keeps the
            # first layer in the rows layout)
            u, r = ops.unit_dirs(xyz, return_norm=True)
            h = l1.forward_dir(r, u, scalar_out=True)                                # [B*N, 128]
            a1 = None                                                                # the rows u (x) h are never needed
        else:
            x = xyz.contiguous().float().reshape(B * N * 3, 1)                       # rows layout with C = 1 IS xyz
            a1 = l1.forward_rows(x)
            if dir_ok:
                u = ops.unit_dirs(xyz)
                h = ops.DirContractFn.apply(a1, u)
        if ops.SW.ENC_LINEAR_REASSOCIATE:
            # first_conv[1] is a linear map whose output f = W1 a1 [B*N*3, 512] feeds maxpool1 and — behind the concatenation —
            # second_conv[0]'s linear maps (pcn.py:168-173): the pool takes f from its fused GEMM without storing it, and
            # second_conv[0] contracts (W W1) with a1 over first_conv[1]'s 128 input channels instead of W with f over 512
            if a1 is None:
                # ... and maxpool1 on the scalars too: a point's score is (h W1^T)(h (Wd W1)^T), no components involved
                g1, _ = ops.DirLinearMaxPoolFn.apply(h, u, W1, self.maxpool1.map_to_dir.weight.detach(), B, N)   # g1 [B*3, 512]
            else:
                g1, _ = ops.VNLinearMaxPoolFn.apply(a1, W1, self.maxpool1.map_to_dir.weight.detach(), B, N)
            if dir_ok:
                f = l2.forward_dir(h, u, global_rows=g1, points_per_group=N, pre_linear=W1)
            else:
                f = l2.forward_rows(a1, global_rows=g1, rows_per_group=N * 3, pre_linear=W1)
        else:
            # as written: f is stored.  When the shapes allow, one GEMM stores f and pools in its epilogue; in any case the pool's
            # sparse gradient is added into second_conv[0]'s input gradient in place
            if ops.VNLinearPoolTapFn.supported(a1, W1, B, N):
                f, g1, _ = ops.VNLinearPoolTapFn.apply(a1, W1, self.maxpool1.map_to_dir.weight.detach(), B, N)
            else:
                f = self.first_conv[1].forward_rows(a1)
                f, g1 = self.maxpool1.forward_rows_tap(f, B, N, producer=(a1, self.first_conv[1]))   # g1 [B*3, 512]
            f = l2.forward_rows(f, global_rows=g1, rows_per_group=N * 3)
        # second_conv[1] + maxpool2 fused: the linear output has no other consumer, so its (one-point-per-channel)
        # gradient is never materialised (ops.VNLinearMaxPoolFn)
        g2, _ = ops.VNLinearMaxPoolFn.apply(f, self.second_conv[1].map_to_feat.weight,
                                            self.maxpool2.map_to_dir.weight.detach(), B, N)    # [B*3, 2*latent]
        h = self.mlp[0].forward_rows(g2)
        h = self.mlp[1].forward_rows(h)
        h = self.mlp[2].forward_rows(h)                                         # [B*3, Mc]
        coarse = h.reshape(B, 3, self.num_coarse).permute(0, 2, 1).contiguous()  # == [B,Mc,3,1].reshape(-1,Mc,3), pcn.py:177
        feature_global = rows_to_global(g2, B)
        if self.num_coarse == 224:
            inp_sparse = fps(xyz.contiguous(), 224)
            coarse_cat = torch.cat([coarse, inp_sparse], dim=1).contiguous()
            return (coarse, coarse_cat), feature_global
        return coarse, feature_global


class VN_FoldingNet(nn.Module):
    """models/pcn.py:319-389."""

    def __init__(self, config, grid_size=4):
        super().__init__()
        self.latent_dim = config.latent_dim
        if config.num_coarse == 448:
            self.num_coarse = config.num_coarse // 2
            self.num_dense = 14336
            self.grid_size = 8
        else:
            self.num_coarse = config.num_coarse
            self.num_dense = 16384
            self.grid_size = 4
        self.final_conv = nn.Sequential(
            VNLinearLeakyReLU(self.latent_dim + 1 + 1, 256, dim=4),
            VNLinearLeakyReLU(256, 256, dim=4),
            VNLinear(256, 1),
        )
        g = self.grid_size
        lin = torch.linspace(-0.05, 0.05, steps=g, dtype=torch.float)
        a = lin.view(1, g).expand(g, g).reshape(1, -1)
        b = lin.view(g, 1).expand(g, g).reshape(1, -1)
        c = torch.zeros_like(a)
        # plain attribute (not a buffer), as in the reference (pcn.py:362): it must not appear in the state_dict
        self.folding_seed = torch.cat([a, b, c], dim=0).reshape(1, 1, 3, -1)

    def forward(self, coarse, feature_global, rot=None):
        B, Mc, _ = coarse.shape
        S = self.grid_size ** 2
        if Mc * S != self.num_dense:
            # the reference hard-codes num_dense (pcn.py:332) and its reshape would fail here; callers running other
            # sizes set decoder.num_dense after construction — accept that as well.
            raise RuntimeError(f"VN_FoldingNet: num_coarse*grid^2 = {Mc * S} != num_dense = {self.num_dense}")
        dev = coarse.device
        if self.folding_seed.device != dev:
            self.folding_seed = self.folding_seed.to(dev)
        seed = self.folding_seed.squeeze(1).transpose(1, 2)                     # [1, S, 3]
        if rot is not None:
            seed = rot.transform_points(seed)                                   # [B, S, 3]  (pcn.py:367-370)
        seed = seed.contiguous().float()
        l0 = self.final_conv[0]
        latent = feature_global.shape[1]
        g_rows = global_to_rows(feature_global.float())
        Wf, Wd = l0.map_to_feat.weight, l0.map_to_dir.weight
        # per-sample part | seed column | coarse column of both weights, each as one [2C, n] matrix (rows: feat then dir)
        Wg, wseed, wcoarse = ops.pack_pd_cols(Wf, Wd, (latent, 1, 1))
        bias = ops.VNLinearFn.apply(g_rows, Wg, None, 0)                        # [B*3, 512] (feat | dir), one product
        wseed, wcoarse = wseed.squeeze(1), wcoarse.squeeze(1)
        coarse = coarse.contiguous().float()
        bn = l0.batchnorm.bn
        l0.batchnorm._tick()
        if ops.FoldBnLeakyFn.coarse_per_chunk(Mc, S) >= min(Mc, 16):
            # fused layer 0: the (p | d) pre-activations are recomputed inside every pass instead of being stored
            f = ops.FoldBnLeakyFn.apply(bias, wseed, wcoarse, seed, coarse, bn.weight, bn.bias, bn.running_mean,
                                        bn.running_var, self.training, l0.negative_slope)
        else:
            pd = ops.FoldFn.apply(bias, wseed, wcoarse, seed, coarse)           # [B*Md*3, 512] = (p | d) of layer 0
            f = ops.VNBnLeakyFn.apply(pd, True, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training,
                                      l0.negative_slope)
        l1 = self.final_conv[1]
        bn1 = l1.batchnorm.bn
        l1.batchnorm._tick()
        # final_conv[1] + final_conv[2] + point_feat fused: the 256-channel activation between them is never stored
        fine = ops.VNLinearLeakyOutFn.apply(f, l1.map_to_feat.weight, l1.map_to_dir.weight, bn1.weight, bn1.bias, bn1.running_mean,
                                            bn1.running_var, self.final_conv[2].map_to_feat.weight, coarse, S, self.training,
                                            l1.negative_slope)                                   # [B*Md*3]
        return fine.reshape(B, Mc * S, 3)
