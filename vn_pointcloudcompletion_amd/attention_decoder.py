"""Attention_VN_FoldingNet decoder: reference class name, constructor signature and ``state_dict`` keys
(/root/reference/models/pcn.py:392-520) on libvnpcc.  The reference's layout quirks are reproduced literally:
the centre coordinates are added through a reshape (not a transpose) scramble (pcn.py:466,469), ``rot`` is ignored,
and ``final_conv`` is constructed but never used.  Both folding stages act on cat([per-point channel, 384 per-centre
channels]) (pcn.py:486,488): the per-centre part becomes a per-group bias (groups of 16 grid samples), so the
[B*1024, 385, 3, 16] tensors are never built."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .pcn import global_to_rows
from .transformer import VN_Block, tokens_to_rows
from .vn_layers import VNLinear, VNLinearLeakyReLU


class Attention_VN_FoldingNet(nn.Module):
    def __init__(self, config, grid_size=4):
        super().__init__()
        self.latent_dim = config.latent_dim
        if config.num_coarse == 448:
            self.num_coarse, self.num_dense, self.grid_size = config.num_coarse // 2, 14336, 8
        else:
            self.num_coarse, self.num_dense, self.grid_size = config.num_coarse, 16384, 4
        self.transformer = nn.ModuleList([
            VN_Block(dim=384, num_heads=8, mlp_ratio=1, qkv_bias=False, qk_scale=1, drop=0, attn_drop=0) for _ in range(2)])
        self.final_conv = nn.Sequential(VNLinearLeakyReLU(self.latent_dim + 1 + 1, 256, dim=4),
                                        VNLinearLeakyReLU(256, 256, dim=4), VNLinear(256, 1))       # unused, as in the reference
        self.downsize_global = VNLinear(2048, 384)
        g = self.grid_size
        lin = torch.linspace(-1., 1., steps=g, dtype=torch.float)
        a = lin.view(1, g).expand(g, g).reshape(1, -1)
        b = lin.view(g, 1).expand(g, g).reshape(1, -1)
        self.folding_seed = torch.cat([a, b, torch.zeros_like(a)], dim=0)                           # [3, S], plain attribute
        in_channel, hidden_dim = 384, 256
        self.vn_folding1 = nn.Sequential(VNLinearLeakyReLU(in_channel + 1, hidden_dim, dim=4),
                                         VNLinearLeakyReLU(hidden_dim, hidden_dim // 2, dim=4), VNLinear(hidden_dim // 2, 1))
        self.vn_folding2 = nn.Sequential(VNLinearLeakyReLU(in_channel + 1, hidden_dim, dim=4),
                                         VNLinearLeakyReLU(hidden_dim, hidden_dim // 2, dim=4), VNLinear(hidden_dim // 2, 1))

    _ones = None             # [384, 1] of ones on the device (the unit weight of the token build), made once
    FUSED_TAIL = True        # stage[1] + stage[2] as one function (ops.VNLinearLeakyOutFn); False: layer by layer (tests)

    @classmethod
    def _fold(cls, stage, point_rows, feat_rows, S):
        h = stage[0].forward_rows(point_rows, global_rows=feat_rows, rows_per_group=S * 3, global_first=False)
        l1, l2 = stage[1], stage[2]
        if not cls.FUSED_TAIL:
            return l2.forward_rows(l1.forward_rows(h))                                               # [P*3, 1]
        # VNLinearLeakyReLU(256, 128) + VNLinear(128, 1) fused as in VN_FoldingNet's tail: the 128-channel activation between them and
        # its gradient are never stored (the upstream gradient of that activation is rank one: g_fold[r] * w_out[c])
        bn1 = l1.batchnorm.bn
        l1.batchnorm._tick()
        fd = ops.VNLinearLeakyOutFn.apply(h, l1.map_to_feat.weight, l1.map_to_dir.weight, bn1.weight, bn1.bias, bn1.running_mean,
                                          bn1.running_var, l2.map_to_feat.weight, None, S, l1.training, l1.negative_slope)
        return fd.view(-1, 1)                                                                        # [P*3, 1]

    def forward(self, coarse, feature_global, rot=None):
        bs, N, _ = coarse.shape
        S = self.grid_size ** 2
        coarse = coarse.float()
        g = self.downsize_global.forward_rows(global_to_rows(feature_global.float()))                # [bs*3, 384]
        # pcn.py:466-469: the centres are added through expand(384) + RESHAPE (memory-order re-chunking) + transpose, which works out to
        # token (b, n), feature (c, t) <- coarse[b].flatten()[t N + n], the same for every channel c; with the per-sample part g (the
        # down-sized global feature, feature index c*3+t, pcn.py:467) the block input in the rows layout is
        #   rows[(b, n, t), c] = s[b, n, t] + g[(b, t), c],   s[b, n, t] = coarse[b].flatten()[t N + n]
        # i.e. a one-channel VNLinear with weight 1 and a per-sample bias: one K = 1 product (vnp_gemm_rank1) instead of two 75 MB copies and
        # an addition forward, three full-size sums backward (tests/test_gpu_attention_emd.py::test_attention_decoder_matches_reference)
        s = coarse.reshape(bs, 3, N).transpose(1, 2).reshape(bs * N * 3, 1)
        if self._ones is None or self._ones.device != coarse.device:
            self._ones = torch.ones((384, 1), device=coarse.device, dtype=torch.float32)
        rows = ops.VNLinearFn.apply(s.contiguous(), self._ones, g, N * 3)                            # [bs*N*3, 384]
        carry = None
        for blk in self.transformer:                                     # a block's output stays the pair (skip stream, MLP branch) until
            rows, carry = blk.forward_rows_split(rows, bs, N, carry=carry)   # the next LayerNorm pass (or the add below) sums it
        if carry is not None:
            rows = rows + carry
        # rows now hold feature_global.reshape(bs,N,-1,3).permute(0,2,3,1) == per-centre features [bs*N, 384, 3]
        dev = coarse.device
        if self.folding_seed.device != dev:
            self.folding_seed = self.folding_seed.to(dev)
        seed_rows = self.folding_seed.t().reshape(1, S * 3, 1).expand(bs * N, -1, -1).reshape(bs * N * S * 3, 1).contiguous()
        fd1 = self._fold(self.vn_folding1, seed_rows, rows, S)
        fd2 = self._fold(self.vn_folding2, fd1, rows, S)                                             # [(bs*N*S)*3, 1]
        rel = fd2.reshape(bs, N, S, 3)
        return (rel + coarse.unsqueeze(2)).reshape(bs, N * S, 3)
