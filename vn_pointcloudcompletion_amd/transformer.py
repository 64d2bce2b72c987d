"""VN transformer block used by Attention_VN_FoldingNet: reference class names, constructor signatures and ``state_dict``
keys (/root/reference/models/transformer.py:25-106) on libvnpcc (VNLayerNorm kernel, batched fp32-MFMA attention)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .vn_layers import VNLayerNorm, VNLinear, VNLinearLeakyReLU, mean_pool


def tokens_to_rows(x, C):
    """[B, N, C*3] tokens (feature index c*3+t, transformer.py:60) -> rows [B*N*3, C]."""
    B, N, _ = x.shape
    return x.reshape(B, N, C, 3).transpose(2, 3).reshape(B * N * 3, C).contiguous()


def rows_to_tokens(rows, B, N):
    C = rows.shape[1]
    return rows.reshape(B, N, 3, C).transpose(2, 3).reshape(B, N, C * 3).contiguous()


def graph_feature_rows(x_rows, knn_index, B, N, k):
    """transformer.py:6-22 on rows: cat(feature - x, x) for the given flat neighbour ids [B*k*N] (b, j, n order)."""
    nbr = knn_index.reshape(B, k, N).permute(0, 2, 1).reshape(B * N, k).int().contiguous()
    C = x_rows.shape[1]
    nb = ops.GatherPointsFn.apply(x_rows, nbr.reshape(-1))                  # [(B*N*k)*3, C]
    xc = x_rows.reshape(B * N, 1, 3, C).expand(-1, k, -1, -1).reshape(B * N * k * 3, C)
    return torch.cat([nb - xc, xc], dim=1)


class Attention(nn.Module):
    """transformer.py:76-106."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)      # unused by forward in the reference as well
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)                        # unused by forward in the reference as well
        self.proj_drop = nn.Dropout(proj_drop)
        self.proj_vnq = VNLinear(dim, dim)
        self.proj_vnk = VNLinear(dim, dim)
        self.proj_vnv = VNLinear(dim, dim)
        ops.pack_hint(self.proj_vnq.map_to_feat.weight, self.proj_vnk.map_to_feat.weight, self.proj_vnv.map_to_feat.weight)
        self.proj_vn = VNLinear(dim, dim)
        if attn_drop or proj_drop:
            raise NotImplementedError("dropout > 0 is not used on the PCNNet paths (rates are 0 in models/pcn.py:408-412)")

    def forward_rows(self, rows, B, N, residual=None):
        # one projection GEMM over [Wq; Wk; Wv] (the three VNLinear modules keep their state_dict keys), attention on the packed result;
        # ``residual``: the block's skip connection, added in the output projection's epilogue
        Wqkv = ops.PackRowsFn.apply(self.proj_vnq.map_to_feat.weight, self.proj_vnk.map_to_feat.weight, self.proj_vnv.map_to_feat.weight)
        qkv = ops.VNLinearFn.apply(rows, Wqkv, None, 0)
        x = ops.VNAttentionPackedFn.apply(qkv, B, N, self.num_heads, self.scale)
        return self.proj_vn.forward_rows(x, residual=residual)

    def forward(self, vn_x):
        from .vn_layers import from_rows, to_rows
        rows, meta = to_rows(vn_x)
        return from_rows(self.forward_rows(rows, vn_x.shape[0], vn_x.shape[3]), meta)


class VN_Block(nn.Module):
    """transformer.py:25-74."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.norm1 = VNLayerNorm(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = VNLayerNorm(dim)
        self.conv1 = VNLinearLeakyReLU(dim * 2, dim)
        self.conv2 = VNLinear(dim * 2, dim)
        self.conv3 = VNLinearLeakyReLU(dim, dim * 2, dim=4)
        self.conv4 = VNLinearLeakyReLU(dim * 2, dim, dim=4)
        self.pool1 = mean_pool

    def forward_rows_split(self, vn_x, B, N, knn_index=None, k=8, carry=None):
        """The block on the input vn_x (+ carry) as the pair (skip stream, MLP branch) whose SUM is its output (see the VN-PoinTr block,
        vn_pointr.VN_Block.forward_rows_split): the adds ride in the LayerNorm passes and the GEMM epilogues."""
        s, norm_x = self.norm1.forward_rows_carry(vn_x, carry)
        if knn_index is None:
            vn_x = self.attn.forward_rows(norm_x, B, N, residual=s)
        else:
            x_1 = self.attn.forward_rows(norm_x, B, N)
            f = graph_feature_rows(norm_x, knn_index, B, N, k)              # [(B*N*k)*3, 2C]
            f = self.conv1.forward_rows(f)
            f = ops.VNMeanPoolFn.apply(f, B * N, k)
            vn_x = self.conv2.forward_rows(torch.cat([x_1, f], dim=1), residual=s)
        vn_x, n2 = self.norm2.forward_rows_carry(vn_x)
        return vn_x, self.conv4.forward_rows(self.conv3.forward_rows(n2))

    def forward_rows(self, vn_x, B, N, knn_index=None, k=8):
        a, b = self.forward_rows_split(vn_x, B, N, knn_index, k)
        return a + b

    def forward(self, x, knn_index=None):
        B, N, _ = x.shape
        rows = tokens_to_rows(x.float(), self.dim)
        return rows_to_tokens(self.forward_rows(rows, B, N, knn_index), B, N)
