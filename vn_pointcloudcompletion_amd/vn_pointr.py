"""VN-PoinTr encoder (BASELINE config 5): reference class names, constructor signatures, ``state_dict`` keys and init
(/root/reference/models/pointr/vn_pointr.py:101-160, 366-722; models/pointr/utils/dgcnn_group.py:112-247) on libvnpcc.

Only the branch that ``PCNNet`` instantiates is executable (``dgcnn='vn_dgcnn', trans='vn_trans', only_coarse=True``,
models/model.py:23-24).  The sub-modules that this branch constructs but never runs (``input_proj``, ``pos_embed``,
``vn_pos_embed``, the 8 ``VN_DecoderBlock``s, ``vn_mlp_query``) are built as parameter containers so
that checkpoints and the seeded initialisation match the reference tensor for tensor; their ``forward`` raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .dgcnn import batch_base, edge_conv_rows, fps_downsample_rows, knn_graph, sampled_coords
from .pcn import rows_to_global
from .vn_layers import (VNLayerNorm, VNLeakyReLU, VNLinear, VNLinearAndLeakyReLU, VNLinearLeakyReLU, VNMaxPool, mean_pool)


def feature_knn_graph(f_rows, B, N, k):
    """Dynamic kNN graph in feature space (dgcnn_group.py:188-196: knn(x, x) on x.view(B, C*3, N)): distances over all
    3*C feature components of a point, which is the rows layout viewed as [B, N, 3*C].  Returns global ids [B*N, k]."""
    with torch.no_grad():
        feats = f_rows.detach().reshape(B, N, -1)
        idx = ops.knn_indices(feats, feats, k)
        return (idx + batch_base(B, N, f_rows.device).view(B, 1, 1)).reshape(B * N, k)


class VN_DGCNN_Grouper(nn.Module):
    """dgcnn_group.py:112-247: 2048 points -> 128 proxies with 128 VN channels; every graph is built in FEATURE space."""
    K = 16

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Sequential(VNLinearLeakyReLU(2, 32))
        self.conv4 = VNLinearLeakyReLU(64, 64)
        self.conv5 = VNLinearLeakyReLU(128, 64)
        self.conv6 = VNLinearLeakyReLU(128, 128)
        self.pool1 = self.pool2 = self.pool3 = self.pool4 = mean_pool

    def forward_rows(self, coords, plan=None):
        """coords [B,N,3] -> (proxy coordinates [B,128,3], proxy features rows [B*128*3, 128]).  ``plan``: an ops.FpsPlan whose first two
        levels are this grouper's samplings (512 of the input, 128 of those)."""
        B, N, _ = coords.shape
        k = self.K
        own = plan is None
        if own:
            plan = ops.FpsPlan(coords, (512, 128))
        f = coords.reshape(B * N * 3, 1)
        f = edge_conv_rows(self.conv1[0], f, knn_graph(coords, k), k)           # features == coordinates: D = 3
        coor_q, f_q = fps_downsample_rows(coords, f, 512, plan.level(0))
        f = edge_conv_rows(self.conv4, f_q, feature_knn_graph(f_q, B, 512, k), k)
        f = edge_conv_rows(self.conv5, f, feature_knn_graph(f, B, 512, k), k)
        coor_q, f_q = fps_downsample_rows(coor_q, f, 128, plan.level(1))
        if own:
            plan.close()
        f = edge_conv_rows(self.conv6, f_q, feature_knn_graph(f_q, B, 128, k), k)
        return coor_q, f

    def forward(self, x):
        """x [B,3,N] -> (coor [B,3,128], f [B,128,3,128]) like the reference."""
        coor_q, f = self.forward_rows(x.transpose(1, 2).contiguous().float())
        B = x.shape[0]
        return coor_q.transpose(1, 2).contiguous(), ops.FromRowsFn.apply(f, B, 128)


class Attention(nn.Module):
    """vn_pointr.py:113-145: VN attention 128 -> 384 (4 heads of 96 channels x 3) -> 128."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)          # never used by forward (as in the reference)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)                            # never used by forward (as in the reference)
        self.proj_drop = nn.Dropout(proj_drop)
        self.proj_vnq = VNLinear(128, 384)
        self.proj_vnk = VNLinear(128, 384)
        self.proj_vnv = VNLinear(128, 384)
        ops.pack_hint(self.proj_vnq.map_to_feat.weight, self.proj_vnk.map_to_feat.weight, self.proj_vnv.map_to_feat.weight)
        self.proj_vn = VNLinear(384, 128)

    def forward_rows(self, rows, B, N, residual=None):
        # one projection GEMM over [Wq; Wk; Wv] (the three VNLinear modules keep their state_dict keys), attention on the packed result;
        # ``residual``: the block's skip connection, added in the output projection's epilogue
        Wqkv = ops.PackRowsFn.apply(self.proj_vnq.map_to_feat.weight, self.proj_vnk.map_to_feat.weight, self.proj_vnv.map_to_feat.weight)
        qkv = ops.VNLinearFn.apply(rows, Wqkv, None, 0)
        return self.proj_vn.forward_rows(ops.VNAttentionPackedFn.apply(qkv, B, N, self.num_heads, self.scale), residual=residual)


class CrossAttention(nn.Module):
    """vn_pointr.py:148-184 — parameter container (only reachable with only_coarse=False)."""

    def __init__(self, dim, out_dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads, self.dim, self.out_dim = num_heads, dim, out_dim
        self.scale = qk_scale or (out_dim // num_heads) ** -0.5
        self.q_map = nn.Linear(dim, out_dim, bias=qkv_bias)
        self.k_map = nn.Linear(dim, out_dim, bias=qkv_bias)
        self.v_map = nn.Linear(dim, out_dim, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(out_dim, out_dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, q, v):
        raise NotImplementedError("CrossAttention is unreachable from PCNNet (only_coarse=True, models/model.py:24)")


class VN_Block(nn.Module):
    """vn_pointr.py:366-412."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = VNLayerNorm(dim // 3)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = VNLayerNorm(dim // 3)
        self.conv1 = VNLinearLeakyReLU(256, 128)
        self.conv2 = VNLinear(256, 128)
        self.conv3 = VNLinearLeakyReLU(128, 256, dim=4)
        self.conv4 = VNLinearLeakyReLU(256, 128, dim=4)
        self.pool1 = mean_pool

    def forward_rows_split(self, vn_x, B, N, nbr=None, k=8, pos=None, carry=None):
        """The block on the input vn_x (+ pos + carry), returned as the pair (skip stream, MLP branch) whose SUM is the block's output:
        the caller hands the pair to the next block as (vn_x, carry), whose first LayerNorm pass adds it up together with the position
        embedding (ops.AddLayerNormFn); both skip connections are added in the epilogue of the VNLinear that ends their branch."""
        s, norm_x = self.norm1.forward_rows_carry(vn_x, pos, carry)
        if nbr is None:
            vn_x = self.attn.forward_rows(norm_x, B, N, residual=s)
        else:
            x_1 = self.attn.forward_rows(norm_x, B, N)
            knn_f = edge_conv_rows(self.conv1, norm_x, nbr, k)                      # graph feature + conv1 + mean pool
            vn_x = self.conv2.forward_rows(torch.cat([x_1, knn_f], dim=1), residual=s)
        vn_x, n2 = self.norm2.forward_rows_carry(vn_x)
        return vn_x, self.conv4.forward_rows(self.conv3.forward_rows(n2))

    def forward_rows(self, vn_x, B, N, nbr=None, k=8):
        a, b = self.forward_rows_split(vn_x, B, N, nbr, k)
        return a + b


class VN_DecoderBlock(nn.Module):
    """vn_pointr.py:256-333 — parameter container (only reachable with only_coarse=False)."""

    def __init__(self, dim, num_heads, dim_q=None, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.self_attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        dim_q = dim_q or dim
        self.norm_q = norm_layer(dim_q)
        self.norm_v = norm_layer(dim)
        self.attn = CrossAttention(dim, dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.merge_map = nn.Linear(dim * 2, dim)
        self.conv1 = VNLinearLeakyReLU(256, 128)
        self.conv2 = VNLinear(256, 128)
        self.conv3 = VNLinearLeakyReLU(256, 128)
        self.conv4 = VNLinear(256, 128)
        self.conv5 = VNLinearLeakyReLU(128, 256, dim=4)
        self.conv6 = VNLinearLeakyReLU(256, 128, dim=4)
        self.pool1 = VNMaxPool(128)
        self.pool2 = VNMaxPool(128)

    def forward(self, q, v, self_knn_index=None, cross_knn_index=None):
        raise NotImplementedError("VN_DecoderBlock is unreachable from PCNNet (only_coarse=True, models/model.py:24)")


class VN_PCTransformer(nn.Module):
    """vn_pointr.py:414-722."""

    def __init__(self, in_chans=3, embed_dim=768, depth=[6, 8], num_heads=4, mlp_ratio=2., qkv_bias=False, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., num_query=224, knn_layer=-1, dgcnn="vn_dgcnn", trans="vn_trans",
                 memory_profile=False, only_coarse=True):
        super().__init__()
        if dgcnn != "vn_dgcnn" or trans != "vn_trans" or not only_coarse:
            raise NotImplementedError("only the branch PCNNet instantiates is built (dgcnn='vn_dgcnn', trans='vn_trans', "
                                      "only_coarse=True; models/model.py:23-24)")
        self.dgcnn, self.trans, self.memory_profile, self.only_coarse = dgcnn, trans, memory_profile, only_coarse
        self.num_features = self.embed_dim = embed_dim
        self.knn_layer = knn_layer
        self.grouper = VN_DGCNN_Grouper()
        self.input_proj = nn.Sequential(nn.Conv1d(384, embed_dim, 1), nn.BatchNorm1d(embed_dim), nn.LeakyReLU(negative_slope=0.2),
                                        nn.Conv1d(embed_dim, embed_dim, 1))                         # unused
        self.vn_pos_embed = nn.Sequential(VNLinearAndLeakyReLU(1, 128, 4), VNLinear(128, 128))     # unused
        self.fourth_vn_pos_embed = nn.Sequential(VNLinearAndLeakyReLU(2, 128, 4), VNLinear(128, 128))
        self.vn_input_proj = nn.Sequential(VNLinearLeakyReLU(128, 128, 4), VNLinear(128, 128))
        self.pos_embed = nn.Sequential(nn.Conv1d(in_chans, 128, 1), nn.BatchNorm1d(128), nn.LeakyReLU(negative_slope=0.2),
                                       nn.Conv1d(128, embed_dim, 1))                                # unused
        self.encoder = nn.ModuleList([
            VN_Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                     drop=drop_rate, attn_drop=attn_drop_rate) for _ in range(depth[0])])
        self.decoder = nn.ModuleList([
            VN_DecoderBlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                            drop=drop_rate, attn_drop=attn_drop_rate) for _ in range(depth[1])])    # unused
        self.vn_increase_dim = nn.Sequential(VNLinearAndLeakyReLU(128, 1024, 4), VNLinear(1024, 1024))
        self.vn_global_pool = VNMaxPool(in_channels=1024)
        self.vn_coarse_pred = nn.Sequential(VNLinear(1024, 512), VNLeakyReLU(512), VNLinear(512, 1024))
        self.vn_mlp_query = nn.Sequential(VNLinearLeakyReLU(1025, 1024, dim=4), VNLinearLeakyReLU(1024, 1024, dim=4),
                                          VNLinear(1024, embed_dim // 3))                           # unused
        self.num_query = num_query
        self.apply(self._init_weights)

    def _init_weights(self, m):
        """vn_pointr.py:544-556 (timm's trunc_normal_ == torch.nn.init.trunc_normal_)."""
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, nn.Conv1d):
            nn.init.xavier_normal_(m.weight.data, gain=1)
        elif isinstance(m, nn.BatchNorm1d):
            nn.init.constant_(m.weight.data, 1)
            nn.init.constant_(m.bias.data, 0)

    def embed_rows(self, inpc, coor, f):
        """vn_pointr.py:585-637: (token rows x [B*nc*3, 128], position rows [B*nc*3, 128]) from the grouper's proxies."""
        B, nc = coor.shape[0], coor.shape[1]
        x = self.vn_input_proj[1].forward_rows(self.vn_input_proj[0].forward_rows(f))
        centers = inpc.mean(dim=1)                                              # [B,3]
        pos_in = torch.stack([coor, centers.unsqueeze(1).expand(-1, nc, -1)], dim=-1).reshape(B * nc * 3, 2)
        pos = self.fourth_vn_pos_embed[1].forward_rows(self.fourth_vn_pos_embed[0].forward_rows(pos_in))
        return x, pos

    def head_rows(self, x, B, nc):
        """vn_pointr.py:655-668, 700-719: vn_increase_dim -> vn_global_pool -> vn_coarse_pred on token rows x [B*nc*3, 128]:
        (coarse [B,1024,3], global_feature [B,1024,3,1])."""
        g = self.vn_increase_dim[0].forward_rows(x)
        pooled, _ = ops.VNLinearMaxPoolFn.apply(g, self.vn_increase_dim[1].map_to_feat.weight,
                                                self.vn_global_pool.map_to_dir.weight.detach(), B, nc)   # [B*3, 1024]
        h = self.vn_coarse_pred[0].forward_rows(pooled)
        h = self.vn_coarse_pred[1].forward_rows(h)
        h = self.vn_coarse_pred[2].forward_rows(h)                              # [B*3, 1024]
        return h.reshape(B, 3, -1).permute(0, 2, 1).contiguous(), rows_to_global(pooled, B)

    def forward(self, inpc):
        """inpc [B,N,3] -> (coarse [B, 1024+224, 3], global_feature [B,1024,3,1])."""
        B = inpc.shape[0]
        inpc = inpc.contiguous().float()
        # the grouper's two samplings and the 224 extra coarse points (vn_pointr.py:720) start now, on a side stream (ops.FpsPlan)
        plan = ops.FpsPlan(inpc, (512, 128), (224,))
        coor, f = self.grouper.forward_rows(inpc, plan)                         # [B,128,3], [B*128*3, 128]
        nc = coor.shape[1]
        x, pos = self.embed_rows(inpc, coor, f)
        nbr = knn_graph(coor, 8)                                                # get_knn_index(coor), k = 8 (vn_pointr.py:10,17-29)
        carry = None
        for i, blk in enumerate(self.encoder):                                  # x = blk(x + pos), vn_pointr.py:700-702
            x, carry = blk.forward_rows_split(x, B, nc, nbr if i < self.knn_layer else None, pos=pos, carry=carry)
        if carry is not None:
            x = x + carry
        coarse, global_feature = self.head_rows(x, B, nc)
        inp_sparse = sampled_coords(inpc, plan.level(2))
        coarse = torch.cat([coarse, inp_sparse], dim=1).contiguous()
        return coarse, global_feature
