"""VN-DGCNN encoder with FPS down-sampling: reference class name, constructor signature and ``state_dict`` keys
(/root/reference/models/dgcnn.py:164-324), on libvnpcc kernels (kNN, FPS, gathers, vnp_gemm GEMMs, VN pointwise).

EdgeConv restructuring (same math): the reference builds the edge tensor cat(feature - x, x) of shape [B,2C,3,N,k]
(dgcnn.py:250-278) and applies W to it; here u = x W1^T and v = x (W2 - W1)^T are computed once per point and the edge
pre-activations are u[neighbour] + v[centre] (ops.EdgeGatherAddFn): k/2 = 8x fewer GEMM FLOPs and no 2C-wide edge tensor.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .pcn import rows_to_global
from .vn_layers import VNLinear, VNLinearLeakyReLU, VNMaxPool, mean_pool


def edge_conv_rows(layer: VNLinearLeakyReLU, x_rows, nbr, k):
    """VNLinearLeakyReLU(2C -> C') on the kNN edge tensor followed by mean_pool over the k neighbours.
    x_rows [P*3, C]; nbr [P,k] global neighbour ids.  Returns [P*3, C']."""
    C = x_rows.shape[1]
    Wf, Wd = layer.map_to_feat.weight, layer.map_to_dir.weight
    Cout = Wf.shape[0]
    bn = layer.batchnorm.bn
    layer.batchnorm._tick()
    if ops.SW.EDGE_FUSED:
        # ONE GEMM for u | v (x is read once, forward and backward), edge pre-activations u[nbr] + v recomputed inside each pass: the
        # [P*k*3, 2C'] tensor is never written
        Wuv = ops.EdgeWeightsFn.apply(Wf, Wd)                                              # [4C', C]: [Wu; Wv]
        if C == 1 and ops.SW.EDGE_RANK1 and nbr.shape[0] * 3 == x_rows.shape[0]:
            # the layer's input is the cloud itself: u | v = x Wuv^T is rebuilt from 12 bytes per point inside the passes
            return ops.EdgeConvRank1Fn.apply(x_rows, Wuv, nbr, bn.weight, bn.bias, bn.running_mean, bn.running_var, layer.training,
                                             layer.negative_slope)
        uv = ops.VNLinearFn.apply(x_rows, Wuv, None, 0)                                    # [P*3, 4C']: u (feat | dir), v (feat | dir)
        return ops.EdgeConvPackedFn.apply(uv, nbr, bn.weight, bn.bias, bn.running_mean, bn.running_var, layer.training,
                                          layer.negative_slope)
    Wu = torch.cat([Wf[:, :C], Wd[:, :C]], dim=0)                           # [2C', C]  (feat | dir)
    Wv = torch.cat([Wf[:, C:] - Wf[:, :C], Wd[:, C:] - Wd[:, :C]], dim=0)
    u = ops.VNLinearFn.apply(x_rows, Wu, None, 0)
    v = ops.VNLinearFn.apply(x_rows, Wv, None, 0)
    pd = ops.EdgeGatherAddFn.apply(u, v, nbr)                              # [P*k*3, 2C']
    f = ops.VNBnLeakyFn.apply(pd, True, bn.weight, bn.bias, bn.running_mean, bn.running_var, layer.training,
                              layer.negative_slope)                         # [P*k*3, C']
    return ops.VNMeanPoolFn.apply(f, x_rows.shape[0] // 3, k)


_BASES = {}


def batch_base(B, N, device):
    """int32 [B] = b * N (first global row id of cloud b), made once per (B, N, device) instead of an arange and a multiply per call."""
    key = (B, N, device)
    t = _BASES.get(key)
    if t is None:
        if len(_BASES) > 64:
            _BASES.clear()
        t = _BASES[key] = torch.arange(B, device=device, dtype=torch.int32) * N
    return t


def knn_graph(coords, k):
    """coords [B,N,3] -> global neighbour ids [B*N, k] int32 (no gradient, dgcnn.py:235-241)."""
    B, N, _ = coords.shape
    with torch.no_grad():
        idx = ops.knn_indices(coords, coords, k)                           # [B,N,k] local
        return (idx + batch_base(B, N, coords.device).view(B, 1, 1)).reshape(B * N, k)


def fps_downsample_rows(coords, f_rows, m, picked=None):
    """dgcnn.py:201-222: FPS on the coordinates, gather coordinates and features.  coords [B,N,3], f_rows [B*N*3, C].
    ``picked``: (idx, coordinates, global ids) of this level from an ops.FpsPlan started earlier."""
    B, N, _ = coords.shape
    if picked is None:
        with torch.no_grad():
            idx = ops.fps_indices(coords, m)                               # [B,m]
            gid = (idx + batch_base(B, N, coords.device).view(B, 1)).reshape(-1)
        new_coords = torch.gather(coords, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
    else:
        idx, new_coords, gid = picked
        if coords.requires_grad:                                           # a caller that differentiates through the sampled coordinates
            new_coords = torch.gather(coords, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
    return new_coords, ops.GatherPointsFn.apply(f_rows, gid)


def sampled_coords(coords, picked):
    """The coordinates of an ops.FpsPlan level (the 224 extra coarse points, dgcnn.py:322, vn_pointr.py:720).  The plan's own gather
    (vnp_gather_coords) has no autograd: a cloud that asks for a gradient gets the differentiable torch.gather of the reference."""
    idx, xyz_m, _ = picked
    if coords.requires_grad:
        return torch.gather(coords, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
    return xyz_m


class VN_DGCNN_fps(nn.Module):
    K = 16       # hard-coded in the reference (dgcnn.py:11,230,251); exposed as an attribute for the k=20 benchmark config

    def __init__(self, config, only_coarse=False):
        super().__init__()
        self.num_coarse = config.num_coarse // 2 if config.num_coarse == 448 else config.num_coarse
        self.k = getattr(config, "knn_k", self.K)
        self.conv1 = nn.Sequential(VNLinearLeakyReLU(2, 32))
        self.conv4 = VNLinearLeakyReLU(64, 64)
        self.conv5 = VNLinearLeakyReLU(128, 128)
        self.conv6 = VNLinearLeakyReLU(256, 512)
        self.conv7 = nn.Sequential(VNLinearLeakyReLU(512, 1024, dim=4), VNLinear(1024, self.num_coarse))
        self.pool1 = self.pool2 = self.pool3 = self.pool4 = mean_pool
        self.pool5 = VNMaxPool(512)

    def forward(self, x):
        B, N, _ = x.shape
        k = self.k
        coords = x.contiguous().float()
        # all samplings (512 of the input, 128 of those, the 224 extra coarse points) start now, on a side stream (ops.FpsPlan)
        plan = ops.FpsPlan(coords, (512, 128), (224,) if self.num_coarse == 224 else ())
        f = coords.reshape(B * N * 3, 1)
        f = edge_conv_rows(self.conv1[0], f, knn_graph(coords, k), k)                  # dynamic graph == coordinates here
        coor_q, f_q = fps_downsample_rows(coords, f, 512, plan.level(0))
        nbr = knn_graph(coor_q, k)
        f = edge_conv_rows(self.conv4, f_q, nbr, k)
        f = edge_conv_rows(self.conv5, f, nbr, k)
        coor_q, f_q = fps_downsample_rows(coor_q, f, 128, plan.level(1))
        f = edge_conv_rows(self.conv6, f_q, knn_graph(coor_q, k), k)                   # [B*128*3, 512]
        g = self.pool5.forward_rows(f, B, 128)                                         # [B*3, 512]
        h = self.conv7[0].forward_rows(g)
        h = self.conv7[1].forward_rows(h)                                              # [B*3, Mc]
        coarse = h.reshape(B, 3, self.num_coarse).permute(0, 2, 1).contiguous()
        global_feature = rows_to_global(g, B)
        if self.num_coarse == 224:
            inp_sparse = sampled_coords(coords, plan.level(2))
            coarse_cat = torch.cat([coarse, inp_sparse], dim=1).contiguous()
            return (coarse, coarse_cat), global_feature
        plan.close()
        return coarse, global_feature
