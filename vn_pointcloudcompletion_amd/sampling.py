"""Drop-in replacements for the two CUDA-only third-party packages the reference imports (README.md:29-31):
``knn_cuda.KNN`` and ``pointnet2_ops.pointnet2_utils.{furthest_point_sample, gather_operation}``, on libvnpcc."""
from __future__ import annotations

import torch

from . import ops


class KNN:
    """``KNN(k, transpose_mode=False)(ref[B,D,Nr], query[B,D,Nq]) -> (dist[B,k,Nq], idx[B,k,Nq])`` as used at
    models/dgcnn.py:11,236 (``transpose_mode=True``: inputs [B,N,D], outputs [B,Nq,k])."""

    def __init__(self, k, transpose_mode=False):
        self.k, self.transpose_mode = k, transpose_mode

    def __call__(self, ref, query):
        if not self.transpose_mode:
            ref, query = ref.transpose(1, 2), query.transpose(1, 2)
        idx, dist = ops.knn_indices(ref, query, self.k, return_dist=True)
        idx = idx.long()
        dist = dist.sqrt()
        if not self.transpose_mode:
            idx, dist = idx.transpose(1, 2).contiguous(), dist.transpose(1, 2).contiguous()
        return dist, idx


def furthest_point_sample(xyz, npoint):
    """xyz [B,N,3] -> idx [B,npoint] int32."""
    return ops.fps_indices(xyz, npoint)


def gather_operation(features, idx):
    """features [B,C,N], idx [B,n] -> [B,C,n] (index plumbing on coordinates / small tensors)."""
    return torch.gather(features, 2, idx.long().unsqueeze(1).expand(-1, features.shape[1], -1))


def gather_points(pc, idx):
    """pc [B,N,3], idx [B,n] -> [B,n,3]."""
    return torch.gather(pc, 1, idx.long().unsqueeze(-1).expand(-1, -1, pc.shape[2]))
