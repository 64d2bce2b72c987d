"""Own counterparts of the three pytorch3d.transforms names the reference's loops use (train.py:20,131-138,
test.py:14,59-65): row-vector convention, ``points @ R``."""
from __future__ import annotations

import math

import torch


def random_rotations(n, dtype=torch.float32, device=None, generator=None):
    """n uniformly random rotation matrices [n,3,3] (normalised Gaussian quaternions, w >= 0)."""
    q = torch.randn((n, 4), dtype=dtype, device=device, generator=generator)
    q = q / q.norm(dim=1, keepdim=True)
    q = torch.where(q[:, :1] < 0, -q, q)
    r, i, j, k = q.unbind(1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1)
    return o.reshape(n, 3, 3)


class Rotate:
    """``Rotate(R)``: ``transform_points(p[B|1,S,3]) -> p @ R`` ([B,S,3])."""

    def __init__(self, R, device=None):
        self.R = R if device is None else R.to(device)

    def to(self, device):
        return Rotate(self.R.to(device))

    def get_matrix(self):
        return self.R

    def transform_points(self, points):
        R = self.R.to(points.dtype)
        if points.dim() == 2:
            return points @ R[0]
        if points.shape[0] == 1 and R.shape[0] != 1:
            points = points.expand(R.shape[0], -1, -1)
        return torch.matmul(points, R)


class RotateAxisAngle(Rotate):
    """Rotation about a coordinate axis by ``angle`` (degrees by default), pytorch3d convention."""

    def __init__(self, angle, axis="X", degrees=True, device=None):
        angle = torch.as_tensor(angle, dtype=torch.float32).reshape(-1)
        if degrees:
            angle = angle * math.pi / 180.0
        c, s = torch.cos(angle), torch.sin(angle)
        one, zero = torch.ones_like(c), torch.zeros_like(c)
        axis = axis.upper()
        if axis == "X":
            rows = (one, zero, zero, zero, c, -s, zero, s, c)
        elif axis == "Y":
            rows = (c, zero, s, zero, one, zero, -s, zero, c)
        elif axis == "Z":
            rows = (c, -s, zero, s, c, zero, zero, zero, one)
        else:
            raise ValueError(f"axis must be X, Y or Z, got {axis}")
        R = torch.stack(rows, -1).reshape(-1, 3, 3).transpose(1, 2)     # row-vector convention: p @ R
        super().__init__(R, device)
