"""Datasets for the train / test loops.

``ShapeNet`` mirrors /root/reference/dataset/shapenet.py:60-102 (PCN ShapeNet: partial clouds resampled to 2048 points,
complete clouds to 16384) with a small built-in PLY reader instead of open3d.  ``SyntheticShapes`` produces the
uniform synthetic clouds used by the benchmark and the tests: partial U(-0.5,0.5)^{2048x3},
complete U(-0.5,0.5)^{16384x3}; there is no dataset on the GPU box.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch
import torch.utils.data as data


def read_ply_xyz(path):
    """Vertex positions of an ASCII or binary-little-endian PLY file as float32 [N,3]."""
    with open(path, "rb") as f:
        header = []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            header.append(line.decode("ascii", "replace").strip())
            if header[-1] == "end_header":
                break
        fmt = [h for h in header if h.startswith("format")][0].split()[1]
        n = 0
        props, in_vertex = [], False
        for h in header:
            tok = h.split()
            if tok[:2] == ["element", "vertex"]:
                n, in_vertex = int(tok[2]), True
            elif tok and tok[0] == "element":
                in_vertex = False
            elif tok and tok[0] == "property" and in_vertex:
                props.append((tok[1], tok[2]))
        names = [p[1] for p in props]
        ix = [names.index(c) for c in ("x", "y", "z")]
        if fmt == "ascii":
            arr = np.loadtxt(f, max_rows=n, ndmin=2)
            return arr[:, ix].astype(np.float32)
        np_types = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
                    "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
                    "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}
        if fmt != "binary_little_endian":
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        dt = np.dtype([(nm, np_types[ty]) for ty, nm in props])
        rec = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
        return np.stack([rec["x"], rec["y"], rec["z"]], 1).astype(np.float32)


class ShapeNet(data.Dataset):
    """dataset/shapenet.py:60-102.  split in {train, valid, test, test_novel}; category 'all' or a category name."""

    CATEGORY_IDS = {"airplane": "02691156", "cabinet": "02933112", "car": "02958343", "chair": "03001627",
                    "lamp": "03636649", "sofa": "04256520", "table": "04379243", "vessel": "04530566",
                    "bus": "02924116", "bed": "02818832", "bookshelf": "02871439", "bench": "02828884",
                    "guitar": "03467517", "motorbike": "03790512", "skateboard": "04225987", "pistol": "03948459"}

    def __init__(self, dataroot, split, category):
        assert split in ["train", "valid", "test", "test_novel"], "split error value!"
        self.dataroot, self.split, self.category = dataroot, split, category
        self.partial_paths, self.complete_paths = self._load_data()

    def __getitem__(self, index):
        if self.split == "train":
            partial_path = self.partial_paths[index].format(random.randint(0, 7))
        else:
            partial_path = self.partial_paths[index]
        partial_pc = self.random_sample(read_ply_xyz(partial_path), 2048)
        complete_pc = self.random_sample(read_ply_xyz(self.complete_paths[index]), 16384)
        return torch.from_numpy(partial_pc), torch.from_numpy(complete_pc)

    def __len__(self):
        return len(self.complete_paths)

    def _load_data(self):
        with open(os.path.join(self.dataroot, f"{self.split}.list"), "r") as f:
            lines = f.read().splitlines()
        if self.category != "all":
            lines = [ln for ln in lines if ln.startswith(self.CATEGORY_IDS[self.category])]
        partial, complete = [], []
        for ln in lines:
            cat, model_id = ln.split("/")
            if self.split == "train":
                partial.append(os.path.join(self.dataroot, self.split, "partial", cat, model_id + "_{}.ply"))
            else:
                partial.append(os.path.join(self.dataroot, self.split, "partial", cat, model_id + ".ply"))
            complete.append(os.path.join(self.dataroot, self.split, "complete", cat, model_id + ".ply"))
        return partial, complete

    @staticmethod
    def random_sample(pc, n):
        idx = np.random.permutation(pc.shape[0])
        if idx.shape[0] < n:
            idx = np.concatenate([idx, np.random.randint(pc.shape[0], size=n - pc.shape[0])])
        return pc[idx[:n]]


class SyntheticShapes(data.Dataset):
    """Uniform random clouds of the PCN shapes: (partial [n_partial,3], complete [n_complete,3]) float32."""

    def __init__(self, length=64, n_partial=2048, n_complete=16384, seed=1234):
        self.length, self.n_partial, self.n_complete, self.seed = length, n_partial, n_complete, seed

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        return (torch.rand(self.n_partial, 3, generator=g) - 0.5, torch.rand(self.n_complete, 3, generator=g) - 0.5)


def make_dataset(config, split, category="all"):
    """``config.data_type == 'synthetic'`` selects SyntheticShapes; anything else the PCN ShapeNet layout under data/PCN
    (train.py:52-53)."""
    if getattr(config, "data_type", "shapenet") == "synthetic":
        length = getattr(config, "synthetic_length", 64)
        if split != "train":
            length = max(length // 4, 1)
        return SyntheticShapes(length, getattr(config, "synthetic_partial", 2048), getattr(config, "synthetic_dense", 16384),
                               seed=1234 + {"train": 0, "valid": 1, "test": 2, "test_novel": 3}[split])
    return ShapeNet("data/PCN", split, category)
