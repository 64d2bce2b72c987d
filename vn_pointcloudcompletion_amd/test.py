"""Evaluation loop with the reference's semantics (/root/reference/test.py:33-136): per-category L1-CD (x1e3),
L2-CD (x1e4) and F-score@0.01 (x1e2).  F-score is computed on the device from the Chamfer distances
(metric.f_score / metric.fscore) instead of per-sample open3d KD-trees; the voxel IoU column (pyntcloud) is not
produced (CPU eval metric, out of scope)."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.utils.data as Data

from .chamfer_distance import ChamferDistance
from .dataset import make_dataset
from .model import PCNNet
from .train import make_rotation

CATEGORIES_PCN = ["airplane", "cabinet", "car", "chair", "lamp", "sofa", "table", "vessel"]
CATEGORIES_PCN_NOVEL = ["bus", "bed", "bookshelf", "bench", "guitar", "motorbike", "skateboard", "pistol"]


def test_single_category(category, model, config, save=True):
    dataset = make_dataset(config, "test", category)
    loader = Data.DataLoader(dataset, batch_size=config.batch_size, shuffle=False)
    cd = ChamferDistance()
    tot = torch.zeros(3, device=config.device)                 # l1_cd, l2_cd, f-score sums (on device)
    with torch.no_grad():
        for p, c in loader:
            p, c = p.to(config.device), c.to(config.device)
            trot = make_rotation(config.test_rotation, p.shape[0], config.device)
            if trot is not None:
                p, c = trot.transform_points(p), trot.transform_points(c)
            if not config.only_coarse:
                _, c_ = model(p, trot)
            else:
                c_, _ = model(p, trot)
            d1, d2 = cd(c_, c)
            tot[0] += torch.sum(torch.mean(torch.sqrt(d1), 1) + torch.mean(torch.sqrt(d2), 1)) / 2      # metric.py:19-23
            tot[1] += torch.sum(torch.mean(d1, 1) + torch.mean(d2, 1))                                 # metric.py:12-16
            prec, rec = (d1 < 1e-4).float().mean(1), (d2 < 1e-4).float().mean(1)                        # th = 0.01 -> th^2
            f = 2 * rec * prec / (rec + prec)
            tot[2] += torch.nan_to_num(f, nan=0.0).sum()
    l1, l2, fs = (tot / len(dataset)).tolist()
    return l1, l2, fs


def test(config, save=False):
    print(config.name)
    print(f"Rotation used during testing: {config.test_rotation}")
    model = PCNNet(config, enc_type=config.enc_type, dec_type=config.dec_type)
    ckpt_path = os.path.join(config.exp_dir, "models/model_best.pth")
    model.load_state_dict(torch.load(ckpt_path, map_location=config.device))
    model.eval()
    print("\033[33m{:20s}{:20s}{:20s}{:20s}\033[0m".format("Category", "L1_CD(1e-3)", "L2_CD(1e-4)", "FScore-0.01(%)"))
    print("\033[33m{:20s}{:20s}{:20s}{:20s}\033[0m".format("--------", "-----------", "-----------", "--------------"))
    categories = CATEGORIES_PCN if config.category == "all" else [config.category]
    rows = []
    for category in categories:
        l1, l2, fs = test_single_category(category, model, config, save)
        print("{:20s}{:<20.4f}{:<20.4f}{:<20.4f}".format(category.title(), 1e3 * l1, 1e4 * l2, 1e2 * fs))
        rows.append((l1, l2, fs))
    if len(rows) > 1:
        m = np.mean(np.array(rows), axis=0)
        print("\033[32m{:20s}{:<20.4f}{:<20.4f}{:<20.4f}\033[0m".format("Average", m[0] * 1e3, m[1] * 1e4, m[2] * 1e2))
    return rows
