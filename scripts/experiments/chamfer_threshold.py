#!/usr/bin/env python3
"""Brute-force vs pruned Chamfer forward on the coarse-loss shapes (1024 / 1248 predicted points against 16384 / 19968 targets)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vn_pointcloudcompletion_amd import ops
from vn_pointcloudcompletion_amd.chamfer_distance import chamfer_3DDist

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

g = torch.Generator().manual_seed(0)
for (B, n, m) in [(32, 16384, 1024), (32, 19968, 1248), (32, 16384, 2048)]:
    gt = (torch.rand(B, n, 3, generator=g) - 0.5).cuda()
    pred = (torch.rand(B, m, 3, generator=g) * 0.6 - 0.3).cuda()
    cd = chamfer_3DDist()
    for thr in (4096, 512):
        ops.SW.CHAMFER_PRUNED_MIN_POINTS = thr
        us = t(lambda: cd(gt, pred))
        print(f"B={B} gt={n} pred={m} pruned-threshold {thr}: {us:.1f} us")
    ops.SW.CHAMFER_PRUNED_MIN_POINTS = 4096
