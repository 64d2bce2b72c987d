#!/usr/bin/env python3
"""The dense Chamfer search alone (B = 32, 16384 x 16384, uniform clouds like the bench's) — for `rocprofv3 --pmc ... -- python3 <this>` and for
wall timing: python3 scripts/experiments/chamfer_pmc.py [iters]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vn_pointcloudcompletion_amd.ops.chamfer import _chamfer_forward

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = torch.Generator().manual_seed(1234)
a = (torch.rand(32, 16384, 3, generator=g) - 0.5).cuda()
b = (torch.rand(32, 16384, 3, generator=g) - 0.5).cuda()
# a second pair shaped like a trained decoder's output: points on a thin shell around the target's points
c = (b + 0.01 * torch.randn(32, 16384, 3, generator=g).cuda()).contiguous()
from vn_pointcloudcompletion_amd import _lib
import ctypes
setv = getattr(_lib.lib(), "vnp_chamfer_set_variant", None)
variants = tuple(range(8)) if setv is not None else (0,)          # a library built with the experiment's vnp_chamfer_set_variant switch
# a blob (what an untrained decoder emits) against a spread target: the heavy-tailed case of the bench's first steps
blob = (0.05 * torch.randn(32, 16384, 3, generator=g)).cuda().contiguous()
for name, (x, y) in {"uniform vs uniform": (a, b), "near copy vs target": (c, b), "blob vs uniform": (blob, b)}.items():
    for rnd in range(3):
        for v in variants:
            if setv is not None:
                setv(ctypes.c_int(v))
            for _ in range(2):
                _chamfer_forward(x, y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                d1, d2, i1, i2 = _chamfer_forward(x, y)
            e1.record()
            torch.cuda.synchronize()
            print(f"{name}: variant {v} round {rnd}: {e0.elapsed_time(e1) / iters:.3f} ms per call (bin + two pruned scans), mean d1 {d1.mean().item():.3e}, checksum {int(i1.long().sum().item()) + int(i2.long().sum().item())} {d1.double().sum().item() + d2.double().sum().item():.12e}")
if setv is not None:
    setv(ctypes.c_int(0))
