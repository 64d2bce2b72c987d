#!/usr/bin/env python3
"""Chunks scanned per query wave of the pruned Chamfer search (B = 32, 16384 x 16384): distribution per cloud pair type."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vn_pointcloudcompletion_amd import _lib
from vn_pointcloudcompletion_amd._lib import call, ptr
g = torch.Generator().manual_seed(1234)
a = (torch.rand(32, 16384, 3, generator=g) - 0.5).cuda()
b = (torch.rand(32, 16384, 3, generator=g) - 0.5).cuda()
c = (b + 0.01 * torch.randn(32, 16384, 3, generator=g).cuda()).contiguous()
blob = (0.05 * torch.randn(32, 16384, 3, generator=g)).cuda().contiguous()
for name, (x, y) in {"uniform vs uniform": (a, b), "near copy vs target": (c, b), "blob vs uniform": (blob, b)}.items():
    B, n, m = x.shape[0], x.shape[1], y.shape[1]
    L = _lib.lib()
    ws = torch.zeros(L.vnp_chamfer_pruned_ws_bytes(B, n, m), device="cuda", dtype=torch.uint8)
    d1 = torch.empty(B, n, device="cuda"); d2 = torch.empty(B, m, device="cuda")
    i1 = torch.empty(B, n, device="cuda", dtype=torch.int32); i2 = torch.empty(B, m, device="cuda", dtype=torch.int32)
    call("vnp_chamfer_fwd_pruned", B, n, ptr(x), m, ptr(y), ptr(d1), ptr(i1), ptr(d2), ptr(i2), ptr(ws))
    torch.cuda.synchronize()
    off = L.vnp_chamfer_pruned_ws_wave_counts(B, n, m)
    wc = ws[off:off + 4 * B * (n // 64 + m // 64)].view(torch.int32).cpu().numpy()
    for d, w in (("1 (x -> y)", wc[:B * (n // 64)]), ("2 (y -> x)", wc[B * (n // 64):])):
        q = np.percentile(w, [50, 90, 99, 99.9, 100])
        print(f"{name}, direction {d}: mean {w.mean():.1f} chunks, p50 {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} p99.9 {q[3]:.0f} max {q[4]:.0f}; waves above 64: {(w > 64).sum()} of {w.size}")
