#!/usr/bin/env python3
"""Pruned Chamfer search against the brute-force kernel on random cases (sizes 1 .. 32768, uniform / clustered / surface / duplicated /
lattice clouds, a few non-finite coordinates): the four outputs bit for bit.  usage: chamfer_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vn_pointcloudcompletion_amd import _lib
from vn_pointcloudcompletion_amd._lib import call, ptr

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())


def cloud(B, n, kind):
    if kind == 0: return torch.rand(B, n, 3, generator=g) - 0.5
    if kind == 1: return 0.02 * torch.randn(B, n, 3, generator=g) + (torch.rand(B, 1, 3, generator=g) - 0.5)          # one blob per sample
    if kind == 2:                                                                                                      # a sphere shell
        v = torch.randn(B, n, 3, generator=g); return 0.4 * v / v.norm(dim=-1, keepdim=True).clamp_min(1e-9)
    if kind == 3:                                                                                                      # heavy duplication
        base = torch.rand(B, max(1, n // 7), 3, generator=g) - 0.5
        return base[:, torch.randint(0, base.shape[1], (n,), generator=g)]
    if kind == 4: return torch.randint(-4, 5, (B, n, 3), generator=g).float() * 0.125                                   # lattice: exact ties everywhere
    return torch.cat([torch.rand(B, n - n // 2, 3, generator=g) * 0.01, torch.rand(B, n // 2, 3, generator=g) * 100], 1)  # two scales


bad = 0
for case in range(cases):
    B = ri(1, 4)
    n = ri(1, 300) if case % 5 == 0 else ri(1, 32768) if case % 7 == 0 else ri(64, 9000)
    m = ri(1, 300) if case % 3 == 0 else ri(1, 32768) if case % 11 == 0 else ri(64, 9000)
    a, b = cloud(B, n, ri(0, 5)), cloud(B, m, ri(0, 5))
    if case % 13 == 0: a[0, ri(0, n - 1), ri(0, 2)] = float("nan")
    if case % 17 == 0: b[0, ri(0, m - 1), ri(0, 2)] = float("inf")
    a, b = a.cuda().contiguous(), b.cuda().contiguous()
    outs = []
    for pruned in (False, True):
        d1 = torch.empty(B, n, device="cuda"); d2 = torch.empty(B, m, device="cuda")
        i1 = torch.empty(B, n, device="cuda", dtype=torch.int32); i2 = torch.empty(B, m, device="cuda", dtype=torch.int32)
        if pruned:
            ws = torch.empty(_lib.lib().vnp_chamfer_pruned_ws_bytes(B, n, m), device="cuda", dtype=torch.uint8)
            call("vnp_chamfer_fwd_pruned", B, n, ptr(a), m, ptr(b), ptr(d1), ptr(i1), ptr(d2), ptr(i2), ptr(ws))
        else:
            ws = torch.empty(max(n, m) * B * 8, device="cuda", dtype=torch.uint8)
            call("vnp_chamfer_fwd", B, n, ptr(a), m, ptr(b), ptr(d1), ptr(i1), ptr(d2), ptr(i2), ptr(ws))
        outs.append((d1, d2, i1, i2))
    torch.cuda.synchronize()
    same = all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(*outs))
    if not same:
        bad += 1
        print(f"case {case}: B={B} n={n} m={m} MISMATCH", [int((x.view(torch.int32) != y.view(torch.int32)).sum()) for x, y in zip(*outs)])
print(f"{cases} cases, {bad} mismatches")
