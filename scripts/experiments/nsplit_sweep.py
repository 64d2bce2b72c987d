#!/usr/bin/env python3
"""Split-K sweep: wall time (HIP events, GEMM + slab reduction) of ops.gemm for the small products of configs 2 and 5 as a function of
nsplit.  usage: python scripts/experiments/nsplit_sweep.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vn_pointcloudcompletion_amd import ops

SHAPES = [  # (tag, op_a, op_b, M, N, K)
    ("cfg5 block wgrad  TN", 1, 1, 384, 128, 12288), ("cfg5 block wgrad  TN", 1, 1, 1152, 128, 12288), ("cfg5 block wgrad  TN", 1, 1, 128, 384, 12288),
    ("cfg5 block wgrad  TN", 1, 1, 512, 128, 12288), ("cfg5 block wgrad  TN", 1, 1, 256, 256, 12288),
    ("cfg2 mlp fwd      NT", 0, 0, 96, 2048, 2048), ("cfg2 mlp dgrad    NN", 0, 1, 96, 2048, 2048), ("cfg2 mlp fwd      NT", 0, 0, 96, 1024, 1024),
    ("cfg2 bias fwd     NT", 0, 0, 96, 2048, 512), ("cfg2 bias dgrad    NN", 0, 1, 96, 512, 1024),
    ("cfg5 block fwd    NT", 0, 0, 12288, 1152, 128), ("cfg5 block fwd    NT", 0, 0, 12288, 128, 384), ("cfg5 block dgrad  NN", 0, 1, 12288, 128, 1152),
]


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for tag, oa, ob, M, N, K in SHAPES:
    A = torch.randn((K, M) if oa else (M, K), device="cuda")
    B = torch.randn((K, N) if ob else (N, K), device="cuda")
    out = torch.empty(M, N, device="cuda")
    auto = timeit(lambda: ops.gemm(A, B, op_a=oa, op_b=ob, out=out))
    res = []
    for ns in (1, 2, 4, 8, 16, 32, 64, 96, 128, 192):
        if ns > (K + 31) // 32:
            continue
        res.append((ns, timeit(lambda: ops.gemm(A, B, op_a=oa, op_b=ob, out=out, nsplit=ns))))
    best = min(res, key=lambda r: r[1])
    print(f"{tag} M={M:6d} N={N:5d} K={K:6d}: auto {auto:6.1f} us | " + " ".join(f"{ns}:{t:.1f}" for ns, t in res) + f" | best nsplit {best[0]} {best[1]:.1f} us", flush=True)
