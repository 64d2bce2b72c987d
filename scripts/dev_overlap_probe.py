#!/usr/bin/env python3
"""Probe (dev tool): can a weight-gradient GEMM (MFMA-bound, 2 waves/SIMD at 208 VGPRs) overlap with an HBM-bound BatchNorm backward
pass when they run on two streams?  Decoder-sized operands.  Prints each alone, back to back, and concurrent."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vn_pointcloudcompletion_amd import ops

P, C, K = 32 * 16384, 256, 256
gpd = torch.randn(P * 3, 2 * C, device="cuda")
x = torch.randn(P * 3, K, device="cuda")
pd = torch.randn(P * 3, 2 * C, device="cuda")
gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
_, mean, rstd = ops._bn_leaky_forward(pd[:, :C], pd[:, C:], C, gamma, beta, rm, rv, True, 0.2)
grow, gcol = torch.randn(P * 3, device="cuda"), torch.randn(C, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def gemm_():
    return ops.gemm(gpd, x, op_a=1, op_b=1)

def bn_():
    return ops._bn_leaky_backward(None, pd[:, :C], pd[:, C:], C, gamma, beta, mean, rstd, True, 0.2, go_row=grow, go_col=gcol, want_gw=True)

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def both():
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sa):
        a = gemm_()
    with torch.cuda.stream(sb):
        b = bn_()
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
    return a, b

with torch.no_grad():
    tg, tb = timed(gemm_), timed(bn_)
    ts = timed(lambda: (gemm_(), bn_()))
    tc = timed(both)
print(f"wgrad GEMM alone {tg:.3f} ms; BN backward alone {tb:.3f} ms; back to back {ts:.3f} ms; on two streams {tc:.3f} ms")
