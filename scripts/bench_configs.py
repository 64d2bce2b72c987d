#!/usr/bin/env python3
"""Step time of the other BASELINE configs (3, 4, 5) at their full sizes on ONE MI355X — parity-test cases, not the bench
line (bench.py measures config 2); this table is what DESIGN.md §6 quotes.
usage: python scripts/bench_configs.py [3 4 5] [--steps N] [--shapes]      (--shapes: kernel time of every matrix product of one step, by shape)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from types import SimpleNamespace
import torch
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.optim import FusedAdam
from vn_pointcloudcompletion_amd.train import settle_gc, train_step

CONFIGS = {
    2: dict(enc="vn_pointnet", dec="vn_foldingnet", B=32, latent=2048, loss="cd", mc=1024),
    3: dict(enc="vn_dgcnn_fps", dec="vn_foldingnet", B=32, latent=512, loss="cd", mc=1024, knn_k=20),
    4: dict(enc="vn_pointnet", dec="attention_vn_foldingnet", B=16, latent=2048, loss="emd", mc=1024),
    5: dict(enc="vn_pointr", dec="vn_foldingnet", B=32, latent=1024, loss="cd", mc=1024),   # B = 256 over 8 GPUs
}
argv = sys.argv[1:]
nsteps = 5
if "--steps" in argv:                       # profiling runs use fewer timed steps
    i = argv.index("--steps"); nsteps = int(argv[i + 1]); del argv[i:i + 2]
shapes = "--shapes" in argv
argv = [a for a in argv if a != "--shapes"]
which = [int(a) for a in argv] or [3, 4, 5]
for cid in which:
    c = CONFIGS[cid]
    cfg = SimpleNamespace(num_coarse=c["mc"], latent_dim=c["latent"], only_coarse=False, enc_pretrained="none", device="cuda:0",
                          coarse_loss=c["loss"], dcd_opts={"alpha": 200, "lambda": 0.5}, knn_k=c.get("knn_k", 16))
    torch.manual_seed(0)
    net = PCNNet(cfg, c["enc"], c["dec"]).train()
    md = getattr(net.decoder, "num_dense", 16384)
    opt = FusedAdam(net.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(1)
    p = (torch.rand(c["B"], 2048, 3, generator=g) - 0.5).cuda()
    gt = (torch.rand(c["B"], md, 3, generator=g) - 0.5).cuda()
    for _ in range(5):                       # warm-up (a one-off ~0.1 s stall half a second into a process's GPU work would land in a shorter one's timed steps)
        train_step(cfg, net, opt, p, gt)
    settle_gc()                              # as train.train() does after its first step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = nsteps
    for _ in range(n):
        loss, _, _ = train_step(cfg, net, opt, p, gt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"config {cid}: {c['enc']} + {c['dec']} ({c['loss']}), B={c['B']}, 2048->{md}: {dt * 1e3:.1f} ms/step, "
          f"{c['B'] * md / dt / 1e6:.2f} M points/s, loss {loss.item():.4f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    if shapes:
        from vn_pointcloudcompletion_amd import ops
        prof = []
        with ops.hooks(gemm_profile=prof):
            train_step(cfg, net, opt, p, gt)
            torch.cuda.synchronize()
        per = {}
        for variant, flops, nsplit, shape, timer in prof:
            if timer.seconds() is None:
                continue
            a = per.setdefault((variant, timer.kind, shape, nsplit), [0.0, 0.0, 0])
            a[0] += flops; a[1] += timer.seconds(); a[2] += 1
        print("  matrix-product kernels: %.2f ms/step in %d launches (kind: 0 split MFMA, 2 streaming, 3 screening pass)"
              % (sum(a[1] for a in per.values()) * 1e3, sum(a[2] for a in per.values())))
        for (variant, kind, shape, nsplit), a in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
            print(f"    {variant} kind={kind} M={shape[0]:8d} N={shape[1]:5d} K={shape[2]:8d} split={nsplit:4d} x{a[2]:2d} {a[1] * 1e3:7.3f} ms "
                  f"{a[0] / a[1] / 1e12:6.1f} TF/s")
    del net, opt
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
