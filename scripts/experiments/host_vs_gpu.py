#!/usr/bin/env python3
"""Host enqueue time against GPU time per step: if the host needs almost as long to ENQUEUE a step as the GPU needs to run it, the queue
runs dry and the step is host-bound.  usage: host_vs_gpu.py [configs...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.optim import FusedAdam
from vn_pointcloudcompletion_amd.train import settle_gc, train_step
CONFIGS = {2: dict(enc="vn_pointnet", dec="vn_foldingnet", B=32, latent=2048, loss="cd"),
           3: dict(enc="vn_dgcnn_fps", dec="vn_foldingnet", B=32, latent=512, loss="cd", knn_k=20),
           4: dict(enc="vn_pointnet", dec="attention_vn_foldingnet", B=16, latent=2048, loss="emd"),
           5: dict(enc="vn_pointr", dec="vn_foldingnet", B=32, latent=1024, loss="cd")}
for cid in [int(a) for a in sys.argv[1:]] or [2, 3, 4, 5]:
    c = CONFIGS[cid]
    cfg = SimpleNamespace(num_coarse=1024, latent_dim=c["latent"], only_coarse=False, enc_pretrained="none", device="cuda:0",
                          coarse_loss=c["loss"], dcd_opts={"alpha": 200, "lambda": 0.5}, knn_k=c.get("knn_k", 16))
    torch.manual_seed(0)
    net = PCNNet(cfg, c["enc"], c["dec"]).train()
    md = getattr(net.decoder, "num_dense", 16384)
    opt = FusedAdam(net.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(1)
    p = (torch.rand(c["B"], 2048, 3, generator=g) - 0.5).cuda(); gt = (torch.rand(c["B"], md, 3, generator=g) - 0.5).cuda()
    for i in range(5):
        train_step(cfg, net, opt, p, gt)
        if i == 0: settle_gc()
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        train_step(cfg, net, opt, p, gt)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): train_step(cfg, net, opt, p, gt)
    torch.cuda.synchronize(); b2b = (time.perf_counter() - t0) / 20 * 1e3
    print(f"config {cid}: host enqueue {sorted(host)[5]:.2f} ms, one isolated step start to drained {sorted(total)[5]:.2f} ms, back-to-back {b2b:.2f} ms/step")
    del net, opt; torch.cuda.empty_cache()
