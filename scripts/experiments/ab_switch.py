#!/usr/bin/env python3
"""A/B of one ops.SW switch in one process: step time of configs 2..5 with the switch at two values, alternating blocks of steps.
usage: ab_switch.py NAME VALUE_A VALUE_B [configs...]        (values are Python literals)"""
import ast, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from types import SimpleNamespace
import torch
from vn_pointcloudcompletion_amd import ops, train as T
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.optim import FusedAdam
CONFIGS = {
    2: dict(enc="vn_pointnet", dec="vn_foldingnet", B=32, latent=2048, loss="cd", mc=1024),
    3: dict(enc="vn_dgcnn_fps", dec="vn_foldingnet", B=32, latent=512, loss="cd", mc=1024, knn_k=20),
    4: dict(enc="vn_pointnet", dec="attention_vn_foldingnet", B=16, latent=2048, loss="emd", mc=1024),
    5: dict(enc="vn_pointr", dec="vn_foldingnet", B=32, latent=1024, loss="cd", mc=1024),
}
name, va, vb = sys.argv[1], ast.literal_eval(sys.argv[2]), ast.literal_eval(sys.argv[3])
for cid in [int(a) for a in sys.argv[4:]] or [2, 3, 4, 5]:
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
    for _ in range(5):
        T.train_step(cfg, net, opt, p, gt)
    res = {0: [], 1: []}
    for rnd in range(6):
        for which, val in enumerate((va, vb)):
            setattr(ops.SW, name, val)
            T.train_step(cfg, net, opt, p, gt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                T.train_step(cfg, net, opt, p, gt)
            torch.cuda.synchronize()
            res[which].append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"config {cid}: {name} = {va!r}: {sorted(res[0])[3]:.2f} ms (min {min(res[0]):.2f});  = {vb!r}: {sorted(res[1])[3]:.2f} ms (min {min(res[1]):.2f})", flush=True)
    del net, opt
    torch.cuda.empty_cache()
