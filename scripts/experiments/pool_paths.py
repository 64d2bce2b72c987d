#!/usr/bin/env python3
"""Which path the VNLinear -> VNMaxPool calls took over N steps of a config (screened search / fallback), and the step times."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from types import SimpleNamespace
import torch
from vn_pointcloudcompletion_amd import ops
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.optim import FusedAdam
from vn_pointcloudcompletion_amd.train import train_step
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 4
C = {2: dict(enc="vn_pointnet", dec="vn_foldingnet", B=32, latent=2048, loss="cd"),
     4: dict(enc="vn_pointnet", dec="attention_vn_foldingnet", B=16, latent=2048, loss="emd")}[cid]
cfg = SimpleNamespace(num_coarse=1024, latent_dim=C["latent"], only_coarse=False, enc_pretrained="none", device="cuda:0",
                      coarse_loss=C["loss"], dcd_opts={"alpha": 200, "lambda": 0.5}, knn_k=16)
torch.manual_seed(0)
net = PCNNet(cfg, C["enc"], C["dec"]).train()
opt = FusedAdam(net.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(1)
p = (torch.rand(C["B"], 2048, 3, generator=g) - 0.5).cuda()
gt = (torch.rand(C["B"], 16384, 3, generator=g) - 0.5).cuda()
for s in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, _, _ = train_step(cfg, net, opt, p, gt)
    torch.cuda.synchronize()
    print(f"step {s:2d}: {(time.perf_counter() - t0) * 1e3:6.1f} ms  loss {loss.item():9.4f}  pool paths {dict(ops.POOL_PATH_COUNTS)}", flush=True)
