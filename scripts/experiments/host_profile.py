#!/usr/bin/env python3
"""Where the HOST spends its time in one training step (cProfile; the GPU runs asynchronously).  usage: host_profile.py [config id]"""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from types import SimpleNamespace
import torch
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.optim import FusedAdam
from vn_pointcloudcompletion_amd.train import train_step

CONFIGS = {
    2: dict(enc="vn_pointnet", dec="vn_foldingnet", B=32, latent=2048, loss="cd", mc=1024),
    3: dict(enc="vn_dgcnn_fps", dec="vn_foldingnet", B=32, latent=512, loss="cd", mc=1024, knn_k=20),
    4: dict(enc="vn_pointnet", dec="attention_vn_foldingnet", B=16, latent=2048, loss="emd", mc=1024),
    5: dict(enc="vn_pointr", dec="vn_foldingnet", B=32, latent=1024, loss="cd", mc=1024),
}
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 5
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
for _ in range(3):
    train_step(cfg, net, opt, p, gt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    train_step(cfg, net, opt, p, gt)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
