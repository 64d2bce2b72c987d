#!/usr/bin/env python3
"""Dev probe: candidates per (sample, channel) and fallback flags of the screened arg-max pool while the weights move (config 2,
synthetic batch, lr 1e-3 so that 80 steps change the network visibly)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from types import SimpleNamespace
import torch
from vn_pointcloudcompletion_amd import ops
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.optim import FusedAdam
from vn_pointcloudcompletion_amd.train import train_step

cfg = SimpleNamespace(num_coarse=1024, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda:0",
                      coarse_loss="cd", dcd_opts={"alpha": 200, "lambda": 0.5})
torch.manual_seed(0)
net = PCNNet(cfg, "vn_pointnet", "vn_foldingnet").train()
opt = FusedAdam(net.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(1)
B = 32
# a structured target (points on a bumpy sphere) instead of uniform noise, so that training has something to fit
v = torch.randn(B, 16384, 3, generator=g); v = v / v.norm(dim=-1, keepdim=True)
gt = (v * (0.35 + 0.05 * torch.sin(7 * v[..., :1]))).cuda()
p = gt[:, torch.randperm(16384, generator=g)[:2048]].contiguous()
stats = []
ops.POOL_SCREEN_STATS = stats
for step in range(int(os.environ.get('STEPS', 80))):
    loss, l1, l2 = train_step(cfg, net, opt, p, gt)
    if step % 10 == 9 or step < 3:
        torch.cuda.synchronize()
        cand, flag, cap = stats[-1].tolist()
        flags = sum(int(s[1].item()) for s in stats)
        print(f"step {step + 1:3d}: loss {loss.item():.4f}  candidates {cand} = {cand / (B * 2048):.2f} per (sample, channel), capacity {cap}, "
              f"fallbacks so far {flags}/{len(stats)}", flush=True)
