#!/usr/bin/env python3
"""Per-phase time (synchronised) of the first steps of config 4: where does a one-off slow step come from?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from types import SimpleNamespace
import torch
from vn_pointcloudcompletion_amd import distributed as D
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.optim import FusedAdam
from vn_pointcloudcompletion_amd.train import compute_losses
if "noticks" in sys.argv:                       # A/B: BatchNorm counters incremented one by one
    from vn_pointcloudcompletion_amd import vn_layers
    vn_layers.begin_deferred_ticks = lambda: 1
if "slowstream" in sys.argv:                    # A/B: torch.cuda.current_stream() per launch
    from vn_pointcloudcompletion_amd import _lib
    _lib._RAW_STREAM = None
cfg = SimpleNamespace(num_coarse=1024, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda:0",
                      coarse_loss="emd", dcd_opts={"alpha": 200, "lambda": 0.5}, knn_k=16)
torch.manual_seed(0)
net = PCNNet(cfg, "vn_pointnet", "attention_vn_foldingnet").train()
opt = FusedAdam(net.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(1)
p = (torch.rand(16, 2048, 3, generator=g) - 0.5).cuda()
gt = (torch.rand(16, 16384, 3, generator=g) - 0.5).cuda()

def tick():
    torch.cuda.synchronize(); return time.perf_counter()
for s in range(8):
    t0 = tick(); opt.zero_grad()
    coarse, fine = net(p); t1 = tick()
    loss, l1, l2 = compute_losses(cfg, coarse, fine, gt); t2 = tick()
    loss.backward(); t3 = tick()
    D.allreduce_gradients(opt); opt.step(); t4 = tick()
    print(f"step {s}: forward {1e3 * (t1 - t0):6.1f}  losses {1e3 * (t2 - t1):6.1f}  backward {1e3 * (t3 - t2):6.1f}  adam {1e3 * (t4 - t3):6.1f} ms   "
          f"mem {torch.cuda.memory_allocated() / 2**30:.2f} / reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
