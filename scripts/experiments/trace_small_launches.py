#!/usr/bin/env python3
"""Which Python lines of the package issue the small ATen launches of a training step (config 5 by default)?
usage: python scripts/experiments/trace_small_launches.py [config id]   -> counts of aten ops per (op, package source line)"""
import os, sys, collections
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
for _ in range(2):
    train_step(cfg, net, opt, p, gt)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

cnt = collections.Counter()


class Tracer(TorchDispatchMode):
    """Every ATen op that reaches the dispatcher (forward, and the Python bodies of the custom Functions' backward), with the innermost
    frame of the package on the Python stack; ops the autograd engine issues itself (AccumulateGrad's add_) have none."""

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        outs = func(*args, **(kwargs or {}))
        flat = []
        for a in list(args) + list((kwargs or {}).values()):
            flat += list(a) if isinstance(a, (list, tuple)) else [a]
        dev = any(isinstance(a, torch.Tensor) and a.is_cuda for a in flat)
        if dev and name.split(".")[0] not in ("view", "_unsafe_view", "reshape", "expand", "permute", "transpose", "slice", "select", "detach", "alias",
                                              "unsqueeze", "squeeze", "as_strided", "empty", "empty_like", "empty_strided", "t", "unbind", "split",
                                              "split_with_sizes", "narrow", "_reshape_alias", "lift_fresh", "sym_size", "sym_stride"):
            where = None
            for fr in reversed(traceback.extract_stack()):
                if "/vn_pointcloudcompletion_amd/" in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            if where is None:
                shp = next((tuple(a.shape) for a in flat if isinstance(a, torch.Tensor)), ())
                where = f"(autograd engine / optimizer) first operand {shp}"
            cnt[(name, where)] += 1
        return outs


with Tracer():
    train_step(cfg, net, opt, p, gt)
torch.cuda.synchronize()
tot = collections.Counter()
for (name, where), n in cnt.items():
    tot[name] += n
print("per op:", dict(tot.most_common()))
for (name, where), n in cnt.most_common(90):
    print(f"{n:5d}  {name:28s} {where}")
