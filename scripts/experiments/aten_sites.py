#!/usr/bin/env python3
"""Which lines of the package call kernel-launching ATen operators in a training step?  Runs a config's step under a TorchDispatchMode
(it sees every ATen call of the forward AND of the backward, with the Python stack that made it) and prints, per (operator, innermost
package frame), the calls per step.  View / metadata / allocation operators launch nothing and are left out.
usage: python scripts/experiments/aten_sites.py <config 2..5>"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from types import SimpleNamespace
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.optim import FusedAdam
from vn_pointcloudcompletion_amd.train import train_step

CONFIGS = {
    2: dict(enc="vn_pointnet", dec="vn_foldingnet", B=32, latent=2048, loss="cd", mc=1024),
    3: dict(enc="vn_dgcnn_fps", dec="vn_foldingnet", B=32, latent=512, loss="cd", mc=1024, knn_k=20),
    4: dict(enc="vn_pointnet", dec="attention_vn_foldingnet", B=16, latent=2048, loss="emd", mc=1024),
    5: dict(enc="vn_pointr", dec="vn_foldingnet", B=32, latent=1024, loss="cd", mc=1024),
}
NO_LAUNCH = {"view", "_unsafe_view", "reshape", "_reshape_alias", "permute", "transpose", "slice", "expand", "detach", "as_strided", "unsqueeze",
             "squeeze", "select", "alias", "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "stride", "size", "numel",
             "split", "split_with_sizes", "unbind", "narrow", "_local_scalar_dense", "t", "lift_fresh", "set_", "record_stream", "chunk",
             "unfold", "diagonal", "is_same_size", "is_contiguous", "sym_size", "sym_stride", "sym_numel", "dim", "result_type", "view_as",
             "_record_function_enter_new", "_record_function_exit", "is_pinned", "is_nonzero", "item", "storage_offset", "sym_storage_offset"}

cid = int(sys.argv[1]) if len(sys.argv) > 1 else 2
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
sites = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        base = name.split(".")[1] if name.count(".") >= 1 else name
        if base not in NO_LAUNCH:
            try:
                f = sys._getframe(1)
            except ValueError:                   # a C++ autograd node on the engine's thread: no Python frame made this call
                f = None
            where = "(autograd engine)"
            while f is not None:
                fn = f.f_code.co_filename
                if "vn_pointcloudcompletion_amd" in fn and "aten_sites" not in fn:
                    where = f"{fn.split('vn_pointcloudcompletion_amd/')[-1]}:{f.f_lineno} {f.f_code.co_name}"
                    break
                f = f.f_back
            if where == "(autograd engine)":      # say what is being accumulated / materialised
                where += " " + " ".join(str(tuple(a.shape)) for a in args if isinstance(a, torch.Tensor))
            sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Census():
    train_step(cfg, net, opt, p, gt)
    torch.cuda.synchronize()
tot = 0
for (name, where), n in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{n:4d}  {name:34s} {where}")
    tot += n
print("total kernel-launching ATen calls:", tot)
