#!/usr/bin/env python3
"""Gathered vs in-place gradient buffers on the real model: which parameters differ after one backward?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from types import SimpleNamespace
import torch
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.optim import FlatBuffers
from vn_pointcloudcompletion_amd.train import compute_losses

cfg = SimpleNamespace(num_coarse=1024, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda:0",
                      coarse_loss="cd", dcd_opts={"alpha": 200, "lambda": 0.5}, knn_k=16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = torch.Generator().manual_seed(1)
p = (torch.rand(B, 2048, 3, generator=g) - 0.5).cuda()
gt = (torch.rand(B, 16384, 3, generator=g) - 0.5).cuda()
res = {}
for mode in (True, False, True):
    torch.manual_seed(0)
    net = PCNNet(cfg, "vn_pointnet", "vn_foldingnet").train()
    fb = FlatBuffers(net.parameters(), gathered=mode)
    names = {id(q): n for n, q in net.named_parameters()}
    for step in range(2):
        fb.zero_grad()
        coarse, fine = net(p)
        loss, _, _ = compute_losses(cfg, coarse, fine, gt)
        loss.backward()
        kinds = {names[id(q)]: (None if q.grad is None else (tuple(q.grad.shape), q.grad.is_contiguous(), q.grad.data_ptr() % 16, q.grad._base is not None))
                 for q in fb.params}
        fb.check_views()
        torch.cuda.synchronize()
    key = "gathered" if mode else "inplace"
    if key in res:
        key += "2"
    res[key] = (fb.flat_grad.clone(), [names[id(q)] for q in fb.params], list(fb.offsets), [q.numel() for q in fb.params], kinds)
a, names_, offs, nums, kinds = res["gathered"]
b = res["inplace"][0]
c = res["gathered2"][0]
print("gathered == inplace:", torch.equal(a, b), " gathered == gathered again:", torch.equal(a, c))
for n, o, k in zip(names_, offs, nums):
    d = (a[o:o + k] - b[o:o + k]).abs().max().item()
    s = b[o:o + k].abs().max().item()
    if d > 0:
        print(f"  {n:50s} max diff {d:.3e} of {s:.3e}   incoming grad {kinds[n]}")
