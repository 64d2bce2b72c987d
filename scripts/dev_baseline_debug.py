import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from types import SimpleNamespace
import numpy as np, torch
from vn_pointcloudcompletion_amd import _lib
from vn_pointcloudcompletion_amd.baselines import FoldingNet
G = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "baselines_small.npz"), allow_pickle=True))
cfg = SimpleNamespace(num_coarse=64, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda:0")
for mode in ("split", "fp32"):
    _lib.set_gemm_mode(mode)
    torch.manual_seed(5)
    dec = FoldingNet(cfg); dec.num_dense = 1024; dec = dec.cuda().train()
    coarse = torch.as_tensor(G["fold.coarse"]).cuda().requires_grad_(True); fg = torch.as_tensor(G["fold.fg"]).cuda().requires_grad_(True)
    fine = dec(coarse, fg)
    print(mode, "fine err", (fine.detach().cpu() - torch.as_tensor(G["fold.fine"])).abs().max().item())
    (fine * torch.as_tensor(G["fold.gf"]).cuda()).sum().backward()
    for name, got, ref in (("g_coarse", coarse.grad, G["fold.g_coarse"]), ("g_fg", fg.grad, G["fold.g_fg"])):
        d = np.abs(got.cpu().numpy() - ref)
        print("   ", name, "max err", d.max(), "rel", d.max() / np.abs(ref).max(), "rms rel", np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean()))
    for (k, p), n in zip(dec.named_parameters(), G["fold.grad_norms"]):
        print("   ", k, "norm", p.grad.double().norm().item(), "ref", n, "rel", abs(p.grad.double().norm().item() - n) / n)
    for k, p in dec.named_parameters():
        if f"fold.gs64.{k}" not in G: continue
        g64, g32 = G[f"fold.gs64.{k}"], G[f"fold.gs32.{k}"].astype(np.float64)
        stride = max(1, -(-p.numel() // 16384))
        mine = p.grad.flatten()[::stride].double().cpu().numpy()
        print("    L2", k, "mine", np.linalg.norm(mine - g64) / np.linalg.norm(g64), "ref32", np.linalg.norm(g32 - g64) / np.linalg.norm(g64))
