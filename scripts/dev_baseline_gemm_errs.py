import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from types import SimpleNamespace
import numpy as np, torch
from vn_pointcloudcompletion_amd import _lib, ops
from vn_pointcloudcompletion_amd.baselines import FoldingNet
G = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "baselines_small.npz"), allow_pickle=True))
cfg = SimpleNamespace(num_coarse=64, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda:0")
orig = ops.gemm
log = []
def spy(A, B, *, op_a=0, op_b=0, out=None, bias=None, rows_per_group=0, accumulate=False, nsplit=None):
    prev = out.double().clone() if (accumulate and out is not None) else None
    c = orig(A, B, op_a=op_a, op_b=op_b, out=out, bias=bias, rows_per_group=rows_per_group, accumulate=accumulate, nsplit=nsplit)
    a64 = A.double().t() if op_a else A.double(); b64 = B.double() if op_b else B.double().t()
    ref = a64 @ b64
    if prev is not None: ref = ref + prev
    mag = a64.abs() @ b64.abs()
    cen = float("nan")
    if ref.shape[0] == 2048:
        r3 = ref.view(2, 1024, -1); c3 = c.double().view(2, 1024, -1)
        rc = r3 - r3.mean(1, keepdim=True); cc = c3 - c3.mean(1, keepdim=True)
        cen = ((cc - rc).norm() / rc.norm()).item()
    log.append((("T" if op_a else "N") + ("N" if op_b else "T"), tuple(ref.shape), a64.shape[1], ((c.double() - ref).norm() / ref.norm()).item(),
                ((c.double() - ref).abs() / mag.clamp_min(1e-300)).max().item(), cen))
    return c
ops.gemm = spy
for mode in ("fp32", "split"):
    _lib.set_gemm_mode(mode); log.clear()
    torch.manual_seed(5)
    dec = FoldingNet(cfg); dec.num_dense = 1024; dec = dec.cuda().train()
    coarse = torch.as_tensor(G["fold.coarse"]).cuda().requires_grad_(True); fg = torch.as_tensor(G["fold.fg"]).cuda().requires_grad_(True)
    fine = dec(coarse, fg)
    (fine * torch.as_tensor(G["fold.gf"]).cuda()).sum().backward()
    print(mode)
    for l in log: print("   %s out %-14s K %5d  rel err %.2e  max err/sum|ab| %.2e  per-sample-centred rel err %.2e" % l)
