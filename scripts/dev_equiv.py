import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.transforms import Rotate, random_rotations
from vn_pointcloudcompletion_amd import ops
torch.manual_seed(0)
cfg = SimpleNamespace(num_coarse=1024, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda")
net = PCNNet(cfg, "vn_pointnet", "vn_foldingnet")
enc = net.encoder
g = torch.Generator().manual_seed(1)
B, N = 4, 512
p = (torch.rand(B, N, 3, generator=g) - 0.5).cuda()
R = random_rotations(B, generator=g).cuda()
pr = torch.matmul(p, R)

def rot_rows(rows, npts):     # rows [(B*npts)*3, C] -> rotate the 3-axis by R[b]
    C = rows.shape[1]
    v = rows.reshape(B, npts, 3, C)
    return torch.einsum("bntc,bts->bnsc", v, R).reshape(B * npts * 3, C)

def stages(x):
    out = {}
    xr = x.contiguous().reshape(B * N * 3, 1)
    a1 = enc.first_conv[0].forward_rows(xr); out["a1"] = (a1, N)
    f = enc.first_conv[1].forward_rows(a1); out["f"] = (f, N)
    g1 = enc.maxpool1.forward_rows(f, B, N, producer=(a1, enc.first_conv[1])); out["g1"] = (g1, 1)
    f3 = enc.second_conv[0].forward_rows(f, global_rows=g1, rows_per_group=N * 3); out["f3"] = (f3, N)
    g2, _ = ops.VNLinearMaxPoolFn.apply(f3, enc.second_conv[1].map_to_feat.weight, enc.maxpool2.map_to_dir.weight.detach(), B, N); out["g2"] = (g2, 1)
    return out

for mode in ("train", "eval"):
    net.train()
    with torch.no_grad():
        net(p)
    if mode == "eval":
        net.eval()
    with torch.no_grad():
        s0, s1 = stages(p), stages(pr)
    for k in s0:
        a, n = s0[k]; b, _ = s1[k]
        err = (b - rot_rows(a, n)).abs()
        print(mode, k, "max err", err.max().item(), "median", err.median().item(), "scale", a.abs().mean().item())
