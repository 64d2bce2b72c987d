"""dev: per-shape GEMM time of one training step of a BASELINE config (usage: dev_gemm_shapes.py <cfg>)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.argv_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
from types import SimpleNamespace
import torch
from vn_pointcloudcompletion_amd import ops
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.optim import FusedAdam
from vn_pointcloudcompletion_amd.train import train_step
C = {2: ("vn_pointnet", "vn_foldingnet", 32, 2048, "cd"), 3: ("vn_dgcnn_fps", "vn_foldingnet", 32, 512, "cd"), 4: ("vn_pointnet", "attention_vn_foldingnet", 16, 2048, "emd"),
     5: ("vn_pointr", "vn_foldingnet", 32, 1024, "cd")}[sys.argv_cfg]
cfg = SimpleNamespace(num_coarse=1024, latent_dim=C[3], only_coarse=False, enc_pretrained="none", device="cuda:0", coarse_loss=C[4],
                      dcd_opts={"alpha": 200, "lambda": 0.5}, knn_k=20 if sys.argv_cfg == 3 else 16)
torch.manual_seed(0)
net = PCNNet(cfg, C[0], C[1]).train()
md = getattr(net.decoder, "num_dense", 16384)
opt = FusedAdam(net.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(1)
p = (torch.rand(C[2], 2048, 3, generator=g) - 0.5).cuda(); gt = (torch.rand(C[2], md, 3, generator=g) - 0.5).cuda()
for _ in range(2): train_step(cfg, net, opt, p, gt)
prof = []; ops.GEMM_PROFILE = prof
train_step(cfg, net, opt, p, gt); torch.cuda.synchronize(); ops.GEMM_PROFILE = None
per = {}
for variant, flops, nsplit, shape, e0, e1 in prof:
    a = per.setdefault((variant, shape, nsplit), [0.0, 0.0, 0]); a[0] += flops; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
tot = sum(a[1] for a in per.values())
print("total GEMM ms/step %.2f in %d launches" % (tot * 1e3, sum(a[2] for a in per.values())))
for (variant, shape, nsplit), a in sorted(per.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {variant} M={shape[0]:8d} N={shape[1]:5d} K={shape[2]:8d} split={nsplit:4d} x{a[2]:2d} {a[1] * 1e3:7.3f} ms {a[0] / a[1] / 1e12:6.1f} TF/s")
