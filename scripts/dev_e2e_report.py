"""Dev aid: per-parameter gradient deviations of the HIP path vs the golden cfg1 fixture, and a float64 autograd
cross-check of single layers (oracle on the GPU in double)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from types import SimpleNamespace
from conftest import load_golden, t
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.loss import cd_loss_L1
from vn_pointcloudcompletion_amd import vn_layers as L
from oracle import vn_oracle as O

E = load_golden("cfg1_e2e.npz")
torch.manual_seed(int(E["seed"]))
cfg = SimpleNamespace(num_coarse=64, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda")
net = PCNNet(cfg, "vn_pointnet", "vn_foldingnet"); net.decoder.num_dense = 1024; net.train()
p, c = t(E["p"]), t(E["c"])
coarse, fine = net(p)
l1, l2 = cd_loss_L1(coarse, c), cd_loss_L1(fine, c)
print("loss1", l1.item(), float(E["loss1"]), float(E["loss1_64"]), "loss2", l2.item(), float(E["loss2"]), float(E["loss2_64"]))
print("coarse err", (coarse.cpu() - torch.as_tensor(E["coarse"])).abs().max().item(), "fine err", (fine.cpu() - torch.as_tensor(E["fine"])).abs().max().item(),
      "fine err vs 64", (fine.cpu().double() - torch.as_tensor(E["fine64"])).abs().max().item())
(l1 + l2).backward()
for (k, prm), n32, n64 in zip(net.named_parameters(), E["grad_norms"], E["grad_norms64"]):
    if n32 < 0: continue
    n = prm.grad.double().norm().item()
    print(f"{k:55s} mine {n:.6e} ref32 {n32:.6e} ref64 {n64:.6e} rel(mine-64) {abs(n-n64)/n64:.2e} rel(32-64) {abs(n32-n64)/n64:.2e}")

# single layer, C_in = 1 (first layer) in isolation vs float64 autograd
torch.manual_seed(0)
m = L.VNLinearLeakyReLU(1, 128, dim=4).cuda().train()
x = (torch.rand(2, 1, 3, 256, device="cuda") - 0.5)
g = torch.randn(2, 128, 3, 256, device="cuda")
xr = x.clone().requires_grad_(True)
y = m(xr); y.backward(g)
wf, wd = m.map_to_feat.weight.detach().double().requires_grad_(True), m.map_to_dir.weight.detach().double().requires_grad_(True)
ga, be = m.batchnorm.bn.weight.detach().double().requires_grad_(True), m.batchnorm.bn.bias.detach().double().requires_grad_(True)
xd = x.double().requires_grad_(True)
yo, _, _ = O.vn_linear_leaky_relu(xd, wf, wd, ga, be, None, None, True)
yo.backward(g.double())
def rel(a, b): return ((a.double() - b).abs().max() / b.abs().max()).item()
print("layer C_in=1: y", rel(y, yo), "gx", rel(xr.grad, xd.grad), "gwf", rel(m.map_to_feat.weight.grad, wf.grad), "gwd", rel(m.map_to_dir.weight.grad, wd.grad),
      "ggamma", rel(m.batchnorm.bn.weight.grad, ga.grad), "gbeta", rel(m.batchnorm.bn.bias.grad, be.grad))
# same in fp32 autograd of the oracle (what the reference's fp32 would give)
wf32, wd32 = m.map_to_feat.weight.detach().clone().requires_grad_(True), m.map_to_dir.weight.detach().clone().requires_grad_(True)
x32 = x.clone().requires_grad_(True)
y32, _, _ = O.vn_linear_leaky_relu(x32, wf32, wd32, m.batchnorm.bn.weight.detach(), m.batchnorm.bn.bias.detach(), None, None, True)
y32.backward(g)
print("fp32 autograd of the oracle vs fp64: gwd", rel(wd32.grad, wd.grad), "gwf", rel(wf32.grad, wf.grad))
