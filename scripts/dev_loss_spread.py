"""dev: cfg-1 golden losses under both GEMM modes (how far rounding-level differences move the chaotic B=2 fixture)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from vn_pointcloudcompletion_amd import _lib
from vn_pointcloudcompletion_amd.loss import cd_loss_L1
import test_gpu_e2e as T
E = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cfg1_e2e.npz"), allow_pickle=True))
for mode in ("fp32", "split"):
    _lib.set_gemm_mode(mode)
    net = T._build(E)
    p, c = torch.as_tensor(E["p"]).cuda(), torch.as_tensor(E["c"]).cuda()
    coarse, fine = net(p)
    l1, l2 = cd_loss_L1(coarse, c).item(), cd_loss_L1(fine, c).item()
    print(mode, "loss1 %.9f (ref %.9f, f64 %.9f)  loss2 %.9f (ref %.9f, f64 %.9f)  |fine-ref|max %.3e  coarse %.3e" % (
        l1, float(E["loss1"]), float(E["loss1_64"]), l2, float(E["loss2"]), float(E["loss2_64"]),
        (fine.cpu() - torch.as_tensor(E["fine"])).abs().max().item(), (coarse.cpu() - torch.as_tensor(E["coarse"])).abs().max().item()))
    d = (fine.cpu().double() - torch.as_tensor(E["fine"]).double())
    print("   fine - ref: mean %.3e rms %.3e max %.3e ; per-axis mean" % (d.mean().item(), d.pow(2).mean().sqrt().item(), d.abs().max().item()), d.mean(dim=(0, 1)).tolist())
    if "fine64" in E:
        d = (fine.cpu().double() - torch.as_tensor(E["fine64"]).double())
        print("   fine - f64: mean %.3e rms %.3e max %.3e" % (d.mean().item(), d.pow(2).mean().sqrt().item(), d.abs().max().item()))
        d = (torch.as_tensor(E["fine"]).double() - torch.as_tensor(E["fine64"]).double())
        print("   ref  - f64: mean %.3e rms %.3e max %.3e" % (d.mean().item(), d.pow(2).mean().sqrt().item(), d.abs().max().item()))
    lref = cd_loss_L1(torch.as_tensor(E["fine"]).cuda(), c).item()
    print("   my chamfer on the reference's fine: %.9f (ref %.9f)" % (lref, float(E["loss2"])))
