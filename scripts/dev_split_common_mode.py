"""dev: accuracy of the ROW-TO-ROW DIFFERENCES of a GEMM whose A rows are a common vector plus a small per-row variation
(what a layer after a batch-statistics normalisation with a dominant common mode looks like)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from vn_pointcloudcompletion_amd import ops, _lib
g = torch.Generator().manual_seed(1)
for (M, N, K, rho) in [(2048, 512, 512, 1e-2), (2048, 512, 512, 1e-3), (1920, 512, 512, 1e-3), (2048, 512, 6144, 1e-3)]:
    common = torch.relu(torch.randn(1, K, generator=g))
    A = (common + rho * torch.randn(M, K, generator=g)).cuda()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    ref = A.double() @ B.double().t()
    refc = ref - ref.mean(0, keepdim=True)
    for mode in ("fp32", "split"):
        _lib.set_gemm_mode(mode)
        c = ops.gemm(A, B).double()
        cc = c - c.mean(0, keepdim=True)
        print((M, N, K, rho), mode, "raw rel err %.3e" % ((c - ref).norm() / ref.norm()).item(),
              " centred rel err %.3e" % ((cc - refc).norm() / refc.norm()).item())
_lib.set_gemm_mode("split")
