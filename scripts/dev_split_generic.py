import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from vn_pointcloudcompletion_amd import ops, _lib
_lib.set_gemm_mode("split")
g = torch.Generator().manual_seed(1)
def rnd(*s): return ((torch.rand(*s, generator=g) * 2 - 1) * torch.exp2(torch.randint(-8, 9, s, generator=g).float())).cuda()
for (M, N, K) in [(200, 37, 37), (2048, 512, 3), (2048, 512, 5), (100, 64, 32), (192, 128, 33), (200, 130, 64), (7, 5, 3)]:
    A = rnd(M, K)
    # NT with B = I (N = K)
    I = torch.eye(K, device="cuda")
    c = ops.gemm(A, I)                       # [M, K]
    print("NT generic A*I", (M, K, K), "exact" if torch.equal(c, A) else "max rel err %.3e" % ((c - A).abs() / A.abs().clamp_min(1e-30)).max().item())
    c = ops.gemm(A, I, op_b=1)               # NN: B [K, K]
    print("NN generic A*I", (M, K, K), "exact" if torch.equal(c, A) else "max rel err %.3e" % ((c - A).abs() / A.abs().clamp_min(1e-30)).max().item())
    At = A.t().contiguous()                  # [K, M] ; TN: C = At^T * I_K?  op_a=1: A is [K, M] -> C [M, N] with B [K, N]
    c = ops.gemm(At, I, op_a=1, op_b=1, nsplit=1)
    print("TN generic A*I", (M, K, K), "exact" if torch.equal(c, A) else "max rel err %.3e" % ((c - A).abs() / A.abs().clamp_min(1e-30)).max().item())
    # general accuracy vs fp64 for NN with small K
    B = rnd(K, N)
    ref = A.double() @ B.double()
    for mode in ("fp32", "split"):
        _lib.set_gemm_mode(mode)
        c = ops.gemm(A, B, op_b=1)
        mag = A.double().abs() @ B.double().abs()
        print("   NN", (M, N, K), mode, "max err/sum|ab| %.3e" % ((c.double() - ref).abs() / mag.clamp_min(1e-300)).max().item())
    _lib.set_gemm_mode("split")
