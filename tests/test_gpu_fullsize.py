"""Full-size (BASELINE config 2: B=32, 2048 -> 1024 -> 16384) checks through size-independent properties, where the CPU
oracle would take minutes: Chamfer identities and symmetry, run-to-run determinism of the forward pass, SO(3)
equivariance of the eval-mode network, finite gradients for every live parameter, Adam step vs torch.optim.Adam."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_chamfer_identities_full_size(lib_built):
    from vn_pointcloudcompletion_amd.chamfer_distance import chamfer_3DDist
    g = torch.Generator().manual_seed(0)
    a = (torch.rand(4, 16384, 3, generator=g) - 0.5).cuda()
    b = (torch.rand(4, 16384, 3, generator=g) - 0.5).cuda()
    d1, d2, i1, i2 = chamfer_3DDist()(a, a)
    ar = torch.arange(16384, device="cuda", dtype=torch.int32).expand(4, -1)
    assert d1.abs().max().item() == 0.0 and torch.equal(i1, ar) and torch.equal(i2, ar)
    d1, d2, i1, i2 = chamfer_3DDist()(a, b)
    e1, e2, j1, j2 = chamfer_3DDist()(b, a)
    assert torch.equal(d1, e2) and torch.equal(d2, e1) and torch.equal(i1, j2) and torch.equal(i2, j1)
    # every reported distance is the distance to the reported index, and no point of a sample beats it (spot check)
    pick = torch.gather(b, 1, i1.long().unsqueeze(-1).expand(-1, -1, 3))
    assert torch.allclose(((a - pick) ** 2).sum(-1), d1, rtol=1e-5, atol=1e-9)
    sub = ((a[:, :256, None, :] - b[:, None, :, :]) ** 2).sum(-1).min(-1)[0]
    assert torch.allclose(sub, d1[:, :256], rtol=1e-5, atol=1e-9)


@pytest.fixture(scope="module")
def net(lib_built):
    from vn_pointcloudcompletion_amd.model import PCNNet
    torch.manual_seed(0)
    cfg = SimpleNamespace(num_coarse=1024, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda")
    return PCNNet(cfg, "vn_pointnet", "vn_foldingnet")


def test_forward_shapes_determinism_and_equivariance(net):
    from vn_pointcloudcompletion_amd.transforms import Rotate, random_rotations
    g = torch.Generator().manual_seed(1)
    p = (torch.rand(8, 2048, 3, generator=g) - 0.5).cuda()
    net.train()
    with torch.no_grad():
        net(p)                                               # populate running statistics
    net.eval()
    with torch.no_grad():
        c0, f0 = net(p)
        c1, f1 = net(p)
    assert c0.shape == (8, 1024, 3) and f0.shape == (8, 16384, 3)
    assert torch.equal(c0, c1) and torch.equal(f0, f1)        # no atomics on the forward path: bitwise reproducible
    # SO(3) equivariance, stage by stage.  Up to the first arg-max pool it is exact to fp32 rounding; a pooled channel whose
    # two best points are within ~1e-7 flips under the rotation's rounding noise, and every flip perturbs the global feature
    # that feeds all later layers (the architecture's own sensitivity: the reference behaves the same), so after the pool
    # only the fraction of agreeing channels is asserted.
    R = random_rotations(8, generator=g).cuda()
    enc = net.encoder
    B, N = 8, 2048

    def rot_rows(rows, npts):
        C = rows.shape[1]
        return torch.einsum("bntc,bts->bnsc", rows.reshape(B, npts, 3, C), R).reshape(B * npts * 3, C)

    def stages(x):
        a1 = enc.first_conv[0].forward_rows(x.contiguous().reshape(B * N * 3, 1))
        f = enc.first_conv[1].forward_rows(a1)
        g1 = enc.maxpool1.forward_rows(f, B, N, producer=(a1, enc.first_conv[1]))
        return a1, f, g1

    with torch.no_grad():
        (a1, f, g1), (a1r, fr_, g1r) = stages(p), stages(torch.matmul(p, R))
    assert (a1r - rot_rows(a1, N)).abs().max().item() < 1e-5
    assert (fr_ - rot_rows(f, N)).abs().max().item() < 1e-5
    agree = ((g1r - rot_rows(g1, 1)).abs().reshape(B, 3, -1).amax(dim=1) < 1e-5).float().mean().item()
    assert agree >= 0.95


def test_train_step_full_size_and_adam_matches_torch(net):
    from vn_pointcloudcompletion_amd.optim import FusedAdam
    from vn_pointcloudcompletion_amd.train import train_step
    cfg = SimpleNamespace(coarse_loss="cd", only_coarse=False, dcd_opts={"alpha": 200, "lambda": 0.5})
    net.train()
    ref = {k: v.detach().clone() for k, v in net.named_parameters()}
    opt = FusedAdam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    g = torch.Generator().manual_seed(2)
    p = (torch.rand(32, 2048, 3, generator=g) - 0.5).cuda()
    c = (torch.rand(32, 16384, 3, generator=g) - 0.5).cuda()
    loss, l1, l2 = train_step(cfg, net, opt, p, c)
    assert torch.isfinite(loss) and 0.0 < l1.item() < 1.0 and 0.0 < l2.item() < 1.0
    grads = {k: v.grad.detach().clone() for k, v in net.named_parameters()}
    for k, gk in grads.items():
        assert torch.isfinite(gk).all(), k
        if "maxpool" in k:
            assert gk.abs().max().item() == 0.0, k            # never receives a gradient (flat zero slice)
        else:
            assert gk.abs().max().item() > 0.0, k
    # the fused step == torch.optim.Adam's first step on the same gradients
    tp = [torch.nn.Parameter(v.clone()) for v in ref.values()]
    for t_, gk in zip(tp, grads.values()):
        t_.grad = gk.clone()
    torch.optim.Adam(tp, lr=1e-4, betas=(0.9, 0.999)).step()
    for (k, v), t_ in zip(net.named_parameters(), tp):
        assert (v.detach() - t_.detach()).abs().max().item() < 1e-7, k
    sd = opt.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 1.0
    opt2 = FusedAdam(net.parameters(), lr=1e-4)
    opt2.load_state_dict(sd)                                   # torch-Adam format round-trips (train.py:255-277)
    assert opt2.flat_buffers[0].step == 1


@pytest.mark.parametrize("enc,dec,B,latent,loss,ceiling_ms", [("vn_dgcnn_fps", "vn_foldingnet", 32, 512, "cd", 90.0),
                                                              ("vn_pointnet", "attention_vn_foldingnet", 16, 2048, "emd", 160.0),
                                                              ("vn_pointr", "vn_foldingnet", 32, 1024, "cd", 170.0)])
def test_other_configs_step_at_full_batch_and_resolution(lib_built, enc, dec, B, latent, loss, ceiling_ms):
    """BASELINE configs 3, 4, 5 at their FULL sizes on one GPU (config 3: B = 32, k = 20; config 4: B = 16 with the EMD coarse
    loss; config 5: the per-GPU shard B = 32 of the B = 256 job, 2048 -> 1248 -> 19968): complete training steps run, the loss
    is finite (and, configs 3 and 4, decreases over a few steps on a fixed batch), every live parameter gets a finite gradient, and the step time
    stays under a loose ceiling (~3x the measured 25 / 51 / 55 ms of DESIGN.md §6 — a regression tripwire, not a benchmark)."""
    import time
    from vn_pointcloudcompletion_amd.model import PCNNet
    from vn_pointcloudcompletion_amd.optim import FusedAdam
    from vn_pointcloudcompletion_amd.train import train_step
    cfg = SimpleNamespace(num_coarse=1024, latent_dim=latent, only_coarse=False, enc_pretrained="none", device="cuda:0",
                          coarse_loss=loss, dcd_opts={"alpha": 200, "lambda": 0.5}, knn_k=20 if enc == "vn_dgcnn_fps" else 16)
    torch.manual_seed(0)
    net = PCNNet(cfg, enc, dec).train()
    md = getattr(net.decoder, "num_dense", 16384)
    opt = FusedAdam(net.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(2)
    p = (torch.rand(B, 2048, 3, generator=g) - 0.5).cuda()
    gt = (torch.rand(B, md, 3, generator=g) - 0.5).cuda()
    first, _, _ = train_step(cfg, net, opt, p, gt)
    assert torch.isfinite(first).item()
    assert torch.isfinite(opt.flat_buffers[0].flat_grad).all().item()
    assert opt.flat_buffers[0].flat_grad.abs().sum().item() > 0
    train_step(cfg, net, opt, p, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        last, _, _ = train_step(cfg, net, opt, p, gt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    print(f"{enc}+{dec} B={B} 2048->{md}: {ms:.1f} ms/step, loss {first.item():.4f} -> {last.item():.4f}, "
          f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    assert torch.isfinite(last).item()
    if enc != "vn_pointr":                    # config 5's feature-space kNN graphs re-wire between steps: six steps at lr 1e-4 on
        assert last.item() < first.item()     # random clouds do not reliably lower its loss (0.170 -> 0.185 measured)
    assert ms < ceiling_ms, ms
    del net, opt
    torch.cuda.empty_cache()


def test_screened_pool_on_surface_like_clouds_needs_no_fallback(net):
    """The candidate count of the screened arg-max pool depends on the data: uniform noise (the benchmark's clouds) leaves ~4 candidates
    per (sample, channel), points sampled from a surface ~20 (neighbouring points have nearly equal scores).  The candidate list must hold
    that without raising the fallback flag."""
    from vn_pointcloudcompletion_amd import _lib, ops
    if _lib.get_gemm_mode() != "split":
        pytest.skip("the screened pool belongs to the split GEMM mode")
    g = torch.Generator().manual_seed(5)
    v = torch.randn(4, 2048, 3, generator=g)
    v = v / v.norm(dim=-1, keepdim=True)
    p = (v * (0.35 + 0.05 * torch.sin(7 * v[..., :1]))).cuda()
    net.train()
    stats = []
    with ops.hooks(pool_screen_stats=stats), torch.no_grad():
        net(p)
    cand, flag, cap = stats[-1].tolist()
    per_pair = cand / (4 * 2048)
    print(f"surface-like clouds: {per_pair:.1f} candidates per (sample, channel), capacity {cap / (4 * 2048):.0f}, fallback flag {flag}")
    assert flag == 0 and per_pair < 32
