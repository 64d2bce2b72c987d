"""The non-equivariant members of the plugin surface (enc_type "dgcnn_fps", dec_type "foldingnet")
against goldens generated from the reference (tests/golden/make_golden.py::main_baselines): same seed -> same weights,
forward, gradients, BatchNorm running statistics, eval mode."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden, relerr, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(lib_built):
    return load_golden("baselines_small.npz")


def _check_params(module, G, tag, seed_tol=1e-9):
    names = [k for k, _ in module.named_parameters()]
    assert names == list(G[tag + ".names"])
    for (k, p), s, a in zip(module.named_parameters(), G[tag + ".weight_sum"], G[tag + ".weight_abssum"]):
        assert abs(p.double().sum().item() - s) <= seed_tol * max(1.0, a), k
        assert abs(p.double().abs().sum().item() - a) <= seed_tol * max(1.0, a), k


def _check_grads(module, G, tag, factor=3.0, floor=1e-5):
    """Parameter gradients against the reference's fp64 run on a strided sample of every gradient: our L2 error may be at
    most `factor` x the reference's own fp32 error on the same sample (plus a `floor` relative term).  Gradients that are
    mathematically zero (a conv bias in front of a train-mode BatchNorm) are pure rounding noise in every implementation:
    they only have to be negligible next to the largest gradient."""
    n64 = G[tag + ".grad_norms64"]
    scale = max(n64)
    for i, (k, p) in enumerate(module.named_parameters()):
        if n64[i] < 1e-9 * scale:
            assert p.grad.double().norm().item() < 1e-5 * scale, k
            continue
        g64, g32 = G[f"{tag}.gs64.{k}"], G[f"{tag}.gs32.{k}"].astype(np.float64)
        stride = max(1, -(-p.numel() // 16384))
        mine = p.grad.flatten()[::stride].double().cpu().numpy()
        ref_err = np.linalg.norm(g32 - g64)
        err = np.linalg.norm(mine - g64)
        assert err <= factor * ref_err + floor * np.linalg.norm(g64), (k, err, ref_err, np.linalg.norm(g64))


def _close_to_fp64(got, ref32, ref64, what, factor=2.0, floor=1e-5):
    """|got - fp64| <= factor * |reference fp32 - fp64| + floor * max|fp64|  (max norm)."""
    gap = np.abs(ref32.astype(np.float64) - ref64).max()
    err = np.abs(got.detach().double().cpu().numpy() - ref64).max()
    assert err <= factor * gap + floor * np.abs(ref64).max(), (what, err, gap)


def test_foldingnet_decoder_matches_reference(G):
    from vn_pointcloudcompletion_amd.baselines import FoldingNet
    cfg = SimpleNamespace(num_coarse=64, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda:0")
    torch.manual_seed(5)
    dec = FoldingNet(cfg)
    dec.num_dense = 64 * 16
    dec = dec.cuda().train()
    _check_params(dec, G, "fold")
    assert "folding_seed" not in dec.state_dict()
    coarse, fg = t(G["fold.coarse"]).requires_grad_(True), t(G["fold.fg"]).requires_grad_(True)
    fine = dec(coarse, fg)
    assert fine.shape == (2, 1024, 3)
    assert (fine.detach().cpu() - torch.as_tensor(G["fold.fine"])).abs().max().item() < 5e-5
    _close_to_fp64(fine, G["fold.fine"], G["fold.fine64"], "fine", factor=3.0)
    (fine * t(G["fold.gf"])).sum().backward()
    # Train-mode backward of this fixture is chaotic at the 1e-7 level (B = 2, a 6144-value global feature that dominates the
    # first BatchNorm's variance): in exact fp64 arithmetic a 6e-8 relative perturbation of inputs and weights moves these
    # gradients by 2e-6, 3e-3 or 8e-3 depending on the draw (ReLU flips coupled through the batch statistics), and the
    # reference's own fp32 gradients sit 1-5e-3 from its fp64 ones.  Every GEMM of the chain is individually as accurate as
    # the fp32-input MFMA's (scripts/dev_baseline_gemm_errs.py), so train mode gets a 2e-2 bound that still catches any wrong
    # formula (O(1) errors); the tight gradient check is the eval-mode one below.
    _close_to_fp64(coarse.grad, G["fold.g_coarse"], G["fold.g_coarse64"], "g_coarse", factor=3.0, floor=2e-2)
    _close_to_fp64(fg.grad, G["fold.g_fg"], G["fold.g_fg64"], "g_fg", factor=3.0, floor=2e-2)
    _check_grads(dec, G, "fold", floor=2e-2)
    for k, v in dec.state_dict().items():
        if "running" in k:
            assert np.abs(v.cpu().numpy() - G["fold.after." + k]).max() < 1e-4 * max(1.0, np.abs(G["fold.after." + k]).max()), k
    # eval mode: BatchNorm on the (just updated) running statistics — no batch coupling, well-conditioned backward
    dec.eval()
    dec.zero_grad()
    ce, fe_in = t(G["fold.coarse"]).requires_grad_(True), t(G["fold.fg"]).requires_grad_(True)
    fe = dec(ce, fe_in)
    assert (fe.detach().cpu() - torch.as_tensor(G["fold.fine_eval"])).abs().max().item() < 5e-5
    (fe * t(G["fold.gf"])).sum().backward()
    _close_to_fp64(ce.grad, G["folde.g_coarse"], G["folde.g_coarse64"], "eval g_coarse", factor=3.0)
    _close_to_fp64(fe_in.grad, G["folde.g_fg"], G["folde.g_fg64"], "eval g_fg", factor=3.0)
    _check_grads(dec, G, "folde")


def test_dgcnn_fps_encoder_matches_reference(G):
    from vn_pointcloudcompletion_amd.baselines import DGCNN_fps
    cfg = SimpleNamespace(num_coarse=32, latent_dim=1024, only_coarse=True, enc_pretrained="none", device="cuda:0")
    torch.manual_seed(int(G["enc.seed"]))
    enc = DGCNN_fps(cfg).cuda().train()
    _check_params(enc, G, "enc")
    x = t(G["enc.x"])
    coarse, g = enc(x)
    assert coarse.shape == (2, 32, 3) and g.shape == (2, 1024)
    # yardstick: the reference's own fp32-vs-fp64 gap on this fixture is ~3e-6
    assert (coarse.detach().cpu() - torch.as_tensor(G["enc.coarse"])).abs().max().item() < 1e-4
    assert (g.detach().cpu() - torch.as_tensor(G["enc.global"])).abs().max().item() < 1e-4
    (coarse * t(G["enc.gc"])).sum().backward(retain_graph=True)
    (g * t(G["enc.gg"])).sum().backward()
    _check_grads(enc, G, "enc")
    for k, v in enc.state_dict().items():
        if "running" in k:
            assert np.abs(v.cpu().numpy() - G["enc.after." + k]).max() < 1e-4 * max(1.0, np.abs(G["enc.after." + k]).max()), k


def test_plugin_surface_builds_every_type(lib_built):
    """PCNNet accepts every enc_type / dec_type of the reference's factory (models/model.py:14-48) and the runnable
    pairs step: (vn_pointnet, foldingnet) end to end with the Chamfer loss, and dgcnn_fps as an only_coarse model."""
    from vn_pointcloudcompletion_amd.loss import cd_loss_L1
    from vn_pointcloudcompletion_amd.model import DECODERS, ENCODERS, PCNNet
    assert set(ENCODERS) == {"dgcnn_fps", "vn_dgcnn_fps", "vn_pointnet", "vn_pointr"}
    assert set(DECODERS) == {"vn_foldingnet", "foldingnet", "attention_vn_foldingnet"}
    g = torch.Generator().manual_seed(3)
    p, c = (torch.rand(2, 256, 3, generator=g) - 0.5).cuda(), (torch.rand(2, 1024, 3, generator=g) - 0.5).cuda()
    cfg = SimpleNamespace(num_coarse=64, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda:0")
    torch.manual_seed(0)
    net = PCNNet(cfg, "vn_pointnet", "foldingnet")
    net.decoder.num_dense = 1024
    net.train()
    coarse, fine = net(p)
    assert coarse.shape == (2, 64, 3) and fine.shape == (2, 1024, 3)
    (cd_loss_L1(coarse, c) + cd_loss_L1(fine, c)).backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in net.decoder.parameters())
    cfg2 = SimpleNamespace(num_coarse=32, latent_dim=1024, only_coarse=True, enc_pretrained="none", device="cuda:0")
    net2 = PCNNet(cfg2, "dgcnn_fps", "foldingnet").train()
    x = (torch.rand(2, 640, 3, generator=g) - 0.5).cuda()
    co, fi = net2(x)
    assert co.shape == (2, 32, 3) and fi is None
    cd_loss_L1(co, c).backward()
    with pytest.raises(Exception):
        PCNNet(cfg, "vn_pointnet++", "foldingnet")                      # models/model.py:20-21


@pytest.mark.parametrize("P,C,act,slope", [(5000, 96, 1, 0.0), (1027, 130, 2, 0.2), (64, 512, 0, 0.0), (3, 7, 2, 0.01)])
def test_batchnorm1d_act_kernels_equal_torch(lib_built, P, C, act, slope):
    """vnp_bn1d_act (BatchNorm1d over point-major rows + ReLU / LeakyReLU / nothing, the glue of the non-equivariant baselines) against
    torch.nn.functional.batch_norm + the activation in float64: outputs, running statistics (momentum 0.1, unbiased variance), dL/dx,
    dL/dgamma, dL/dbeta in training mode; outputs and dL/dx in eval mode; ragged chunks, a strided input view."""
    import torch.nn.functional as F
    from vn_pointcloudcompletion_amd.baselines import _bn_rows
    g = torch.Generator().manual_seed(P + C)
    wide = (torch.randn(P, C + 5, generator=g) * 3 + 1.5).cuda()
    x0 = wide[:, 2:2 + C]
    go = torch.randn(P, C, generator=g).cuda()

    def actf(t):
        return F.relu(t) if act == 1 else F.leaky_relu(t, slope) if act == 2 else t
    for training in (True, False):
        torch.manual_seed(1)
        bn = torch.nn.BatchNorm1d(C).cuda()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5); bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2.0)
        ref = torch.nn.BatchNorm1d(C).cuda().double()
        ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
        bn.train(training); ref.train(training)
        x = x0.clone().requires_grad_(True)
        y = _bn_rows(bn, x, act=act, slope=slope)
        y.backward(go)
        xd = x0.double().clone().requires_grad_(True)
        yd = actf(ref(xd))
        yd.backward(go.double())
        assert relerr(y, yd) < 2e-6, training
        assert relerr(x.grad, xd.grad) < 2e-5, training
        if training:
            assert relerr(bn.weight.grad, ref.weight.grad) < 2e-5 and relerr(bn.bias.grad, ref.bias.grad) < 2e-5
            assert relerr(bn.running_mean, ref.running_mean) < 2e-6 and relerr(bn.running_var, ref.running_var) < 2e-6
            assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("B,Nk,Nq,k,C,G", [(2, 300, 300, 16, 32, 4), (3, 640, 128, 16, 128, 4), (1, 70, 70, 5, 24, 3), (2, 200, 100, 8, 64, 4)])
def test_edge_groupnorm_leaky_max_equals_torch(lib_built, B, Nk, Nq, k, C, G):
    """vnp_edge_gn_max / vnp_edge_gn_gather (the EdgeConv tail of the non-equivariant DGCNN_fps: GroupNorm(G) + LeakyReLU(0.2) + max over
    the k neighbours of e(q, j) = u[nbr[q][j]] + v[q], models/dgcnn.py:36-54, 101-124) against the same expression on the materialised
    [B, C, Nq, k] edge tensor in float64 torch: output, dL/du (gathered over the sorted reverse adjacency), dL/dv, dL/dgamma, dL/dbeta;
    a block of queries that does not fill 64, C below 64, a neighbour listed twice by one query, and bit-identical repeats."""
    import torch.nn.functional as F
    from vn_pointcloudcompletion_amd.baselines import _EdgeGnLeakyMaxFn
    g = torch.Generator().manual_seed(B * Nk + C)
    u0, v0 = torch.randn(B * Nk, C, generator=g), torch.randn(B * Nq, C, generator=g)
    idx = torch.stack([torch.stack([torch.randperm(Nk, generator=g)[:k] for _ in range(Nq)]) for _ in range(B)])      # [B,Nq,k] distinct
    idx[0, 0, 1] = idx[0, 0, 0]                                                          # one query names a neighbour twice
    gidx = (idx + torch.arange(B).view(B, 1, 1) * Nk).reshape(B * Nq, k).int()
    gam0, bet0 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    go = torch.randn(B * Nq, C, generator=g)

    def run():
        u, v = u0.cuda().requires_grad_(True), v0.cuda().requires_grad_(True)
        gam, bet = gam0.cuda().requires_grad_(True), bet0.cuda().requires_grad_(True)
        out = _EdgeGnLeakyMaxFn.apply(u, v, gidx.cuda(), gam, bet, B, Nq, G, 1e-5, 0.2)
        out.backward(go.cuda())
        return out.detach(), u.grad, v.grad, gam.grad, bet.grad
    a, b = run(), run()
    for x, y in zip(a, b):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    ud, vd = u0.double().requires_grad_(True), v0.double().requires_grad_(True)
    gd, bd = gam0.double().requires_grad_(True), bet0.double().requires_grad_(True)
    e = ud[gidx.long().reshape(-1)].view(B, Nq, k, C) + vd.view(B, Nq, 1, C)             # [B,Nq,k,C]
    y = F.group_norm(e.permute(0, 3, 1, 2), G, gd, bd, 1e-5)                             # [B,C,Nq,k]
    ref = F.leaky_relu(y, 0.2).max(dim=-1)[0].permute(0, 2, 1).reshape(B * Nq, C)
    ref.backward(go.double())
    assert relerr(a[0], ref) < 5e-6
    for name, mine, want in zip(("gu", "gv", "ggamma", "gbeta"), a[1:], (ud.grad, vd.grad, gd.grad, bd.grad)):
        assert relerr(mine, want) < 5e-5, (name, relerr(mine, want))


def test_bias_act_and_rows_max_equal_torch(lib_built):
    """vnp_bias_act (bias + ReLU / LeakyReLU on rows, its input and bias gradients) and vnp_rows_max (max over a sample's rows) against torch."""
    import torch.nn.functional as F
    from vn_pointcloudcompletion_amd.baselines import _BiasActFn, _RowsMaxFn
    g = torch.Generator().manual_seed(4)
    for P, C, act, slope in ((37, 20, 1, 0.0), (3000, 130, 2, 0.2), (5, 1024, 0, 0.0)):
        x0, b0, go = torch.randn(P, C, generator=g), torch.randn(C, generator=g), torch.randn(P, C, generator=g)
        x, b = x0.cuda().requires_grad_(True), b0.cuda().requires_grad_(True)
        y = _BiasActFn.apply(x, b, act, slope)
        y.backward(go.cuda())
        xd, bd = x0.double().requires_grad_(True), b0.double().requires_grad_(True)
        t_ = xd + bd
        yd = F.relu(t_) if act == 1 else F.leaky_relu(t_, slope) if act == 2 else t_
        yd.backward(go.double())
        assert relerr(y, yd) < 1e-6 and relerr(x.grad, xd.grad) < 1e-6 and relerr(b.grad, bd.grad) < 2e-6
    B, n, C = 3, 128, 1024
    h0, go = torch.randn(B * n, C, generator=g), torch.randn(B, C, generator=g)
    h = h0.cuda().requires_grad_(True)
    m = _RowsMaxFn.apply(h, B, n)
    m.backward(go.cuda())
    hd = h0.double().requires_grad_(True)
    md = hd.view(B, n, C).max(dim=1)[0]
    md.backward(go.double())
    assert torch.equal(m.detach().cpu().double(), md.detach()) and torch.equal(h.grad.cpu().double(), hd.grad)
