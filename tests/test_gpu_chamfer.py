"""Chamfer HIP kernels vs (a) the golden outputs of the reference's own pure-PyTorch equivalent
(chamfer_python.distChamfer, the relation its unit_test.py:22-33 asserts: sq-err < 1e-8 and exact idx) and
(b) the C oracle (bit-exact dist and idx: same fmaf formula, first-minimum tie break)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(lib_built):
    return load_golden("chamfer.npz")


def _fwd(a, b):
    from vn_pointcloudcompletion_amd.chamfer_distance import chamfer_3DDist
    return chamfer_3DDist()(a, b)


def test_reference_unit_test_shapes(G):
    a, b = t(G["a"]).requires_grad_(True), t(G["b"]).requires_grad_(True)
    d1, d2, i1, i2 = _fwd(a, b)
    # unit_test.py:23-33: mean squared error sums < 1e-8, indices exactly equal
    assert ((d1.cpu() - torch.as_tensor(G["d1"])) ** 2).mean().item() + ((d2.cpu() - torch.as_tensor(G["d2"])) ** 2).mean().item() < 1e-8
    assert np.array_equal(i1.cpu().numpy(), G["i1"]) and np.array_equal(i2.cpu().numpy(), G["i2"])
    ((d1 * t(G["g1"])).sum() + (d2 * t(G["g2"])).sum()).backward()
    assert np.abs(a.grad.cpu().numpy() - G["ga"]).max() < 1e-5
    assert np.abs(b.grad.cpu().numpy() - G["gb"]).max() < 1e-5


def test_golden_1024_vs_16384(G):
    d1, d2, i1, i2 = _fwd(t(G["a2"]), t(G["b2"]))
    assert np.abs(d1.cpu().numpy() - G["e1"]).max() < 1e-6 and np.abs(d2.cpu().numpy() - G["e2"]).max() < 1e-6
    assert np.array_equal(i1.cpu().numpy(), G["j1"]) and np.array_equal(i2.cpu().numpy(), G["j2"])


@pytest.mark.parametrize("B,n,m", [(1, 1, 1), (2, 5, 1030), (3, 1025, 7), (2, 2048, 4096), (1, 4100, 1023)])
def test_bit_exact_vs_oracle(lib_built, B, n, m):
    from oracle import chamfer_oracle as CO
    g = torch.Generator().manual_seed(B * 1000 + n + m)
    a = torch.rand(B, n, 3, generator=g) - 0.5
    b = torch.rand(B, m, 3, generator=g) - 0.5
    d1, d2, i1, i2 = _fwd(a.cuda(), b.cuda())
    o1, o2, j1, j2 = CO.chamfer_forward(a.numpy(), b.numpy())
    assert np.array_equal(d1.cpu().numpy(), o1) and np.array_equal(d2.cpu().numpy(), o2)
    assert np.array_equal(i1.cpu().numpy(), j1) and np.array_equal(i2.cpu().numpy(), j2)


def test_ties_first_minimum_wins(lib_built):
    from oracle import chamfer_oracle as CO
    # duplicated reference points (across LDS tiles and reference slices): the smallest index must win
    base = torch.rand(1, 300, 3) - 0.5
    b = base.repeat(1, 12, 1)                       # 3600 points, each 12 times
    a = base[:, :64] + 1e-3
    d1, d2, i1, i2 = _fwd(a.cuda(), b.cuda())
    o1, o2, j1, j2 = CO.chamfer_forward(a.numpy(), b.numpy())
    assert np.array_equal(i1.cpu().numpy(), j1) and np.array_equal(i2.cpu().numpy(), j2)
    assert (i1.cpu().numpy() < 300).all()


def test_backward_vs_oracle(lib_built):
    from oracle import chamfer_oracle as CO
    g = torch.Generator().manual_seed(5)
    a = (torch.rand(2, 700, 3, generator=g) - 0.5).cuda().requires_grad_(True)
    b = (torch.rand(2, 1500, 3, generator=g) - 0.5).cuda().requires_grad_(True)
    d1, d2, i1, i2 = _fwd(a, b)
    g1, g2 = torch.randn(2, 700, generator=g).cuda(), torch.randn(2, 1500, generator=g).cuda()
    ((d1 * g1).sum() + (d2 * g2).sum()).backward()
    oa, ob = CO.chamfer_backward(a.detach().cpu().numpy(), b.detach().cpu().numpy(), g1.cpu().numpy(), g2.cpu().numpy(),
                                 i1.cpu().numpy(), i2.cpu().numpy())
    # float atomics: summation order differs from the sequential oracle -> tolerance, not bit equality
    assert np.abs(a.grad.cpu().numpy() - oa).max() < 1e-5 and np.abs(b.grad.cpu().numpy() - ob).max() < 1e-5


def test_losses_match_oracle_formulas(lib_built):
    from oracle import vn_oracle as O
    from vn_pointcloudcompletion_amd import loss, metric
    g = torch.Generator().manual_seed(9)
    a = (torch.rand(3, 256, 3, generator=g) - 0.5)
    b = (torch.rand(3, 512, 3, generator=g) - 0.5)
    d1, d2, _, _ = O.chamfer_torch(a, b)
    assert abs(loss.cd_loss_L1(a.cuda(), b.cuda()).item() - O.cd_loss_l1(d1, d2).item()) < 1e-6
    assert abs(loss.cd_loss_L2(a.cuda(), b.cuda()).item() - O.cd_loss_l2(d1, d2).item()) < 1e-6
    assert abs(metric.l1_cd(a.cuda(), b.cuda()).item() - O.l1_cd_metric(d1, d2).item()) < 1e-5
    assert abs(metric.l2_cd(a.cuda(), b.cuda()).item() - O.l2_cd_metric(d1, d2).item()) < 1e-5


def test_cpu_tensor_raises(lib_built):
    from vn_pointcloudcompletion_amd import _lib
    with pytest.raises(_lib.VnpError):
        _fwd(torch.rand(1, 4, 3), torch.rand(1, 5, 3))


def test_dcd_and_fscore_match_oracle(lib_built):
    """calc_dcd (metrics/loss.py:77-118) and fscore (ChamferDistancePytorch/fscore.py) vs the oracle restatement, incl. grads."""
    from oracle import vn_oracle as O
    from vn_pointcloudcompletion_amd import loss, metric
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(3, 200, 3, generator=g) - 0.5)
    gt = (torch.rand(3, 500, 3, generator=g) - 0.5)
    xr = x.cuda().requires_grad_(True)
    l, cd_p, cd_t = loss.calc_dcd(xr, gt.cuda(), alpha=200, n_lambda=0.5)
    xo = x.clone().requires_grad_(True)
    lo, po, to = O.calc_dcd_oracle(xo, gt, alpha=200, n_lambda=0.5)
    assert (l.cpu() - lo).abs().max().item() < 1e-5 and (cd_p.cpu() - po).abs().max().item() < 1e-6
    assert (cd_t.cpu() - to).abs().max().item() < 1e-6
    l.mean().backward()
    lo.mean().backward()
    assert (xr.grad.cpu() - xo.grad).abs().max().item() < 1e-5 * max(1.0, xo.grad.abs().max().item())
    d1, d2, _, _ = O.chamfer_torch(x, gt)
    f, p1, p2 = metric.fscore(d1.cuda(), d2.cuda(), 0.01)
    fo, _, _ = O.fscore_oracle(d1, d2, 0.01)
    assert (f.cpu() - fo).abs().max().item() < 1e-6
    assert abs(metric.f_score(x[0].numpy(), gt[0].numpy(), th=0.1) - 2 * ((d2[0] < 0.01).float().mean() * (d1[0] < 0.01).float().mean()
               / ((d2[0] < 0.01).float().mean() + (d1[0] < 0.01).float().mean())).item()) < 1e-6
