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


@pytest.mark.parametrize("tag", ["a", "b"])
def test_dcd_and_fscore_match_reference_golden(lib_built, tag):
    """PINNED: tests/golden/dcd.npz holds the outputs of the reference's own calc_cd / calc_dcd (utils/loss.py:14-74) and
    fscore (fscore.py:3-16), imported and run by make_golden.py::main_dcd.  Tolerance 2e-6 absolute on losses / distances
    (the reference's distChamfer forms |x|^2+|y|^2-2xy in fp64; ours is the direct fp32 form), indices exact."""
    from vn_pointcloudcompletion_amd import loss, metric
    G = load_golden("dcd.npz")
    x, gt = t(G[f"x_{tag}"]), t(G[f"gt_{tag}"])
    for j in range(3):
        alpha, lam, non_reg = G[f"opts_{tag}{j}"]
        xr = x.clone().requires_grad_(True)
        res = loss.calc_dcd(xr, gt, alpha=float(alpha), n_lambda=float(lam), return_raw=True, non_reg=bool(non_reg))
        assert len(res) == 7
        assert np.abs(res[0].detach().cpu().numpy() - G[f"loss_{tag}{j}"]).max() < 2e-6, j
        res[0].sum().backward()
        ref = G[f"gx_{tag}{j}"]
        assert np.abs(xr.grad.cpu().numpy() - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), j
        if j == 0:
            assert (np.abs(res[1].detach().cpu().numpy() - G[f"cd_p_{tag}"]) / np.maximum(G[f"cd_p_{tag}"], 1.0)).max() < 2e-6
            assert (np.abs(res[2].detach().cpu().numpy() - G[f"cd_t_{tag}"]) / np.maximum(G[f"cd_t_{tag}"], 1.0)).max() < 2e-6
            for ours, ref in ((res[3], G[f"d1_{tag}"]), (res[4], G[f"d2_{tag}"])):      # case b holds distances ~70: relative
                assert (np.abs(ours.detach().cpu().numpy() - ref) / np.maximum(ref, 1.0)).max() < 2e-6
            assert np.array_equal(res[5].cpu().numpy(), G[f"i1_{tag}"]) and np.array_equal(res[6].cpu().numpy(), G[f"i2_{tag}"])
    d1, d2 = t(G[f"d1_{tag}"]), t(G[f"d2_{tag}"])
    for th in (0.001, 0.0001, 0.01):
        f, p1, p2 = metric.fscore(d1, d2, th)
        assert np.abs(torch.stack([f, p1, p2]).cpu().numpy() - G[f"f_{tag}_{th}"]).max() < 1e-6, th   # 0/0 -> 0 included (case b)
    sep_p, sep_t, f1 = loss.calc_cd(x, gt, calc_f1=True, separate=True)
    assert (np.abs(sep_p.cpu().numpy() - G[f"sep_p_{tag}"]) / np.maximum(G[f"sep_p_{tag}"], 1.0)).max() < 2e-6
    assert (np.abs(sep_t.cpu().numpy() - G[f"sep_t_{tag}"]) / np.maximum(G[f"sep_t_{tag}"], 1.0)).max() < 2e-6
    assert np.abs(f1.cpu().numpy() - G[f"f1_{tag}"]).max() <= 1.0 / 128 + 1e-7     # threshold 1e-4 sits on distance noise: one point may flip


@pytest.mark.parametrize("B,N,T,lam", [(3, 1000, 37, 0.5), (2, 16384, 16384, 1.0), (1, 5, 3, 0.5), (4, 2049, 700, 2.0)])
def test_dcd_side_histogram_and_mean(lib_built, B, N, T, lam):
    """vnp_dcd_side through the C ABI: the histogram of the arg-min indices equals torch.bincount per sample (exact: integer atomics),
    loss and d loss / d dist equal the reference's formula (metrics/loss.py:99-113) evaluated in torch fp32 / fp64; ragged sizes, heavy
    collisions (T << N) and targets nobody hits."""
    from vn_pointcloudcompletion_amd._lib import call, ptr
    g = torch.Generator().manual_seed(B * 1000 + N + T)
    dist = (torch.rand(B, N, generator=g) * 0.02).cuda()
    idx = torch.randint(0, max(1, T - 1), (B, N), generator=g, dtype=torch.int32).cuda()          # target T - 1 is never hit
    hits = torch.full((B, T), -7, device="cuda", dtype=torch.int32)                               # scratch need not be zeroed
    loss = torch.empty(B, device="cuda")
    gfac = torch.empty(B, N, device="cuda")
    alpha, frac = 200.0, T / N
    call("vnp_dcd_side", B, N, T, ptr(dist), ptr(idx), alpha, lam, frac, ptr(hits), ptr(loss), ptr(gfac))
    ref_hits = torch.stack([torch.bincount(idx[b].long(), minlength=T) for b in range(B)])
    assert torch.equal(hits.long(), ref_hits)
    d64 = dist.double().requires_grad_(True)
    w = (ref_hits.gather(1, idx.long()).float() ** lam + 1e-6) ** (-1) * frac
    ref = (1.0 - torch.exp(-alpha * d64) * w.double()).mean(dim=1)
    assert (loss.double() - ref).abs().max().item() < 2e-6
    ref.sum().backward()
    assert (gfac.double() - d64.grad).abs().max().item() < 2e-6 * max(1.0, d64.grad.abs().max().item())
    loss2 = torch.empty(B, device="cuda")
    call("vnp_dcd_side", B, N, T, ptr(dist), ptr(idx), alpha, lam, frac, ptr(hits), ptr(loss2), None)   # no gradient wanted
    assert torch.equal(loss, loss2)                                                                # and deterministic


# ------------------------------------------------------------------------------------------- pruned search (dense clouds)
def _both_paths(a, b):
    """(brute-force kernel outputs, pruned kernel outputs) through the C ABI, same inputs."""
    from vn_pointcloudcompletion_amd import _lib
    from vn_pointcloudcompletion_amd._lib import call, ptr
    B, n, _ = a.shape
    m = b.shape[1]
    outs = []
    for pruned in (False, True):
        d1 = torch.empty(B, n, device="cuda"); d2 = torch.empty(B, m, device="cuda")
        i1 = torch.empty(B, n, device="cuda", dtype=torch.int32); i2 = torch.empty(B, m, device="cuda", dtype=torch.int32)
        if pruned:
            nb = _lib.lib().vnp_chamfer_pruned_ws_bytes(B, n, m)
            assert nb > 0
            ws = torch.empty((nb + 7) // 8, device="cuda", dtype=torch.int64)
            call("vnp_chamfer_fwd_pruned", B, n, ptr(a), m, ptr(b), ptr(d1), ptr(i1), ptr(d2), ptr(i2), ptr(ws))
        else:
            ws = torch.empty(B * max(n, m), device="cuda", dtype=torch.int64)
            call("vnp_chamfer_fwd", B, n, ptr(a), m, ptr(b), ptr(d1), ptr(i1), ptr(d2), ptr(i2), ptr(ws))
        outs.append((d1, d2, i1, i2))
    return outs


def _assert_same_bits(x, y):
    for u, v in zip(x, y):
        if u.dtype == torch.float32:
            assert torch.equal(u.view(torch.int32), v.view(torch.int32))         # bit pattern, so NaN == NaN too
        else:
            assert torch.equal(u, v)


def _surface(g, B, n):
    """points on a bumpy sphere: what completed shapes look like (empty interior, thin shell)."""
    v = torch.randn(B, n, 3, generator=g)
    v = v / v.norm(dim=-1, keepdim=True)
    return v * (0.4 + 0.05 * torch.sin(9 * v[..., :1]))


_CLOUDS = {
    "uniform": lambda g, B, n: torch.rand(B, n, 3, generator=g) - 0.5,
    "surface": _surface,
    "blob": lambda g, B, n: 0.01 * torch.randn(B, n, 3, generator=g) + 0.3,                 # compact, far from the other cloud
    "lattice": lambda g, B, n: torch.randint(0, 12, (B, n, 3), generator=g).float() * 0.125,  # many exactly equal distances
    "line": lambda g, B, n: torch.cat([torch.rand(B, n, 1, generator=g), torch.zeros(B, n, 2)], -1),   # degenerate extent
    "huge": lambda g, B, n: (torch.rand(B, n, 3, generator=g) - 0.5) * 1e18,
}


@pytest.mark.parametrize("qa,qb", [("uniform", "uniform"), ("surface", "surface"), ("blob", "uniform"), ("uniform", "blob"),
                                   ("lattice", "lattice"), ("line", "surface"), ("huge", "huge"), ("blob", "blob")])
@pytest.mark.parametrize("B,n,m", [(2, 4096, 4096), (1, 16384, 14336), (3, 5000, 4097), (1, 19968, 32768)])
def test_pruned_equals_brute_force_bits(lib_built, qa, qb, B, n, m):
    g = torch.Generator().manual_seed(n + m + len(qa) * 7 + len(qb))
    a, b = _CLOUDS[qa](g, B, n).cuda().contiguous(), _CLOUDS[qb](g, B, m).cuda().contiguous()
    brute, pruned = _both_paths(a, b)
    _assert_same_bits(brute, pruned)


def test_pruned_ties_duplicates_and_oracle(lib_built):
    """Every point present 4 times (up-sampling by repetition) and queries that coincide with references: the lowest
    original index must win although chunks are visited out of index order; also checked against the C oracle."""
    from oracle import chamfer_oracle as CO
    g = torch.Generator().manual_seed(11)
    base = torch.rand(2, 1100, 3, generator=g) - 0.5
    b = base.repeat(1, 4, 1)[:, torch.randperm(4400, generator=g)]                # 4400 refs, duplicates scattered
    a = torch.cat([base[:, :600], base.repeat(1, 4, 1)[:, :3600] + 1e-4], 1)       # 4200 queries, 600 exactly on a reference
    brute, pruned = _both_paths(a.cuda().contiguous(), b.cuda().contiguous())
    _assert_same_bits(brute, pruned)
    o1, o2, j1, j2 = CO.chamfer_forward(a.numpy(), b.contiguous().numpy())
    assert np.array_equal(pruned[0].cpu().numpy(), o1) and np.array_equal(pruned[1].cpu().numpy(), o2)
    assert np.array_equal(pruned[2].cpu().numpy(), j1) and np.array_equal(pruned[3].cpu().numpy(), j2)


def test_pruned_non_finite_inputs_match_brute_force(lib_built):
    g = torch.Generator().manual_seed(12)
    a = torch.rand(1, 4096, 3, generator=g) - 0.5
    b = torch.rand(1, 4096, 3, generator=g) - 0.5
    a[0, 17, 1] = float("nan"); a[0, 900, 0] = float("inf"); b[0, 5, 2] = float("nan"); b[0, 4000, 0] = -float("inf")
    brute, pruned = _both_paths(a.cuda(), b.cuda())
    _assert_same_bits(brute, pruned)


def test_pruned_is_the_path_dense_clouds_take(lib_built):
    """ops.ChamferFn routes clouds of >= 4096 points through the pruned kernels (and says how to opt out)."""
    from vn_pointcloudcompletion_amd import ops
    assert ops.SW.CHAMFER_PRUNED_MIN_POINTS == 4096
    g = torch.Generator().manual_seed(13)
    a, b = (torch.rand(2, 4096, 3, generator=g) - 0.5).cuda(), (torch.rand(2, 8192, 3, generator=g) - 0.5).cuda()
    d1, d2, i1, i2 = _fwd(a.clone().requires_grad_(True), b)
    brute, _ = _both_paths(a, b)
    _assert_same_bits(brute, (d1.detach(), d2.detach(), i1, i2))


@pytest.mark.parametrize("B,n,m", [(2, 100, 200), (3, 1024, 4096), (2, 4096, 4096), (1, 5, 3)])
def test_fused_chamfer_l1_loss_equals_the_operator_followed_by_the_formula(lib_built, B, n, m):
    """loss.cd_loss_L1 as ONE Function (ops.ChamferL1LossFn: search, vnp_cd_l1_loss, and backward vnp_cd_l1_grad -> vnp_chamfer_bwd) against
    the reference's formula (metrics/loss.py:28-31) written with ATen operators on ChamferDistance's outputs: the loss within 2e-7 relative
    (double sums here, torch's float tree there), the gradients of both clouds bit for bit (the distance gradients are the same three
    IEEE operations in the same order), with a cotangent other than 1, brute-force and pruned searches."""
    from vn_pointcloudcompletion_amd import loss as L
    g = torch.Generator().manual_seed(n + m)
    a0, b0 = (torch.rand(B, n, 3, generator=g) - 0.5).cuda(), (torch.rand(B, m, 3, generator=g) - 0.5).cuda()
    a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    fused = L.cd_loss_L1(a, b)
    (fused * 0.7).backward()
    a2, b2 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    d1, d2 = L.CD(a2, b2)
    plain = (torch.mean(torch.sqrt(d1)) + torch.mean(torch.sqrt(d2))) / 2.0
    (plain * 0.7).backward()
    assert fused.dim() == 0 and abs(fused.item() - plain.item()) <= 2e-7 * abs(plain.item())
    assert torch.equal(a.grad, a2.grad) and torch.equal(b.grad, b2.grad)
    # only one of the operator's two distance outputs used: the other's gradient arrives as None (no zero-filled stand-ins any more)
    a3, b3 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    L.CD(a3, b3)[0].sum().backward()
    a4, b4 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    d1, d2 = L.CD(a4, b4)
    (d1.sum() + 0.0 * d2.sum()).backward()
    assert torch.equal(a3.grad, a4.grad) and torch.equal(b3.grad, b4.grad)
