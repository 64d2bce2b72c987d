"""VN layer modules on the HIP path vs the golden fixtures generated from the reference
(tests/golden/layers.npz, make_golden.py section 1).  Inputs / weights / cotangents are the reference's;
tolerance 2e-5 relative to the tensor's max (fp32, K <= 8; the reference-vs-oracle gap is of the same order)."""
import numpy as np
import pytest
import torch

from conftest import fp32_mode_slack, load_golden, relerr, require_split_gemm, t

pytestmark = pytest.mark.gpu
TOL = 3e-5


@pytest.fixture(scope="module")
def G(lib_built):
    return load_golden("layers.npz")


def _run(mod, x, g):
    mod = mod.cuda()
    mod.train()
    xr = t(x).requires_grad_(True)
    y = mod(xr)
    y.backward(t(g))
    return y, xr.grad


@pytest.mark.parametrize("tag", ["d4", "d5"])
def test_vnlinear(G, tag):
    from vn_pointcloudcompletion_amd.vn_layers import VNLinear
    w = G[f"lin_{tag}_w"]
    m = VNLinear(w.shape[1], w.shape[0])
    m.map_to_feat.weight.data.copy_(torch.as_tensor(w))
    y, gx = _run(m, G[f"lin_{tag}_x"], G[f"lin_{tag}_g"])
    assert relerr(y, G[f"lin_{tag}_y"]) < TOL
    assert relerr(gx, G[f"lin_{tag}_gx"]) < TOL
    assert relerr(m.map_to_feat.weight.grad, G[f"lin_{tag}_gw"]) < TOL


@pytest.mark.parametrize("tag", ["d4", "d5"])
def test_vnleakyrelu(G, tag):
    from vn_pointcloudcompletion_amd.vn_layers import VNLeakyReLU
    w = G[f"leaky_{tag}_w"]
    m = VNLeakyReLU(w.shape[1])
    m.map_to_dir.weight.data.copy_(torch.as_tensor(w))
    y, gx = _run(m, G[f"x_{tag}"], G[f"leaky_{tag}_g"])
    assert relerr(y, G[f"leaky_{tag}_y"]) < TOL
    assert relerr(gx, G[f"leaky_{tag}_gx"]) < TOL
    assert relerr(m.map_to_dir.weight.grad, G[f"leaky_{tag}_gw"]) < TOL


@pytest.mark.parametrize("tag,dim", [("d4", 4), ("d5", 5)])
def test_vnbatchnorm(G, tag, dim):
    from vn_pointcloudcompletion_amd.vn_layers import VNBatchNorm
    C = G[f"bn_{tag}_gamma"].shape[0]
    m = VNBatchNorm(C, dim=dim)
    m.bn.weight.data.copy_(torch.as_tensor(G[f"bn_{tag}_gamma"]))
    m.bn.bias.data.copy_(torch.as_tensor(G[f"bn_{tag}_beta"]))
    y, gx = _run(m, G[f"x_{tag}"], G[f"bn_{tag}_g"])
    assert relerr(y, G[f"bn_{tag}_y"]) < TOL
    assert relerr(gx, G[f"bn_{tag}_gx"]) < 1e-4
    assert relerr(m.bn.weight.grad, G[f"bn_{tag}_ggamma"]) < TOL
    assert relerr(m.bn.bias.grad, G[f"bn_{tag}_gbeta"]) < TOL
    assert relerr(m.bn.running_mean, G[f"bn_{tag}_rm"]) < TOL
    assert relerr(m.bn.running_var, G[f"bn_{tag}_rv"]) < TOL
    assert int(m.bn.num_batches_tracked) == 1
    m.eval()
    with torch.no_grad():
        ye = m(t(G[f"x_{tag}"]))
    assert relerr(ye, G[f"bn_{tag}_y_eval"]) < TOL


@pytest.mark.parametrize("tag,dim", [("d4", 4), ("d5", 5)])
def test_vnlinearleakyrelu(G, tag, dim):
    from vn_pointcloudcompletion_amd.vn_layers import VNLinearLeakyReLU
    wf = G[f"ll_{tag}_wf"]
    m = VNLinearLeakyReLU(wf.shape[1], wf.shape[0], dim=dim)
    m.map_to_feat.weight.data.copy_(torch.as_tensor(wf))
    m.map_to_dir.weight.data.copy_(torch.as_tensor(G[f"ll_{tag}_wd"]))
    m.batchnorm.bn.weight.data.copy_(torch.as_tensor(G[f"ll_{tag}_gamma"]))
    m.batchnorm.bn.bias.data.copy_(torch.as_tensor(G[f"ll_{tag}_beta"]))
    y, gx = _run(m, G[f"x_{tag}"], G[f"ll_{tag}_g"])
    assert relerr(y, G[f"ll_{tag}_y"]) < TOL
    assert relerr(gx, G[f"ll_{tag}_gx"]) < 1e-4
    assert relerr(m.map_to_feat.weight.grad, G[f"ll_{tag}_gwf"]) < 1e-4
    assert relerr(m.map_to_dir.weight.grad, G[f"ll_{tag}_gwd"]) < 1e-4
    assert relerr(m.batchnorm.bn.weight.grad, G[f"ll_{tag}_ggamma"]) < 1e-4
    assert relerr(m.batchnorm.bn.bias.grad, G[f"ll_{tag}_gbeta"]) < 1e-4
    assert relerr(m.batchnorm.bn.running_mean, G[f"ll_{tag}_rm"]) < TOL
    assert relerr(m.batchnorm.bn.running_var, G[f"ll_{tag}_rv"]) < TOL
    m.eval()
    with torch.no_grad():
        ye = m(t(G[f"x_{tag}"]))
    assert relerr(ye, G[f"ll_{tag}_y_eval"]) < TOL


@pytest.mark.parametrize("tag,dim", [("d4", 4), ("d5", 5)])
def test_vnlinearandleakyrelu(G, tag, dim):
    from vn_pointcloudcompletion_amd.vn_layers import VNLinearAndLeakyReLU
    wl = G[f"lal_{tag}_wl"]
    m = VNLinearAndLeakyReLU(wl.shape[1], wl.shape[0], dim=dim, use_batchnorm="none")
    m.linear.map_to_feat.weight.data.copy_(torch.as_tensor(wl))
    m.leaky_relu.map_to_dir.weight.data.copy_(torch.as_tensor(G[f"lal_{tag}_wd"]))
    y, gx = _run(m, G[f"x_{tag}"], G[f"lal_{tag}_g"])
    assert relerr(y, G[f"lal_{tag}_y"]) < TOL
    assert relerr(gx, G[f"lal_{tag}_gx"]) < TOL
    assert relerr(m.linear.map_to_feat.weight.grad, G[f"lal_{tag}_gwl"]) < TOL
    assert relerr(m.leaky_relu.map_to_dir.weight.grad, G[f"lal_{tag}_gwd"]) < TOL


@pytest.mark.parametrize("tag", ["d4", "d5"])
def test_vnmaxpool(G, tag):
    from vn_pointcloudcompletion_amd.vn_layers import VNMaxPool
    w = G[f"pool_{tag}_w"]
    m = VNMaxPool(w.shape[0])
    m.map_to_dir.weight.data.copy_(torch.as_tensor(w))
    y, gx = _run(m, G[f"x_{tag}"], G[f"pool_{tag}_g"])
    assert tuple(y.shape) == G[f"pool_{tag}_y"].shape
    assert relerr(y, G[f"pool_{tag}_y"]) < TOL             # margins at this size are >> fp32 noise: same arg-max
    assert relerr(gx, G[f"pool_{tag}_gx"]) < TOL
    assert m.map_to_dir.weight.grad is None                 # arg-max detaches the direction weights


def test_layout_round_trip(lib_built):
    from vn_pointcloudcompletion_amd.vn_layers import from_rows, to_rows
    x = torch.randn(3, 5, 3, 7, 2, device="cuda")
    rows, meta = to_rows(x)
    assert rows.shape == (3 * 7 * 2 * 3, 5)
    # rows[(b*N+n)*3+t][c] == x[b][c][t][n]
    xf = x.reshape(3, 5, 3, 14)
    assert torch.equal(rows.reshape(3, 14, 3, 5), xf.permute(0, 3, 2, 1))
    assert torch.equal(from_rows(rows, meta), x)


def test_mean_pool(lib_built):
    from vn_pointcloudcompletion_amd.vn_layers import mean_pool
    x = torch.randn(2, 6, 3, 9, 4, device="cuda", requires_grad=True)
    y = mean_pool(x)
    ref = x.detach().mean(-1)
    assert relerr(y, ref) < 1e-6
    g = torch.randn_like(y)
    y.backward(g)
    assert relerr(x.grad, (g / 4).unsqueeze(-1).expand_as(x)) < 1e-6


@pytest.mark.parametrize("nf", [False, True])
def test_vnstdfeature(G, nf):
    from oracle.vn_oracle import closed_form_fill
    from vn_pointcloudcompletion_amd.vn_layers import VNStdFeature
    tag = "std_nf" if nf else "std"
    m = VNStdFeature(8, dim=4, normalize_frame=nf)
    closed_form_fill(m.state_dict())
    m = m.cuda().train()
    y, z = m(t(G[f"{tag}_x"]))
    assert relerr(y, G[f"{tag}_y"]) < 1e-4 and relerr(z, G[f"{tag}_z"]) < 1e-4


def _pool_three_ways(f, W, Wd, G, gs, stats=None):
    """(out, idx, grad f, grad W) of VNLinear -> VNMaxPool through the screened search, the full fused epilogue and the two-GEMM +
    pool-kernel path."""
    from vn_pointcloudcompletion_amd import ops
    res = {}
    for name, fused, screened, pas in (("screened", True, True, "fp16"), ("screened_bf16x3", True, True, "bf16x3"), ("fused", True, False, "fp16"),
                                       ("unfused", False, False, "fp16")):
        with ops.switches(POOL_FUSED_EPILOGUE=fused, POOL_SCREENED=screened, POOL_SCREEN_PASS=pas), \
                ops.hooks(pool_screen_stats=stats if name == "screened" else None):
            fx, Wx = f.cuda().requires_grad_(True), W.cuda().requires_grad_(True)
            out, idx = ops.VNLinearMaxPoolFn.apply(fx, Wx, Wd.cuda(), G, gs)
            (out * torch.arange(out.numel(), device="cuda").reshape(out.shape).cos()).sum().backward()
            res[name] = (out.detach(), idx, fx.grad, Wx.grad)
    return res


def _assert_pool_bits_equal(res):
    for name in ("screened", "screened_bf16x3", "fused"):
        assert torch.equal(res[name][1], res["unfused"][1]), name                                  # indices
        assert torch.equal(res[name][0].view(torch.int32), res["unfused"][0].view(torch.int32)), name   # pooled vectors, bit patterns
        assert torch.equal(res[name][2], res["unfused"][2]) and torch.equal(res[name][3], res["unfused"][3]), name


@pytest.mark.parametrize("G,gs,C,K", [(3, 64, 128, 64), (2, 2048, 256, 96), (1, 96, 64, 32), (5, 32, 192, 160), (2, 2048, 128, 1024)])
def test_linear_pool_fused_epilogue_equals_unfused(lib_built, G, gs, C, K):
    """VNLinear -> VNMaxPool with the pool in the GEMM epilogue — the exact screened search (vnp_linear_pool_screened_fwd, the
    default) and the full 6-product epilogue (vnp_linear_pool_fwd) — against the two-GEMM + pool-kernel path: same arg-max indices
    and pooled vectors, bit for bit, including exact ties (duplicated points: the lowest index wins) — and the same gradients,
    since backward only needs the indices."""
    require_split_gemm()
    g = torch.Generator().manual_seed(G * 100 + gs + C + K)
    f = torch.randn(G * gs * 3, K, generator=g)
    f[3 * (gs // 2):3 * (gs // 2) + 3] = f[0:3]                  # point gs/2 duplicates point 0 (group 0): exact ties
    f[3 * (gs - 1):3 * gs] = f[3:6]                              # and the last point duplicates point 1
    W = (torch.randn(C, K, generator=g) / K ** 0.5)
    Wd = (torch.randn(C, C, generator=g) / C ** 0.5)
    stats = []
    res = _pool_three_ways(f, W, Wd, G, gs, stats)
    _assert_pool_bits_equal(res)
    # the ties really occurred and were resolved to the lower index
    i0 = res["screened"][1][0]
    assert (i0 != gs // 2).all() and (i0 != gs - 1).all()
    if stats:                                                    # shapes the fused kernels do not cover take the unfused path
        cand, flag, cap = stats[0].tolist()
        print(f"screened pool G={G} gs={gs} C={C} K={K}: {cand} candidates for {G * C} (group, channel) pairs, fallback flag {flag}")
        assert flag == 0 and G * C <= cand <= cap
    else:
        assert (G * gs * 3) % 192 != 0


def test_screened_pool_adversarial_near_ties(lib_built):
    """Scores that differ in their last bits only: every point is a tiny perturbation (1e-7 relative: single ulps of the
    features) of one of four base points, so each channel has hundreds of candidates inside any error bound and the decision rests
    entirely on the exact candidate evaluation reproducing the full kernels' bits (and on the fallback when the list overflows)."""
    require_split_gemm()
    g = torch.Generator().manual_seed(5)
    G, gs, C, K = 2, 256, 128, 256
    base = torch.randn(4, 3, K, generator=g)
    pick = torch.randint(0, 4, (G * gs,), generator=g)
    f = base[pick] * (1.0 + 1.2e-7 * torch.randint(-2, 3, (G * gs, 3, K), generator=g).float())
    f = f.reshape(G * gs * 3, K).contiguous()
    W = torch.randn(C, K, generator=g) / K ** 0.5
    Wd = torch.randn(C, C, generator=g) / C ** 0.5
    # three near-tied points that dominate the pool (3x the norm of the rest: their scores are ~9x larger), so that the
    # candidate list does NOT overflow and the exact candidate evaluation — not the fallback — decides between last-bit differences
    few = torch.randn(G * gs, 3, K, generator=g)
    dom = 3.0 * torch.randn(1, 3, K, generator=g)
    for b in range(G):
        for n in (5, 70, 140):                                   # in three different 32-point blocks (two in one block would make
            few[b * gs + n] = dom[0] * (1.0 + 1.2e-7 * torch.randint(-2, 3, (3, K), generator=g).float())   # the whole block a candidate)
    few = few.reshape(G * gs * 3, K).contiguous()
    for cap_note, ff in (("three dominant near-tied points", few), ("near ties", f), ("all points identical", base[0].reshape(1, 3, K).expand(G * gs, 3, K).reshape(G * gs * 3, K).contiguous())):
        stats = []
        res = _pool_three_ways(ff, W, Wd, G, gs, stats)
        _assert_pool_bits_equal(res)
        cand, flag, cap = stats[0].tolist()
        print(f"{cap_note}: {cand} candidates (capacity {cap}), fallback flag {flag}")
        if cap_note == "three dominant near-tied points":
            assert flag == 0 and cand > 1.5 * G * C              # several candidates per pair, all evaluated exactly
            win = res["screened"][1]
            assert ((win == 5) | (win == 70) | (win == 140)).float().mean().item() > 0.15
        if cap_note == "all points identical":
            assert flag == 1                                     # 256 exact ties per channel > 8 per pair: the predicated full kernel ran
            assert (res["screened"][1] == 0).all()               # and the lowest index won everywhere


@pytest.mark.parametrize("scale", [3e-6, 1e-3, 3e4])
def test_screened_pool_fp16_range_edges(lib_built, scale):
    """The fp16 bracketing pass outside fp16's comfortable range: features of magnitude 3e-6 (fp16 subnormals: the bound's absolute
    term; the hardware multiplies them exactly, profiles/r02_mfma_f16_subnormals.txt), 1e-3, and 3e4 (products of the direction map
    overflow fp16's 65504 for some operands: non-finite bound -> fallback).  Whatever the pass does, the result is the exact one."""
    require_split_gemm()
    g = torch.Generator().manual_seed(17)
    G, gs, C, K = 2, 256, 128, 128
    f = torch.randn(G * gs * 3, K, generator=g) * scale
    W = torch.randn(C, K, generator=g) / K ** 0.5
    Wd = torch.randn(C, C, generator=g) / C ** 0.5
    stats = []
    res = _pool_three_ways(f, W, Wd, G, gs, stats)
    _assert_pool_bits_equal(res)
    cand, flag, cap = stats[0].tolist()
    print(f"scale {scale:g}: {cand} candidates (capacity {cap}), fallback flag {flag}")


def test_screened_pool_random_shapes_and_magnitudes(lib_built):
    """Fuzz: 24 random problems — group counts, group sizes (one to 64 blocks of 32 points), channel counts, contraction lengths,
    feature / weight magnitudes over 6 decades, random duplicated points, heavy-tailed features — both bracketing passes against the full
    six-product epilogue (itself tested against the two-GEMM path above): indices and pooled vectors bit for bit."""
    require_split_gemm()
    from vn_pointcloudcompletion_amd import ops
    rng = torch.Generator().manual_seed(2024)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=rng).item())
    flags = 0
    for trial in range(24):
        G = ri(1, 6)
        gs = 32 * ri(1, 64) if trial % 3 else 32 * [1, 2, 64][ri(0, 2)]
        while (G * gs * 3) % 192:                                   # shapes the fused kernels cover
            G += 1
        C, K = 64 * ri(1, 5), 32 * ri(1, 40)
        fs, ws = 10.0 ** (ri(-30, 30) / 10.0), 10.0 ** (ri(-20, 10) / 10.0)
        f = torch.randn(G * gs * 3, K, generator=rng) * fs
        if trial % 4 == 1:
            f = f * torch.randn(G * gs * 3, 1, generator=rng).pow(4)          # heavy tails: a few dominant points
        if trial % 4 == 2 and gs > 32:
            src = torch.randint(0, G * gs, (G * gs // 8,), generator=rng)      # duplicated points: exact ties
            dst = torch.randint(0, G * gs, (G * gs // 8,), generator=rng)
            f.view(G * gs, 3, K)[dst] = f.view(G * gs, 3, K)[src]
        W = torch.randn(C, K, generator=rng) * ws / K ** 0.5
        Wd = torch.randn(C, C, generator=rng) / C ** 0.5
        out = {}
        for name, screened, pas in (("fp16", True, "fp16"), ("bf16x3", True, "bf16x3"), ("full", False, "fp16")):
            st = []
            with ops.switches(POOL_SCREENED=screened, POOL_SCREEN_PASS=pas), ops.hooks(pool_screen_stats=st if screened else None):
                out[name] = ops.VNLinearMaxPoolFn.apply(f.cuda(), W.cuda(), Wd.cuda(), G, gs)
            if st:
                flags += int(st[0][1].item())
        for name in ("fp16", "bf16x3"):
            assert torch.equal(out[name][1], out["full"][1]), (trial, name, G, gs, C, K, fs, ws)
            assert torch.equal(out[name][0].view(torch.int32), out["full"][0].view(torch.int32)), (trial, name, G, gs, C, K, fs, ws)
    print(f"fallback flag raised in {flags} of 48 screened calls")


@pytest.mark.parametrize("G,gs,C,K", [(2, 256, 128, 128), (3, 1024, 256, 192), (4, 2048, 512, 256)])
def test_screen_pass_tiles_write_the_same_slots(lib_built, G, gs, C, K):
    """The fp16 screening pass on its three tiles — 384 x 256 with LDS-DMA staging (screen_xwide_kernel.h, what maxpool2 takes), 192 x 256
    (screen_wide_kernel.h) and 192 x 128 — on shapes all three accept: the same candidate count (a sum over every slot's bracket: it moves
    if one bound of one slot moves), no fallback, the same picks and pooled vectors bit for bit, equal to the full six-product epilogue's.
    (On maxpool2's own shape the slots of the two wide tiles were compared word for word: profiles/r04_screen_ablation.txt.)"""
    require_split_gemm()
    from vn_pointcloudcompletion_amd import ops
    assert (G * gs * 3) % 384 == 0 and (2 * C) % 256 == 0 and K % 64 == 0
    g = torch.Generator().manual_seed(G * gs + C + K)
    f = (torch.randn(G * gs * 3, K, generator=g) * torch.exp2(torch.randint(-3, 4, (G * gs * 3, 1), generator=g).float())).cuda()
    W = (torch.randn(C, K, generator=g) / K ** 0.5).cuda()
    Wd = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    res = {}
    for tile in ("auto", "wide", "narrow"):
        st = []
        with ops.switches(POOL_SCREEN_TILE=tile, POOL_SCREEN_MIN_K=0), ops.hooks(pool_screen_stats=st):
            out, idx = ops.VNLinearMaxPoolFn.apply(f, W, Wd, G, gs)
        res[tile] = (out, idx, st[0].tolist())
    with ops.switches(POOL_SCREENED=False):
        full = ops.VNLinearMaxPoolFn.apply(f, W, Wd, G, gs)
    for tile in ("auto", "wide", "narrow"):
        out, idx, (cand, flag, cap) = res[tile]
        assert flag == 0 and cand == res["narrow"][2][0], (tile, cand, res["narrow"][2])
        assert torch.equal(idx, full[1]) and torch.equal(out.view(torch.int32), full[0].view(torch.int32)), tile


def test_screened_pool_backs_off_after_a_fallback(lib_built):
    """After a call whose candidate list overflowed, the next calls of that shape skip the bracketing pass (the flag travels to the host
    through a non-blocking copy: no sync in the forward) and the search is tried again once the back-off has run out."""
    require_split_gemm()
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(9)
    G, gs, C, K = 2, 64, 64, 32
    same = torch.randn(1, 3, K, generator=g).expand(G * gs, 3, K).reshape(G * gs * 3, K).contiguous().cuda()   # every point identical: overflow
    W, Wd = (torch.randn(C, K, generator=g) / K ** 0.5).cuda(), (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    key = (torch.cuda.current_device(), G, gs, C, K)
    ops._screen_state.pop(key, None)
    try:
        paths = []
        with ops.switches(POOL_SCREEN_BACKOFF=3, POOL_SCREEN_MIN_K=0):        # (a K this small takes the full epilogue by default)
            for _ in range(7):
                tr = {}
                with ops.hooks(trace=tr):
                    ops.VNLinearMaxPoolFn.apply(same, W, Wd, G, gs)
                    torch.cuda.synchronize()                   # lets the flag of this call be visible to the next one
                paths.append("screened" if "linear_pool_screened" in tr else "full")
        assert paths == ["screened", "full", "full", "full", "screened", "full", "full"], paths
    finally:
        ops._screen_state.pop(key, None)


def test_screened_pool_non_finite_inputs_take_the_fallback(lib_built):
    require_split_gemm()
    g = torch.Generator().manual_seed(6)
    G, gs, C, K = 2, 64, 64, 64
    f = torch.randn(G * gs * 3, K, generator=g)
    f[17, 3] = float("nan")
    f[250, 9] = float("inf")
    W = torch.randn(C, K, generator=g) / K ** 0.5
    Wd = torch.randn(C, C, generator=g) / C ** 0.5
    from vn_pointcloudcompletion_amd import ops
    stats = []
    out = {}
    for name, screened in (("screened", True), ("fused", False)):
        with ops.switches(POOL_SCREENED=screened), ops.hooks(pool_screen_stats=stats if screened else None):
            out[name] = ops.VNLinearMaxPoolFn.apply(f.cuda(), W.cuda(), Wd.cuda(), G, gs)
    assert stats[0][1].item() == 1
    assert torch.equal(out["screened"][1], out["fused"][1])
    assert torch.equal(out["screened"][0].view(torch.int32), out["fused"][0].view(torch.int32))


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("rank1", [False, True])
def test_bn_backward_two_pass_equals_bwd1_bwd2(lib_built, training, rank1):
    """vnp_vn_bn_leaky_bwd_pass (sums, then final gradients with phase 2 applied on the fly) against vnp_vn_bn_leaky_bwd1 ->
    vnp_vn_bn_bwd2: the same arithmetic term for term, so dL/dp, dL/dd, dL/dgamma, dL/dbeta (and dL/dw_out in the rank-1 form of the
    fused decoder tail) agree to the last bits of fp32 contraction (1e-6 relative)."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(31)
    P, C = 3000, 96
    pd = torch.randn(P * 3, 2 * C, generator=g).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    _, mean, rstd = ops._bn_leaky_forward(pd[:, :C], pd[:, C:], C, gamma, beta, rm, rv, True, 0.2)
    go = torch.randn(P * 3, C, generator=g).cuda()
    grow, gcol = torch.randn(P * 3, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    res = {}
    for two in (True, False):
        ops.SW.BN_BWD_TWO_PASS = two
        try:
            if rank1:
                res[two] = ops._bn_leaky_backward(None, pd[:, :C], pd[:, C:], C, gamma, beta, mean, rstd, training, 0.2,
                                                  go_row=grow, go_col=gcol, want_gw=True)
            else:
                res[two] = ops._bn_leaky_backward(go, pd[:, :C], pd[:, C:], C, gamma, beta, mean, rstd, training, 0.2)     # (matrix form: both take bwd1 -> bwd2 today)
        finally:
            ops.SW.BN_BWD_TWO_PASS = True
    for a, b in zip(res[True], res[False]):
        assert relerr(a, b) < 2e-6


@pytest.mark.parametrize("P,rpg", [(4096, 2048 * 3), (3000, 1500 * 3), (6144, 3072 * 3)])
def test_bn_backward_fused_per_sample_colsum(lib_built, P, rpg):
    """The per-sample bias gradient (column sums of dL/dp | dL/dd per sample and component) taken inside the final-gradient pass
    (vnp_vn_bn_leaky_bwd_pass, colsum) against the separate pass over the gradient; P = 3000: a sample is not a whole number of chunks,
    so the separate pass is used on both sides."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(P)
    C = 96
    pd = torch.randn(P * 3, 2 * C, generator=g).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    _, mean, rstd = ops._bn_leaky_forward(pd[:, :C], pd[:, C:], C, gamma, beta, rm, rv, True, 0.2)
    go = torch.randn(P * 3, C, generator=g).cuda()
    res = {}
    for two in (True, False):
        ops.SW.BN_BWD_TWO_PASS = two
        try:
            res[two] = ops._bn_leaky_backward(go, pd[:, :C], pd[:, C:], C, gamma, beta, mean, rstd, True, 0.2, colsum_rows_per_group=rpg)
        finally:
            ops.SW.BN_BWD_TWO_PASS = True
    assert len(res[True]) == 4 and res[True][3].shape == (P * 3 // rpg * 3, 2 * C)
    for a, b in zip(res[True], res[False]):
        assert relerr(a, b) < 5e-6
    ref = res[False][0].double().reshape(P * 3 // rpg, rpg // 3, 3, 2 * C).sum(1).reshape(-1, 2 * C)
    assert relerr(res[True][3], ref) < 1e-5


def test_pool_tap_adds_into_the_other_consumers_gradient(lib_built):
    """VNMaxPoolTapFn (encoder maxpool1: the pooled tensor also feeds second_conv[0]) against the plain pool + autograd's own sum of
    the two gradients: same pooled output and indices, same gradient for the shared input (one writer per element: bit-equal)."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(8)
    G, gs, C, K = 3, 128, 64, 32
    a = torch.randn(G * gs * 3, K, generator=g).cuda()
    W = (torch.randn(C, K, generator=g) / K ** 0.5).cuda()
    Wd = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    W2 = (torch.randn(48, C, generator=g) / C ** 0.5).cuda()
    cot1, cot2 = torch.randn(G * 3, C, generator=g).cuda(), torch.randn(G * gs * 3, 48, generator=g).cuda()
    res = {}
    for tap in (False, True):
        ar = a.clone().requires_grad_(True)
        f = ops.VNLinearFn.apply(ar, W, None, 0)
        if tap:
            f2, pooled, idx = ops.VNMaxPoolTapFn.apply(f, Wd, G, gs, ar.detach(), W)
        else:
            pooled, idx = ops.VNMaxPoolFn.apply(f, Wd, G, gs, ar.detach(), W)
            f2 = f
        y = ops.VNLinearFn.apply(f2, W2, None, 0)                      # the other consumer
        ((pooled * cot1).sum() + (y * cot2).sum()).backward()
        res[tap] = (pooled.detach(), idx, ar.grad)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert torch.allclose(res[True][2], res[False][2], rtol=1e-6, atol=1e-6)
    # pooled alone (no other consumer): the dense fallback inside the tap
    ar = a.clone().requires_grad_(True)
    _, pooled, _ = ops.VNMaxPoolTapFn.apply(ops.VNLinearFn.apply(ar, W, None, 0), Wd, G, gs)
    (pooled * cot1).sum().backward()
    br = a.clone().requires_grad_(True)
    pooled2, _ = ops.VNMaxPoolFn.apply(ops.VNLinearFn.apply(br, W, None, 0), Wd, G, gs)
    (pooled2 * cot1).sum().backward()
    assert torch.allclose(ar.grad, br.grad, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("G,gs,C,K", [(2, 2048, 512, 128), (3, 64, 128, 64), (1, 192, 64, 32)])
def test_linear_pool_tap_fused_equals_linear_then_tap(lib_built, G, gs, C, K):
    """VNLinearPoolTapFn (one GEMM that stores y = x W^T and pools in its epilogue: vnp_linear_pool_store_fwd) against VNLinearFn ->
    VNMaxPoolTapFn: y, pooled vectors and indices bit for bit (the same MFMA sequence per element), gradients to fp32 summation order."""
    require_split_gemm()
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(G + gs + C + K)
    a = torch.randn(G * gs * 3, K, generator=g).cuda()
    a[3 * (gs // 2):3 * (gs // 2) + 3] = a[0:3]                    # an exact tie in group 0
    W = (torch.randn(C, K, generator=g) / K ** 0.5).cuda()
    Wd = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    W2 = (torch.randn(40, C, generator=g) / C ** 0.5).cuda()
    cot1, cot2 = torch.randn(G * 3, C, generator=g).cuda(), torch.randn(G * gs * 3, 40, generator=g).cuda()
    assert ops.VNLinearPoolTapFn.supported(a, W, G, gs)
    res = {}
    for fused in (True, False):
        ar, Wr = a.clone().requires_grad_(True), W.clone().requires_grad_(True)
        if fused:
            y, pooled, idx = ops.VNLinearPoolTapFn.apply(ar, Wr, Wd, G, gs)
        else:
            y0 = ops.VNLinearFn.apply(ar, Wr, None, 0)
            y, pooled, idx = ops.VNMaxPoolTapFn.apply(y0, Wd, G, gs, ar.detach(), Wr.detach())
        z = ops.VNLinearFn.apply(y, W2, None, 0)
        ((pooled * cot1).sum() + (z * cot2).sum()).backward()
        res[fused] = (y.detach().clone(), pooled.detach(), idx, ar.grad, Wr.grad)
    assert torch.equal(res[True][2], res[False][2])
    assert torch.equal(res[True][0].view(torch.int32), res[False][0].view(torch.int32))
    assert torch.equal(res[True][1].view(torch.int32), res[False][1].view(torch.int32))
    assert relerr(res[True][3], res[False][3]) < 1e-6 and relerr(res[True][4], res[False][4]) < 1e-6


@pytest.mark.parametrize("B,N,Mc", [(2, 256, 64), (2, 2048, 128)])
def test_encoder_linear_reassociation_equals_as_written(lib_built, B, N, Mc):
    """VN_PointNet with second_conv[0] folded onto first_conv[1]'s input channels ((W2 W1) a, ops.SW.ENC_LINEAR_REASSOCIATE: the 512-channel
    map is never stored) — on per-point scalars (ops.SW.ENC_DIR_LAYER, vnp_vn_dir_layer: every vector behind the single-channel first layer
    is parallel to its point's direction) and in the rows layout — against the evaluation order of models/pcn.py:168-173 (W2 (W1 a),
    VNP_ENC_AS_WRITTEN=1): outputs equal to fp32 re-association noise (1e-5 of the largest element), every parameter gradient equal to
    1e-4 of its largest element (1e-3 in the coarse head), running statistics to 1e-6.

    Behind the single-channel first layer a channel's score in maxpool1 is a smooth function of |x| alone, so the two points next to its
    maximum tie to rounding (differences O(d|x|^2)) and ~0.4 % of the picks depend on the evaluation order — as they do against the
    reference (tests/test_gpu_fullres_parity.py: 99.5 % identical picks); one moved pooled vector changes the B = 2 batch statistics of
    every later layer.  Each order therefore runs twice: FREE (its own picks: at most 1 % of maxpool1's may differ from the as-written
    run's, everything finite) and with the as-written run's picks FORCED (ops.hooks(pool_idx_force=...)), and on the forced run every output,
    gradient and running statistic is asserted — on every cloud, not only on a tie-free one."""
    from types import SimpleNamespace
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.pcn import VN_PointNet
    torch.manual_seed(5)
    enc = VN_PointNet(SimpleNamespace(num_coarse=Mc), latent_dim=256).cuda().train()
    sd0 = {k: v.clone() for k, v in enc.state_dict().items()}
    g = torch.Generator().manual_seed(B * N)
    # (no point at the origin here: its score in maxpool1 is exactly 0 and ties, to rounding, with channels whose other scores are all
    # tiny — the arg-max would then depend on the evaluation order; the origin is covered by test_dir_layer_equals_rows_layer)
    xyz = (torch.rand(B, N, 3, generator=g) - 0.5).cuda()
    cot_c, cot_g = torch.randn(B, Mc, 3, generator=g).cuda(), torch.randn(B, 512, 3, 1, generator=g).cuda()

    def run(fold, dirl, force=None):
        enc.load_state_dict(sd0)
        enc.zero_grad(set_to_none=True)
        tr = {}
        left = [i.clone() for i in force] if force is not None else None
        with ops.switches(ENC_LINEAR_REASSOCIATE=fold, ENC_DIR_LAYER=dirl), ops.hooks(trace=tr, pool_idx_force=left):
            coarse, fg = enc(xyz)
            ((coarse * cot_c).sum() + (fg * cot_g).sum()).backward()
        assert not left, "a forced index tensor was not consumed"
        assert (tr.get("linear_folded_onto_producer", 0) == 1) == fold and (tr.get("dir_layer", 0) == 1) == dirl, tr
        return (coarse.detach(), fg.detach(), [i.clone() for i in tr["pool_idx"]],
                {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None},
                {k: v.clone() for k, v in enc.state_dict().items() if "running" in k})

    ref = run(False, False)
    assert len(ref[2]) == 2
    for mode, (fold, dirl) in {"dir": (True, True), "rows": (True, False)}.items():
        free = run(fold, dirl)
        flips1 = int((free[2][0] != ref[2][0]).sum())
        assert flips1 <= max(2, ref[2][0].numel() // 100), (mode, flips1)
        assert all(torch.isfinite(t_).all() for t_ in free[:2])
        r = run(fold, dirl, force=ref[2])
        assert all(torch.equal(a, b) for a, b in zip(r[2], ref[2]))
        eo = (relerr(r[0], ref[0]), relerr(r[1], ref[1]))
        assert r[3].keys() == ref[3].keys()
        eg = sorted(((relerr(gr, ref[3][k]), k) for k, gr in r[3].items()), reverse=True)
        print(f"{mode}: {flips1} free maxpool1 picks differ; forced: coarse {eo[0]:.1e}, global feature {eo[1]:.1e}; worst gradients: "
              + ", ".join(f"{k} {e:.1e}" for e, k in eg[:3]))
        assert max(eo) < 1e-5, (mode, eo)           # fp32 re-association noise behind two BatchNorm layers over B*N norms
        for e, k in eg:
            # 1e-4, and 1e-3 for the coarse head's mlp (96 rows): there a leaky projection whose <q, d> is within rounding of zero
            # switches branch between two evaluation orders, and its direction weight's gradient moves by that one term
            assert e < (1e-3 if k.startswith("mlp.") else 1e-4) * fp32_mode_slack(), (mode, k, e)
        for k, v in r[4].items():
            assert relerr(v, ref[4][k]) < 1e-6, (mode, k)


@pytest.mark.parametrize("G,gs,C,K", [(2, 2048, 512, 128), (3, 100, 40, 24)])
def test_dir_pool_equals_rows_pool(lib_built, G, gs, C, K):
    """DirLinearMaxPoolFn (VNLinear -> VNMaxPool on per-point scalars: scores y y') against VNLinearMaxPoolFn on the expanded rows
    u (x) h: same picks (generic h: no rounding-level ties), pooled vectors, dL/dh and dL/dW; the origin (u = 0) scores 0 in both."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(G + gs + C)
    P = G * gs
    xyz = torch.randn(P, 3, generator=g)
    xyz[3] = 0.0
    u = ops.unit_dirs(xyz.cuda())
    h = torch.randn(P, K, generator=g).cuda()
    W = (torch.randn(C, K, generator=g) / K ** 0.5).cuda()
    Wd = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    cot = torch.randn(G * 3, C, generator=g).cuda()
    hr, Wr = h.clone().requires_grad_(True), W.clone().requires_grad_(True)
    out, idx = ops.DirLinearMaxPoolFn.apply(hr, u, Wr, Wd, G, gs)
    (out * cot).sum().backward()
    hr2, Wr2 = h.clone().requires_grad_(True), W.clone().requires_grad_(True)
    out2, idx2 = ops.VNLinearMaxPoolFn.apply(ops.DirExpandFn.apply(hr2, u), Wr2, Wd, G, gs)
    (out2 * cot).sum().backward()
    assert torch.equal(idx, idx2)
    assert relerr(out, out2) < 2e-6 and relerr(hr.grad, hr2.grad) < 1e-5 and relerr(Wr.grad, Wr2.grad) < 1e-5


def test_first_layer_on_scalars_equals_rows(lib_built):
    """VN_PointNet.first_conv[0] (ONE input channel: the point) on the per-point scalar |x| with scalar output (forward_dir,
    scalar_out) against the rows layout: h = sum_t u_t a1_t to rounding, and the same weight / BatchNorm gradients and running
    statistics for a random cotangent on the rows output."""
    from types import SimpleNamespace
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.pcn import VN_PointNet
    torch.manual_seed(7)
    l1 = VN_PointNet(SimpleNamespace(num_coarse=64), latent_dim=64).first_conv[0].cuda().train()
    sd0 = {k: v.clone() for k, v in l1.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(2, 1024, 3, generator=g) - 0.5
    xyz[1, 7] = 0.0
    xyz = xyz.cuda()
    P = 2048
    cot = torch.randn(P * 3, 128, generator=g).cuda()
    u, r = ops.unit_dirs(xyz, return_norm=True)
    assert relerr(u * r, xyz.reshape(-1, 3)) < 1e-6
    a1 = l1.forward_rows(xyz.reshape(P * 3, 1).contiguous())
    (a1 * cot).sum().backward()
    g_rows = {k: p.grad.clone() for k, p in l1.named_parameters()}
    st_rows = {k: v.clone() for k, v in l1.state_dict().items() if "running" in k}
    l1.load_state_dict(sd0); l1.zero_grad(set_to_none=True)
    h = l1.forward_dir(r, u, scalar_out=True)
    assert h.shape == (P, 128)
    assert relerr(ops.DirExpandFn.apply(h, u), a1) < 2e-6 and relerr(h, ops.DirContractFn.apply(a1.detach(), u)) < 2e-6
    (ops.DirExpandFn.apply(h, u) * cot).sum().backward()
    for k, p in l1.named_parameters():
        # the weights of this layer are the two worst-conditioned tensors of the network (tests/test_gpu_fullres_parity.py): 1e-3
        assert relerr(p.grad, g_rows[k]) < (1e-3 if "map_to" in k else 1e-4), (k, relerr(p.grad, g_rows[k]))
    for k, v in l1.state_dict().items():
        if "running" in k:
            assert relerr(v, st_rows[k]) < 1e-6, k


@pytest.mark.parametrize("P,K,C,ppg", [(512, 32, 64, 256), (4096, 128, 192, 2048), (300, 16, 40, 0)])
def test_dir_layer_equals_rows_layer(lib_built, P, K, C, ppg):
    """vnp_vn_dir_layer / VNDirLinearLeakyFn on per-point scalars h against VNLinearLeakyFn on the expanded rows u (x) h: output, input
    gradient (contracted with u), weight / BatchNorm / bias gradients and running statistics; train and eval mode."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(P + K)
    xyz = torch.randn(P, 3, generator=g)
    xyz[5] = 0.0
    u = ops.unit_dirs(xyz.cuda())
    assert torch.allclose(u[5], torch.zeros(3, device="cuda")) and relerr(u.norm(dim=1)[6:], torch.ones(P - 6)) < 1e-6
    # feature scalars v_p = h Wf^T bounded away from zero (same-signed terms, a random sign per channel): where v_p is the tiny
    # remainder of cancelling terms, the rows path's p = (u_t h) Wf^T leaves the direction u by its rounding error / |p| and its
    # gradient — of order 1 / |p| perpendicular to p — leaks into the component along u, while here p is parallel to u by
    # construction and the contraction with u removes that part exactly (measured on such a sample: |p| = 1.8e-5, out differs by 1 %
    # between the two paths and the rows path's input gradient is dominated by that one element)
    h = (torch.rand(P, K, generator=g) + 0.5).cuda()
    Wf = ((torch.rand(C, K, generator=g) + 0.1) / K * (torch.randint(0, 2, (C, 1), generator=g) * 2 - 1)).cuda()
    Wd = (torch.randn(C, K, generator=g) / K ** 0.5).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    G = P // ppg if ppg else 0
    bias = torch.randn(G * 3, 2 * C, generator=g).cuda() if ppg else None
    cot = torch.randn(P * 3, C, generator=g).cuda()
    if not ops.VNDirLinearLeakyFn.supported(P, C, ppg, bias is not None):
        pytest.skip("chunk does not divide the group")
    for training in (True, False):
        outs = {}
        for name in ("dir", "rows"):
            hr, Wfr, Wdr, gr, br = (t.clone().requires_grad_(True) for t in (h, Wf, Wd, gamma, beta))
            bir = bias.clone().requires_grad_(True) if bias is not None else None
            rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda") * 0.7
            if name == "dir":
                out = ops.VNDirLinearLeakyFn.apply(hr, u, Wfr, Wdr, gr, br, rm, rv, bir, ppg, training, 0.2)
            else:
                rows = (u.unsqueeze(2) * hr.unsqueeze(1)).reshape(P * 3, K)                 # test-side expansion (torch glue)
                out = ops.VNLinearLeakyFn.apply(rows, Wfr, Wdr, gr, br, rm, rv, bir, ppg * 3, training, 0.2)
            (out * cot).sum().backward()
            outs[name] = [out.detach(), hr.grad, Wfr.grad, Wdr.grad, gr.grad, br.grad, rm, rv] + ([bir.grad] if bias is not None else [])
        names = ["out", "gh", "gWf", "gWd", "ggamma", "gbeta", "running_mean", "running_var", "gbias"]
        # relative L2 error of each tensor
        def l2(a, b):
            a, b = a.double().cpu(), b.double().cpu()
            return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        errs = {nm: l2(a, b) for nm, a, b in zip(names, outs["dir"], outs["rows"])}
        print(f"training={training}: " + ", ".join(f"{k} {e:.1e}" for k, e in errs.items()))
        if bias is None and training:
            # scalar_out: the output as per-point scalars (the first layer of VN_PointNet) = the contraction of the rows output with u
            hr, Wfr, Wdr = (t_.clone().requires_grad_(True) for t_ in (h, Wf, Wd))
            rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda") * 0.7
            so = ops.VNDirLinearLeakyFn.apply(hr, u, Wfr, Wdr, gamma, beta, rm, rv, None, 0, True, 0.2, False, True)
            want = (outs["dir"][0].reshape(P, 3, C) * u.unsqueeze(2)).sum(1)
            assert so.shape == (P, C) and l2(so, want) < 2e-6
            (so * (cot.reshape(P, 3, C) * u.unsqueeze(2)).sum(1)).sum().backward()          # dL/dout_t = u_t g  with  g = u . cot
            back = (u.unsqueeze(2) * (cot.reshape(P, 3, C) * u.unsqueeze(2)).sum(1).unsqueeze(1)).reshape(P * 3, C)
            hr2, Wfr2, Wdr2 = (t_.clone().requires_grad_(True) for t_ in (h, Wf, Wd))
            rm2, rv2 = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda") * 0.7
            ro = ops.VNDirLinearLeakyFn.apply(hr2, u, Wfr2, Wdr2, gamma, beta, rm2, rv2, None, 0, True, 0.2)
            (ro * back).sum().backward()
            assert l2(hr.grad, hr2.grad) < 2e-5 and l2(Wfr.grad, Wfr2.grad) < 2e-5 and l2(Wdr.grad, Wdr2.grad) < 2e-5
        for nm, e in errs.items():
            # gradients: 3e-4 — with same-signed features the norms of a channel vary little over the batch, and the BatchNorm backward's
            # cancelling sums amplify the two paths' different rounding (measured 9e-5 on the input gradient of the third case)
            assert e < (2e-5 if nm in ("out", "running_mean", "running_var") else 3e-4), (training, nm, e)


@pytest.mark.parametrize("nb,n,W", [(3, 1248, 40), (2, 2048, 64), (2, 1031, 16), (1, 7, 5), (4, 4992, 24)])
def test_reduce_rows_any_count(lib_built, nb, n, W):
    """Partial-sum reduction for counts that are not multiples of 256 (config 5 folds 1248 coarse points) and for primes."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(n)
    part = torch.randn(nb, n, W, generator=g).cuda()
    out = ops.reduce_rows(part, nb, n, W)
    assert relerr(out, part.double().sum(1)) < 1e-6


@pytest.mark.parametrize("R,K,C,rpg", [(3072, 64, 64, 0), (6144, 96, 128, 3072), (9216, 32, 192, 1536)])
def test_gemm_epilogue_stats_equal_separate_pass(lib_built, R, K, C, rpg):
    """vnp_gemm_vn_stats: [p | d] and the per-1024-point Welford partials of |p| from the GEMM's accumulators, against vnp_gemm
    followed by vnp_vn_norm_stats: identical [p | d] bits (same GEMM arithmetic, rows only permuted inside the tile) and
    statistics equal to fp32 rounding; through VNLinearLeakyFn the layer output and all gradients agree."""
    require_split_gemm()
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd._lib import call, ptr
    g = torch.Generator().manual_seed(R + C)
    x = torch.randn(R, K, generator=g).cuda()
    W = (torch.randn(2 * C, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(R // rpg * 3, 2 * C, generator=g).cuda() if rpg else None
    pd_ref = torch.empty(R, 2 * C, device="cuda")
    ops.gemm(x, W, out=pd_ref, bias=bias, rows_per_group=rpg, nsplit=1)
    part_ref = torch.empty(R // 3072, C, 2, device="cuda")
    call("vnp_vn_norm_stats", ptr(pd_ref), 2 * C, R // 3, C, ptr(part_ref), 0)
    pd = torch.empty(R, 2 * C, device="cuda")
    part = ops.gemm_pd_stats(x, W, pd, bias, rpg, C, True)
    assert part is not None and torch.equal(pd, pd_ref)
    assert (part[..., 0] - part_ref[..., 0]).abs().max().item() <= 2e-6 * part_ref[..., 0].abs().max().item()
    assert (part[..., 1] - part_ref[..., 1]).abs().max().item() <= 2e-5 * part_ref[..., 1].abs().max().item()
    # whole layer, fused statistics on / off
    res = []
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    go = torch.randn(R, C, generator=g).cuda()
    for on in (True, False):
        with ops.switches(STATS_IN_EPILOGUE=on):
            xx = x.clone().requires_grad_(True)
            Wf, Wd = W[:C].clone().requires_grad_(True), W[C:].clone().requires_grad_(True)
            rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
            out = ops.VNLinearLeakyFn.apply(xx, Wf, Wd, gam, bet, rm, rv, bias, rpg, True, 0.2)
            (out * go).sum().backward()
            res.append((out.detach(), xx.grad, Wf.grad, Wd.grad, rm, rv))
    for a, b in zip(*res):
        assert (a - b).abs().max().item() <= 2e-5 * (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize("B,Mc,S,C,batched,training", [(2, 64, 16, 256, False, True),      # one 1024-point chunk per sample
                                                       (3, 96, 16, 128, True, True),       # Mc*S not a chunk multiple: 48 coarse points per chunk
                                                       (2, 78, 16, 64, False, True),       # config-5-like Mc = 2*3*13: 39 per chunk
                                                       (2, 40, 4, 72, True, False)])       # eval mode, C not a multiple of 64
def test_fold_layer0_fused_equals_unfused(lib_built, B, Mc, S, C, batched, training):
    """FoldingNet layer 0 with the (p | d) pre-activations recomputed inside each pass (ops.FoldBnLeakyFn / vnp_fold_fused) against the
    materialising pair FoldFn + VNBnLeakyFn, for chunkings where a sample is not a whole number of 1024-point chunks: forward bit-identical
    (same arithmetic per element; statistics merged from different chunk sizes agree to rounding), every gradient to 2e-5."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + Mc)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    bias0, ws0, wc0 = rn(B * 3, 2 * C), rn(2 * C), rn(2 * C)
    seed0, coarse0 = rn(B if batched else 1, S, 3) * 0.3, rn(B, Mc, 3) * 0.5
    gamma0, beta0 = rn(C).abs() + 0.5, rn(C) * 0.1
    go = rn(B * Mc * S * 3, C)
    res = {}
    for fused in (True, False):
        leaves = [x.clone().requires_grad_(True) for x in (bias0, ws0, wc0, coarse0, gamma0, beta0)]
        bias, ws, wc, coarse, gamma, beta = leaves
        rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
        if fused:
            assert ops.FoldBnLeakyFn.coarse_per_chunk(Mc, S) * S <= 1024 and Mc % ops.FoldBnLeakyFn.coarse_per_chunk(Mc, S) == 0
            out = ops.FoldBnLeakyFn.apply(bias, ws, wc, seed0, coarse, gamma, beta, rm, rv, training, 0.2)
        else:
            pd = ops.FoldFn.apply(bias, ws, wc, seed0, coarse)
            out = ops.VNBnLeakyFn.apply(pd, True, gamma, beta, rm, rv, training, 0.2)
        out.backward(go)
        res[fused] = [out.detach(), rm, rv] + [x.grad for x in leaves]
    names = ["out", "running_mean", "running_var", "g_bias", "g_wseed", "g_wcoarse", "g_coarse", "g_gamma", "g_beta"]
    for n, a, b in zip(names, res[True], res[False]):
        if not training and n in ("g_gamma", "g_beta"):
            continue
        assert relerr(a, b) < (2e-6 if n in ("out", "running_mean", "running_var") else 3e-5), (n, relerr(a, b))


@pytest.mark.parametrize("nb,npts,W,weighted", [(300, 16, 96, False), (64, 5, 40, True), (3, 2000, 72, False), (100, 128, 64, True), (70, 129, 64, False)])
def test_colsum_short_and_long_groups(lib_built, nb, npts, W, weighted):
    """vnp_colsum: per-group column sums of [nb][npts][W] (optionally weighted per row) — the wave-per-group kernel for short groups
    (<= 128 rows, >= 64 groups) and the chunked one otherwise — against torch in float64."""
    from vn_pointcloudcompletion_amd import _lib
    from vn_pointcloudcompletion_amd._lib import call, ptr
    g = torch.Generator().manual_seed(nb + npts)
    x = torch.randn(nb, npts, W, generator=g).cuda()
    w = torch.randn(nb, npts, generator=g).cuda() if weighted else None
    nch = (npts + _lib.chunk_points() - 1) // _lib.chunk_points()
    part = torch.zeros(nb, nch, W, device="cuda")
    call("vnp_colsum", ptr(x), npts * W, W, nb, npts, W, ptr(w) if weighted else None, ptr(part))
    ref = (x.double() * (w.double()[..., None] if weighted else 1.0)).sum(1)
    assert relerr(part.sum(1), ref) < 2e-6


def test_split_arithmetic_edge_cases_through_a_layer(lib_built):
    """The three documented edge cases of the default GEMM arithmetic (fp32 operands split exactly into three bf16 terms)
    at NETWORK level, through one VNLinearLeakyReLU (GEMM -> BatchNorm over norms -> leaky projection), against the same layer in
    VNP_GEMM_MODE=fp32 (fp32-input MFMA) and against float64:
      * an Inf operand element: the split forms Inf - Inf in its remainder planes and yields NaN where an fp32 product yields Inf; either
        way the output rows that touch the element are non-finite in BOTH modes and every other row agrees — the failure is loud and local;
      * fp32 DENORMAL operand elements lose their low planes (bf16 has no more exponent range than fp32): their contribution is below
        2^-126 * |w| and the outputs agree with float64 to fp32 rounding;
      * finite elements above 3.39e38 (the largest bf16) round to Inf in the first plane: as Inf in the split mode, whatever fp32 gives."""
    from vn_pointcloudcompletion_amd import _lib, ops
    from vn_pointcloudcompletion_amd.vn_layers import VNLinearLeakyReLU
    g = torch.Generator().manual_seed(31)
    P, Cin, Cout = 512, 64, 96
    x0 = torch.randn(P * 3, Cin, generator=g)
    torch.manual_seed(7)
    layer = VNLinearLeakyReLU(Cin, Cout, dim=4).cuda().eval()           # eval: BatchNorm on running statistics, rows independent
    with torch.no_grad():
        layer.batchnorm.bn.running_mean.uniform_(0.5, 1.5); layer.batchnorm.bn.running_var.uniform_(0.5, 1.5)

    def run(x, mode):
        keep = _lib.get_gemm_mode()
        _lib.set_gemm_mode(mode)
        try:
            with torch.no_grad():
                return layer.forward_rows(x.cuda()).cpu()
        finally:
            _lib.set_gemm_mode(keep)

    # denormals: a whole channel of them, and a few scattered
    xd = x0.clone()
    xd[:, 3] = 1e-40
    xd[5, 7] = -3e-42
    ys, yf = run(xd, "split"), run(xd, "fp32")
    assert torch.isfinite(ys).all() and torch.isfinite(yf).all()
    xz = xd.clone(); xz[:, 3] = 0.0; xz[5, 7] = 0.0                    # what the low planes' loss amounts to at most: those elements as zeros
    yz = run(xz, "split")
    assert relerr(ys, yf) < 2e-5 and relerr(ys, yz) < 1e-6
    # Inf and an above-bf16-range element: rows 30 .. 32 (point 10) and 60 .. 62 (point 20)
    for bad in (float("inf"), 3.4e38):
        xi = x0.clone()
        xi[31, 5] = bad
        xi[61, 9] = -bad
        ys, yf = run(xi, "split"), run(xi, "fp32")
        hit = torch.zeros(P * 3, dtype=torch.bool)
        hit[30:33] = True; hit[60:63] = True                            # the leaky projection mixes the three rows of a point
        assert not torch.isfinite(ys[hit]).all(), bad                  # the split mode: NaN on the rows of the two points
        if bad == float("inf"):
            assert not torch.isfinite(yf[hit]).all()                   # (3.4e38 is a finite fp32: the fp32-input mode may stay finite there)
        assert torch.isfinite(ys[~hit]).all() and torch.isfinite(yf[~hit]).all()
        assert relerr(ys[~hit], yf[~hit]) < 2e-5


def test_pack_pd_cols_and_leaky_dir_equal_their_unfused_forms(lib_built):
    """ops.PackPdColsFn (column ranges of a layer's two weights as [2C, n] matrices / adjacent halves) against slicing and concatenating
    with autograd, values and both weight gradients; ops.cat_rows returns a VIEW for the halves; and ops.VNLeakyDirFn (direction map +
    leaky projection, the two gradient paths of x joined in the epilogue of a product) against VNLinearFn -> VNLeakyFn: the same bits."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(5)
    C, K = 24, 2050
    Wf0, Wd0 = torch.randn(C, K, generator=g).cuda(), torch.randn(C, K, generator=g).cuda()
    cots = [torch.randn(2 * C, 2048, generator=g).cuda(), torch.randn(2 * C, 1, generator=g).cuda(), torch.randn(C, 1, generator=g).cuda(),
            torch.randn(C, 1, generator=g).cuda()]
    Wf, Wd = Wf0.clone().requires_grad_(True), Wd0.clone().requires_grad_(True)
    Wg, ws, hf, hd = ops.pack_pd_cols(Wf, Wd, (2048, 1, 1), (False, False, True))
    assert ops.cat_rows(hf, hd).data_ptr() == hf.data_ptr() and ops.cat_rows(hf, hd).shape == (2 * C, 1)
    torch.autograd.backward([Wg, ws, hf, hd], cots)
    Wf2, Wd2 = Wf0.clone().requires_grad_(True), Wd0.clone().requires_grad_(True)
    ref = [torch.cat([Wf2[:, :2048], Wd2[:, :2048]]), torch.cat([Wf2[:, 2048:2049], Wd2[:, 2048:2049]]), Wf2[:, 2049:], Wd2[:, 2049:]]
    torch.autograd.backward(ref, cots)
    for a, b in zip((Wg, ws, hf, hd), ref):
        assert torch.equal(a, b)
    assert torch.equal(Wf.grad, Wf2.grad) and torch.equal(Wd.grad, Wd2.grad)

    for R, Cc in ((96, 256), (3 * 700, 64)):
        x0, W0, go = torch.randn(R, Cc, generator=g).cuda(), torch.randn(Cc, Cc, generator=g).cuda() * 0.1, torch.randn(R, Cc, generator=g).cuda()
        x, W = x0.clone().requires_grad_(True), W0.clone().requires_grad_(True)
        y = ops.VNLeakyDirFn.apply(x, W, 0.2)
        y.backward(go)
        x2, W2 = x0.clone().requires_grad_(True), W0.clone().requires_grad_(True)
        y2 = ops.VNLeakyFn.apply(x2, ops.VNLinearFn.apply(x2, W2, None, 0), 0.2)
        y2.backward(go)
        assert torch.equal(y, y2) and torch.equal(W.grad, W2.grad)
        assert torch.equal(x.grad, x2.grad)


@pytest.mark.parametrize("G,C,gs", [(32, 2048, 2048), (5, 1000, 77), (3, 4096, 16384), (2, 1, 9), (4, 70, 3)])
def test_sort_groups_is_the_stable_sort(lib_built, G, C, gs):
    """vnp_sort_groups (every group's arg-max indices ascending, with the channels they came from, ties in channel order: what the sparse
    pool backward walks) against torch.sort(stable=True) per row, many ties included."""
    from vn_pointcloudcompletion_amd.ops.pool import sort_groups
    g = torch.Generator().manual_seed(G + C)
    idx = torch.randint(0, gs, (G, C), generator=g, dtype=torch.int32).cuda()
    pts, order = sort_groups(idx, gs)
    rp, ro = torch.sort(idx, dim=1, stable=True)
    assert pts.dtype == torch.int32 and order.dtype == torch.int32
    assert torch.equal(pts, rp.int()) and torch.equal(order, ro.int())
