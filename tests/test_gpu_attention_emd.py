"""BASELINE config 4 path: Earth Mover's Distance kernels vs the C oracle (oracle/emd_oracle.c — PARITY UNPINNED: the
reference has no EMD test / CPU equivalent; tolerance 2e-4 relative because the reference and the HIP
kernels use the fast __expf), VNLayerNorm / attention pieces vs float64 torch, and Attention_VN_FoldingNet vs
tests/golden/attention_small.npz (reference run; stage-wise teacher-forced at 2e-4, end to end at
max(10 x the reference's own fp32-vs-fp64 gap, 5e-3) — the chain amplifies rounding to ~1e-3 in the reference itself)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import fp32_mode_slack, load_golden, relerr, t

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,n,m", [(2, 256, 256), (1, 300, 150), (2, 128, 512), (1, 1024, 1024)])
def test_emd_vs_c_oracle(lib_built, B, n, m):
    from oracle import chamfer_oracle as CO
    from vn_pointcloudcompletion_amd.emd import EarthMoverDistance
    g = torch.Generator().manual_seed(n + m)
    a = (torch.rand(B, n, 3, generator=g) - 0.5)
    b = (torch.rand(B, m, 3, generator=g) - 0.5)
    ar, br = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    cost = EarthMoverDistance()(ar, br)
    match = CO.emd_approxmatch(a.numpy(), b.numpy())
    ref_cost = CO.emd_matchcost(a.numpy(), b.numpy(), match)
    assert relerr(cost, ref_cost) < 2e-4
    gc = torch.rand(B, generator=g) + 0.5
    (cost * gc.cuda()).sum().backward()
    g1, g2 = CO.emd_matchcost_backward(gc.numpy(), a.numpy(), b.numpy(), match)
    assert relerr(ar.grad, g1) < 5e-4 and relerr(br.grad, g2) < 5e-4


def _approxmatch(a, b):
    """match [B, m, n] straight from the C ABI (the tensor emd_cuda.approxmatch_forward returns, emd.cpp:8-10)."""
    from vn_pointcloudcompletion_amd._lib import call, ptr
    B, n, _ = a.shape
    m = b.shape[1]
    match = torch.empty((B, m, n), device="cuda", dtype=torch.float32)
    temp = torch.empty((B, (n + m) * 2), device="cuda", dtype=torch.float32)
    call("vnp_emd_approxmatch", B, n, m, ptr(a), ptr(b), ptr(match), ptr(temp))
    return match


@pytest.mark.parametrize("n,m", [(1024, 1024), (512, 1024), (1024, 256)])
def test_emd_match_invariants(lib_built, n, m):
    """EMD parity is UNPINNED (no reference fixture, nvcc absent), so beside the C restatement the transport plan is held to
    the invariants the algorithm of emd_kernel.cu:25-158 guarantees by construction, at the size train.py:148-150 uses:
      * match >= 0;
      * each xyz1 point ships multiL = max(1, m/n) units and each xyz2 point receives multiR = max(1, n/m) units — after
        the final level-0 pass (exp(0) = 1: all remaining mass is spread) both marginals are met to 1e-3;
      * relabelling the points of either cloud permutes the plan and leaves the cost unchanged up to fp32 summation order —
        which the 30 dependent passes (each subtracting what the previous one shipped from `remain`) amplify: measured 4e-4
        on plan entries (which are <= 1) at 1024 x 1024, bound 2e-3; cost 1e-3 relative."""
    from vn_pointcloudcompletion_amd.emd import EarthMoverDistance
    g = torch.Generator().manual_seed(n * 3 + m)
    a = (torch.rand(2, n, 3, generator=g) - 0.5).cuda()
    b = (torch.rand(2, m, 3, generator=g) - 0.5).cuda()
    match = _approxmatch(a, b)
    multiL = 1.0 if n >= m else float(m // n)
    multiR = float(n // m) if n >= m else 1.0
    assert match.min().item() >= 0.0 and torch.isfinite(match).all()
    assert (match.sum(1) - multiL).abs().max().item() < 1e-3            # shipped by each xyz1 point (sum over xyz2)
    assert (match.sum(2) - multiR).abs().max().item() < 1e-3            # received by each xyz2 point
    pa, pb = torch.randperm(n, generator=g).cuda(), torch.randperm(m, generator=g).cuda()
    match_p = _approxmatch(a[:, pa].contiguous(), b[:, pb].contiguous())
    assert (match_p - match[:, pb][:, :, pa]).abs().max().item() < 2e-3 * max(multiL, multiR)
    c0 = EarthMoverDistance()(a, b)
    c1 = EarthMoverDistance()(a[:, pa].contiguous(), b[:, pb].contiguous())
    assert relerr(c1, c0) < 1e-3
    # the cost is the plan's transport cost with SQUARED distances: sum_lk match[l,k] |xyz1_k - xyz2_l|^2  (emd_kernel.cu:224-225)
    d2 = (a[:, None, :, :] - b[:, :, None, :]).pow(2).sum(-1)
    assert relerr(c0, (match.double() * d2.double()).sum((1, 2))) < 1e-4


def test_emd_surface(lib_built):
    from vn_pointcloudcompletion_amd import _lib
    from vn_pointcloudcompletion_amd.emd import EarthMoverDistance
    from vn_pointcloudcompletion_amd.loss import emd_loss
    a, b = torch.rand(64, 3).cuda(), torch.rand(64, 3).cuda()
    assert EarthMoverDistance()(a, b).shape == (1,)                          # 2-D inputs are unsqueezed (emd.py:38-41)
    assert emd_loss(a[None], b[None]).dim() == 0
    assert EarthMoverDistance()(a, a).item() < 1e-4                          # identical clouds: zero transport cost
    with pytest.raises(_lib.VnpError):
        EarthMoverDistance()(torch.rand(1, 8, 3), torch.rand(1, 8, 3))       # CPU tensors: no fallback


def test_layernorm_module_vs_golden(lib_built):
    from vn_pointcloudcompletion_amd.vn_layers import VNLayerNorm
    G = load_golden("layers.npz")
    m = VNLayerNorm(G["ln_gamma"].shape[0])
    m.layer_norm.weight.data.copy_(torch.as_tensor(G["ln_gamma"]))
    m.layer_norm.bias.data.copy_(torch.as_tensor(G["ln_beta"]))
    m = m.cuda()
    x = t(G["x_d4"]).requires_grad_(True)
    y = m(x)
    assert relerr(y, G["ln_y"]) < 3e-5
    y.backward(t(G["ln_g"]))
    assert relerr(x.grad, G["ln_gx"]) < 1e-4
    assert relerr(m.layer_norm.weight.grad, G["ln_ggamma"]) < 3e-5 and relerr(m.layer_norm.bias.grad, G["ln_gbeta"]) < 3e-5


def test_batched_gemm_and_attention_core(lib_built):
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(1)
    A, Bm = torch.randn(6, 70, 40, generator=g).cuda(), torch.randn(6, 50, 40, generator=g).cuda()
    assert relerr(ops.gemm_batched(A, Bm), A.double() @ Bm.double().transpose(1, 2)) < 2e-5
    Bn = torch.randn(6, 40, 33, generator=g).cuda()
    assert relerr(ops.gemm_batched(A, Bn, op_b=1), A.double() @ Bn.double()) < 2e-5
    At = torch.randn(6, 40, 70, generator=g).cuda()
    assert relerr(ops.gemm_batched(At, Bn, op_a=1, op_b=1), At.double().transpose(1, 2) @ Bn.double()) < 2e-5
    # attention core vs float64 torch (models/transformer.py:91-104)
    B, N, C, H, scale = 2, 37, 24, 4, 0.7
    q, k, v = (torch.randn(B * N * 3, C, generator=g).cuda().requires_grad_(True) for _ in range(3))
    out = ops.VNAttentionFn.apply(q, k, v, B, N, H, scale)
    go = torch.randn(B * N * 3, C, generator=g).cuda()
    out.backward(go)

    def heads(r):
        return r.double().reshape(B, N, 3, H, C // H).permute(0, 3, 1, 4, 2).reshape(B, H, N, -1)
    qd, kd, vd = (x.detach().clone().requires_grad_(True) for x in (q, k, v))
    o = ((heads(qd) @ heads(kd).transpose(-1, -2)) * scale).softmax(-1) @ heads(vd)
    o = o.reshape(B, H, N, C // H, 3).permute(0, 2, 4, 1, 3).reshape(B * N * 3, C)
    assert relerr(out, o) < 2e-5
    o.backward(go.double())
    assert relerr(q.grad, qd.grad) < 5e-5 and relerr(k.grad, kd.grad) < 5e-5 and relerr(v.grad, vd.grad) < 5e-5


@pytest.fixture(scope="module")
def A(lib_built):
    return load_golden("attention_small.npz")


def _build(A):
    from vn_pointcloudcompletion_amd.attention_decoder import Attention_VN_FoldingNet
    torch.manual_seed(int(A["seed"]))
    dec = Attention_VN_FoldingNet(SimpleNamespace(num_coarse=128, latent_dim=2048))
    names = [k for k, _ in dec.named_parameters()]
    assert names == list(A["grad_names"])
    for (k, p), s, a in zip(dec.named_parameters(), A["weight_sum"], A["weight_abssum"]):
        assert abs(p.double().sum().item() - s) <= 1e-9 * max(1.0, a), k
    return dec.cuda().train()


def test_vn_blocks_teacher_forced(A):
    dec = _build(A)
    with torch.no_grad():
        b0 = dec.transformer[0](t(A["tok_in"]))
        b1 = dec.transformer[1](t(A["blk0_out"]))
    assert relerr(b0, A["blk0_out"]) < 2e-4
    assert relerr(b1, A["blk1_out"]) < 2e-4


def test_attention_decoder_matches_reference(A):
    dec = _build(A)
    coarse, fg = t(A["coarse"]).requires_grad_(True), t(A["fg"]).requires_grad_(True)
    fine = dec(coarse, fg)
    assert fine.shape == (2, 2048, 3)
    gap = np.abs(A["fine"].astype(np.float64) - A["fine64"]).max()
    tol = max(10 * gap, 5e-3)
    assert (fine.detach().cpu().double() - torch.as_tensor(A["fine64"])).abs().max().item() < tol
    fine.backward(t(A["g"]))
    # the decoder amplifies rounding to ~1e-3 in the reference itself (fp32 vs fp64 runs) and the gradients are of order 1e4: 5e-2 on the
    # split GEMM; the fp32-input MFMA mode sums in another order than the CPU reference and gets the documented slack (measured 5.8e-2)
    gtol = 5e-2 * (1.0 if fp32_mode_slack() == 1.0 else 2.0)
    bad = []
    for (k, prm), n in zip(dec.named_parameters(), A["grad_norms"]):
        if n < 0:
            assert prm.grad is None, k                     # final_conv, attn.qkv/proj, conv1/conv2 are never used
            continue
        if abs(prm.grad.double().norm().item() - n) > gtol * n:
            bad.append((k, prm.grad.double().norm().item(), n))
    assert not bad, bad
    assert relerr(coarse.grad, A["gcoarse"]) < gtol and relerr(fg.grad, A["gfg"]) < gtol


def test_pcnnet_config4_step(lib_built):
    """vn_pointnet + attention_vn_foldingnet + EMD coarse loss: one optimizer step runs and every used parameter moves."""
    from vn_pointcloudcompletion_amd.model import PCNNet
    from vn_pointcloudcompletion_amd.optim import FusedAdam
    from vn_pointcloudcompletion_amd.train import train_step
    torch.manual_seed(0)
    cfg = SimpleNamespace(num_coarse=64, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda",
                          coarse_loss="emd", dcd_opts={"alpha": 200, "lambda": 0.5})
    net = PCNNet(cfg, "vn_pointnet", "attention_vn_foldingnet").train()
    opt = FusedAdam(net.parameters(), lr=1e-3)
    p = torch.rand(2, 256, 3, device="cuda") - 0.5
    c = torch.rand(2, 1024, 3, device="cuda") - 0.5
    before = net.decoder.vn_folding2[2].map_to_feat.weight.detach().clone()
    loss, l1, l2 = train_step(cfg, net, opt, p, c)
    assert torch.isfinite(loss) and l1.item() > 0 and l2.item() > 0
    assert not torch.equal(before, net.decoder.vn_folding2[2].map_to_feat.weight)


@pytest.mark.parametrize("L", [64, 128, 256, 512, 1024, 2048, 192, 4096])
def test_softmax_rows_forward_backward(lib_built, L):
    """vnp_softmax_fwd / vnp_softmax_bwd (the register-resident kernels for rows of 128 ... 2048 elements, the streaming ones otherwise)
    against torch in float64: P = softmax(S * scale), dS = P * (dP - sum dP P) * scale."""
    from vn_pointcloudcompletion_amd._lib import call, ptr
    g = torch.Generator().manual_seed(L)
    R, scale = 37, 0.37
    S = (torch.randn(R, L, generator=g) * 3).cuda()
    dP = torch.randn(R, L, generator=g).cuda()
    Sd = S.double().requires_grad_(True)
    Pd = (Sd * scale).softmax(-1)
    Pd.backward(dP.double())
    P = S.clone()
    call("vnp_softmax_fwd", ptr(P), R, L, scale)
    assert (P.double() - Pd.detach()).abs().max().item() < 2e-7
    dS = dP.clone()
    call("vnp_softmax_bwd", ptr(P), ptr(dS), R, L, scale)
    assert (dS.double() - Sd.grad).abs().max().item() < 5e-7 * max(1.0, Sd.grad.abs().max().item())


@pytest.mark.parametrize("B,N,C,H", [(2, 37, 384, 8), (1, 5, 12, 4), (3, 64, 96, 2), (1, 3, 2304, 8)])
def test_head_permutes_are_the_index_permutation(lib_built, B, N, C, H):
    """vnp_rows_to_heads / vnp_heads_to_rows (LDS-staged transposes): heads[(b, h)][n][cl * 3 + t] == rows[(b, n, t)][h * ch + cl] exactly,
    for token counts that do not fill the last workgroup, one to eight tokens per workgroup, and row strides wider than C on either side."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(B + N + C + H)
    wide = torch.randn(B * N * 3, C + 8, generator=g).cuda()
    rows = wide[:, 4:4 + C]                                      # a column slice: row stride C + 8
    ch = C // H
    want = rows.view(B, N, 3, H, ch).permute(0, 3, 1, 4, 2).reshape(B * H, N, ch * 3)
    heads = ops._to_heads(rows, B, N, H)
    assert torch.equal(heads, want)
    back = torch.full((B * N * 3, C + 8), 7.0, device="cuda")
    ops._from_heads(heads, B, N, C, H, out=back[:, 2:2 + C])
    assert torch.equal(back[:, 2:2 + C], rows)
    assert (back[:, :2] == 7.0).all() and (back[:, 2 + C:] == 7.0).all()          # nothing written outside the view


@pytest.mark.parametrize("B,N,C,H", [(2, 37, 384, 8), (1, 130, 12, 4), (3, 64, 96, 2), (1, 128, 384, 6), (2, 70, 384, 4)])
def test_padded_and_feature_major_head_layouts(lib_built, B, N, C, H):
    """vnp_rows_to_heads_pad / vnp_rows_to_heads_fm / vnp_heads_fm_to_rows: the permutation of vnp_rows_to_heads into token-major slices of
    ldh >= Dh floats and into feature-major [fr >= Dh][N] panels, padding written as zeros, ragged token tiles, strided row views."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(B + N + C + H)
    wide = torch.randn(B * N * 3, 3 * C, generator=g).cuda()
    rows = wide[:, C:2 * C]                                      # the k third of a packed q | k | v matrix
    ch = C // H
    Dh, ldh, fr = ops._attn_dims(C, H)
    assert ldh % 32 == 0 and fr % 96 == 0 and ldh >= Dh and fr >= Dh
    want = rows.view(B, N, 3, H, ch).permute(0, 3, 1, 4, 2).reshape(B * H, N, Dh)
    tm = ops._heads_tm(rows, B, N, H, ldh)
    assert torch.equal(tm[:, :, :Dh], want) and (tm[:, :, Dh:] == 0).all()
    fm = ops._heads_fm(rows, B, N, H, fr)
    assert torch.equal(fm[:, :Dh, :], want.transpose(1, 2)) and (fm[:, Dh:, :] == 0).all()
    back = torch.full((B * N * 3, 3 * C), 7.0, device="cuda")
    ops._rows_from_fm(fm, B, N, C, H, out=back[:, 2 * C:])
    assert torch.equal(back[:, 2 * C:], rows) and (back[:, :2 * C] == 7.0).all()


@pytest.mark.parametrize("B,N,C,H", [(2, 128, 384, 8), (2, 256, 384, 6), (1, 37, 24, 4), (2, 128, 384, 4)])
def test_attention_layouts_agree(lib_built, B, N, C, H):
    """The attention core on padded token-major / feature-major head tensors (every product on the aligned kernels) against the
    rounds 1-2 form (head dimension on N / K, bounds-checked kernels) and against float64 torch: same values within fp32 rounding
    (the products of out, dq, dk, dv swap their operand roles, so the six-product sums meet in another order), packed and unpacked."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(N + C)
    qkv = torch.randn(B * N * 3, 3 * C, generator=g).cuda()
    go = torch.randn(B * N * 3, C, generator=g).cuda()
    scale = 0.3
    res = {}
    floor = ops.SW.ATTN_FM_MIN_FLOP
    for tm in (False, True):
        ops.SW.ATTN_TOKEN_MAJOR, ops.SW.ATTN_FM_MIN_FLOP = tm, 0.0              # no size floor: these small cases take the padded layouts too
        try:
            x = qkv.clone().requires_grad_(True)
            out = ops.VNAttentionPackedFn.apply(x, B, N, H, scale)
            out.backward(go)
            q, k, v = (qkv[:, i * C:(i + 1) * C].contiguous().requires_grad_(True) for i in range(3))
            out2 = ops.VNAttentionFn.apply(q, k, v, B, N, H, scale)
            out2.backward(go)
            res[tm] = (out.detach(), x.grad, out2.detach(), torch.cat([q.grad, k.grad, v.grad], dim=1))
        finally:
            ops.SW.ATTN_TOKEN_MAJOR, ops.SW.ATTN_FM_MIN_FLOP = False, floor
    assert not ops._attn_token_major(16, 1024, 384, 8) and ops._attn_token_major(32, 128, 384, 4)       # configs 4 and 5
    assert torch.equal(res[False][0], res[False][2]) and torch.equal(res[False][1], res[False][3])        # packed == unpacked
    assert relerr(res[False][0], res[True][0]) < 2e-6 and relerr(res[False][1], res[True][1]) < 5e-6

    def heads(r):
        return r.double().reshape(B, N, 3, H, C // H).permute(0, 3, 1, 4, 2).reshape(B, H, N, -1)
    xd = qkv.double().clone().requires_grad_(True)
    o = ((heads(xd[:, :C]) @ heads(xd[:, C:2 * C]).transpose(-1, -2)) * scale).softmax(-1) @ heads(xd[:, 2 * C:])
    o = o.reshape(B, H, N, C // H, 3).permute(0, 2, 4, 1, 3).reshape(B * N * 3, C)
    o.backward(go.double())
    assert relerr(res[False][0], o) < 2e-5 and relerr(res[False][1], xd.grad) < 5e-5


@pytest.mark.parametrize("G,S,C,needs_x", [(128, 16, 256, True), (96, 16, 64, False), (64, 4, 128, True)])
def test_group_fold_fused_equals_the_unfused_layer(lib_built, G, S, C, needs_x):
    """GroupFoldBnLeakyFn (vnp_group_fold_fused: the folding stages' first layer without its [P*3, 2C] pre-activation tensor) against the
    same VNLinearLeakyReLU evaluated by VNLinearLeakyFn (outer product + BatchNorm / leaky passes): outputs, running statistics and
    every gradient — the per-point input's (one partial per 64-column block, summed), the point channel's weight columns, the group
    features' (through the per-group bias), the group weights', gamma / beta — in training mode, and the outputs in eval mode."""
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.vn_layers import VNLinearLeakyReLU
    g = torch.Generator().manual_seed(G + S + C)
    cg = 24
    x0 = torch.randn(G * S * 3, 1, generator=g).cuda()
    f0 = torch.randn(G * 3, cg, generator=g).cuda()
    go = torch.randn(G * S * 3, C, generator=g).cuda()
    res = {}
    for fused in (True, False):
        ops.SW.GROUP_FOLD = fused
        try:
            torch.manual_seed(3)
            layer = VNLinearLeakyReLU(cg + 1, C, dim=4).cuda().train()
            with torch.no_grad():
                layer.batchnorm.bn.weight.uniform_(0.5, 1.5); layer.batchnorm.bn.bias.uniform_(-0.3, 0.3)
            x = x0.clone().requires_grad_(needs_x)
            f = f0.clone().requires_grad_(True)
            tr = {}
            with ops.hooks(trace=tr):
                y = layer.forward_rows(x, global_rows=f, rows_per_group=S * 3, global_first=False)
            assert ("group_fold_fused" in tr) == fused
            y.backward(go)
            grads = [f.grad] + ([x.grad] if needs_x else []) + [p.grad for p in layer.parameters()]
            stats = [layer.batchnorm.bn.running_mean.clone(), layer.batchnorm.bn.running_var.clone()]
            layer.eval()
            with torch.no_grad():
                ye = layer.forward_rows(x0, global_rows=f0, rows_per_group=S * 3, global_first=False)
            res[fused] = (y.detach(), grads, stats, ye)
        finally:
            ops.SW.GROUP_FOLD = True
    assert relerr(res[True][0], res[False][0]) < 2e-6 and relerr(res[True][3], res[False][3]) < 2e-6
    for a, b in zip(res[True][2], res[False][2]):
        assert relerr(a, b) < 1e-6
    assert len(res[True][1]) == len(res[False][1])
    for a, b in zip(res[True][1], res[False][1]):
        assert a.shape == b.shape and relerr(a, b) < 3e-5, (a.shape, relerr(a, b))


def test_attention_decoder_fused_stage_tails_equal_layer_by_layer(lib_built):
    """Attention_VN_FoldingNet with each folding stage's VNLinearLeakyReLU(256, 128) + VNLinear(128, 1) as one function
    (ops.VNLinearLeakyOutFn without a residual) against the layer-by-layer evaluation: outputs and every parameter gradient."""
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.attention_decoder import Attention_VN_FoldingNet
    g = torch.Generator().manual_seed(11)
    coarse = (torch.rand(2, 128, 3, generator=g) - 0.5).cuda()
    feat = torch.randn(2, 2048, 3, generator=g).cuda()
    go = torch.randn(2, 128 * 16, 3, generator=g).cuda()
    res = {}
    for fused in (True, False):
        Attention_VN_FoldingNet.FUSED_TAIL = fused
        try:
            torch.manual_seed(4)
            dec = Attention_VN_FoldingNet(SimpleNamespace(num_coarse=128, latent_dim=2048)).cuda().train()
            c = coarse.clone().requires_grad_(True)
            tr = {}
            with ops.hooks(trace=tr):
                out = dec(c, feat)
            assert tr.get("decoder_tail_fused", 0) == (2 if fused else 0) and tr.get("group_fold_fused", 0) == 2
            out.backward(go)
            res[fused] = (out.detach(), c.grad, [(n, p.grad) for n, p in dec.named_parameters() if p.grad is not None])
        finally:
            Attention_VN_FoldingNet.FUSED_TAIL = True
    assert relerr(res[True][0], res[False][0]) < 5e-6 and relerr(res[True][1], res[False][1]) < 5e-5
    assert [n for n, _ in res[True][2]] == [n for n, _ in res[False][2]]
    # the two evaluations sum the BatchNorm / leaky partials over different partitions of the points (the fused tail's passes and the
    # layer-by-layer kernels choose their chunks separately): gradients agree to the rounding the network amplifies — measured up to 2e-4
    # of a tensor's largest entry (vn_folding2.1.map_to_dir.weight) with 16-point chunks, 6e-5 with 64-point ones
    for (n, a), (_, b) in zip(res[True][2], res[False][2]):
        assert relerr(a, b) < 5e-4, (n, relerr(a, b))


@pytest.mark.parametrize("B,N,C,H", [(2, 37, 384, 4), (1, 128, 384, 4), (3, 5, 12, 2)])
def test_grouped_head_permutes_equal_three_single_ones(lib_built, B, N, C, H):
    """vnp_rows_to_heads_grouped / vnp_heads_to_rows_grouped (q | k | v side by side in the rows -> three head tensors and back, one
    launch each) against three vnp_rows_to_heads / vnp_heads_to_rows calls on the column thirds: the same floats in the same places,
    also with a row stride wider than the matrix."""
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd._lib import call, ptr
    g = torch.Generator().manual_seed(C + N)
    wide = torch.randn(B * N * 3, 3 * C + 8, generator=g).cuda()
    qkv = wide[:, :3 * C]
    heads = torch.empty((3, B * H, N, (C // H) * 3), device="cuda")
    call("vnp_rows_to_heads_grouped", ptr(qkv), qkv.stride(0), B, N, C, H, 3, ptr(heads))
    for i in range(3):
        assert torch.equal(heads[i], ops._to_heads(qkv[:, i * C:(i + 1) * C], B, N, H))
    back = torch.full_like(wide, 7.0)
    call("vnp_heads_to_rows_grouped", ptr(heads), B, N, C, H, 3, ptr(back), back.stride(0))
    assert torch.equal(back[:, :3 * C], qkv) and bool((back[:, 3 * C:] == 7.0).all())


@pytest.mark.parametrize("M,N,K,nsplit", [(12288, 128, 384, 1), (12288, 128, 384, 4), (96, 256, 2048, 16), (100, 130, 70, 1), (384, 128, 12288, 32)])
def test_gemm_with_a_residual_in_the_epilogue(lib_built, M, N, K, nsplit):
    """vnp_gemm_add (C = A B^T + R, the skip connection of a transformer block taken in the epilogue of the VNLinear that ends the
    branch, models/pointr/vn_pointr.py:401-410) against vnp_gemm followed by an addition: the same bits (one fp32 addition of the same two
    numbers either way), aligned and ragged shapes, with split-K (the slab reduction adds R), R with a wider row stride, and R = C."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(M + K)
    op_a = 1 if K > 4 * M else 0
    A = (torch.randn(K, M, generator=g) if op_a else torch.randn(M, K, generator=g)).cuda()
    Bm = (torch.randn(K, N, generator=g) if op_a else torch.randn(N, K, generator=g)).cuda()
    Rw = torch.randn(M, N + 4, generator=g).cuda()
    R = Rw[:, :N]
    from vn_pointcloudcompletion_amd import _lib
    mode0 = _lib.get_gemm_mode()
    try:
        for mode in ("split", "fp32"):                   # both product kernels carry the residual epilogue
            _lib.set_gemm_mode(mode)
            want = ops.gemm(A, Bm, op_a=op_a, op_b=op_a, nsplit=nsplit) + R
            got = ops.gemm(A, Bm, op_a=op_a, op_b=op_a, nsplit=nsplit, resid=R)
            assert torch.equal(got, want), mode
            inplace = R.contiguous()
            ops.gemm(A, Bm, op_a=op_a, op_b=op_a, nsplit=nsplit, resid=inplace, out=inplace)
            assert torch.equal(inplace, want), mode
    finally:
        _lib.set_gemm_mode(mode0)


@pytest.mark.parametrize("P,C,na", [(4096, 128, 2), (37, 384, 1), (5, 12, 0), (300, 128, 2)])
def test_add_layernorm_equals_the_adds_and_the_layernorm(lib_built, P, C, na):
    """ops.AddLayerNormFn (s = x + a1 + a2, y = VNLayerNorm(s), and in the backward the skip connection's gradient added inside the
    LayerNorm's pass) against ATen additions around ops.VNLayerNormFn: s, y and every gradient bit for bit."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(P + C)
    leaves = [torch.randn(P * 3, C, generator=g).cuda() for _ in range(1 + na)]
    gam0, bet0 = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    gs, gy = torch.randn(P * 3, C, generator=g).cuda(), torch.randn(P * 3, C, generator=g).cuda()

    def run(fused):
        xs = [v.clone().requires_grad_(True) for v in leaves]
        gam, bet = gam0.clone().requires_grad_(True), bet0.clone().requires_grad_(True)
        if fused:
            s, y = ops.AddLayerNormFn.apply(xs[0], xs[1] if na > 0 else None, xs[2] if na > 1 else None, gam, bet)
        else:
            s = xs[0]
            for v in xs[1:]:
                s = s + v
            y = ops.VNLayerNormFn.apply(s, gam, bet)
        torch.autograd.backward([s, y], [gs, gy])
        return [s.detach(), y.detach()] + [v.grad for v in xs] + [gam.grad, bet.grad]
    for a, b in zip(run(True), run(False)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("use", ["s", "y"])
@pytest.mark.parametrize("na", [0, 2])
def test_add_layernorm_with_one_output_unused(lib_built, use, na):
    """Only the skip connection s, or only the normalised y, reaches the loss: the Function sees None for the other cotangent (no
    zero-filled stand-in), absent addends get no gradient, and every gradient equals the unfused form bit for bit."""
    from vn_pointcloudcompletion_amd import ops
    P, C = 130, 128
    g = torch.Generator().manual_seed(5)
    leaves = [torch.randn(P * 3, C, generator=g).cuda() for _ in range(1 + na)]
    gam0, bet0 = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    cot = torch.randn(P * 3, C, generator=g).cuda()

    def run(fused):
        xs = [v.clone().requires_grad_(True) for v in leaves]
        gam, bet = gam0.clone().requires_grad_(True), bet0.clone().requires_grad_(True)
        if fused:
            s, y = ops.AddLayerNormFn.apply(xs[0], xs[1] if na > 0 else None, xs[2] if na > 1 else None, gam, bet)
        else:
            s = xs[0] * 1.0
            for v in xs[1:]:
                s = s + v
            y = ops.VNLayerNormFn.apply(s, gam, bet)
        ((s if use == "s" else y) * cot).sum().backward()
        return [v.grad for v in xs] + [gam.grad, bet.grad]
    for a, b in zip(run(True), run(False)):
        if use == "s" and (a is None or b is None):              # gamma / beta: no gradient (None) or an exact zero
            assert (a is None or not a.any()) and (b is None or not b.any())
        else:
            assert torch.equal(a, b)


def test_coarse_emd_loss_beside_the_decoder_is_the_same_step(lib_built):
    """train.train_step starts an EMD coarse loss on a side HIP stream the moment the encoder's output exists (PCNNet's on_coarse hook) and
    lets it run beside the decoder.  From the same state and batches: two runs taken that way agree bit for bit over three steps (a
    missing stream dependency — a loss computed on stale coordinates, a gradient read before it was written — would differ from run to
    run or be plainly wrong), and they agree with the run that computes the loss after the forward on the main stream: the three losses
    bit for bit, the flat gradient to rounding (the loss's node is created earlier in the graph, so autograd adds the coarse cloud's three
    gradient contributions in another order)."""
    import copy
    from vn_pointcloudcompletion_amd import train as T
    from vn_pointcloudcompletion_amd.model import PCNNet
    from vn_pointcloudcompletion_amd.optim import FusedAdam
    cfg = SimpleNamespace(num_coarse=1024, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda",
                          coarse_loss="emd", dcd_opts={"alpha": 200, "lambda": 0.5})
    torch.manual_seed(0)
    ref_net = PCNNet(cfg, "vn_pointnet", "attention_vn_foldingnet").train()
    g = torch.Generator().manual_seed(3)
    batches = [((torch.rand(2, 2048, 3, generator=g) - 0.5).cuda(), (torch.rand(2, 16384, 3, generator=g) - 0.5).cuda()) for _ in range(3)]
    runs = []
    old = T.COARSE_LOSS_AHEAD
    try:
        for ahead in (("emd",), ("emd",), ()):
            T.COARSE_LOSS_AHEAD = ahead
            net = copy.deepcopy(ref_net)
            opt = FusedAdam(net.parameters(), lr=1e-3)
            rec = []
            for p, c in batches[:1 if ahead == () else 3]:       # later steps start from weights that already differ by rounding
                losses = T.train_step(cfg, net, opt, p, c)
                torch.cuda.synchronize()
                rec.append((torch.stack(losses).clone(), opt.flat_buffers[0].flat_grad.clone()))
            runs.append(rec)
    finally:
        T.COARSE_LOSS_AHEAD = old
    for (la, ga), (lb, gb) in zip(runs[0], runs[1]):
        assert torch.equal(la, lb) and torch.equal(ga.view(torch.int32), gb.view(torch.int32))
    (la, ga), (lc, gc) = runs[0][0], runs[2][0]
    assert torch.equal(la, lc)
    assert relerr(ga, gc) < 1e-5, relerr(ga, gc)
