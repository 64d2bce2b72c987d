"""Tensor-wise GRADIENT parity of BASELINE configs 4 and 5 against the teacher-forced oracle in fp32 and fp64 — the pattern
tests/test_gpu_fullres_parity.py uses for configs 2 and 3, applied to the two sub-networks those tests do not reach:

* config 4's decoder, Attention_VN_FoldingNet (/root/reference/models/pcn.py:392-520, models/transformer.py:25-106), with the
  attention on the padded token-major / feature-major head layouts the benchmark shape takes;
* config 5's encoder, VN_PCTransformer with its feature-space grouper (/root/reference/models/pointr/vn_pointr.py:366-412, 575-722,
  models/pointr/utils/dgcnn_group.py:112-247), at the benchmark's resolution (2048 points -> 512 -> 128 proxies).

The oracle's forward of both is pinned stage by stage against the imported reference (tests/golden/make_golden.py, attention_small.npz /
pointr_small.npz); its backward is torch autograd through that same restatement, which is what the reference's backward is.

Both networks are discontinuous in their discrete choices (kNN tables in feature space, FPS picks, the arg-max pool): the reference's own
fp32 and fp64 runs differ by O(1) end to end for that reason alone.  So every discrete choice of the HIP run is recorded
(ops.hooks(trace=...)) and the oracle is teacher-forced with it (oracle.Teacher), after asserting that each forced choice is one the
oracle's own arithmetic could have made: a neighbour set whose k-th distance is within 1e-4 relative of the oracle's own k-th distance,
an arg-max within 2e-6 of the score range of the oracle's maximum; the share of choices that are identical outright is printed and
bounded.  With the choices pinned, every output, every parameter gradient and every input gradient is compared as a whole tensor:
relative L2 error against the fp64 oracle, bounded by 2x the fp32 oracle's own error against fp64 (the reference's arithmetic is the
yardstick), with a floor written at the assert.  No gradient of a BASELINE configuration is checked by its norm only.
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import fp32_mode_slack, load_golden, t

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    ref = ref.double()
    return ((a.detach().cpu().double() - ref).norm() / ref.norm().clamp_min(1e-30)).item()


def _oracle_sd(sd0, dtype):
    sd = {k: (v.detach().clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
    return {k: v.requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}


def _compare_gradients(named_ours, g32, g64, k_ref, floor, what):
    """named_ours: [(name, gradient tensor or None)]; g32 / g64: name -> oracle gradient (or None).  Prints the worst rows, asserts all."""
    rows = []
    for name, g in named_ours:
        if g64[name] is None:
            assert g is None or g.abs().max().item() == 0.0, f"{what}: {name} receives a gradient here but not in the reference"
            continue
        assert g is not None, f"{what}: {name} receives no gradient here but one in the reference"
        assert g.shape == g64[name].shape, (name, g.shape, g64[name].shape)
        rows.append((_rel(g, g64[name]), _rel(g32[name], g64[name]), name))
    rows.sort(reverse=True)
    for ours, ref, name in rows[:6]:
        print(f"  {what} grad {name}: rel L2 vs fp64 oracle: ours {ours:.2e}, fp32 oracle {ref:.2e}")
    bad = [(name, ours, ref) for ours, ref, name in rows if ours > max(k_ref * ref, floor)]
    assert not bad, (what, bad)
    return rows


# ------------------------------------------------------------------------------------------------ config 4: attention decoder
def test_cfg4_attention_decoder_gradients_vs_oracle(lib_built):
    """Attention_VN_FoldingNet, B = 2, 256 centres -> 4096 points (the folding stages' GEMMs at 1024 centres cost the fp64 oracle minutes;
    every code path of the benchmark shape is kept: attention on the padded / feature-major layouts via ATTN_FM_MIN_FLOP = 0, the fused
    group-fold first layers, the fused stage tails — asserted through the trace)."""
    from oracle import vn_oracle as O
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.attention_decoder import Attention_VN_FoldingNet
    B, Mc = 2, 256
    torch.manual_seed(1)
    dec = Attention_VN_FoldingNet(SimpleNamespace(num_coarse=Mc, latent_dim=2048)).cuda().train()
    sd0 = {"decoder." + k: v.detach().cpu().clone() for k, v in dec.state_dict().items()}
    g = torch.Generator().manual_seed(21)
    coarse0 = 0.4 * (torch.rand(B, Mc, 3, generator=g) - 0.5)
    fg0 = 0.3 * torch.randn(B, 2048, 3, 1, generator=g)
    cot = torch.randn(B, Mc * 16, 3, generator=g)

    coarse, fg = coarse0.clone().cuda().requires_grad_(True), fg0.clone().cuda().requires_grad_(True)
    tr = {}
    with ops.switches(ATTN_FM_MIN_FLOP=0.0), ops.hooks(trace=tr):
        fine = dec(coarse, fg)
        fine.backward(cot.cuda())
        torch.cuda.synchronize()
    assert tr.get("group_fold_fused", 0) == 2 and tr.get("decoder_tail_fused", 0) == 2, tr
    assert not ops._attn_token_major(16, 1024, 384, 8)                         # the benchmark shape takes the layouts forced above

    def oracle(dtype):
        sd = _oracle_sd(sd0, dtype)
        c, f = coarse0.clone().to(dtype).requires_grad_(True), fg0.clone().to(dtype).requires_grad_(True)
        new_stats = {}
        fo = O.attention_vn_foldingnet_forward(sd, c, f, True, new_stats=new_stats)
        (fo * cot.to(dtype)).sum().backward()
        grads = {k: v.grad for k, v in sd.items() if v.requires_grad}
        grads["<coarse>"], grads["<feature_global>"] = c.grad, f.grad
        return SimpleNamespace(fine=fo.detach(), grads=grads, stats=new_stats)
    o32, o64 = oracle(torch.float32), oracle(torch.float64)

    e_f, ref_f = _rel(fine, o64.fine), _rel(o32.fine, o64.fine)
    print(f"cfg 4 decoder: fine rel L2 vs fp64 oracle: ours {e_f:.2e}, fp32 oracle {ref_f:.2e}")
    k = 2.0 * fp32_mode_slack()
    assert e_f <= max(k * ref_f, 1e-5), (e_f, ref_f)
    named = [("decoder." + n, p.grad) for n, p in dec.named_parameters()] + [("<coarse>", coarse.grad), ("<feature_global>", fg.grad)]
    # floor 2e-3: the level the reference's own fp32 arithmetic reaches on this decoder (six train-mode BatchNorms over few hundred
    # samples amplify rounding to ~1e-3: tests/golden/make_golden.py::main_attention)
    rows = _compare_gradients(named, o32.grads, o64.grads, k, 2e-3, "cfg 4")
    assert len(rows) >= 40                                                      # every live tensor of the decoder was compared
    sd = dec.state_dict()
    for name, v in o64.stats.items():
        ours = sd[name[len("decoder."):]].detach().cpu().double()
        assert ((ours - v.detach()).abs().max() / v.detach().abs().max().clamp_min(1e-12)).item() < 2e-5 * fp32_mode_slack(), name


def test_cfg4_fixture_gradient_heads_elementwise():
    """attention_small.npz stores the first 16 elements of every parameter gradient of the REFERENCE's fp32 run (`grad_heads`): compared
    element by element here (a sign error or a transposed weight gradient of equal norm fails), at the accuracy the reference's own fp32
    run has on this decoder (its fp32 and fp64 clouds differ by ~1e-3 relative)."""
    from vn_pointcloudcompletion_amd.attention_decoder import Attention_VN_FoldingNet
    A = load_golden("attention_small.npz")
    torch.manual_seed(int(A["seed"]))
    dec = Attention_VN_FoldingNet(SimpleNamespace(num_coarse=128, latent_dim=2048)).cuda().train()
    coarse, fg = t(A["coarse"]).requires_grad_(True), t(A["fg"]).requires_grad_(True)
    dec(coarse, fg).backward(t(A["g"]))
    worst = []
    for (name, prm), n, head in zip(dec.named_parameters(), A["grad_norms"], A["grad_heads"]):
        if n < 0:
            assert prm.grad is None, name
            continue
        m = min(16, prm.numel())
        mine = prm.grad.flatten()[:m].double().cpu().numpy()
        ref = head[:m].astype(np.float64)
        # per-element error against the size of an average gradient element (norm / sqrt(numel)) and of the head itself
        scale = max(np.abs(ref).max(), n / np.sqrt(prm.numel()))
        worst.append((np.abs(mine - ref).max() / scale, name))
    worst.sort(reverse=True)
    print("cfg 4 fixture gradient heads, worst:", ", ".join(f"{n} {e:.1e}" for e, n in worst[:4]))
    # measured 3.6e-2 on the worst element (transformer.1.conv3.map_to_feat.weight), 1.5e-2 .. 1.7e-2 on the next three: single elements
    # of the reference's FP32 run, which itself sits ~1e-3 from its fp64 run at the output and more in the gradients; the tight
    # tensor-wise bound is the fp64 comparison above
    tol = 6e-2 * fp32_mode_slack()
    assert worst[0][0] < tol, worst[:4]
    gtol = 1e-2 * fp32_mode_slack()
    assert _rel(coarse.grad, torch.as_tensor(A["gcoarse"])) < gtol and _rel(fg.grad, torch.as_tensor(A["gfg"])) < gtol


# ------------------------------------------------------------------------------------------------ config 5: VN-PoinTr encoder
def test_cfg5_vn_pointr_encoder_gradients_stage_by_stage(lib_built):
    """VN_PCTransformer as PCNNet builds it, B = 2, 2048 points -> 512 -> 128 proxies.

    MEASURED (scripts/experiments/debug_cfg5.py, MI355X): with EVERY discrete choice teacher-forced (all five kNN tables — the three
    feature-space ones included — and the FPS picks came out 100 % identical to the fp64 oracle's own) the encoder still amplifies
    rounding ~10x per transformer block in the reference's own arithmetic: the fp32 oracle is 3e-6 from the fp64 oracle after the
    input projection, 3e-4 after block 0, 3e-2 after block 2, 0.56 after block 5 and 0.97 in front of the pool (the HIP path: 1.2x those
    figures at every stage).  An end-to-end gradient comparison would therefore compare noise with noise.  So the chain is cut where the
    reference's own error is still small, each stage gets the fp64 oracle's stage INPUT (cast to fp32) on both sides, a random
    cotangent on its output, and its output, parameter gradients and input gradients are compared tensor-wise:
    grouper | input projection + position embedding | each of the six VN_Blocks (block 0 with its k = 8 graph) | vn_increase_dim ->
    global pool (indices teacher-forced after the score check) -> coarse head.  Bound per tensor: relative L2 error against the fp64
    oracle <= 2x the fp32 oracle's own, floor 1e-4 (gradients) / 1e-5 (outputs).  Measured on MI355X: every one of the 146 gradient tensors
    between 2e-6 and 3.5e-5, at or below the fp32 oracle's own error on the same tensor (worst: grouper conv1 BatchNorm bias 3.5e-5 vs 5.5e-5).
    kNN / FPS conventions are the oracle's (third-party sources absent: PARITY UNPINNED for tie order)."""
    from oracle import vn_oracle as O
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.transformer import rows_to_tokens, tokens_to_rows
    from vn_pointcloudcompletion_amd.vn_layers import to_rows
    from vn_pointcloudcompletion_amd.vn_pointr import VN_PCTransformer
    B, N, NC = 2, 2048, 128
    torch.manual_seed(2)
    net = VN_PCTransformer(in_chans=3, embed_dim=384, depth=[6, 8], drop_rate=0., num_query=224, knn_layer=1, dgcnn="vn_dgcnn",
                           trans="vn_trans", memory_profile=False, only_coarse=True).cuda().train()
    sd0 = {"encoder." + k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(22)
    x0 = torch.rand(B, N, 3, generator=g) - 0.5
    k = 2.0 * fp32_mode_slack()

    # ---- the HIP run's discrete choices, and the fp64 oracle chain forced with them: the stage inputs of everything below
    tr = {}
    with torch.no_grad(), ops.hooks(trace=tr):
        coarse_e2e, gf_e2e = net(x0.cuda())
    knn, fps = [i.cpu() for i in tr["knn_idx"]], [i.cpu() for i in tr["fps_idx"]]
    assert [tuple(i.shape) for i in knn] == [(B, 2048, 16), (B, 512, 16), (B, 512, 16), (B, 128, 16), (B, 128, 8)], [i.shape for i in knn]
    assert [tuple(i.shape) for i in fps] == [(B, 512), (B, 128), (B, 224)]
    chain = {}
    for dtype in (torch.float32, torch.float64):
        T = O.Teacher(knn=knn, fps=fps)
        st = {}
        with torch.no_grad():
            co, go = O.vn_pctransformer_forward(_oracle_sd(sd0, dtype), x0.to(dtype), True, teacher=T, trace=st)
        assert T.exhausted()
        chain[dtype] = (st, co, go, T.report)
    st64 = chain[torch.float64][0]
    for kind, shape, same, gap in chain[torch.float64][3]:
        print(f"  cfg 5 forced {kind} {shape}: identical to the fp64 oracle's own {same:.4%}, worst relative gap {gap:.2e}")
        assert same >= 0.97 and (kind != "knn" or gap <= 1e-4), (kind, shape, same, gap)
    print("  cfg 5 end to end (informational — chaotic in the reference itself): coarse ours / fp32 oracle vs fp64 oracle "
          f"{_rel(coarse_e2e, chain[torch.float64][1]):.2e} / {_rel(chain[torch.float32][1], chain[torch.float64][1]):.2e}")

    def f32(key):
        return st64[key].float()

    def stage(name, prefix, hip, oracle, floor_out=1e-5):
        """hip() -> (outputs, input grads, module); oracle(dtype) -> (outputs, input grads, grads by name)."""
        outs, gin, module = hip()
        res = {dt: oracle(dt) for dt in (torch.float32, torch.float64)}
        o32, o64 = res[torch.float32], res[torch.float64]
        for i, (a, b32, b64) in enumerate(zip(outs, o32[0], o64[0])):
            e, ref = _rel(a, b64), _rel(b32, b64)
            print(f"cfg 5 {name}: output {i} rel L2 vs fp64 oracle: ours {e:.2e}, fp32 oracle {ref:.2e}")
            assert e <= max(k * ref, floor_out), (name, i, e, ref)
        named = [(prefix + n, p.grad) for n, p in module.named_parameters()] + [(f"<input {i}>", gi) for i, gi in enumerate(gin)]
        g32 = dict(o32[2], **{f"<input {i}>": gi for i, gi in enumerate(o32[1])})
        g64 = dict(o64[2], **{f"<input {i}>": gi for i, gi in enumerate(o64[1])})
        g64 = {n: g64.get(n) for n, _ in named}
        rows = _compare_gradients(named, g32, g64, k, 1e-4, f"cfg 5 {name}")
        module.zero_grad(set_to_none=True)
        return len(rows)

    def own_grads(sd):
        return {n: v.grad for n, v in sd.items() if v.requires_grad}

    compared = 0
    # ---- stage G: the grouper (dgcnn_group.py:214-247), all four graphs and both FPS stages forced
    cot_f = torch.randn(B, 128, 3, NC, generator=g)

    def hip_grouper():
        x = x0.clone().cuda().requires_grad_(True)
        with ops.hooks(knn_force=[i.clone() for i in knn[:4]]):
            coor, f = net.grouper.forward_rows(x)
            fr = ops.FromRowsFn.apply(f, B, NC)
            (fr * cot_f.cuda()).sum().backward()
        return (fr, coor), (x.grad,), net.grouper

    def ora_grouper(dtype):
        sd = _oracle_sd(sd0, dtype)
        xo = x0.clone().to(dtype).requires_grad_(True)
        coor, f = O.vn_dgcnn_grouper_forward(sd, "encoder.grouper", xo, True, 16, teacher=O.Teacher(knn=knn[:4], fps=fps[:2]))
        (f * cot_f.to(dtype)).sum().backward()
        return (f.detach(), coor.detach()), (xo.grad,), own_grads(sd)
    compared += stage("grouper", "encoder.grouper.", hip_grouper, ora_grouper)

    # ---- stage E: vn_input_proj and fourth_vn_pos_embed (vn_pointr.py:585-637) on the fp64 oracle's proxies
    cot_x, cot_p = torch.randn(B, NC, 384, generator=g), torch.randn(B, NC, 384, generator=g)

    class _Embed(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.vn_input_proj, self.fourth_vn_pos_embed = net.vn_input_proj, net.fourth_vn_pos_embed

    def hip_embed():
        f = f32("grouper_f").cuda().requires_grad_(True)
        coor = f32("grouper_coor").cuda().requires_grad_(True)
        x = x0.clone().cuda().requires_grad_(True)
        xr, pr = net.embed_rows(x, coor, to_rows(f)[0])
        xt, pt = rows_to_tokens(xr, B, NC), rows_to_tokens(pr, B, NC)
        ((xt * cot_x.cuda()).sum() + (pt * cot_p.cuda()).sum()).backward()
        return (xt, pt), (f.grad, coor.grad, x.grad), _Embed()

    def ora_embed(dtype):
        sd = _oracle_sd(sd0, dtype)
        f, coor = f32("grouper_f").to(dtype).requires_grad_(True), f32("grouper_coor").to(dtype).requires_grad_(True)
        xo = x0.clone().to(dtype).requires_grad_(True)
        xt, pt = O.pointr_embed(sd, "encoder", f, coor, xo, True)
        ((xt * cot_x.to(dtype)).sum() + (pt * cot_p.to(dtype)).sum()).backward()
        return (xt.detach(), pt.detach()), (f.grad, coor.grad, xo.grad), own_grads(sd)
    compared += stage("embed", "encoder.", hip_embed, ora_embed)

    # ---- stages B0 .. B5: the six VN_Blocks (vn_pointr.py:366-412), each on the fp64 oracle's input tokens (x + pos)
    nbr = (knn[4].cuda() + (torch.arange(B, device="cuda", dtype=torch.int32) * NC).view(B, 1, 1)).reshape(B * NC, 8).contiguous()
    for i in range(6):
        tin = (st64["tok_in" if i == 0 else f"blk{i - 1}"] + st64["pos"]).float()
        cot_b = torch.randn(B, NC, 384, generator=g)

        def hip_block():
            tk = tin.clone().cuda().requires_grad_(True)
            out = rows_to_tokens(net.encoder[i].forward_rows(tokens_to_rows(tk, 128), B, NC, nbr if i < 1 else None), B, NC)
            (out * cot_b.cuda()).sum().backward()
            return (out,), (tk.grad,), net.encoder[i]

        def ora_block(dtype):
            sd = _oracle_sd(sd0, dtype)
            tk = tin.clone().to(dtype).requires_grad_(True)
            out = O.pointr_vn_block(sd, f"encoder.encoder.{i}", tk, knn[4].long() if i < 1 else None, True)
            (out * cot_b.to(dtype)).sum().backward()
            return (out.detach(),), (tk.grad,), own_grads(sd)
        compared += stage(f"block {i}", f"encoder.encoder.{i}.", hip_block, ora_block)

    # ---- stage H: vn_increase_dim -> vn_global_pool -> vn_coarse_pred (vn_pointr.py:655-668, 700-719) on the fp64 oracle's blk5 tokens
    cot_c, cot_g = torch.randn(B, 1024, 3, generator=g), torch.randn(B, 1024, 3, 1, generator=g)
    pool_idx = []

    class _Head(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.vn_increase_dim, self.vn_global_pool, self.vn_coarse_pred = net.vn_increase_dim, net.vn_global_pool, net.vn_coarse_pred

    def hip_head():
        tk = f32("blk5").cuda().requires_grad_(True)
        htr = {}
        with ops.hooks(trace=htr):
            c, gf = net.head_rows(tokens_to_rows(tk, 128), B, NC)
            ((c * cot_c.cuda()).sum() + (gf * cot_g.cuda()).sum()).backward()
        pool_idx.extend(i_.cpu() for i_ in htr["pool_idx"])
        return (c, gf), (tk.grad,), _Head()

    reports = []

    def ora_head(dtype):
        sd = _oracle_sd(sd0, dtype)
        tk = f32("blk5").to(dtype).requires_grad_(True)
        T = O.Teacher(pool=[pool_idx[0]])
        c, gf = O.pointr_head(sd, "encoder", tk, True, teacher=T)
        ((c * cot_c.to(dtype)).sum() + (gf * cot_g.to(dtype)).sum()).backward()
        reports.append((dtype, T.report[0]))
        return (c.detach(), gf.detach()), (tk.grad,), own_grads(sd)
    compared += stage("head", "encoder.", hip_head, ora_head)
    for dtype, (kind, shape, same, gap) in reports:
        print(f"  cfg 5 head, forced pool vs the {dtype} oracle's own: identical {same:.4%}, worst (best - chosen) / range {gap:.2e}")
        if dtype == torch.float64:
            assert gap <= 2e-6 and same >= 0.98, (same, gap)

    # every live parameter of the encoder was compared in exactly one stage (the never-live set is the reference's, SURVEY.md §8e:
    # tests/test_gpu_pointr.py::test_encoder_end_to_end_structure)
    print(f"cfg 5: {compared} gradient tensors compared (parameters + stage inputs)")
    assert compared >= 100, compared
