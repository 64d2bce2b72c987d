"""Full-RESOLUTION parity of the headline configuration (BASELINE config 2: vn_pointnet + vn_foldingnet, 2048 -> 1024 ->
16384) against the CPU oracle — the reference's train.py:143-167 on identical inputs and weights, at B = 2 so that the
oracle finishes in well under a minute.

Why a second end-to-end test beside test_gpu_e2e.py (config 1, 256 -> 64 -> 1024): at fixture size the layers have
R = 1536 rows, so the code paths the benchmark actually runs — BatchNorm statistics in the GEMM epilogue (needs whole
1024-point chunks), the exact screened arg-max pool fused into the GEMM epilogue, the fused FoldingNet layer 0 and decoder tail, the
pruned Chamfer search — are never taken there.  This test asserts (ops.hooks(trace=...)) that they ARE taken here.

The two arg-max pools are discontinuous: two points whose scores agree to ~1e-7 relative swap under any change of
summation order (the reference's own CPU BLAS vs einsum do that too), and the pooled VECTOR then changes
by O(1).  The oracle is therefore teacher-forced with OUR indices, after checking that every forced choice is a maximum
of the ORACLE's scores to within 2e-6 of the channel's score range (i.e. each index is one the reference's arithmetic
could have produced) and that at least 99 % of the indices are identical outright.  Everything downstream of the pools
is then compared directly; tolerances are written at each assert (north star: Chamfer-L1 within 1e-5).
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import fp32_mode_slack

pytestmark = pytest.mark.gpu

B, N, MC, MD = 2, 2048, 1024, 16384


def _cd_l1_oracle(O, CO, pred, gt):
    """metrics/loss.py:20-31 with the nearest neighbours found by oracle/chamfer_oracle.c (bit-exact restatement of
    chamfer3D.cu) and the distances re-formed differentiably from those indices."""
    _, _, i1, i2 = CO.chamfer_forward(pred.detach().float().numpy(), gt.float().numpy())
    d1, d2 = O.chamfer_from_idx(pred, gt, i1, i2)
    return O.cd_loss_l1(d1, d2)


@pytest.fixture(scope="module")
def run(lib_built):
    from oracle import chamfer_oracle as CO
    from oracle import vn_oracle as O
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.loss import cd_loss_L1
    from vn_pointcloudcompletion_amd.model import PCNNet

    torch.manual_seed(0)
    cfg = SimpleNamespace(num_coarse=MC, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda")
    net = PCNNet(cfg, "vn_pointnet", "vn_foldingnet").train()
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    p = torch.rand(B, N, 3, generator=g) - 0.5
    c = torch.rand(B, MD, 3, generator=g) - 0.5

    tr = {}
    with ops.hooks(trace=tr):
        coarse, fine = net(p.cuda())
        loss1, loss2 = cd_loss_L1(coarse, c.cuda()), cd_loss_L1(fine, c.cuda())
        (loss1 + loss2).backward()
        torch.cuda.synchronize()
    idx1, idx2 = [i.cpu().long() for i in tr["pool_idx"]]

    # ---- oracle, pools teacher-forced with our indices: in fp32 (the reference's arithmetic) and in fp64 (the yardstick that
    # tells rounding noise of the fp32 reference apart from error of the HIP path)
    def oracle(dtype):
        sd = {k: (v.detach().clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
        sd = {k: v.requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
        otr, new_stats = {}, {}
        pp, cc = p.to(dtype), c.to(dtype)
        co, go = O.vn_pointnet_forward(sd, pp, True, new_stats=new_stats, pool_idx=(idx1, idx2), trace=otr)
        fo = O.vn_foldingnet_forward(sd, co, go, None, True, new_stats=new_stats)
        with torch.no_grad():
            # decoder alone on OUR coarse and the oracle's global feature (isolates the decoder from encoder rounding)
            fo_tf = O.vn_foldingnet_forward(sd, coarse.detach().cpu().to(dtype), go.detach(), None, True)
        l1o, l2o = _cd_l1_oracle(O, CO, co, cc), _cd_l1_oracle(O, CO, fo, cc)
        (l1o + l2o).backward()
        return SimpleNamespace(otr=otr, co=co.detach(), fo=fo.detach(), fo_tf=fo_tf, loss=(l1o.item(), l2o.item()), sd=sd,
                               new_stats=new_stats)
    o32, o64 = oracle(torch.float32), oracle(torch.float64)
    return SimpleNamespace(net=net, tr=tr, idx=(idx1, idx2), coarse=coarse.detach().cpu(), fine=fine.detach().cpu(),
                           loss=(loss1.item(), loss2.item()), o32=o32, o64=o64, sd0=sd0, p=p, c=c)


def test_benchmark_code_paths_were_taken(run):
    from vn_pointcloudcompletion_amd import _lib
    if _lib.get_gemm_mode() != "split":
        pytest.skip("VNP_GEMM_MODE=fp32: the fused epilogues belong to the split kernel; the parity asserts below still run")
    tr = run.tr
    # final_conv[1]: statistics in the GEMM epilogue; first_conv[0] and second_conv[0] run on per-point scalars (vnp_vn_dir_layer)
    assert tr.get("pd_gemm_stats_epilogue", 0) == 1 and "pd_gemm_plain" not in tr, tr
    assert tr.get("dir_layer", 0) == 1 and tr.get("dir_layer_scalar_out", 0) == 1, tr
    # maxpool2 (K = 1024): screened search; maxpool1 (K = 128): on per-point scalars, its 512-channel input never stored and
    # second_conv[0] folded onto first_conv[1]'s input channels
    assert tr.get("linear_pool_screened", 0) == 1 and "linear_pool_unfused" not in tr and tr.get("dir_pool", 0) == 1, tr
    assert tr.get("linear_folded_onto_producer", 0) == 1 and "linear_pool_tap_fused" not in tr and "pool_tap" not in tr, tr
    assert tr.get("fold_layer0_fused", 0) == 1 and "fold_layer0_unfused" not in tr, tr
    assert tr.get("decoder_tail_fused", 0) == 1, tr
    assert tr.get("chamfer_pruned", 0) == 1 and tr.get("chamfer_brute", 0) == 1, tr             # dense loss / coarse loss


def test_pool_choices_are_maxima_of_the_oracle_scores(run):
    for name, idx in (("1", run.idx[0]), ("2", run.idx[1])):
        dot = run.o64.otr["dot" + name]                                    # [B, C, N] oracle scores (fp64: exact ranking)
        chosen = dot.gather(-1, idx.unsqueeze(-1)).squeeze(-1)
        best, lo = dot.max(-1)[0], dot.min(-1)[0]
        gap = ((best - chosen) / (best - lo).clamp_min(1e-20)).max().item()
        agree = (run.o32.otr["idx" + name] == idx).float().mean().item()      # vs the fp32 oracle's own arg-max
        print(f"pool{name}: identical indices {agree:.4%}, worst (best - chosen)/range {gap:.2e}")
        assert gap <= 2e-6, (name, gap)
        assert agree >= 0.99, (name, agree)


def _maxerr(a, b):
    return (a.double() - b.double()).abs().max().item()


def test_outputs_and_chamfer_l1_match_oracle(run):
    o32, o64 = run.o32, run.o64
    ec, ec64, ref_c = _maxerr(run.coarse, o32.co), _maxerr(run.coarse, o64.co), _maxerr(o32.co, o64.co)
    ef, ef64, ref_f = _maxerr(run.fine, o32.fo), _maxerr(run.fine, o64.fo), _maxerr(o32.fo, o64.fo)
    tf64, ref_tf = _maxerr(run.fine, o64.fo_tf), _maxerr(o32.fo_tf, o64.fo_tf)
    d1, d2 = abs(run.loss[0] - o32.loss[0]), abs(run.loss[1] - o32.loss[1])
    e1, e2 = abs(run.loss[0] - o64.loss[0]), abs(run.loss[1] - o64.loss[1])
    print(f"coarse: |ours-o32| {ec:.2e} |ours-o64| {ec64:.2e} (|o32-o64| {ref_c:.2e});  fine: |ours-o32| {ef:.2e} |ours-o64| {ef64:.2e} "
          f"(|o32-o64| {ref_f:.2e});  decoder alone on our coarse: |ours-o64| {tf64:.2e} (|o32-o64| {ref_tf:.2e})")
    print(f"CD-L1 coarse {run.loss[0]:.7f} vs o32 {o32.loss[0]:.7f} (diff {d1:.1e}) vs o64 {o64.loss[0]:.7f} (diff {e1:.1e}); "
          f"dense {run.loss[1]:.7f} vs o32 {o32.loss[1]:.7f} (diff {d2:.1e}) vs o64 {o64.loss[1]:.7f} (diff {e2:.1e})")
    # the north-star criterion on both heads at full resolution, against the fp32 oracle AND the exact (fp64) value
    assert d1 < 1e-5 and d2 < 1e-5 and e1 < 1e-5 and e2 < 1e-5
    # clouds: no further from the exact result than 1.5x the fp32 reference arithmetic itself is (+ 2e-6 absolute)
    k = 1.5 * fp32_mode_slack()
    assert ec64 <= k * ref_c + 2e-6, (ec64, ref_c)
    assert ef64 <= k * ref_f + 2e-6, (ef64, ref_f)
    assert tf64 <= k * ref_tf + 2e-6, (tf64, ref_tf)
    # and absolutely close to the fp32 oracle (coordinates are O(0.3)).  The dense cloud's bound is the sum of both sides' distances from
    # the exact result: the fp32 oracle's own is 5e-5 .. 6e-5 here and ours 3e-5 .. 5.4e-5, largest over 98304 coordinates
    assert ec < 1e-5 and ef < 2e-4


def test_running_statistics_match_oracle(run):
    sd = run.net.state_dict()
    for k, v in run.o64.new_stats.items():
        ours = sd[k].detach().cpu().double()
        err = ((ours - v.detach()).abs().max() / v.detach().abs().max().clamp_min(1e-12)).item()
        assert err < 1e-5 * fp32_mode_slack(), (k, err)            # (fp32-input MFMA mode: 1.5e-5 on the decoder's second layer)
        assert int(sd[k.rsplit(".", 1)[0] + ".num_batches_tracked"]) == 1


def test_parameter_gradients_match_oracle(run):
    """Every parameter's gradient against the oracle's autograd through the same (teacher-forced) network: relative L2 error
    of the whole tensor against the fp64 oracle, bounded by 2x the fp32 oracle's own error against fp64 (+ 1e-4) or by 2e-3 (see
    below): the HIP path must be as accurate as the reference's arithmetic."""
    rows = []
    for k, prm in run.net.named_parameters():
        g32, g64 = run.o32.sd[k].grad, run.o64.sd[k].grad
        if "maxpool" in k:
            assert g64 is None and (prm.grad is None or prm.grad.abs().max().item() == 0.0), k
            continue
        gm = prm.grad.detach().cpu().double()
        den = g64.norm().clamp_min(1e-30)
        rows.append((((gm - g64).norm() / den).item(), ((g32.double() - g64).norm() / den).item(), k))
    rows.sort(reverse=True)
    for ours, ref, k in rows:
        print(f"grad {k}: rel L2 err vs fp64 oracle: ours {ours:.2e}, fp32 oracle {ref:.2e}")
    # where the error of the decoder's first layer sits: per-sample (global-feature) columns | seed column | coarse column
    for k in ("decoder.final_conv.0.map_to_feat.weight", "decoder.final_conv.0.map_to_dir.weight"):
        gm = dict(run.net.named_parameters())[k].grad.detach().cpu().double()
        g32, g64 = run.o32.sd[k].grad.double(), run.o64.sd[k].grad
        for name, sl in (("global", slice(0, 2048)), ("seed", slice(2048, 2049)), ("coarse", slice(2049, 2050))):
            den = g64[:, sl].norm().clamp_min(1e-30)
            print(f"  {k}[{name}]: |g64| {den:.3e}  ours {((gm - g64)[:, sl].norm() / den):.2e}  fp32 oracle {((g32 - g64)[:, sl].norm() / den):.2e}")
        e = (gm - g64)[:, :2048]
        print(f"  {k}[global]: error rows (channels) norm max/median {e.norm(dim=1).max():.2e}/{e.norm(dim=1).median():.2e}; "
              f"cos(ours-g64, g32-g64) = {(e * (g32 - g64)[:, :2048]).sum() / (e.norm() * (g32 - g64)[:, :2048].norm()):.3f}; "
              f"cos(ours-g64, g64) = {(e * g64[:, :2048]).sum() / (e.norm() * g64[:, :2048].norm()):.3f}")
    # Both sides' errors are fp32 rounding noise amplified by a network that is ill-conditioned at B = 2 (train-mode BatchNorm + arg-max
    # pools).  Measured (see the per-block lines printed above): practically the whole error of decoder.final_conv.0.map_to_feat sits in
    # ONE of its 256 channels (error norm 0.29 against a median of 3e-4 per channel) — a channel whose norms barely vary over the batch, so
    # that rstd amplifies the rounding of |p| — and through the global feature's gradient that one channel sets the error of every encoder
    # tensor: ours 1.1e-3 .. 1.4e-3 in both evaluation orders of the encoder (VNP_ENC_AS_WRITTEN=0/1), IEEE or hardware division in the
    # gradient kernels, fp32 or double partial sums alike, while the fp32 oracle's own error on the same tensors was 4.7e-4 in one run
    # and 8.6e-4 in another (it moves with which of two near-tied points a pool picked and with the oracle's thread count; its worst
    # tensors reach 3.4e-3).  So: within 2x the fp32 oracle's error on the tensor, or under 2e-3 — the level the reference's own fp32
    # arithmetic reaches on this network.
    for ours, ref, k in rows:
        assert ours <= max(2.0 * fp32_mode_slack() * ref + 1e-4, 2e-3), (k, ours, ref)


def test_pruned_chamfer_16384_squared_bit_exact_vs_c_oracle(lib_built):
    """268 M pairs: the pruned exact search against oracle/chamfer_oracle.c, every distance bit and every index."""
    from oracle import chamfer_oracle as CO
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.chamfer_distance import chamfer_3DDist
    g = torch.Generator().manual_seed(3)
    a = torch.rand(1, MD, 3, generator=g) - 0.5
    b = (torch.rand(1, MD, 3, generator=g) - 0.5) * torch.tensor([1.0, 0.6, 0.3])          # anisotropic: uneven Morton bins
    tr = {}
    with ops.hooks(trace=tr):
        d1, d2, i1, i2 = chamfer_3DDist()(a.cuda(), b.cuda())
    assert tr == {"chamfer_pruned": 1}
    o1, o2, j1, j2 = CO.chamfer_forward(a.numpy(), b.numpy())
    assert np.array_equal(d1.cpu().numpy(), o1) and np.array_equal(d2.cpu().numpy(), o2)
    assert np.array_equal(i1.cpu().numpy(), j1) and np.array_equal(i2.cpu().numpy(), j2)


# ------------------------------------------------------------------------------------------------ config 3 at full resolution
def test_cfg3_dgcnn_full_resolution_vs_oracle(lib_built):
    """BASELINE config 3 (vn_dgcnn_fps + vn_foldingnet, k = 16 as the reference hard-codes, latent 512) at 2048 -> 1024 -> 16384, B = 2,
    against the oracle in fp32 and fp64: kNN graphs on 2048 / 512 / 128 points, both FPS stages, the four fused EdgeConv layers, pool5
    (teacher-forced after the score check), the decoder and both Chamfer-L1 heads.  The kNN / FPS conventions are the oracle's
    (third-party sources absent: PARITY UNPINNED for tie order); everything else is the reference's arithmetic."""
    from oracle import chamfer_oracle as CO
    from oracle import vn_oracle as O
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.loss import cd_loss_L1
    from vn_pointcloudcompletion_amd.model import PCNNet

    torch.manual_seed(3)
    cfg = SimpleNamespace(num_coarse=MC, latent_dim=512, only_coarse=False, enc_pretrained="none", device="cuda", knn_k=16)
    net = PCNNet(cfg, "vn_dgcnn_fps", "vn_foldingnet").train()
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(12)
    p = torch.rand(B, N, 3, generator=g) - 0.5
    c = torch.rand(B, MD, 3, generator=g) - 0.5
    tr = {}
    with ops.hooks(trace=tr):
        coarse, fine = net(p.cuda())
        loss1, loss2 = cd_loss_L1(coarse, c.cuda()), cd_loss_L1(fine, c.cuda())
        (loss1 + loss2).backward()
        torch.cuda.synchronize()
    assert tr.get("fold_layer0_fused", 0) == 1 and tr.get("decoder_tail_fused", 0) == 1 and tr.get("chamfer_pruned", 0) == 1, tr
    (idx5,) = [i.cpu().long() for i in tr["pool_idx"]]

    def oracle(dtype):
        sd = {k: (v.detach().clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
        sd = {k: v.requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
        otr = {}
        pp, cc = p.to(dtype), c.to(dtype)
        co, go = O.vn_dgcnn_fps_forward(sd, pp, True, 16, pool_idx=idx5, trace=otr)
        fo = O.vn_foldingnet_forward(sd, co, go, None, True)
        l1o, l2o = _cd_l1_oracle(O, CO, co, cc), _cd_l1_oracle(O, CO, fo, cc)
        (l1o + l2o).backward()
        return SimpleNamespace(otr=otr, co=co.detach(), fo=fo.detach(), loss=(l1o.item(), l2o.item()), sd=sd)
    o32, o64 = oracle(torch.float32), oracle(torch.float64)

    dot = o64.otr["dot5"]
    chosen = dot.gather(-1, idx5.unsqueeze(-1)).squeeze(-1)
    gap = ((dot.max(-1)[0] - chosen) / (dot.max(-1)[0] - dot.min(-1)[0]).clamp_min(1e-20)).max().item()
    agree = (o32.otr["idx5"] == idx5).float().mean().item()
    ec, ec64, ref_c = _maxerr(coarse.detach().cpu(), o32.co), _maxerr(coarse.detach().cpu(), o64.co), _maxerr(o32.co, o64.co)
    ef, ef64, ref_f = _maxerr(fine.detach().cpu(), o32.fo), _maxerr(fine.detach().cpu(), o64.fo), _maxerr(o32.fo, o64.fo)
    d1, d2 = abs(loss1.item() - o32.loss[0]), abs(loss2.item() - o32.loss[1])
    e1, e2 = abs(loss1.item() - o64.loss[0]), abs(loss2.item() - o64.loss[1])
    print(f"cfg 3: pool5 identical {agree:.4%}, worst score gap {gap:.1e}; coarse |ours-o32| {ec:.2e} |ours-o64| {ec64:.2e} (|o32-o64| {ref_c:.2e}); "
          f"fine {ef:.2e} / {ef64:.2e} ({ref_f:.2e}); CD-L1 diffs vs o32 {d1:.1e} / {d2:.1e}, vs o64 {e1:.1e} / {e2:.1e}")
    assert gap <= 2e-6 and agree >= 0.99
    assert d1 < 1e-5 and d2 < 1e-5 and e1 < 1e-5 and e2 < 1e-5
    # clouds against the exact result.  `coarse`: within 3x the fp32 oracle's own worst-element error (measured 7.4e-6 .. 9.3e-6 vs 7.7e-6).
    # `fine`: the decoder's BatchNorm layers amplify rounding, and the worst element of 98 k coordinates is a noisy statistic of that —
    # two builds of the encoder's statistics pass that differ only in the ORDER of a sum (4 or 8 partials per channel) measured 1.2e-4 and
    # 2.7e-4 against the oracle's own 5.9e-5 — so the worst element gets 8x, and the bound that carries the claim is the RMS error over
    # all coordinates, within 3x the fp32 oracle's
    rms = lambda a, b: (a.double() - b.double()).pow(2).mean().sqrt().item()
    rf64, rref = rms(fine.detach().cpu(), o64.fo), rms(o32.fo, o64.fo)
    print(f"cfg 3: fine rms error vs fp64 {rf64:.2e} (fp32 oracle's own {rref:.2e})")
    assert ec64 <= 3.0 * fp32_mode_slack() * ref_c + 5e-6 and ef64 <= 8.0 * fp32_mode_slack() * ref_f + 5e-6
    assert rf64 <= 3.0 * fp32_mode_slack() * rref + 1e-6
    rows = []
    for k, prm in net.named_parameters():
        g64 = o64.sd[k].grad
        if g64 is None:
            assert prm.grad is None or prm.grad.abs().max().item() == 0.0, k
            continue
        den = g64.norm().clamp_min(1e-30)
        rows.append((((prm.grad.detach().cpu().double() - g64).norm() / den).item(), ((o32.sd[k].grad.double() - g64).norm() / den).item(), k))
    rows.sort(reverse=True)
    for ours, ref, k in rows[:4]:
        print(f"  grad {k}: rel L2 err vs fp64 oracle: ours {ours:.2e}, fp32 oracle {ref:.2e}")
    # every gradient within 3x the fp32 oracle's own error on THAT tensor — or, for a tensor on which the oracle happened to land close
    # (conv1's 64 weights: 1.0e-3 where the same network's conv7 gradients carry 4-5e-3 in the oracle's own fp32 run), within 1.5x the
    # oracle's worst tensor.  Measured: two builds of the conv1 passes that differ only in the order of a sum gave 3e-3 and 5.0e-3 there
    worst_ref = max(ref for _, ref, _ in rows)
    for ours, ref, k in rows:
        assert ours <= max(3.0 * fp32_mode_slack() * ref + 2e-4, 1.5 * fp32_mode_slack() * worst_ref), (k, ours, ref, worst_ref)
