"""Extra legs of bench.py (N = 1 only, outside the timed region): the other BASELINE configurations at their full sizes
(`configs`), and `roofline_other` entries for the non-matrix kernels those configurations add — each kernel run alone on step-sized
operands and timed with HIP events on its launch stream, priced on what it executes against the roof that bounds it.

Peaks (/opt/skills/guides/MI355X_MICROARCH.md): HBM 8 TB/s; L2 34.5 TB/s aggregate (8 XCDs x 4 MiB); fp32 VALU 78.6 T lane-ops/s
(157.3 TF / 2); transcendentals (v_exp_f32: 8 issue cycles per wave against 4 for v_fma_f32... i.e. a quarter of the 32 lanes/clk/SIMD
FMA rate) 19.7 T/s.
"""
import time
from types import SimpleNamespace

PEAK_HBM_GBS = 8000.0
PEAK_L2_GBS = 34500.0
PEAK_VALU_TLANEOPS = 78.65
PEAK_EXP_T = PEAK_VALU_TLANEOPS / 4.0
PEAK_SPLIT_TFLOPS = 2500.0 / 6.0

CONFIGS = {
    3: dict(enc="vn_dgcnn_fps", dec="vn_foldingnet", B=32, latent=512, loss="cd", mc=1024, knn_k=20,
            name="vn_dgcnn_fps + vn_foldingnet (kNN k=20 edge-conv), B=32, 2048->16384"),
    4: dict(enc="vn_pointnet", dec="attention_vn_foldingnet", B=16, latent=2048, loss="emd", mc=1024,
            name="vn_pointnet + attention_vn_foldingnet + EMD coarse loss, B=16, 2048->16384"),
    5: dict(enc="vn_pointr", dec="vn_foldingnet", B=32, latent=1024, loss="cd", mc=1024,
            name="vn_pointr + vn_foldingnet, B=32 per GPU (the 8-GPU config's shard of B=256), 2048->19968"),
}


def _events(fn, iters, warm=2):
    import torch
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def other_configs(steps=5, warmup=3, which=(3, 4, 5)):
    """One full training step (forward, both losses, backward, fused Adam) of BASELINE configs 3, 4, 5 at full batch and resolution on one
    GPU: `warmup` untimed + `steps` timed steps each (wall clock around synchronised steps, like the headline line).  The dominant kernel
    of all three is the split GEMM family (the same arithmetic as config 2's); its executed 2MNK / kernel time comes from the library's
    HIP events around each launch (vnp_gemm_profile_next) and reproduces from the vnp_gemm_split* rows of profiles/rNN_cfgK_kernel_stats.csv."""
    import torch
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.model import PCNNet
    from vn_pointcloudcompletion_amd.optim import FusedAdam
    from vn_pointcloudcompletion_amd.train import settle_gc, train_step
    out = {}
    for cid in which:
        c = CONFIGS[cid]
        cfg = SimpleNamespace(num_coarse=c["mc"], latent_dim=c["latent"], only_coarse=False, enc_pretrained="none", device="cuda:0",
                              coarse_loss=c["loss"], dcd_opts={"alpha": 200, "lambda": 0.5}, knn_k=c.get("knn_k", 16))
        torch.manual_seed(0)
        net = PCNNet(cfg, c["enc"], c["dec"]).train()
        md = getattr(net.decoder, "num_dense", 16384)
        opt = FusedAdam(net.parameters(), lr=1e-4)
        g = torch.Generator().manual_seed(1)
        p = (torch.rand(c["B"], 2048, 3, generator=g) - 0.5).cuda()
        gt = (torch.rand(c["B"], md, 3, generator=g) - 0.5).cuda()
        for i in range(warmup):
            train_step(cfg, net, opt, p, gt)
            if i == 0:
                settle_gc()                                        # as train.train() does after its first step
        torch.cuda.synchronize()
        # the step time on un-instrumented steps; the GEMM roofline on `steps` more steps with the per-launch timers attached (they cost
        # a step of config 5 — 150 matrix-product launches — about a millisecond, scripts/experiments/hook_overhead.py)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, _, _ = train_step(cfg, net, opt, p, gt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        prof = []
        with ops.hooks(gemm_profile=prof):
            train_step(cfg, net, opt, p, gt)
            torch.cuda.synchronize()
            ops.reserve_timers(len(prof) * steps + 8)
            del prof[:]
            t1 = time.perf_counter()
            for _ in range(steps):
                train_step(cfg, net, opt, p, gt)
            torch.cuda.synchronize()
            dt_hooked = (time.perf_counter() - t1) / steps
        recs = [(f, t.seconds()) for _, f, _, _, t in prof if t.kind == 0 and t.seconds() is not None]
        fl, tm = sum(f for f, _ in recs), sum(t for _, t in recs)
        allm = sum(t.seconds() for *_, t in prof if t.seconds() is not None)
        out[f"cfg{cid}"] = {
            "workload": c["name"] + ", full train step (fwd + losses + bwd + Adam), fp32",
            "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 3), "points_per_s": c["B"] * md / dt,
            "ms_per_step_with_launch_timers": round(dt_hooked * 1e3, 3),
            "loss": float(loss.item()), "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            "dominant_kernel": "vnp_gemm_split_kernel<*> + vnp_gemm_split_wide_kernel<*> + vnp_gemm_split_wide_tn_kernel<*> "
                               "(six bf16 MFMA products per fp32 product)",
            "roofline": {"bound": "mfma", "achieved": round(fl / tm / 1e12, 2) if tm > 0 else None, "peak": round(PEAK_SPLIT_TFLOPS, 2),
                         "unit": "TFLOP/s", "frac": round(fl / tm / 1e12 / PEAK_SPLIT_TFLOPS, 4) if tm > 0 else None,
                         "ms_per_step": round(tm / steps * 1e3, 3), "launches_per_step": len(recs) // steps,
                         "all_matrix_kernels_ms_per_step": round(allm / steps * 1e3, 3),
                         "note": "executed 2MNK of the split-GEMM launches / their kernel time (library HIP events), taken on `steps` further steps "
                                 "after the timed ones: ms_per_step carries no instrumentation"}}
        del net, opt, p, gt, prof
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
    return out


def roofline_other_configs():
    """Non-matrix kernels of configs 3, 4, 5, each alone on the operands its step hands it."""
    import torch
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd._lib import call, ptr
    out = []
    g = torch.Generator().manual_seed(9)

    # ---- EMD approxmatch (config 4's coarse loss: B = 16, 1024 x 1024): 10 levels x 3 passes, one exp per (level, pass, pair)
    B, n, m = 16, 1024, 1024
    a, b = (torch.rand(B, n, 3, generator=g) - 0.5).cuda(), (torch.rand(B, m, 3, generator=g) - 0.5).cuda()
    match = torch.empty((B, m, n), device="cuda")
    temp = torch.empty((B, (n + m) * 2), device="cuda")
    dt = _events(lambda: call("vnp_emd_approxmatch", B, n, m, ptr(a), ptr(b), ptr(match), ptr(temp)), 10)
    nexp = 30.0 * B * n * m
    out.append({"kernel": "emd_pass_a / emd_pass_b / emd_pass_c_* kernels behind vnp_emd_approxmatch (B = 16, 1024 x 1024; emd_kernel.cu:25-158)",
                "bound": "valu-transcendental", "ms": round(dt * 1e3, 4), "achieved": round(nexp / dt / 1e12, 3), "peak": round(PEAK_EXP_T, 2),
                "unit": "T exp/s", "frac": round(nexp / dt / 1e12 / PEAK_EXP_T, 4), "exp_evaluations": nexp,
                "note": "30 DEPENDENT passes (10 levels x 3) of n*m exponentials per sample, each pass its own launch(es): 41 launches, "
                        "1.0 M pairs per sample and pass — the chain's launch boundaries and the per-pass reductions, not the exponentials, "
                        "set the time"})

    # ---- FPS (configs 3 and 5: 2048 -> 512 and 512 -> 128 per sample, one workgroup per sample): M - 1 dependent arg-max rounds
    for B_, N_, M_ in ((32, 2048, 512), (32, 512, 128)):
        x = (torch.rand(B_, N_, 3, generator=g) - 0.5).cuda()
        dt = _events(lambda: ops.fps_indices(x, M_), 10)
        per = dt / (M_ - 1)
        floor = 0.30e-6
        out.append({"kernel": f"vnp_fps_small_kernel (furthest point sampling {N_} -> {M_}, B = {B_}: one workgroup per sample)",
                    "bound": "latency", "ms": round(dt * 1e3, 4), "achieved": round(per * 1e9, 1), "peak": floor * 1e9, "unit": "ns per round",
                    "frac": round(floor / per, 4), "rounds": M_ - 1,
                    "note": "a round = distance update of N / threads points per thread, DPP arg-max in the wave, one LDS exchange between the "
                            "waves, broadcast of the pick: a dependent chain; peak = 0.30 us per round, the floor of that chain at 2.4 GHz "
                            "(two workgroup barriers, two 6-step cross-lane reductions, one LDS round trip and the pick's coordinate load "
                            "~ 700 cycles); frac = floor / measured"})

    # ---- fused EdgeConv (config 3: conv1 on 65536 points x k = 20, 32 channels; conv5 on 16384 points, 128 channels): L2 gathers
    for P, k, C, tag in ((32 * 2048, 20, 32, "conv1"), (32 * 512, 20, 128, "conv5")):
        uv = torch.randn(P * 3, 4 * C, generator=torch.Generator().manual_seed(P)).cuda()
        per_sample = P // 32
        nbr = (torch.randint(0, per_sample, (32, per_sample, k), generator=g, dtype=torch.int32)
               + (torch.arange(32, dtype=torch.int32) * per_sample).view(32, 1, 1)).reshape(P, k).cuda()
        gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")

        def fwd():
            with torch.no_grad():
                ops.EdgeConvPackedFn.apply(uv, nbr, gam, bet, rm, rv, True, 0.2)
        dt = _events(fwd, 10)
        # two passes (statistics, apply), each gathers u (feat | dir: 2C floats) for every edge row and reads v once; the second writes out
        gathered = 2.0 * (P * k * 3 * 2 * C * 4 + P * 3 * 2 * C * 4) + P * 3 * C * 4
        out.append({"kernel": f"vnp_edge_fused_kernel<0> + <1> (fused EdgeConv forward, config 3 {tag}: {P} points x k = {k}, {C} channels)",
                    "bound": "l2", "ms": round(dt * 1e3, 4), "achieved": round(gathered / dt / 1e9, 1), "peak": PEAK_L2_GBS, "unit": "GB/s",
                    "frac": round(gathered / dt / 1e9 / PEAK_L2_GBS, 4), "algorithmic_bytes": gathered,
                    "hbm_GBs": round((P * 3 * 4 * C * 4 * 2.0 + P * 3 * C * 4) / dt / 1e9, 1),
                    "note": "bytes = every edge's neighbour row gathered once per pass (2 passes) + v + out; the neighbour rows of a sample "
                            "(0.2-1.6 MB) live in the XCD's L2, so the gathers are priced on the L2's 34.5 TB/s, HBM traffic rides along"})

    # ---- attention softmax and VNLayerNorm (config 4: 16 x 8 heads x 1024 x 1024 scores; 16 x 1024 tokens x 384 channels): HBM
    R, L = 16 * 8 * 1024, 1024
    S = torch.randn(R, L, device="cuda")
    dP = torch.randn(R, L, device="cuda")
    dt = _events(lambda: call("vnp_softmax_fwd", ptr(S), R, L, 1.0), 10)
    out.append({"kernel": "vnp_softmax_fwd_reg_kernel (config 4 attention: 131072 rows of 1024 scores, in place)", "bound": "hbm",
                "ms": round(dt * 1e3, 4), "achieved": round(2.0 * R * L * 4 / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(2.0 * R * L * 4 / dt / 1e9 / PEAK_HBM_GBS, 4), "algorithmic_bytes": 2.0 * R * L * 4, "note": "read S, write P"})
    dt = _events(lambda: call("vnp_softmax_bwd", ptr(S), ptr(dP), R, L, 1.0), 10)
    out.append({"kernel": "vnp_softmax_bwd_reg_kernel (same shape)", "bound": "hbm", "ms": round(dt * 1e3, 4),
                "achieved": round(3.0 * R * L * 4 / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(3.0 * R * L * 4 / dt / 1e9 / PEAK_HBM_GBS, 4), "algorithmic_bytes": 3.0 * R * L * 4,
                "note": "read P and dP, write dS"})
    del S, dP
    P_, C_ = 16 * 1024, 384
    x = torch.randn(P_ * 3, C_, device="cuda").requires_grad_(True)
    gam = torch.ones(C_, device="cuda", requires_grad=True)
    bet = torch.zeros(C_, device="cuda", requires_grad=True)
    go = torch.randn(P_ * 3, C_, device="cuda")

    def ln_fwd():
        with torch.no_grad():
            ops.VNLayerNormFn.apply(x, gam, bet)
    dt = _events(ln_fwd, 20)
    e = P_ * 3 * C_ * 4.0
    out.append({"kernel": "vnp_ln_fwd_kernel (VNLayerNorm, config 4: 16384 tokens x 384 channels)", "bound": "hbm", "ms": round(dt * 1e3, 4),
                "achieved": round(2 * e / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(2 * e / dt / 1e9 / PEAK_HBM_GBS, 4),
                "algorithmic_bytes": 2 * e, "note": "read x, write out (75 MB each: a launch of ~30 us sits on the ramp of the HBM curve)"})
    y = ops.VNLayerNormFn.apply(x, gam, bet)

    def ln_bwd():
        torch.autograd.grad(y, (x, gam, bet), go, retain_graph=True)
    dt = _events(ln_bwd, 20)
    out.append({"kernel": "vnp_ln_bwd_kernel + its partial-sum reduction (same shape)", "bound": "hbm", "ms": round(dt * 1e3, 4),
                "achieved": round(3 * e / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(3 * e / dt / 1e9 / PEAK_HBM_GBS, 4),
                "algorithmic_bytes": 3 * e, "note": "read go and x, write gx; includes the autograd dispatch of one Function"})
    return out
