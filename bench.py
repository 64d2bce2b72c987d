#!/usr/bin/env python3
"""Benchmark of the hot path: one full training step (forward, 2x Chamfer-L1, backward, gradient all-reduce, Adam)
of vn_pointnet + vn_foldingnet on synthetic (B,2048,3) -> (B,16384,3) clouds, fp32, B = 32 per GPU (BASELINE config 2).

    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: this process starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (the dominant kernel, vnp_gemm_split_kernel: EXECUTED
work over KERNEL time — HIP events recorded by the library around each launch of that kernel on its launch stream, so the figure is
Σflops / Σduration over the vnp_gemm_split_kernel / vnp_gemm_split_wide_kernel rows of a rocprofv3 kernel trace of the same command, profiles/rNN_bench_kernel_stats.csv),
`roofline_other` (the arg-max pool's screening pass on the fp16 pipe it uses; Chamfer — the pruned search priced on the pairs it
actually evaluates, counted by the kernel — and kNN on the fp32 VALU bound; the BatchNorm / leaky passes and the streaming kernels of
the degenerate matrix shapes on the HBM bound), `chamfer_op` (the one shape the reference tree publishes a number for) and
`cpu_baseline` (the torch-CPU oracle on a bounded sample).  No entry prices a kernel on work it does not execute; the algorithmic and
SURVEY §8(d) views of the matrix work ride along under `roofline.views` without a `frac` of their own.

GEMM arithmetic: fp32 operands and fp32 results; by default each operand is split EXACTLY into three bf16
terms and six bf16 MFMA products are accumulated in fp32 (error vs fp64 at or below the fp32-input MFMA's — asserted by
tests/test_gpu_gemm.py::test_split_is_fp32_grade).  The roofline is priced on the bf16 pipe actually used: peak = 2500/6
TFLOP/s of fp32-equivalent work (the executed bf16 rate is 6 x achieved).  At N = 1 the same step is also timed with the fp32-input
MFMA kernels (VNP_GEMM_MODE=fp32) and reported beside it as `fp32_mfma_mode`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_FP32_VALU_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 vector peak (FMA = 2 flops/lane/clk)
PEAK_BF16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: bf16 dense
PEAK_HBM_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E spec (≈6300 achievable)
SURVEY_FLOP_PER_CLOUD = 212.0e9          # matrix FLOPs per completed cloud and training step, network as written
SPLIT_PRODUCTS = 6                     # bf16 MFMA products per fp32 product in the split kernel
METRIC = "completed points/sec/GPU + Chamfer-L1, 2048→16384, B=32, 1/2/4/8 MI355X"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--num-coarse", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-mode", action="store_true", help="skip the extra fp32-input-MFMA comparison steps (clean profiles)")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline_other / chamfer_op / configs (clean profiles)")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs 3, 4, 5 leg (`configs`)")
    ap.add_argument("--sustain", type=float, default=6.0, help="seconds of back-to-back steps after the timed region (`sustained`; 0: skip)")
    ap.add_argument("--shapes", action="store_true", help="print the kernel time of every matrix product of the timed steps, by shape (stderr)")
    ap.add_argument("--deadline", type=float, default=900.0,
                    help="self-launched ranks (--gpus N without a launcher): wall-clock limit in seconds after which the launcher "
                         "terminates every rank and exits non-zero (a rank stuck in the rendezvous would otherwise hold the others)")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="launcher / rendezvous / bucketed all-reduce rehearsal on CPU ranks over gloo with a toy torch model: "
                         "no HIP kernel runs and the printed line is marked invalid (tests/test_bench_launch.py)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------- rank launcher
def launch_ranks(n, deadline_s):
    """`python bench.py --gpus N` with no launcher: start N rank processes (one per GPU, LOCAL_RANK = GPU ordinal) as children
    of this process, which itself never touches the GPU.  Rank 0 inherits stdout and prints the JSON line.  Exit status is
    non-zero if any rank fails, fewer than N join (the survivors are terminated instead of waiting in the rendezvous), or the
    run is not over after ``deadline_s`` seconds (a rank that is alive but stuck — it never joins, or hangs in a collective —
    is not an exit: without the deadline the launcher would poll until the driver's own limit)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    failed, timed_out = None, False
    t_end = time.monotonic() + deadline_s
    while failed is None and any(p.poll() is None for p in procs):
        time.sleep(0.2)
        failed = next((r for r, p in enumerate(procs) if p.poll() not in (None, 0)), None)
        if failed is None and time.monotonic() > t_end:
            timed_out = True
            break
    if failed is not None or timed_out:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
        if timed_out:
            alive = [r for r, p in enumerate(procs) if p.returncode not in (0,)]
            print(f"bench.py: not finished after the {deadline_s:.0f} s deadline (ranks still running then: {alive}); all ranks terminated",
                  file=sys.stderr)
            return 2
        print(f"bench.py: rank {failed} exited with status {procs[failed].returncode}; {n} ranks were required", file=sys.stderr)
        return 1
    return 0


# ----------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(threads):
    """The torch-CPU oracle (reference-as-written formulation: the 12.9 GB/32 broadcast tensors, eager layer chains) on a
    bounded sample of the same workload, as SURVEY.md §8d specifies: forward + 2x Chamfer-L1 + backward (no optimizer) at
    full resolution for B = 1 and B = 2, three repeats each, best time kept; nearest neighbours by oracle/chamfer_oracle.c
    (OpenMP, all `threads` cores; the reference's own CPU Chamfer, distChamfer, takes ~80 s per 16384^2 cloud)."""
    import torch
    from oracle import chamfer_oracle as CO
    from oracle import vn_oracle as O
    from vn_pointcloudcompletion_amd.model import PCNNet
    torch.set_num_threads(threads)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    cfg = SimpleNamespace(num_coarse=1024, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cpu")
    torch.manual_seed(0)
    net = PCNNet(cfg, "vn_pointnet", "vn_foldingnet")          # parameter container only: never run on the CPU
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(1234)

    def cd_l1(pred, gt):
        _, _, i1, i2 = CO.chamfer_forward(pred.detach().numpy(), gt.numpy())
        return O.cd_loss_l1(*O.chamfer_from_idx(pred, gt, i1, i2))

    per_b = {}
    for B in (1, 2):
        p = torch.rand(B, 2048, 3, generator=g) - 0.5
        c = torch.rand(B, 16384, 3, generator=g) - 0.5
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            coarse, fine, _ = O.pcnnet_forward(sd, p, None, True)
            (cd_l1(coarse, c) + cd_l1(fine, c)).backward()
            times.append(time.perf_counter() - t0)
            for v in sd.values():
                v.grad = None
        per_b[B] = min(times)
    best_b = max(per_b, key=lambda b: b * 16384.0 / per_b[b])
    return {"value": best_b * 16384.0 / per_b[best_b], "unit": "points/s", "cores": threads, "threads_used": threads,
            "cores_available": os.cpu_count(), "kind": "port",
            "nearest_neighbour_search": "oracle/chamfer_oracle.c (OpenMP restatement of chamfer3D.cu), NOT the reference's pure-PyTorch "
                                        "distChamfer (~80 s per 16384^2 cloud): a faster CPU baseline than the reference's own",
            "s_per_step": {f"B={b}": round(t, 2) for b, t in per_b.items()},
            "s_per_cloud": {f"B={b}": round(t / b, 2) for b, t in per_b.items()},
            "sample": "torch-CPU oracle, B=1 and B=2 of the B=32 step at full resolution (2048->1024->16384), fwd+2xCD-L1+bwd, "
                      f"best of 3 each; value = the better of the two ({best_b * 16384.0 / per_b[best_b]:.0f} points/s at B={best_b}); "
                      "per-cloud work is independent of B apart from BatchNorm statistics, so points/s extrapolates linearly"}


# ----------------------------------------------------------------------------------------------- extra legs (N = 1 only)
def _time_events(fn, iters, warm=2):
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


def chamfer_op_bench():
    """The one operator number the reference tree publishes: Chamfer 3-D forward + backward on p1 = 32x2000x3, p2 = 32x1000x3
    (extensions/ChamferDistancePytorch/unit_test.py:38-61; README.md:49-57: 1.4 ms on an unnamed NVIDIA GPU, CUDA 10.1).
    Same protocol: 100 iterations of forward + backward of (mean(dist1) + mean(dist2)).  Context, not a target."""
    import torch
    from vn_pointcloudcompletion_amd.chamfer_distance import chamfer_3DDist
    g = torch.Generator().manual_seed(0)
    p1 = torch.rand(32, 2000, 3, generator=g).cuda().requires_grad_(True)
    p2 = torch.rand(32, 1000, 3, generator=g).cuda().requires_grad_(True)
    op = chamfer_3DDist()

    def fwd_bwd():
        d1, d2, _, _ = op(p1, p2)
        (d1.mean() + d2.mean()).backward()
    dt = _time_events(fwd_bwd, 100, warm=5)

    def fwd():
        with torch.no_grad():
            op(p1, p2)
    dtf = _time_events(fwd, 100, warm=5)
    return {"shape": "p1=32x2000x3, p2=32x1000x3, fp32", "fwd_bwd_ms": round(dt * 1e3, 4), "fwd_ms": round(dtf * 1e3, 4), "iters": 100,
            "reference_published_ms": 1.4, "reference_hardware": "unnamed NVIDIA GPU, CUDA 10.1, PyTorch 1.4 (ChamferDistancePytorch/README.md:49-57)",
            "note": "wall time per iteration incl. autograd and the ATen mean/backward glue, HIP events on the launch stream"}


def roofline_other(B, N, Mc, Md):
    """Kernels of the step that are NOT matrix-bound, each run alone on step-sized operands and timed with HIP events:
    Chamfer / kNN on the fp32 VALU bound (pair evaluations/s; ~10 lane-ops per pair), the BatchNorm / leaky / fold passes on
    the HBM bound (algorithmic bytes: every operand read or written once)."""
    import torch
    from vn_pointcloudcompletion_amd import ops
    out = []
    g = torch.Generator().manual_seed(5)
    dense = (torch.rand(B, Md, 3, generator=g) - 0.5).cuda()
    gt = (torch.rand(B, Md, 3, generator=g) - 0.5).cuda()
    coarse = (torch.rand(B, Mc, 3, generator=g) - 0.5).cuda()
    part = (torch.rand(B, N, 3, generator=g) - 0.5).cuda()

    def valu(name, fn, pairs, hbm_bytes, note):
        dt = _time_events(fn, 10)
        ops_per_pair = 10.0                                    # 3 sub, 1 mul + 2 fma, compare/select pair, index bookkeeping
        out.append({"kernel": name, "bound": "valu", "ms": round(dt * 1e3, 4), "pairs_per_s": pairs / dt,
                    "achieved": round(pairs * ops_per_pair / dt / 1e12, 3), "peak": PEAK_FP32_VALU_TFLOPS / 2, "unit": "T lane-op/s",
                    "frac": round(pairs * ops_per_pair / dt / 1e12 / (PEAK_FP32_VALU_TFLOPS / 2), 4),
                    "hbm_GBs": round(hbm_bytes / dt / 1e9, 2), "note": note})

    with torch.no_grad():
        valu("vnp_chamfer_fwd (brute force; coarse loss 1024 x 16384, both directions)", lambda: ops.ChamferFn.apply(coarse, gt),
             2.0 * B * Mc * Md, B * (Mc + Md) * 12 + B * (Mc + Md) * 8,
             "pairs = 2*n*m per cloud (two one-sided scans, like chamfer3D.cu); peak = 78.6 T lane-op/s (one VALU op per lane per clock)")
        # the pruned search is priced on the pairs it EVALUATES: the kernel counts them (ws + 8, include/vnpcc.h)
        stats = []
        with ops.hooks(chamfer_stats=stats):
            ops.ChamferFn.apply(dense, gt)
        torch.cuda.synchronize()
        visited = float(stats[0].item())
        valu("vnp_chamfer_pruned_kernel + vnp_chamfer_bin_kernel (dense loss 16384 x 16384, uniform random clouds)",
             lambda: ops.ChamferFn.apply(dense, gt), visited, B * 2 * Md * 12 + B * 2 * Md * 8,
             "exact search over Morton-binned chunks with box lower bounds; pairs = the (query, target) evaluations the kernel counted "
             "(bit-identical outputs to the full scan: tests/test_gpu_chamfer.py); time includes the binning pass")
        out[-1]["pairs_evaluated"] = visited
        out[-1]["pairs_of_a_full_scan"] = 2.0 * B * Md * Md
        out[-1]["visited_fraction"] = round(visited / (2.0 * B * Md * Md), 4)
        out[-1]["full_scan_equivalent_pairs_per_s"] = 2.0 * B * Md * Md / (out[-1]["ms"] * 1e-3)
        wc = stats[1].float()                                  # chunks of 64 targets scanned by each wave of 64 queries (256 possible)
        half = wc.numel() // 2
        out[-1]["chunks_scanned_per_wave"] = {
            name: {"mean": round(w_.mean().item(), 1), "median": w_.median().item(), "p99": w_.kthvalue(int(0.99 * w_.numel())).values.item(),
                   "max": w_.max().item(),
                   "waves_with_0_16_32_64_128_to_256_chunks": [int(((w_ >= lo) & (w_ < hi)).sum()) for lo, hi in ((0, 16), (16, 32), (32, 64), (64, 128), (128, 257))]}
            for name, w_ in (("queries = prediction", wc[:half]), ("queries = ground truth", wc[half:]))}
        out[-1]["note"] += ("; a wave's walk is a dependent chain, so the launch lasts as long as its longest chain: chunks_scanned_per_wave shows "
                            "the imbalance (mean vs max)")
        valu("vnp_knn_coords_kernel (k = 16 graph on 2048 x 2048 coordinates, config 3 first layer)", lambda: ops.knn_indices(part, part, 16),
             1.0 * B * N * N, B * N * 12 * 2 + B * N * 16 * 4,
             "one wave per query: distances in registers, threshold selection (k-th smallest lane minimum), no distance matrix; "
             "~10 lane-ops per pair for the distance + ~15 per pair-equivalent for the two cross-lane sorts and the compaction")

    # HBM-bound passes on decoder-sized operands: P = B*Md points, C = 256 channels, (p | d) pairs [P*3, 512]
    P, C = B * Md, 256
    pd = torch.randn(P * 3, 2 * C, device="cuda")
    go = torch.randn(P * 3, C, device="cuda")
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    res = {}

    def fwd():
        res["out"], res["mean"], res["rstd"] = ops._bn_leaky_forward(pd[:, :C], pd[:, C:], C, gamma, beta, rm, rv, True, 0.2)

    def bwd():
        ops._bn_leaky_backward(go, pd[:, :C], pd[:, C:], C, gamma, beta, res["mean"], res["rstd"], True, 0.2)
    with torch.no_grad():
        e = P * 3 * C * 4.0                                    # bytes of one [P*3, C] fp32 map
        for name, fn, nbytes, note in (
                ("VNBatchNorm + leaky forward (vnp_vn_norm_stats + vnp_vn_bn_leaky_fwd), [B*16384*3, 256]", fwd, 4 * e,
                 "read p (statistics), read p and d, write out"),
                ("VNBatchNorm + leaky backward (vnp_vn_bn_leaky_bwd1 + vnp_vn_bn_bwd2)", bwd, 8 * e,
                 "bwd1: read go, p, d, write gq, gd; bwd2: read gq, p, write gq")):
            dt = _time_events(fn, 5)
            out.append({"kernel": name, "bound": "hbm", "ms": round(dt * 1e3, 4), "achieved": round(nbytes / dt / 1e9, 1), "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": round(nbytes / dt / 1e9 / PEAK_HBM_GBS, 4), "algorithmic_bytes": nbytes, "note": note})
    return out


# ----------------------------------------------------------------------------------------------- CPU rehearsal of the N-rank path
def rehearse_cpu(args):
    """`--rehearse-cpu`: the launcher, the rendezvous, the weight broadcast, the bucketed backward-overlapped all-reduce and
    the barrier / max-over-ranks timing protocol, on CPU ranks over gloo with a toy torch model standing in for the network
    (no HIP kernel can run here).  The printed line is NOT a measurement (`valid: false`)."""
    import torch
    from vn_pointcloudcompletion_amd import distributed as D
    from vn_pointcloudcompletion_amd.optim import FlatBuffers
    if os.environ.get("VNP_BENCH_FAIL_RANK") == os.environ.get("RANK"):
        raise SystemExit(3)                                      # test hook: a rank that dies before the rendezvous
    if os.environ.get("VNP_BENCH_STUCK_RANK") == os.environ.get("RANK"):
        time.sleep(3600)                                         # test hook: a rank that is alive but never joins
    rank, _, world = D.init_from_env(backend="gloo")
    if world != args.gpus or world < 2:
        raise SystemExit(f"--rehearse-cpu needs --gpus N >= 2 ranks (got --gpus {args.gpus}, WORLD_SIZE={world})")
    torch.manual_seed(100 + rank)
    model = torch.nn.Sequential(torch.nn.Linear(48, 64, bias=False), torch.nn.Linear(64, 64, bias=False), torch.nn.Linear(64, 3, bias=False))
    D.broadcast_model(model)
    fb = FlatBuffers(model.parameters())
    red = D.OverlappedGradReducer(fb, bucket_bytes=4096, gap_bytes=0)
    holder = SimpleNamespace(flat_buffers=[fb], grad_reducers=[red])
    x = torch.randn(args.batch, 48, generator=torch.Generator().manual_seed(1234 + rank))
    first_wire = red.wire_bytes()
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        fb.zero_grad()
        model[2](model[0](x)).pow(2).mean().backward()          # model[1] never receives a gradient (like maxpool*.map_to_dir)
        D.allreduce_gradients(holder)
    D.barrier()
    elapsed = D.allreduce_scalar_max(time.perf_counter() - t0)
    same = torch.stack([fb.flat_grad.double().sum(), fb.flat_grad.double().abs().sum()])
    lo, hi = same.clone(), same.clone()
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / (args.warmup + args.steps) * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "none (CPU rehearsal of the multi-rank path over gloo)", "valid": False,
                          "config": {"workload": "toy torch model: launcher + rendezvous + bucketed all-reduce rehearsal, no HIP kernels",
                                     "global_batch": args.batch * world, "parallelism": f"dp{world}"},
                          "ranks_joined": torch.distributed.get_world_size(),
                          "allreduce": {"backend": "gloo", "bytes_per_step_first": first_wire, "bytes_per_step": red.wire_bytes(),
                                        "buckets": len(red.buckets), "replicas_agree": bool(torch.equal(lo, hi))}}))
    D.barrier()
    D.shutdown()


# ----------------------------------------------------------------------------------------------- the benchmark
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  Nothing above has touched the GPU (torch is not even imported yet).
        raise SystemExit(launch_ranks(args.gpus, args.deadline))
    if args.rehearse_cpu:
        return rehearse_cpu(args)

    import torch
    from vn_pointcloudcompletion_amd import _lib, ops
    from vn_pointcloudcompletion_amd import distributed as D
    from vn_pointcloudcompletion_amd.model import PCNNet
    from vn_pointcloudcompletion_amd.optim import FusedAdam
    from vn_pointcloudcompletion_amd.train import settle_gc, train_step

    rank, local_rank, world = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would misreport n_gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(device)
    _lib.lib()

    B, N, Mc = args.batch, 2048, args.num_coarse
    Md = Mc * 16
    cfg = SimpleNamespace(num_coarse=Mc, latent_dim=2048, only_coarse=False, enc_pretrained="none", device=device,
                          coarse_loss="cd", dcd_opts={"alpha": 200, "lambda": 0.5})
    torch.manual_seed(0)
    model = PCNNet(cfg, "vn_pointnet", "vn_foldingnet")
    model.decoder.num_dense = Md
    model.train()
    D.broadcast_model(model)
    opt = FusedAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.999))
    overlap = D.collectives_on() and os.environ.get("VNP_OVERLAP_ALLREDUCE", "1") != "0"
    if overlap:
        opt.enable_overlapped_allreduce()            # bucketed all-reduce overlapped with backward (distributed.py)
    g = torch.Generator().manual_seed(1234 + rank)
    p = (torch.rand(B, N, 3, generator=g) - 0.5).to(device)
    c = (torch.rand(B, Md, 3, generator=g) - 0.5).to(device)

    # W untimed warm-up steps.  The LAST of them runs with the roofline's per-launch timers attached: it counts the matrix-product launches
    # of a step, and steps x that many timers (two HIP events each) are created before the timed region instead of inside it
    probe = []
    for i in range(args.warmup):
        if i == args.warmup - 1:
            with ops.hooks(gemm_profile=probe):
                train_step(cfg, model, opt, p, c)
        else:
            train_step(cfg, model, opt, p, c)
        if i == 0:
            settle_gc()          # as train.train() does after its first step: no 70-90 ms full collection of the interpreter's heap later on
    if args.warmup == 0:
        settle_gc()
    torch.cuda.synchronize()
    ops.reserve_timers((len(probe) or 64) * (args.steps // 4 + 1) + 8)
    del probe
    D.barrier()
    torch.cuda.synchronize()
    prof = []
    waits = [] if (D.collectives_on() and overlap) else None
    D.EXPOSED_WAIT = waits
    # Per-launch HIP events INSIDE the timed region, on every PROF_STRIDE-th step of it: a timed dispatch ends with a completion signal and
    # a release of its own, and instrumenting all ~45 matrix-product launches of all K steps cost the region 0.25 ms per step (config 2;
    # 0.9 ms on config 5: scripts/experiments/hook_overhead.py) — 1.4 % of the number the region exists to measure.  Sampled steps are
    # ordinary steps of the same loop; the averages below are over their launches.
    PROF_STRIDE = 4 if args.steps >= 8 else 1
    prof_steps = len(range(0, args.steps, PROF_STRIDE))
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % PROF_STRIDE == 0:
            with ops.hooks(gemm_profile=prof):
                loss, loss1, loss2 = train_step(cfg, model, opt, p, c)
        else:
            loss, loss1, loss2 = train_step(cfg, model, opt, p, c)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    D.EXPOSED_WAIT = None
    elapsed = D.allreduce_scalar_max(elapsed)
    ranks_joined = D.world_size()
    if D.collectives_on():
        wire = sum(r.wire_bytes() for r in opt.grad_reducers) if overlap else sum(fb.numel * 4 for fb in opt.flat_buffers)
        exposed = D.allreduce_scalar_max(sum(a.elapsed_time(b) for a, b in waits) / max(args.steps, 1)) if waits else None
        allreduce = {"backend": torch.distributed.get_backend(), "bytes_per_step": wire,
                     "exposed_wait_ms_per_step": exposed,          # max over ranks: compute stream parked behind collectives after backward
                     "mode": ("32 MB buckets over the live parameters, overlapped with backward" if overlap
                              else "one flat all-reduce after backward"),
                     "buckets": sum(len(r.buckets) for r in opt.grad_reducers) if overlap else len(opt.flat_buffers)}
    else:
        allreduce = {"backend": None, "bytes_per_step": 0, "mode": "none (single rank)"}

    pool_paths = dict(ops.POOL_PATH_COUNTS)
    # one more step (untimed) with the pruned Chamfer's diagnostics attached: pairs evaluated and chunks scanned per query wave on the
    # step's OWN clouds (prediction vs ground truth), the imbalance the launch time follows
    chamfer_step = None
    if world == 1:
        cst = []
        with ops.hooks(chamfer_stats=cst):
            train_step(cfg, model, opt, p, c)
        torch.cuda.synchronize()
        if len(cst) >= 2:
            wc = cst[-1].float()
            half = wc.numel() // 2
            chamfer_step = {"pairs_evaluated": float(cst[-2].item()), "pairs_of_a_full_scan": 2.0 * B * Md * Md,
                            "chunks_scanned_per_wave_of_256": {
                                name: {"mean": round(w_.mean().item(), 1), "p99": w_.kthvalue(max(1, int(0.99 * w_.numel()))).values.item(), "max": w_.max().item(),
                                       "waves_with_0_16_32_64_128_to_256_chunks": [int(((w_ >= lo) & (w_ < hi)).sum()) for lo, hi in ((0, 16), (16, 32), (32, 64), (64, 128), (128, 257))]}
                                for name, w_ in (("queries = prediction", wc[:half]), ("queries = ground truth", wc[half:]))}}

    # the same step for ~6 s without a break, outside the timed region: what the chip holds once its clock / temperature have settled (the
    # timed region is 20 steps = 0.4 s), and a window long enough for a GPU-activity sampler with a period of seconds to see this run
    sustained = None
    if not args.no_extras and args.sustain > 0:
        torch.cuda.synchronize()
        D.barrier()
        n_sus, t1 = 0, time.perf_counter()
        while True:
            for _ in range(25):
                train_step(cfg, model, opt, p, c)
            n_sus += 25
            torch.cuda.synchronize()
            stop = (time.perf_counter() - t1) >= args.sustain
            if D.allreduce_scalar_max(1.0 if stop else 0.0) > 0.0:          # every rank leaves after the same number of steps
                break
        D.barrier()
        dt_sus = D.allreduce_scalar_max(time.perf_counter() - t1)
        sustained = {"seconds": round(dt_sus, 2), "steps": n_sus, "ms_per_step": dt_sus / n_sus * 1e3,
                     "points_per_s": world * B * Md * n_sus / dt_sus,
                     "note": "the timed step repeated back to back after the timed region (not `value`): sustained clocks"}

    # the same step on the fp32-input MFMA kernels, outside the timed region (single GPU only; a comparison, not `value`)
    fp32_mode = None
    gemm_mode = _lib.get_gemm_mode()
    if world == 1 and gemm_mode == "split" and not args.no_fp32_mode:
        _lib.set_gemm_mode("fp32")
        for _ in range(3):                           # the unfused pool path of this mode allocates 2 x 1.6 GB on first use
            train_step(cfg, model, opt, p, c)
        torch.cuda.synchronize()
        prof32 = []
        n32 = max(2, min(5, args.steps))
        with ops.hooks(gemm_profile=prof32):
            t1 = time.perf_counter()
            for _ in range(n32):
                train_step(cfg, model, opt, p, c)
            torch.cuda.synchronize()
            dt32 = time.perf_counter() - t1
        _lib.set_gemm_mode("split")
        mf = [(x[1], x[4].seconds()) for x in prof32 if x[4].kind == 1]
        f32 = sum(f for f, _ in mf)
        t32 = sum(t for _, t in mf)
        fp32_mode = {"ms_per_step": dt32 / n32 * 1e3, "value": B * Md * n32 / dt32, "steps": n32, "warmup": 3,
                     "roofline": {"bound": "mfma", "kernel": "vnp_gemm_kernel<*> (v_mfma_f32_32x32x2_f32)",
                                  "achieved": round(f32 / t32 / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(f32 / t32 / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), "ms_per_step": t32 / n32 * 1e3,
                                  "launches_per_step": len(mf) // n32}}

    screen_entry, degenerate_entry = None, None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * Md * args.steps / elapsed
        # every matrix-product launch of the timed steps: (variant, 2MNK, nsplit, shape, timer); timer.kind says what its dominant
        # kernel was (include/vnpcc.h: vnp_gemm_profile_next) and timer.seconds() how long THAT KERNEL ran
        recs = [(v, f, ns, shp, t.kind, t.seconds()) for v, f, ns, shp, t in prof]
        main_kind = 0 if gemm_mode == "split" else 1
        mfma = [r for r in recs if r[4] == main_kind and r[5] is not None]           # launches of the dominant kernel
        tot_f, tot_t, nlaunch = sum(r[1] for r in mfma), sum(r[5] for r in mfma), len(mfma)
        achieved = tot_f / tot_t / 1e12 if tot_t > 0 else 0.0
        agg = {}
        for v, f, ns, shp, kind, t in mfma:
            a = agg.setdefault(v, [0.0, 0.0, 0])
            a[0] += f; a[1] += t; a[2] += 1
        if args.shapes:
            per = {}
            for v, f, ns, shp, kind, t in recs:
                if t is None:
                    continue
                a = per.setdefault((v, kind, shp, ns), [0.0, 0.0, 0])
                a[0] += f; a[1] += t; a[2] += 1
            for (v, kind, shp, ns), a in sorted(per.items(), key=lambda kv: -kv[1][1]):
                print(f"  {v} kind={kind} M={shp[0]:8d} N={shp[1]:5d} K={shp[2]:8d} split={ns:4d} x{a[2] // prof_steps:2d}/step "
                      f"{a[1] / prof_steps * 1e3:7.3f} ms/step {a[0] / a[1] / 1e12:6.1f} TF/s", file=sys.stderr)
        # the screening pass of the exact arg-max pool (its own kernel, ONE fp16 product — or three bf16 products — per operand pair)
        scr = [r for r in recs if r[4] in (3, 4) and r[5] is not None]
        if scr:
            nprod = 1.0 if scr[0][4] == 3 else 3.0
            sf, st = sum(r[1] for r in scr) * nprod, sum(r[5] for r in scr)
            speak = PEAK_BF16_MFMA_TFLOPS                    # fp16 and bf16 dense MFMA peaks are equal
            screen_entry = {"kernel": ("vnp_screen_f16_xwide_kernel (384 x 256 tile, LDS-DMA staging; other shapes: vnp_screen_f16_wide_kernel / vnp_gemm_split_kernel<..., EPI = 4>): screening pass of the exact arg-max pool "
                                       "(VNLinear -> VNMaxPool), ONE v_mfma_f32_16x16x32_f16 product per operand pair on fp16 operand images"
                                       if nprod == 1.0 else
                                       "vnp_gemm_split_kernel<..., EPI = 3>: screening pass of the exact arg-max pool, three bf16 products"),
                            "bound": "mfma", "ms": round(st / len(scr) * 1e3, 4), "launches_per_step": len(scr) // max(prof_steps, 1),
                            "achieved": round(sf / st / 1e12, 1), "peak": speak, "unit": "TFLOP/s", "frac": round(sf / st / 1e12 / speak, 4),
                            "shape": list(scr[0][3]),
                            "note": "executed flops = 2 M N K x products issued, kernel time from the library's HIP events; the candidate passes "
                                    "behind it (select, sort, exact, final: gathers, 0.8 ms per step) are not matrix-pipe work and are not priced here"}
        deg = [r for r in recs if r[4] == 2 and r[5] is not None]
        if deg:
            db = sum(4.0 * (m * k + n * k + m * n) for _, _, _, (m, n, k), _, _ in deg)
            dt_ = sum(r[5] for r in deg)
            degenerate_entry = {"kernel": "vnp_gemm_rank1 / matvec / wcolsum kernels (K = 1, N = 1 or M = 1 products: streaming, no MFMA)",
                                "bound": "hbm", "ms": round(dt_ / len(deg) * 1e3, 4), "launches_per_step": len(deg) // max(prof_steps, 1),
                                "achieved": round(db / dt_ / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(db / dt_ / 1e9 / PEAK_HBM_GBS, 4),
                                "algorithmic_bytes": db / max(len(deg), 1), "note": "bytes = every operand and the result once"}
        # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction);
        # bench.py cannot run the counters on itself, so this is read from profiles/ (null when absent)
        traffic, traffic_src = None, None
        try:
            import glob
            summ = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_summary.json")))[-1]
            tr = json.load(open(summ)).get("traffic", {})
            knames = ("vnp_gemm_split_kernel", "vnp_gemm_split_wide_kernel", "vnp_gemm_split_wide_tn_kernel") if gemm_mode == "split" else ("vnp_gemm_kernel",)
            fam = [v for k, v in tr.items() if any(n in k for n in knames)]
            num = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in fam)
            den = sum(v["launches"] for v in fam)
            if den:
                traffic, traffic_src = num / den, os.path.relpath(summ, ROOT)
        except Exception:
            pass
        alg_bytes = sum(4.0 * (m * k + n * k + m * n) for _, _, _, (m, n, k), _, _ in mfma) / max(nlaunch, 1)
        if gemm_mode == "split":
            kernel = ("vnp_gemm_split_kernel<*> + vnp_gemm_split_wide_kernel<*> + vnp_gemm_split_wide_tn_kernel<*> (one arithmetic, three tile shapes: fp32 operands split exactly "
                      "into 3 bf16 terms; 6 x v_mfma_f32_16x16x32_bf16 per fp32 product block, fp32 accumulate)")
            peak = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS
        else:
            kernel, peak = "vnp_gemm_kernel<*> (v_mfma_f32_32x32x2_f32)", PEAK_FP32_MFMA_TFLOPS
        # side views (no frac: they are not rates of executed work): every matrix product of the step at its algorithmic 2MNK over the
        # time of ALL matrix-product kernels (the screened pool counts for the full product it replaces), and SURVEY 8(d)'s as-written count
        all_t = sum(r[5] for r in recs if r[5] is not None)
        all_f = sum(r[1] for r in recs if r[5] is not None)
        roofline = {"bound": "mfma", "kernel": kernel,
                    "achieved": round(achieved, 2), "peak": round(peak, 2), "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4),
                    "executed_bf16_PFLOPs": round(achieved * (SPLIT_PRODUCTS if gemm_mode == "split" else 1) / 1e3, 3),
                    "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
                    # the same launches against the OTHER roof: the decoder's K = 256 / 512 products move 5.5-6.2 GB each for 0.41 TFLOP, 75 flop per
                    # byte — the ridge point of (417 TF/s split-bf16, 5.5 TB/s deliverable) is 76 — so they are bound by both at once
                    "hbm_view": ({"traffic_GBs": round(traffic / (tot_t / max(nlaunch, 1)) / 1e9, 1), "peak": PEAK_HBM_GBS,
                                  "frac": round(traffic / (tot_t / max(nlaunch, 1)) / 1e9 / PEAK_HBM_GBS, 4),
                                  "flop_per_byte": round(tot_f / max(nlaunch, 1) / traffic, 1),
                                  "ridge_flop_per_byte": round(peak * 1e12 / (PEAK_HBM_GBS * 1e9), 1),
                                  "note": "PMC fabric bytes per launch (profiles/) / live average kernel time of this run"} if traffic else None),
                    "launches_per_step": nlaunch // max(prof_steps, 1),
                    "flop_per_launch": tot_f / max(nlaunch, 1),
                    "avg_launch_ms": tot_t / max(nlaunch, 1) * 1e3,
                    "ms_per_step": tot_t / prof_steps * 1e3,
                    "sampled_steps": prof_steps,
                    "timing": f"HIP events stamped by each launch of this kernel on its launch stream (vnp_gemm_profile_next) on every {PROF_STRIDE}th step of the "
                              f"timed region ({prof_steps} of {args.steps} steps): "
                              "kernel time only — no slab reduction, no launch gap; reproduce as sum(2MNK) / sum(duration) over the "
                              "vnp_gemm_split_kernel / vnp_gemm_split_wide_kernel / vnp_gemm_split_wide_tn_kernel rows of profiles/rNN_bench_kernel_stats.csv",
                    "note": ("achieved = EXECUTED fp32-equivalent flops (2MNK of the launches of this kernel; each is six bf16 MFMA products) / "
                             "kernel time.  Not in it: the arg-max pool's screening pass (own kernel, one fp16 product: roofline_other), the "
                             "streaming kernels of the degenerate shapes (roofline_other), slab reductions.  "
                             f"peak = bf16 dense MFMA peak ({PEAK_BF16_MFMA_TFLOPS:.0f}) / {SPLIT_PRODUCTS} products; the fp32-input MFMA peak is "
                             f"{PEAK_FP32_MFMA_TFLOPS}") if gemm_mode == "split" else "achieved = executed 2MNK flops / kernel time",
                    "by_variant": {k: {"tflops": round(a[0] / a[1] / 1e12, 2), "ms_per_step": round(a[1] / prof_steps * 1e3, 3),
                                       "launches_per_step": a[2] // max(prof_steps, 1)} for k, a in agg.items() if a[1] > 0},
                    "views": {"all_matrix_products_algorithmic": {
                                  "tflops": round(all_f / all_t / 1e12, 2) if all_t > 0 else None, "ms_per_step": all_t / prof_steps * 1e3,
                                  "note": "2MNK of EVERY matrix product of the step (the screened arg-max pool counted at the full product whose "
                                          "result it reproduces bit for bit) / kernel time of all of them: a rate of useful work, not of MFMA work"},
                              # 212 GFLOP of matrix work per completed cloud and training step for the network AS WRITTEN
                              "survey_8d": ({"flop_per_step": SURVEY_FLOP_PER_CLOUD * B,
                                             "tflops": round(SURVEY_FLOP_PER_CLOUD * B / (all_t / prof_steps) / 1e12, 2),
                                             "note": "SURVEY 8(d): 212 GF per cloud and step for the network as written / matrix-kernel time per "
                                                     "step; DESIGN.md §4 removed two thirds of the as-written products, so this is "
                                                     "reference-equivalent work, not executed work"}
                                            if (Mc == 1024 and all_t > 0) else None)}}
        # the whole step against the HBM roofline (north_star: throughput "as fraction of the HBM roofline"): fabric bytes of one step
        # (every kernel's FETCH_SIZE x 2 + WRITE_SIZE, profiles/ — bench.py cannot run counters on itself) / this run's step time
        hbm_roofline = None
        try:
            sb = json.load(open(summ)).get("step_fabric_bytes")
            if sb and gemm_mode == "split":
                hbm_roofline = {"bytes_per_step": sb, "source": os.path.relpath(summ, ROOT), "achieved": round(sb / (ms * 1e-3) / 1e9, 1),
                                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(sb / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                "bytes_per_completed_point": round(sb / (B * Md), 1),
                                "note": "PMC fabric bytes of ALL kernels of one step (recorded passes under profiles/) over this run's wall time per "
                                        "step; the step is matrix-pipe bound (roofline), so this fraction is a description, not a target: the "
                                        "HBM-bound kernels themselves run at 4.7-5.6 TB/s (roofline_other), 0.9-1.0 of the chip's measured "
                                        "mixed read/write ceiling (5.0-5.7 TB/s by box: profiles/r02_hbm_yardstick.txt, r05_hbm_yardstick.txt)"}
        except Exception:
            pass
        steps_total = args.warmup + args.steps
        pool_path = {"calls_since_start": pool_paths, "steps_since_start": steps_total,
                     "note": "VNLinear -> VNMaxPool (maxpool2) calls by path over warm-up + timed steps: `screened` = exact screened search, "
                             "`full_epilogue` = six-product pool epilogue (search skipped: back-off after a device-side fallback, or K < 256: "
                             "maxpool1 in the rows form), `fallback_flags_seen` = device-side fallbacks reported back so far"}
        out = {"metric": METRIC, "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "dtype_note": ("fp32 operands and fp32 results.  GEMM mode 'split' (default): every operand element is split EXACTLY into three bf16 "
                              "terms and six bf16 MFMA products are accumulated in fp32 — error against fp64 at or below an fp32-input MFMA's "
                              "(tests/test_gpu_gemm.py::test_split_is_fp32_grade); three documented edge cases (Inf operand -> NaN where an fp32 "
                              "product gives Inf, |x| > 3.39e38 likewise, fp32 denormals lose their low planes), exercised at layer level by "
                              "tests/test_gpu_layers.py::test_split_arithmetic_edge_cases_through_a_layer.  The fp32-input MFMA mode "
                              "(VNP_GEMM_MODE=fp32) is reported beside it as fp32_mfma_mode; it has no fused statistics / pool epilogues")
                             if gemm_mode == "split" else "fp32-input MFMA kernels (v_mfma_f32_32x32x2_f32)",
               "config": {"workload": f"vn_pointnet+vn_foldingnet train step (fwd+2xCD-L1+bwd+allreduce+Adam), B={B}/GPU, "
                                      f"{N}->{Mc}->{Md} pts, fp32 (GEMM mode: {gemm_mode})", "global_batch": B * world,
                          "parallelism": f"dp{world}", "allreduce": allreduce["mode"]},
               "points_per_s_per_gpu": value / world, "ranks_joined": ranks_joined, "allreduce": allreduce,
               "loss": {"total": loss.item(), "coarse_cd_l1": loss1.item(), "dense_cd_l1": loss2.item()},
               "roofline": roofline, "hbm_roofline": hbm_roofline, "pool_path": pool_path, "chamfer_dense_loss_in_step": chamfer_step, "sustained": sustained}
        if fp32_mode is not None:
            out["fp32_mfma_mode"] = fp32_mode
        if world == 1 and not args.no_extras:
            del model, opt
            torch.cuda.empty_cache()
            import bench_extras
            legs = [("roofline_other", lambda: roofline_other(B, N, Mc, Md) + bench_extras.roofline_other_configs()),
                    ("chamfer_op", chamfer_op_bench)]
            if not args.no_configs:
                # the other BASELINE configurations at full batch and resolution: 5 untimed + 5 timed steps each
                legs.append(("configs", lambda: bench_extras.other_configs(steps=5, warmup=5)))
            for key, fn in legs:
                try:
                    out[key] = fn()
                except Exception as e:                              # an extra leg must never hide the GPU number
                    out[key] = {"failed": f"{type(e).__name__}: {e}"}
        # kernels of the timed step itself that are not the dominant kernel, priced on what they execute
        step_extra = [e for e in (screen_entry, degenerate_entry) if e is not None]
        if step_extra:
            out["roofline_other"] = step_extra + (out["roofline_other"] if isinstance(out.get("roofline_other"), list) else [])
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(min(16, os.cpu_count() or 1))      # the GPU box gives 16 cores per GPU
            except Exception as e:                                  # the baseline must never hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "points/s", "cores": min(16, os.cpu_count() or 1),
                                       "cores_available": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(out))
    D.barrier()
    D.shutdown()


if __name__ == "__main__":
    main()
