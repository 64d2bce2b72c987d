"""Two ranks on ONE MI355X (RCCL refuses two ranks on one device): the real model, FusedAdam and the backward-overlapped
bucketed gradient averaging (distributed.OverlappedGradReducer: hooks on the autograd device thread, bucket order, never-
used parameters) against a single-process run that computes each shard's gradients separately and averages them by hand.
The transport is injected: each bucket is staged through the host and averaged with a CPU gloo all-reduce (this build's
gloo hangs on repeated device-tensor collectives, and RCCL itself cannot be exercised on a one-GPU box).  Everything else
is exactly what `bench.py --gpus N` runs."""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cfg():
    return SimpleNamespace(num_coarse=64, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cuda:0",
                           coarse_loss="cd", dcd_opts={"alpha": 200, "lambda": 0.5})


def _data():
    g = torch.Generator().manual_seed(5)
    return (torch.rand(4, 128, 3, generator=g) - 0.5), (torch.rand(4, 1024, 3, generator=g) - 0.5)


def _model(cfg):
    from vn_pointcloudcompletion_amd.model import PCNNet
    torch.manual_seed(3)
    m = PCNNet(cfg, "vn_pointnet", "vn_foldingnet")
    m.decoder.num_dense = cfg.num_coarse * 16
    m.train()
    return m


def _worker(rank, world, port, q):
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)                  # a stuck rank reports where and dies instead of hanging the box
    try:
        os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        from vn_pointcloudcompletion_amd import distributed as D
        from vn_pointcloudcompletion_amd.optim import FusedAdam
        from vn_pointcloudcompletion_amd.train import train_step
        torch.cuda.set_device(0)
        D.init_from_env(backend="gloo")
        cfg = _cfg()
        model = _model(cfg)
        # same seed -> identical replicas (no device-tensor broadcast over gloo)

        def host_staged_average(flat):
            h = flat.detach().cpu()
            dist.all_reduce(h)
            flat.copy_((h / world).to(flat.device))
            return None
        opt = FusedAdam(model.parameters(), lr=0.0)      # lr 0: every step sees the same weights -> the same gradients
        red = opt.enable_overlapped_allreduce(bucket_bytes=16 << 20, transport=host_staged_average)[0]
        p, c = _data()
        lo, hi = D.shard_batch(4)
        p, c = p[lo:hi].cuda(), c[lo:hi].cuda()
        launched, grads = [], []
        for _ in range(3):
            opt.zero_grad()
            from vn_pointcloudcompletion_amd.train import compute_losses
            coarse, fine = model(p)
            loss, _, _ = compute_losses(cfg, coarse, fine, c)
            loss.backward()
            launched.append(red.launched)
            D.allreduce_gradients(opt)
            grads.append(opt.flat_buffers[0].flat_grad.cpu().numpy().copy())      # numpy: pickled by value (no fd passing)
            opt.step()
        # two more steps in which RANK 1 ALONE produces no gradient for one parameter: its slice must hold rank 0's
        # contribution / world after BOTH steps — a slice the gather left "known zero" on rank 1 holds the average after the collective
        # and has to be zero-filled again before the next one, or the stale average is reduced a second time (1.5x at step two)
        fb = opt.flat_buffers[0]
        victim = model.decoder.final_conv[0].batchnorm.bn.weight
        vi = next(i for i, q_ in enumerate(fb.params) if q_ is victim)
        lo = fb.offsets[vi]
        skipped = []
        for _ in range(2):
            opt.zero_grad()
            victim.requires_grad_(rank != 1)
            coarse, fine = model(p)
            loss, _, _ = compute_losses(cfg, coarse, fine, c)
            loss.backward()
            victim.requires_grad_(True)
            D.allreduce_gradients(opt)
            skipped.append(fb.flat_grad[lo:lo + victim.numel()].cpu().numpy().copy())
            opt.step()
        torch.cuda.synchronize()
        q.put((rank, "ok", grads, launched, len(red.buckets), skipped))
        D.barrier()
        dist.destroy_process_group()
    except Exception as e:                                            # report instead of hanging the parent
        import traceback
        q.put((rank, "error", f"{type(e).__name__}: {e}\n{traceback.format_exc()}", None, None, None))


def test_two_ranks_overlapped_allreduce_matches_manual_average(lib_built):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    import queue
    import time
    out, t0 = [], time.time()
    while len(out) < world and time.time() - t0 < 90:
        try:
            out.append(q.get(timeout=2))
        except queue.Empty:
            if not any(pr.is_alive() for pr in procs) and q.empty():
                break
    for pr in procs:
        pr.join(10)
        if pr.is_alive():
            pr.terminate()
    assert len(out) == world, f"only {len(out)} of {world} ranks reported (exit codes {[pr.exitcode for pr in procs]})"
    out.sort(key=lambda o: o[0])
    errs = [o for o in out if o[1] != "ok"]
    if errs and "gloo" in errs[0][2].lower() and "cuda" in errs[0][2].lower():
        pytest.skip("this gloo build has no device-tensor support: " + errs[0][2].splitlines()[0])
    assert not errs, errs[0][2]
    assert all(pr.exitcode == 0 for pr in procs)
    import numpy as np
    for g0, g1 in zip(out[0][2], out[1][2]):
        assert np.array_equal(g0, g1)                                    # both replicas hold the same averaged gradient
    # a parameter without a gradient on rank 1 for two steps: both ranks hold the same slice, and the second step's equals the first's
    for r in range(world):
        a, b = out[r][5]
        assert np.array_equal(out[0][5][0], out[1][5][0]) and np.abs(a).max() > 0
        assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max(), (r, np.abs(a - b).max(), np.abs(a).max())
    assert out[0][4] >= 3                                             # several buckets at this bucket size
    assert out[0][3][0] == 0 and all(n >= 1 for n in out[0][3][1:])   # from step 2 on, buckets leave during backward

    # single-process reference: per-shard gradients (per-replica BatchNorm statistics), averaged by hand
    from vn_pointcloudcompletion_amd.optim import FusedAdam
    from vn_pointcloudcompletion_amd.train import compute_losses
    cfg = _cfg()
    model = _model(cfg)
    opt = FusedAdam(model.parameters(), lr=0.0)
    fb = opt.flat_buffers[0]
    p, c = _data()
    acc = torch.zeros_like(fb.flat_grad)
    for r in range(world):
        opt.zero_grad()
        coarse, fine = model(p[2 * r:2 * r + 2].cuda())
        loss, _, _ = compute_losses(cfg, coarse, fine, c[2 * r:2 * r + 2].cuda())
        loss.backward()
        fb.check_views()                                                 # gathered gradients: bring them to the flat buffer first
        acc += fb.flat_grad / world
    ref = acc.cpu()
    scale = ref.abs().max().item()
    for g in out[0][2]:
        err = (torch.as_tensor(g) - ref).abs().max().item()
        assert err < 1e-5 * scale, (err, scale)


def test_rccl_code_path_with_one_forced_rank(lib_built):
    """RCCL refuses two ranks on one device, so the real transport is exercised with ONE rank and VNP_FORCE_COLLECTIVES=1:
    process-group creation with the nccl backend, weight broadcast, the bucketed ncclAvg all-reduces launched asynchronously
    from the autograd hooks, barrier and scalar max — through bench.py itself, in a fresh process (small batch)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VNP_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "2", "--batch", "2",
                        "--no-cpu-baseline", "--no-fp32-mode"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["config"]["allreduce"].startswith("32 MB buckets") and out["n_gpus"] == 1
    assert out["value"] > 0 and all(v == v for v in out["loss"].values())          # finite losses
