"""Training loop with the reference's semantics (/root/reference/train.py:45-279) on the HIP path, plus the
data-parallel step the reference lacks (one process per GPU, RCCL gradient all-reduce; distributed.py).

Differences from the reference, all on purpose:
  * ``FusedAdam`` (flat buffers, one kernel) instead of torch.optim.Adam — same update rule and state_dict format;
  * losses are accumulated on the device and read back once per logging interval instead of 6 ``.item()`` syncs per
    iteration (train.py:168,175-176,184-186);
  * scalars go to ``<exp_dir>/{train,val}/scalars.jsonl`` (tensorboardX is not required); the per-epoch PNG
    (visualization/, matplotlib) is not produced;
  * under torch.distributed every rank loads and trains on its shard of each global batch (ShardedBatchSampler: uneven
    shards are loss-weighted so the averaged gradient is the global-batch gradient up to the BatchNorm statistics, which each rank
    takes over its own shard — no SyncBN), the per-epoch training loss is the weighted sum over all ranks, validation is sharded
    the same way behind a broadcast of rank 0's BatchNorm running statistics and its sums all-reduced, rank 0 logs and checkpoints;
  * the shuffle order of epoch e is a function of (seed, e), so a resumed run continues the uninterrupted run's data order
    instead of replaying epoch 0's (the reference's generator is unseeded);
  * on --resume the StepLR schedule continues at the absolute epoch (lr at epoch e is that of an uninterrupted run); the
    reference builds a fresh StepLR after loading the optimizer (train.py:93), which restarts the 50-epoch count at the resume point.
"""
from __future__ import annotations

import logging
import os
from copy import deepcopy

import torch
from torch.utils.data.dataloader import DataLoader

from . import distributed as D
from .dataset import make_dataset
from .experiments import ScalarWriter, get_num_params_total
from .loss import calc_dcd, cd_loss_L1, emd_loss
from .metric import l1_cd
from .model import PCNNet
from .optim import FusedAdam
from .transforms import Rotate, RotateAxisAngle, random_rotations

log = logging.getLogger("train")
log_dataset = logging.getLogger("dataset")


def make_rotation(kind, batch, device):
    """train.py:130-134."""
    if kind == "z":
        return RotateAxisAngle(angle=torch.rand(batch) * 360, axis="Z", degrees=True).to(device)
    if kind == "so3":
        return Rotate(R=random_rotations(batch)).to(device)
    return None


def coarse_loss(config, coarse_pred, c):
    """train.py:146-155: the loss on the coarse cloud."""
    if config.coarse_loss == "cd":
        return cd_loss_L1(coarse_pred, c)
    if config.coarse_loss == "emd":
        return emd_loss(coarse_pred, c[:, :1024, :])
    if config.coarse_loss == "dcd":
        opts = config.dcd_opts
        t_alpha = opts["alpha"] if isinstance(opts, dict) else opts.alpha
        n_lambda = opts["lambda"] if isinstance(opts, dict) else getattr(opts, "lambda")
        loss1, _, _ = calc_dcd(coarse_pred, c, alpha=t_alpha, n_lambda=n_lambda)
        return loss1.mean()
    raise ValueError("Not implemented loss {}".format(config.coarse_loss))


def compute_losses(config, coarse_pred, dense_pred, c, loss1=None):
    """train.py:146-167: returns (loss, loss1, loss2).  ``loss1``: the coarse loss when it has been computed already (train_step starts it
    beside the decoder)."""
    if loss1 is None:
        loss1 = coarse_loss(config, coarse_pred, c)
    if config.only_coarse:
        return loss1, loss1, torch.zeros((), device=loss1.device)
    loss2 = cd_loss_L1(dense_pred, c)
    return loss1 + loss2, loss1, loss2


def make_scheduler(optimizer, start_epoch=0):
    """train.py:93 — StepLR(step_size=50, gamma=0.8) — positioned at ``start_epoch`` WITHOUT touching the learning rate:
    ``optimizer.load_state_dict`` has already restored the decayed ``lr`` (and ``initial_lr``), so replaying the schedule would
    decay it a second time.  Decay boundaries stay at absolute multiples of 50 epochs, as in an uninterrupted run."""
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=50, gamma=0.8, last_epoch=-1)
    if start_epoch:
        scheduler.last_epoch = start_epoch
        scheduler._last_lr = [g["lr"] for g in optimizer.param_groups]
    return scheduler


_ONES = {}


def _one_like(loss):
    """The cotangent 1.0 of a 0-dim loss, made once per (device, dtype) instead of a ones_like launch per step."""
    key = (loss.device, loss.dtype)
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones((), device=loss.device, dtype=loss.dtype)
    return t


_LOSS_STREAMS = {}
# which coarse losses are started on a side HIP stream beside the decoder's forward.  Measured on MI355X, alternating blocks of steps in one
# process (scripts/experiments/ab_coarse_ahead.py): EMD — 41 dependent, latency-bound launches — 24.58 -> 23.93 ms per step of config 4;
# the Chamfer losses — one VALU-bound search that fills the chip — LOSE when interleaved with the decoder (config 2 17.16 -> 17.60 ms,
# config 3 17.42 -> 17.61): they stay on the main stream
COARSE_LOSS_AHEAD = ("emd",)


def train_step(config, model, optimizer, p, c, trot=None, loss_weight=1.0):
    """One iteration of the hot loop (train.py:140-173): zero_grad, forward, losses, backward, gradient all-reduce
    (world > 1), Adam.  Returns the three loss tensors (on device, no host sync).  ``loss_weight`` scales the gradient of an
    uneven data-parallel shard (ShardedBatchSampler.weight); the returned losses are unscaled.
    The coarse loss depends on the encoder's output alone: an EMD loss (41 dependent launches: latency) is started on a side stream the
    moment that output exists (PCNNet's ``on_coarse`` hook) and runs beside the decoder; autograd runs its backward on the same side
    stream (COARSE_LOSS_AHEAD)."""
    optimizer.zero_grad()
    ahead = config.coarse_loss in COARSE_LOSS_AHEAD and p.is_cuda and isinstance(model, PCNNet) and not model.only_coarse
    if ahead:
        main = torch.cuda.current_stream(p.device)
        side = _LOSS_STREAMS.get(p.device.index)
        if side is None:
            side = _LOSS_STREAMS[p.device.index] = torch.cuda.Stream(device=p.device)
        box = []

        def on_coarse(coarse_pred):
            side.wait_stream(main)                       # the encoder's launches produced coarse_pred on the main stream
            coarse_pred.record_stream(side)
            c.record_stream(side)
            with torch.cuda.stream(side):
                box.append(coarse_loss(config, coarse_pred, c))
        coarse_pred, dense_pred = model(p, trot, on_coarse=on_coarse)
        main.wait_stream(side)
        if box:
            box[0].record_stream(main)
        loss, loss1, loss2 = compute_losses(config, coarse_pred, dense_pred, c, loss1=box[0] if box else None)
    else:
        coarse_pred, dense_pred = model(p, trot)
        loss, loss1, loss2 = compute_losses(config, coarse_pred, dense_pred, c)
    (loss if loss_weight == 1.0 else loss * loss_weight).backward(gradient=_one_like(loss))
    D.allreduce_gradients(optimizer)
    optimizer.step()
    return loss.detach(), loss1.detach(), loss2.detach()


def settle_gc():
    """Python's cyclic collector walks every tracked object of the process on a full collection: ~270 k objects once torch, the model, the
    optimizer and the first step's autograd graph exist — a 70-90 ms host stall during which no kernel is launched (measured on config 3:
    one at step ~50 of every process, scripts/experiments/step_times_cfg3.py).  After set-up everything long-lived moves to the permanent
    generation (gc.freeze), so later collections only look at what a step creates; called after the first training step."""
    import gc
    gc.collect()
    gc.freeze()


def train(config, args):
    rank, world = D.get_rank(), D.world_size()
    is_main = rank == 0
    model_dir = os.path.join(config.exp_dir, "models")
    optim_dir = os.path.join(config.exp_dir, "optimizer")
    train_writer = ScalarWriter(os.path.join(config.exp_dir, "train")) if is_main else None
    val_writer = ScalarWriter(os.path.join(config.exp_dir, "val")) if is_main else None

    log_dataset.info("Loading Data...")
    train_dataset = make_dataset(config, "train", config.category)
    val_dataset = make_dataset(config, "valid", config.category)
    seed = int(getattr(config, "seed", 0))
    train_sampler = D.ShardedBatchSampler(len(train_dataset), config.batch_size, shuffle=True, seed=seed, mode="train")
    val_sampler = D.ShardedBatchSampler(len(val_dataset), config.batch_size, shuffle=False, mode="eval")
    if world > 1 and config.batch_size % world:
        log.warning(f"batch_size {config.batch_size} is not a multiple of the world size {world}: shards are uneven "
                    f"({config.batch_size // world} or {config.batch_size // world + 1} samples per rank, loss-weighted)"
                    + ("; batches smaller than the world size are replicated on every rank" if config.batch_size < world else ""))
    log_dataset.info("Dataset loaded!")

    model = PCNNet(config, enc_type=config.enc_type, dec_type=config.dec_type)
    if config.freeze_decoder:
        for param in model.decoder.parameters():
            param.requires_grad = False
    if config.freeze_encoder:
        for param in model.encoder.parameters():
            param.requires_grad = False
    D.broadcast_model(model)

    optimizer = FusedAdam(model.parameters(), lr=config.lr, betas=(0.9, 0.999))
    if D.collectives_on() and os.environ.get("VNP_OVERLAP_ALLREDUCE", "1") != "0":
        optimizer.enable_overlapped_allreduce()      # VNP_OVERLAP_ALLREDUCE=0: one all-reduce after backward instead
    start_epoch, best_cd_l1, best_epoch_l1 = 0, 1e8, -1
    if args.resume:
        model_path = os.path.join(config.exp_dir, "models/model_last.pth")
        optim_path = os.path.join(config.exp_dir, "optimizer/optim_last.pth")
        if os.path.exists(model_path) and os.path.exists(optim_path):
            model.load_state_dict(torch.load(model_path, map_location=config.device))
            optim_dict = torch.load(optim_path, map_location=config.device)
            optimizer.load_state_dict(optim_dict["optim_state_dict"])
            start_epoch = optim_dict["epoch"] + 1
            best_cd_l1, best_epoch_l1 = optim_dict["best_metrics"], optim_dict["best_epoch"]
            log.info(f"[RESUME INFO] resume ckpts @ {start_epoch - 1} epoch( best_metrics = {str(best_cd_l1 * 1e3):s})")
        else:
            log.info(f"Tried to resume training from experiment: {config.exp_dir}, however, model.pth or optim.pth "
                     "not existant. Train from start")
    else:
        log.info(f"Start a brand new experiment: {config.run_name}")
    scheduler = make_scheduler(optimizer, start_epoch)
    # worker seeds derive from this generator: tie them to (seed, start_epoch) so a resumed run does not replay epoch 0's
    train_loader = DataLoader(train_dataset, batch_sampler=train_sampler, num_workers=config.num_workers,
                              generator=torch.Generator().manual_seed(seed + start_epoch))
    val_loader = DataLoader(val_dataset, batch_sampler=val_sampler, num_workers=config.num_workers)

    n_batches = len(train_loader)
    step = max(n_batches // config.log_frequency, 1)
    log.info(f"Model total params: {get_num_params_total(model)}")
    log.info(f"Producing coarse only: {config.only_coarse}")
    log.info(f"Producing num of coarse points: {config.num_coarse}")

    for epoch in range(start_epoch, config.max_epochs + 1):
        model.train()
        train_sampler.set_epoch(epoch)
        sums = torch.zeros(3, device=config.device)          # total, coarse, dense — accumulated on the device
        for i, (p, c) in enumerate(train_loader):             # this rank's shard of global batch i
            p, c = p.to(config.device), c.to(config.device)
            trot = make_rotation(config.rotation, p.shape[0], config.device)
            if trot is not None:
                p, c = trot.transform_points(p), trot.transform_points(c)
            wgt = train_sampler.weight(i)
            loss, loss1, loss2 = train_step(config, model, optimizer, p, c, trot, wgt)
            if epoch == start_epoch and i == 0:
                settle_gc()                                   # model, optimizer state and the step's long-lived objects exist now
            sums += torch.stack([loss, loss1, loss2]) * (1.0 if wgt is None else wgt)
            if is_main and (i + 1) % step == 0:
                l, l1, l2 = loss.item(), loss1.item(), loss2.item()
                log.info("Training Epoch [{:03d}/{:03d}] - Iteration [{:03d}/{:03d}]: coarse loss = {:.6f}, dense loss = {:.6f}, "
                         "total loss = {:.6f}, lr = {:.6f}".format(epoch, config.max_epochs, i + 1, n_batches, l1 * 1e3, l2 * 1e3,
                                                                    l * 1e3, scheduler.get_last_lr()[0]))
                ts = epoch * n_batches + i
                train_writer.add_scalar("Loss/Batch/Coarse", l1, ts)
                train_writer.add_scalar("Loss/Batch/Dense", l2, ts)
                train_writer.add_scalar("Loss/Batch/Total", l, ts)
        scheduler.step()
        if world > 1:
            D.allreduce_sum_(sums)          # sum over ranks of weight x shard mean = world x the global-batch mean (weights average 1)
            sums /= world
        tot, co, de = (sums / max(n_batches, 1)).tolist()
        if is_main:
            log.info("Training Epoch [{:03d}/{:03d}]: Coarse Loss = {:.6f}, Dense Loss = {:.6f}, Total Loss = {:.6f}".format(
                epoch, config.max_epochs, co * 1e3, de * 1e3, tot * 1e3))
            train_writer.add_scalar("Loss/Epoch/Coarse", co * 1e3, epoch)
            train_writer.add_scalar("Loss/Epoch/Dense", de * 1e3, epoch)
            train_writer.add_scalar("Loss/Epoch/Total", tot * 1e3, epoch)

        # evaluation (train.py:198-250): every rank evaluates its shard of each validation batch and the per-sample sums are
        # all-reduced once (no rank idles in a barrier while rank 0 walks the whole validation set).  The all-reduced step keeps the
        # PARAMETERS of the replicas identical, not their BatchNorm running statistics (each rank updates them from its own shard):
        # rank 0's buffers — the ones model_last / model_best.pth will hold — are broadcast first, so the reduced metric and the
        # best-checkpoint choice describe the model that is saved
        D.sync_buffers(model)
        model.eval()
        val = torch.zeros(2, device=config.device)
        with torch.no_grad():
            for p, c in val_loader:
                p, c = p.to(config.device), c.to(config.device)
                trot = make_rotation(config.val_rotation, p.shape[0], config.device)
                if trot is not None:
                    p, c = trot.transform_points(p), trot.transform_points(c)
                coarse_pred, dense_pred = model(p, trot)
                val[0] += l1_cd(coarse_pred, c)
                if not config.only_coarse:
                    val[1] += l1_cd(dense_pred, c)
        D.allreduce_sum_(val)
        if is_main:
            vc, vd = (val / len(val_dataset)).tolist()
            vt = vc if config.only_coarse else vc + vd
            val_writer.add_scalar("Loss/Epoch/Coarse", vc * 1e3, epoch)
            val_writer.add_scalar("Loss/Epoch/Dense", vd * 1e3, epoch)
            val_writer.add_scalar("Loss/Epoch/Total", vt * 1e3, epoch)
            log.info("Validate Epoch [{:03d}/{:03d}]: Coarse Loss = {:.6f}, Dense Loss = {:.6f}, Total Loss = {:.6f}".format(
                epoch, config.max_epochs, vc * 1e3, vd * 1e3, vt * 1e3))
            ckpt = {"epoch": epoch, "optim_state_dict": None, "best_metrics": best_cd_l1, "best_epoch": best_epoch_l1}
            if vt < best_cd_l1:
                best_epoch_l1, best_cd_l1 = epoch, vt
                torch.save(model.state_dict(), os.path.join(model_dir, "model_best.pth"))
                log.info(f"Save checkpoint at {os.path.join(model_dir, 'model_best.pth')}")
                ckpt.update(optim_state_dict=deepcopy(optimizer.state_dict()), best_metrics=best_cd_l1, best_epoch=best_epoch_l1)
                torch.save(ckpt, os.path.join(optim_dir, "optim_best.pth"))
            torch.save(model.state_dict(), os.path.join(model_dir, "model_last.pth"))
            log.info(f"Save checkpoint at {os.path.join(model_dir, 'model_last.pth')}")
            ckpt.update(optim_state_dict=deepcopy(optimizer.state_dict()), best_metrics=best_cd_l1, best_epoch=best_epoch_l1)
            torch.save(ckpt, os.path.join(optim_dir, "optim_last.pth"))
        D.barrier()
    if is_main:
        log.info("Best l1 cd model in epoch {}, the minimum l1 cd is {}".format(best_epoch_l1, best_cd_l1 * 1e3))
        train_writer.close()
        val_writer.close()
