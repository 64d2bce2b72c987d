"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm,
"gloo" on CPU for the tests).  The reference has no multi-GPU code; what the path needs is small:

  * every rank holds a full replica and processes its own shard of the batch (samples are independent; train-mode
    BatchNorm statistics stay per replica, i.e. the single-GPU semantics at the per-GPU batch size);
  * once per step the flat gradient buffer of ``FusedAdam`` (optim.py) is averaged with ONE all-reduce
    (22.5 M fp32 = 90 MB for vn_pointnet+vn_foldingnet).  Parameters that never get a gradient
    (``maxpool*.map_to_dir``) live in the same buffer as zeros, so every rank reduces the same static layout and
    nothing like DDP's ``find_unused_parameters`` is needed;
  * rank 0's initial weights are broadcast so replicas start identical.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them).
    Returns (rank, local_rank, world_size).  A single process needs no group."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def barrier():
    if world_size() > 1:
        dist.barrier()


def shard_batch(global_batch: int):
    """Weak-scaling / sharding rule: contiguous, equal shards of the global batch (B_global = per_gpu * world)."""
    w, r = world_size(), get_rank()
    if global_batch % w:
        raise ValueError(f"global batch {global_batch} is not divisible by the world size {w}")
    per = global_batch // w
    return r * per, (r + 1) * per


def broadcast_tensors(tensors, src=0):
    if world_size() > 1:
        for t in tensors:
            dist.broadcast(t, src=src)


def broadcast_model(model, src=0):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if world_size() > 1:
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=src)


def allreduce_mean_(flat: torch.Tensor):
    """In-place average of a flat buffer over all ranks (RCCL ncclAvg; SUM + scale on gloo)."""
    w = world_size()
    if w == 1:
        return flat
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / w)
    return flat


def allreduce_gradients(optimizer_or_params):
    """Average gradients across ranks: one collective per flat buffer when given a FusedAdam, else per parameter
    (skipping ``grad is None`` consistently — the set is static for this model family)."""
    if world_size() == 1:
        return
    flats = getattr(optimizer_or_params, "flat_buffers", None)
    if flats is not None:
        for fb in flats:
            fb.check_views()
            allreduce_mean_(fb.flat_grad)
        return
    for p in optimizer_or_params:
        if p.grad is not None:
            allreduce_mean_(p.grad)


def allreduce_scalar_max(x: float) -> float:
    if world_size() == 1:
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()
