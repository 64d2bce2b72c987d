"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm,
"gloo" on CPU for the tests).  The reference has no multi-GPU code; what the path needs is small:

  * every rank holds a full replica and processes its own shard of the batch (samples are independent; train-mode
    BatchNorm statistics stay per replica, i.e. the single-GPU semantics at the per-GPU batch size);
  * once per step the flat gradient buffer of ``FusedAdam`` (optim.py) is averaged (22.5 M fp32 = 90 MB for
    vn_pointnet+vn_foldingnet): either with ONE all-reduce after backward, or — ``OverlappedGradReducer`` — as a few large
    contiguous buckets of that buffer, each all-reduced asynchronously as soon as backward has produced its last
    gradient, so that the exchange of the decoder / coarse-head gradients (60 % of the bytes) runs over xGMI while the
    encoder's backward GEMMs are still computing.  xGMI rings are per-link bound, so buckets are few and large
    (default 32 MB).  Parameters that never get a gradient (``maxpool*.map_to_dir``) live in the same buffer as zeros, so
    every rank reduces the same static layout and nothing like DDP's ``find_unused_parameters`` is needed;
  * rank 0's initial weights are broadcast so replicas start identical.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
import torch.utils.data


# VNP_FORCE_COLLECTIVES=1: create the process group and issue every collective even when there is a single rank — the only way
# to exercise the RCCL code path (communicator, ncclAvg, asynchronous all-reduce from autograd hooks) on a one-GPU box
FORCE_COLLECTIVES = os.environ.get("VNP_FORCE_COLLECTIVES") == "1"


def collectives_on():
    """True when gradient / scalar collectives have to be issued (more than one rank, or forced for testing)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


EXPOSED_WAIT = None


def init_from_env(backend=None, timeout_s=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them).
    Returns (rank, local_rank, world_size).  A single process needs no group.
    The rendezvous and every collective carry an explicit timeout (``timeout_s``, else VNP_DIST_TIMEOUT_S, else 300 s): when a rank never
    joins, or drops out of a collective, the others raise instead of waiting for torch's default (10-30 minutes)."""
    rank, local_rank, world = env_world()
    if (world > 1 or FORCE_COLLECTIVES) and not dist.is_initialized():
        import datetime
        if torch.cuda.is_available():
            from . import _lib
            _lib.build()             # a stale library is compiled BEFORE the rendezvous: a cold hipcc run must not eat a collective's timeout
            _lib.lib()
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        if timeout_s is None:
            timeout_s = float(os.environ.get("VNP_DIST_TIMEOUT_S", "300"))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def barrier():
    if collectives_on():
        dist.barrier()


def shutdown():
    """Destroy the process group (if any) before the interpreter exits."""
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def shard_batch(global_batch: int):
    """Weak-scaling / sharding rule: contiguous, equal shards of the global batch (B_global = per_gpu * world)."""
    w, r = world_size(), get_rank()
    if global_batch % w:
        raise ValueError(f"global batch {global_batch} is not divisible by the world size {w}")
    per = global_batch // w
    return r * per, (r + 1) * per


class ShardedBatchSampler(torch.utils.data.Sampler):
    """Batch sampler of the data-parallel loops: the GLOBAL batches are those of a single-process run (``batch_size`` indices
    in an order derived from ``seed + epoch``, identical on every rank), and each rank yields only ITS contiguous slice of
    each global batch — so a rank reads and decodes 1/world of the data instead of the whole batch.

    A global batch of b samples is split as evenly as b allows (rank r gets [r*b//world, (r+1)*b//world): the reference's
    batch_size 12 on 8 GPUs gives shards of 1 or 2 samples); ``weight(i)`` is the factor a rank must apply to its mean loss of
    batch i so that the all-reduce AVERAGE of the gradients equals the gradient of the global-batch mean — for everything in the
    network that treats samples independently.  Train-mode VNBatchNorm does not: its statistics are taken per SHARD (there is no
    SyncBN), so with shards of 1-2 samples (batch_size 12 on 8 GPUs) training numerics differ from the single-process run whose
    BatchNorm saw all 12; bench.py's weak scaling (32 samples per GPU) keeps the single-GPU statistics.  When a batch has
    fewer samples than ranks (a short final batch), ``mode='train'`` replicates it on every rank (weight 1: same step on all
    ranks, the collectives stay aligned) and ``mode='eval'`` gives it to the first ranks only (no collectives inside an
    evaluation loop, sums are all-reduced once at the end).
    """

    def __init__(self, n, batch_size, shuffle, seed=0, rank=None, world=None, mode="train"):
        self.n, self.batch_size, self.shuffle, self.seed, self.mode = n, batch_size, shuffle, seed, mode
        self.rank = get_rank() if rank is None else rank
        self.world = world_size() if world is None else world
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def global_batches(self):
        order = (torch.randperm(self.n, generator=torch.Generator().manual_seed(self.seed + self.epoch)) if self.shuffle
                 else torch.arange(self.n)).tolist()
        return [order[s:s + self.batch_size] for s in range(0, self.n, self.batch_size)]

    def _slice(self, b):
        return self.rank * b // self.world, (self.rank + 1) * b // self.world

    def weight(self, i):
        b = min(self.batch_size, self.n - i * self.batch_size)
        if b < self.world:
            return 1.0
        lo, hi = self._slice(b)
        return (hi - lo) * self.world / b

    def __iter__(self):
        for idx in self.global_batches():
            b = len(idx)
            if b < self.world:
                if self.mode == "train":
                    yield idx
                elif self.rank < b:
                    yield [idx[self.rank]]
                continue
            lo, hi = self._slice(b)
            yield idx[lo:hi]

    def __len__(self):
        nb = (self.n + self.batch_size - 1) // self.batch_size
        if self.mode != "train":
            last = self.n - (nb - 1) * self.batch_size
            if last < self.world and self.rank >= last:
                return nb - 1
        return nb


def allreduce_sum_(t: torch.Tensor):
    """In-place SUM of a small tensor over all ranks (evaluation totals)."""
    if collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_tensors(tensors, src=0):
    if collectives_on():
        for t in tensors:
            dist.broadcast(t, src=src)


def broadcast_model(model, src=0):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if collectives_on():
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=src)


def sync_buffers(model, src=0):
    """Make every replica's BUFFERS (BatchNorm running statistics, num_batches_tracked) rank ``src``'s.  The gradient all-reduce keeps
    the parameters identical; the running statistics are updated by each rank from its own shard and drift apart, so before a sharded
    evaluation — and before rank 0 saves a checkpoint that the reported metric is supposed to describe — they are made equal again."""
    if collectives_on():
        with torch.no_grad():
            for t in model.buffers():
                dist.broadcast(t.data, src=src)


def allreduce_mean_(flat: torch.Tensor):
    """In-place average of a flat buffer over all ranks (RCCL ncclAvg; SUM + scale on gloo)."""
    w = world_size()
    if not collectives_on():
        return flat
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / w)
    return flat


class OverlappedGradReducer:
    """Bucketed, backward-overlapped averaging of one FlatBuffers gradient buffer.

    Buckets are contiguous element ranges of the flat buffer, cut from its END (parameters are registered in forward
    order, so backward completes them roughly from the end) at >= ``bucket_bytes`` each.  Every parameter carries a
    post-accumulate-grad hook; when the last live parameter of a bucket has its gradient, and every later bucket has
    already been launched (collectives are issued in the same fixed order on every rank), the bucket's slice is
    all-reduced with ``async_op=True``: RCCL's stream waits for the compute queued so far, later compute does not wait
    for RCCL.  ``finish()`` launches whatever is left and waits.

    Parameters whose hook did not fire in the first backward never receive a gradient (the set is static for this model
    family: the arg-max pools' ``map_to_dir``, and in ``vn_pointr`` the 26 M parameters of modules ``forward`` never runs,
    SURVEY.md §8e).  After the first step the buckets are re-cut over the LIVE parameters only: never-live ranges leave the
    wire (config 5: 130 MB -> 26 MB per step; config 2: 90 -> 72 MB), except gaps below ``gap_bytes`` inside a bucket, which
    travel as zeros because one larger collective beats two small ones.  The first step ships everything.  If a never-live
    parameter later does receive a gradient, or any parameter receives one after its bucket was sent, ``finish`` completes the
    step's collectives (so the other ranks are not left waiting inside theirs) and then raises.
    """

    def __init__(self, fb, bucket_bytes=32 << 20, transport=None, gap_bytes=64 << 10):
        """``transport(flat_slice) -> work-or-None`` replaces the default asynchronous all-reduce (tests: a host-staged
        collective); it must leave the AVERAGE in ``flat_slice`` once the returned work (if any) has been waited for."""
        self.fb = fb
        self.transport = transport
        self.bucket_bytes = bucket_bytes
        self.gap_bytes = gap_bytes
        self._idx = {id(q): i for i, q in enumerate(fb.params)}
        self.live = None                        # set of id(param) that fired in the first backward
        self._cut_buckets()
        self.handles = [p.register_post_accumulate_grad_hook(self._hook) for p in fb.params]
        self._begin()

    def _cut_buckets(self):
        """(first param index, last+1, element lo, element hi) in launch order, over the live parameters (all of them until
        the first backward has shown which are live)."""
        fb = self.fb
        n = len(fb.params)
        ends = list(fb.offsets[1:]) + [fb.numel]
        alive = [self.live is None or id(p) in self.live for p in fb.params]
        self.buckets = []
        i = n
        while i > 0:
            if not alive[i - 1]:
                i -= 1
                continue
            hi_i = lo_i = i                     # bucket = params [lo_i, hi_i), grown downwards from a live parameter
            lo_i -= 1
            j = lo_i
            while j > 0 and (ends[hi_i - 1] - fb.offsets[lo_i]) * 4 < self.bucket_bytes:
                j -= 1
                if alive[j]:
                    lo_i = j                    # dead parameters between j and the old lo_i ride along (gap <= gap_bytes)
                elif (fb.offsets[lo_i] - fb.offsets[j]) * 4 > self.gap_bytes:
                    break                       # a large never-live range: cut here, it stays off the wire
            self.buckets.append((lo_i, hi_i, fb.offsets[lo_i], ends[hi_i - 1]))
            i = lo_i
        self.bucket_of = {}
        for b, (lo_i, hi_i, _, _) in enumerate(self.buckets):
            for k in range(lo_i, hi_i):
                self.bucket_of[id(fb.params[k])] = b

    def wire_bytes(self):
        """Bytes each rank contributes to the all-reduces of one step with the current bucket layout."""
        return sum((hi - lo) * 4 for _, _, lo, hi in self.buckets)

    def _begin(self):
        self.fired = set()
        self.launched = 0                       # buckets [0, launched) are in flight
        self.works = []
        self.late = False
        if self.live is None:
            self.pending = None
        else:
            self.pending = [sum(1 for i in range(lo_i, hi_i) if id(self.fb.params[i]) in self.live)
                            for lo_i, hi_i, _, _ in self.buckets]

    def _launch_ready(self):
        while self.launched < len(self.buckets) and self.pending is not None and self.pending[self.launched] == 0:
            self._launch(self.launched)

    def _launch(self, b):
        lo_i, hi_i, lo, hi = self.buckets[b]
        if getattr(self.fb, "gathered", False):
            self.fb.gather(range(lo_i, hi_i))            # the bucket's gradient tensors -> their slices: one launch, then the collective
        flat = self.fb.flat_grad[lo:hi]
        if self.transport is not None:
            self.works.append((self.transport(flat), None))
        elif dist.get_backend() == "nccl":
            self.works.append((dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True), None))
        else:
            self.works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat))
        if hasattr(self.fb, "mark_written"):
            self.fb.mark_written(range(lo_i, hi_i))          # the slices hold the ranks' average from here on, not this rank's zeros
        self.launched = b + 1

    def _hook(self, p):
        if not collectives_on():
            return
        b = self.bucket_of.get(id(p))
        if b is None:
            self.late = True                                 # a never-live parameter (off the wire) received a gradient
            return
        if id(p) in self.fired:
            if b < self.launched:
                self.late = True                             # a second accumulation after the bucket went on the wire
            return
        self.fired.add(id(p))
        if (not getattr(self.fb, "gathered", False) and p.grad is not None
                and p.grad.data_ptr() != self.fb.flat_grad.data_ptr() + self.fb.offsets[self._index(p)] * 4):
            self.fb.view(self.fb.flat_grad, self._index(p)).copy_(p.grad)        # autograd replaced the view: re-home it
            p.grad = self.fb.view(self.fb.flat_grad, self._index(p))             # (gathered buffers: _launch brings the whole bucket home)
        if b < self.launched:
            self.late = True                                 # its bucket is already on the wire
            return
        if self.pending is not None:
            if id(p) in self.live:
                self.pending[b] -= 1
                self._launch_ready()
            else:
                self.late = True                             # never-live parameter riding in a bucket's gap turned live

    def _index(self, p):
        return self._idx[id(p)]

    def finish(self):
        """Call after backward: send the remaining buckets, wait for all of them, re-arm for the next step."""
        if collectives_on():
            self.fb.check_views()
            while self.launched < len(self.buckets):
                self._launch(self.launched)
            w = world_size()
            # EXPOSED_WAIT (bench.py, N > 1): a list that receives one (start, end) event pair per step bracketing the waits on the compute
            # stream: the time the stream sits behind collectives that backward did not cover
            ev = EXPOSED_WAIT is not None and torch.cuda.is_available() and dist.get_backend() == "nccl"
            if ev:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            for work, flat in self.works:
                if work is not None:
                    work.wait()
            if ev:
                e1.record()
                EXPOSED_WAIT.append((e0, e1))
            for work, flat in self.works:
                if flat is not None:
                    flat.mul_(1.0 / w)
            if self.late:
                # every collective of this step has been issued and completed on this rank, so the peers finish their
                # step too instead of hanging inside an all-reduce; the averaged gradients are incomplete -> stop here
                self._begin()
                raise RuntimeError("a parameter received its gradient after its bucket had been all-reduced (or a parameter "
                                   "that never had a gradient got one): the set of parameters that receive gradients "
                                   "changed; build a new OverlappedGradReducer")
            if self.live is None:
                self.live = set(self.fired)
                self._cut_buckets()                          # from now on only live ranges travel
        self._begin()

    def remove(self):
        for h in self.handles:
            h.remove()
        self.handles = []


def allreduce_gradients(optimizer_or_params):
    """Average gradients across ranks: the optimizer's OverlappedGradReducer when it has one (already running since
    backward), else one collective per flat buffer when given a FusedAdam, else per parameter (skipping ``grad is None``
    consistently — the set is static for this model family)."""
    reducers = getattr(optimizer_or_params, "grad_reducers", None)
    if reducers:
        for r in reducers:
            r.finish()
        return
    if not collectives_on():
        return
    flats = getattr(optimizer_or_params, "flat_buffers", None)
    if flats is not None:
        for fb in flats:
            fb.check_views()
            allreduce_mean_(fb.flat_grad)
            fb.mark_written()
        return
    for p in optimizer_or_params:
        if p.grad is not None:
            allreduce_mean_(p.grad)


def allreduce_scalar_max(x: float) -> float:
    if not collectives_on():
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()
