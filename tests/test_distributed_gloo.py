"""CPU, world_size 2 over gloo: the data-parallel plumbing of distributed.py (sharding rule, weight broadcast,
flat-buffer gradient averaging with never-used parameters present)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from vn_pointcloudcompletion_amd import distributed as D
    from vn_pointcloudcompletion_amd.optim import FlatBuffers
    r, lr, w = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and D.world_size() == world and D.get_rank() == rank
    lo, hi = D.shard_batch(8)
    assert (lo, hi) == (rank * 4, rank * 4 + 4)
    with pytest.raises(ValueError):
        D.shard_batch(7)
    torch.manual_seed(100 + rank)                                   # different init on every rank ...
    model = torch.nn.Sequential(torch.nn.Linear(6, 4, bias=False), torch.nn.Linear(4, 4, bias=False), torch.nn.Linear(4, 1, bias=False))
    D.broadcast_model(model)                                        # ... identical after the broadcast
    w0 = model[0].weight.detach().clone()
    fb = FlatBuffers(model.parameters())

    class Holder:                                                   # what allreduce_gradients needs from FusedAdam
        flat_buffers = [fb]
    torch.manual_seed(7)
    x = torch.randn(8, 6)[lo:hi]                                    # each rank: its shard of the same global batch
    y = model[2](model[0](x))                                       # model[1] is never used: grad stays zero (not None)
    y.sum().backward()
    local = fb.flat_grad.clone()
    D.allreduce_gradients(Holder)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered) / world
    ok = torch.allclose(fb.flat_grad, expect, atol=1e-6) and fb.view(fb.flat_grad, 1).abs().sum().item() == 0.0
    mx = D.allreduce_scalar_max(float(rank + 1))

    # bucketed, backward-overlapped averaging: several buckets (tiny bucket size), one never-used parameter, three steps
    # (step 1 learns which parameters receive gradients; later steps launch buckets from inside backward)
    class Holder2:
        flat_buffers = [fb]
        grad_reducers = [D.OverlappedGradReducer(fb, bucket_bytes=8)]
    red = Holder2.grad_reducers[0]
    ok = ok and len(red.buckets) == 3 and red.buckets[0][3] == fb.numel and red.buckets[-1][2] == 0
    for step in range(3):
        fb.zero_grad()
        torch.manual_seed(20 + step)
        xs = torch.randn(8, 6)[lo:hi]
        model[2](model[0](xs)).pow(2).sum().backward()
        launched_in_backward = red.launched
        local = fb.flat_grad.clone() if step == 0 else None      # later steps: slices are being reduced already
        # expected value from an independent computation of both ranks' local gradients
        exp = torch.zeros_like(fb.flat_grad)
        for rr in range(world):
            m2 = torch.nn.Sequential(torch.nn.Linear(6, 4, bias=False), torch.nn.Linear(4, 4, bias=False), torch.nn.Linear(4, 1, bias=False))
            m2.load_state_dict(model.state_dict())
            torch.manual_seed(20 + step)
            m2[2](m2[0](torch.randn(8, 6)[rr * 4:rr * 4 + 4])).pow(2).sum().backward()
            off = 0
            for prm, o in zip(m2.parameters(), fb.offsets):
                if prm.grad is not None:
                    exp[o:o + prm.numel()] += prm.grad.reshape(-1) / world
        D.allreduce_gradients(Holder2)
        ok = ok and torch.allclose(fb.flat_grad, exp, atol=1e-6) and fb.view(fb.flat_grad, 1).abs().sum().item() == 0.0
        if step == 0:
            ok = ok and launched_in_backward == 0 and red.live is not None and len(red.live) == 2
            # the never-used parameter has left the wire: two buckets, (24 + 4) floats instead of 44
            ok = ok and len(red.buckets) == 2 and red.wire_bytes() == (24 + 4) * 4
        else:
            ok = ok and launched_in_backward >= 1                   # model[2]'s bucket left during backward
    # the never-used parameter suddenly receives a gradient: every rank completes the step's collectives, then raises
    fb.zero_grad()
    model[2](model[1](model[0](torch.randn(4, 6)))).sum().backward()
    try:
        D.allreduce_gradients(Holder2)
        ok = False
    except RuntimeError as e:
        ok = ok and "set of parameters" in str(e)
    D.barrier()                                                     # nobody is stuck inside an all-reduce
    red.remove()
    # BatchNorm running statistics drift apart between replicas (each rank updates them from its own shard); sync_buffers makes a
    # sharded evaluation describe rank 0's model: after it two ranks' eval-mode outputs on the same input agree with rank 0 alone
    bn = torch.nn.BatchNorm1d(4)
    bn.train()
    torch.manual_seed(300 + rank)
    bn(torch.randn(16, 4) * (rank + 1) + rank)                     # different running_mean / running_var on every rank
    mine = [b.clone() for b in bn.buffers()]
    D.sync_buffers(bn)
    got = [torch.zeros_like(b) for b in mine]
    for g_, b_ in zip(got, mine):
        g_.copy_(b_)
        dist.broadcast(g_, src=0)                                   # rank 0's pre-sync values
    ok = ok and all(torch.equal(a, b) for a, b in zip(bn.buffers(), got))
    bn.eval()
    xe = torch.arange(8.0).reshape(2, 4)
    ye = bn(xe).detach()
    y0 = ye.clone()
    dist.broadcast(y0, src=0)
    ok = ok and torch.equal(ye, y0)
    q.put((rank, ok, w0.sum().item(), mx))
    D.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_data_parallel():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get() for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    out.sort()
    assert all(o[1] for o in out)
    assert abs(out[0][2] - out[1][2]) < 1e-7                        # same weights on both ranks
    assert out[0][3] == out[1][3] == 2.0


def test_bucket_cut_skips_never_live_ranges():
    """Bucket layout of OverlappedGradReducer without a process group: large never-live ranges leave the wire, small ones
    ride along inside a bucket, never-live parameters at either end are trimmed."""
    from vn_pointcloudcompletion_amd import distributed as D
    from vn_pointcloudcompletion_amd.optim import FlatBuffers
    sizes = [100, 8, 100, 5000, 100, 100, 4000]                     # elements; [1] small dead gap, [3] and [6] large dead ranges
    params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    fb = FlatBuffers(params)
    red = D.OverlappedGradReducer(fb, bucket_bytes=1000, gap_bytes=64)
    assert red.wire_bytes() == fb.numel * 4                         # before the first backward everything travels
    red.live = {id(params[i]) for i in (0, 2, 4, 5)}
    red._cut_buckets()
    spans = [(lo_i, hi_i) for lo_i, hi_i, _, _ in red.buckets]
    assert spans == [(4, 6), (0, 3)]                                # [6] trimmed, [3] cut out, [1] (32 B) rides in a bucket
    assert red.wire_bytes() == (200 + 208) * 4
    assert set(red.bucket_of) == {id(params[i]) for i in (0, 1, 2, 4, 5)}
    red.remove()
