"""dev probe: does this build's gloo all-reduce HIP tensors between two processes on one GPU?  Hard timeouts everywhere."""
import os, sys, socket, datetime, time
import torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=30))
        t = torch.full((1024,), float(rank + 1), device="cuda")
        w = dist.all_reduce(t, async_op=True)
        w.wait()
        torch.cuda.synchronize()
        q.put((rank, "ok", t[0].item()))
    except Exception as e:
        q.put((rank, "error", repr(e)))

if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, port, q)) for r in range(2)]
    for p in ps: p.start()
    t0 = time.time(); got = []
    while len(got) < 2 and time.time() - t0 < 90:
        try: got.append(q.get(timeout=5))
        except Exception: pass
    print("results:", got, "elapsed %.1f" % (time.time() - t0))
    for p in ps:
        p.join(5)
        if p.is_alive(): p.terminate()
