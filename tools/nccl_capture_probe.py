"""Can an RCCL all-reduce be captured INSIDE a linear HIP graph (one-rank group), and is the graph still on the fast
launch path?  Replays: host time per replay with / without the collective in the chain."""
import os, sys, time
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(9_000_000, device=dev)
y = torch.zeros(4096, device=dev)
dist.all_reduce(x); torch.cuda.synchronize()
def body(with_coll):
    for _ in range(300):
        y.add_(1.0)
    if with_coll:
        dist.all_reduce(x, op=dist.ReduceOp.AVG)
    for _ in range(300):
        y.add_(1.0)
cs = torch.cuda.Stream()
for with_coll in (False, True):
    with torch.cuda.stream(cs):
        body(with_coll)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=cs, capture_error_mode="thread_local"):
            body(with_coll)
    except Exception as e:
        print("capture with collective=%s FAILED: %r" % (with_coll, e)); continue
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    host = (time.perf_counter() - t0) / 20 * 1e3
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 20 * 1e3
    print("collective in the chain: %-5s  host %.3f ms per replay, %.3f ms per replay in total; x[0] = %.3f" % (with_coll, host, tot, float(x[0])))
dist.destroy_process_group()
