"""Is a graph-replayed train step slow because a process group exists, or because two processes share one GPU?
  python tools/graph_dist_probe.py none|nccl1|gloo1      one process, no group / a one-rank RCCL group / a one-rank gloo group
  python tools/graph_dist_probe.py pair                  two independent processes (no group) on the one GPU, side by side"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode == "pair":
    ps = [subprocess.Popen([sys.executable, __file__, "none"]) for _ in range(2)]
    sys.exit(max(p.wait() for p in ps))
import torch
import torch.distributed as dist
if mode in ("nccl1", "gloo1"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl" if mode == "nccl1" else "gloo", rank=0, world_size=1)
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
dev = torch.device("cuda", 0)
tr = Trainer(dev, distributed=dist.is_initialized())
batch = tr.resident(make_training_batch(range(4), 20000, device=dev))
for _ in range(3):
    tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(8):
        tr.train_step(batch, next_batch=batch)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%s pid %d: %.2f ms/step (host %.2f), distributed=%s graphs=%d" % (mode, os.getpid(), (time.perf_counter() - t0) / 8 * 1e3,
          host / 8 * 1e3, tr.distributed, len(tr._graphs)), flush=True)
if dist.is_initialized():
    dist.destroy_process_group()
