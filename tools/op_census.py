"""Census of the torch-level ops in one train step (count + device time), to find small-kernel
launch overhead worth fusing.  Uses torch.profiler with stack grouping by python source line."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer

batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0")
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(batch)
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ev if e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.count)
print("%-28s %6s %10s  %s" % ("op", "count", "dev us", "stack"))
seen = 0
for e in rows[:90]:
    st = [s for s in e.stack if "graspbalance_amd" in s or "bench" in s or "torch/optim" in s or "torch/nn/modules" in s][:3]
    st = [s.split("/root/repo/")[-1][:70] for s in st]
    print("%-28s %6d %10.1f  %s" % (e.key[:28], e.count, e.self_device_time_total, " <- ".join(st)))
