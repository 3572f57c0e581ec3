"""Where does a train step's wall time go: host enqueue time vs GPU busy time, and which host
sections block (hidden synchronisations).  Prints per-section host ms and the sync'd step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer

batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0")
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()

# (1) python-return time vs synchronised time
ret, tot = [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ret.append((t1 - t0) * 1e3)
    tot.append((t2 - t0) * 1e3)
print("host return %.1f ms | with sync %.1f ms (isolated steps)" % (sum(ret) / 6, sum(tot) / 6))

# (2) GPU busy time via the profiler
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        tr.train_step(batch)
    torch.cuda.synchronize()
ev = prof.key_averages()
gpu = sum(e.self_device_time_total for e in ev) / 3e3
print("GPU busy (sum of kernel durations) %.1f ms/step" % gpu)
rows = sorted(ev, key=lambda e: -e.self_cpu_time_total)[:25]
for e in rows:
    print("%-60s cpu %8.2f ms  n=%d" % (e.key[:60], e.self_cpu_time_total / 3e3, e.count // 3))
