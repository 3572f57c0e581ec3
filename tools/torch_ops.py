"""Which torch-native ops still launch kernels in one train step (eager execution), by op name, input shapes and - for
forward ops - the graspbalance_amd source line; backward ops are named by their autograd node."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer

batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0", graph=False)
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as p:
    tr.train_step(batch)
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for e in p.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
        continue
    dev = sum(k.duration for k in e.kernels)
    if not e.kernels:
        continue
    where = ""
    for s in (e.stack or []):
        if "graspbalance_amd" in s:
            where = s.split("graspbalance_amd/")[-1]
            break
    if not where:
        par = e.cpu_parent
        while par is not None and not ("Backward" in par.name or "Optimizer" in par.name):
            par = par.cpu_parent
        where = par.name if par is not None else "?"
    shapes = str([s for s in (e.input_shapes or []) if s])[:60]
    r = rows[(e.name, shapes, where)]
    r[0] += len(e.kernels)
    r[1] += dev
tot_n = sum(r[0] for r in rows.values())
tot_t = sum(r[1] for r in rows.values())
print("%d torch-native launches, %.3f ms of kernel time in one step" % (tot_n, tot_t / 1e3))
for (name, shapes, where), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%3d %7.1f us  %-22s %-60s %s" % (n, t, name, shapes, where))
