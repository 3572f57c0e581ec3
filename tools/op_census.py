"""Census of the torch-level (non-gb) ops in one train step by input shape: count + device time."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_step(batch)
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.self_device_time_total > 0]
ev.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in ev)
print("aten ops with device time: %.2f ms, %d launches-ish" % (tot / 1e3, sum(e.count for e in ev)))
for e in ev[:60]:
    print("%-28s n=%4d dev %8.1f us  %s" % (e.key[:28], e.count, e.self_device_time_total, str(e.input_shapes)[:110]))
