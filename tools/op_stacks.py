"""Torch (aten) ops of one launch-by-launch train step that touch device tensors, attributed to the python line of this
package that issued them (a TorchDispatchMode logging the innermost graspbalance_amd frame): which lines still go
through torch kernels, and how often."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0", graph=False)
for _ in range(3):
    tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
SKIP = ("aten.view", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.permute", "aten.slice", "aten.select", "aten.expand",
        "aten.unsqueeze", "aten.squeeze", "aten.detach", "aten.alias", "aten.as_strided", "aten.empty", "aten.reshape",
        "aten.unbind", "aten.split", "aten.narrow", "aten.is_", "aten.size", "aten.stride", "aten._local_scalar_dense",
        "aten.lift_fresh", "aten.new_empty", "aten.empty_like", "aten.unfold", "aten.chunk", "aten.diagonal")
acc = collections.Counter()

class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            dev = any(torch.is_tensor(a) and a.is_cuda for a in args)
            if dev or "zeros" in name or "full" in name or "ones" in name or "arange" in name:
                site = "(no package frame: autograd engine / optimizer)"
                for fr in reversed(traceback.extract_stack(limit=40)):
                    if "graspbalance_amd/" in fr.filename:
                        site = "%s:%d" % (fr.filename.split("graspbalance_amd/")[-1], fr.lineno)
                        break
                acc[(site, name)] += 1
        return func(*args, **(kwargs or {}))

with Log():
    tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
by_site = collections.Counter()
for (site, name), n in acc.items():
    by_site[site] += n
print("aten ops on device tensors (views / allocations excluded): %d" % sum(acc.values()))
for site, n in by_site.most_common(70):
    ops = ", ".join("%s x%d" % (nm.replace("aten.", "").replace(".default", ""), c) for (s2, nm), c in sorted(acc.items(), key=lambda kv: -kv[1]) if s2 == site)
    print("%4d  %-44s %s" % (n, site, ops[:150]))
