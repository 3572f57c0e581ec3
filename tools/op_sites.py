"""Torch (non-gb) kernels of a train step by section: launches and device time, top ops per section."""
import os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from graspbalance_amd import fused_mlp
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
from graspbalance_amd.loss import get_loss
from graspbalance_amd.label_generation import process_grasp_labels

batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0", graph=False)
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
net = tr.net
report = {}


class Sec:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.p = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True)
        self.p.__enter__()

    def __exit__(self, *a):
        torch.cuda.synchronize()
        self.p.__exit__(*a)
        ev = [e for e in self.p.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.self_device_time_total > 0]
        report[self.name] = ev


fused_mlp.begin_step(tr.device)
ep = dict(batch)
with fused_mlp.deferred_counters():
    with Sec("backbone fwd"):
        feats, xyz, ep = net.view_estimator.FeatureExtraction(ep['point_clouds'], ep)
    with Sec("view head fwd"):
        ep = net.view_estimator.GraspableClasification(xyz, feats, ep)
    with Sec("label matching"):
        ep = process_grasp_labels(ep)
    with Sec("stage-2 fwd"):
        ep = net.grasp_generator(ep)
with Sec("loss fwd"):
    loss, ep = get_loss(ep)
with Sec("backward"):
    loss.backward()
with Sec("optimizer"):
    tr.grads.reduce(); tr.optimizer.step(); tr.grads.zero_grad(); tr.scheduler.step()
for name, ev in report.items():
    n = sum(e.count for e in ev)
    us = sum(e.self_device_time_total for e in ev)
    print("== %-16s %4d launches %8.1f us" % (name, n, us))
    for e in sorted(ev, key=lambda e: -e.self_device_time_total)[:60]:
        print("      %-26s n=%3d %7.1f us  %s" % (e.key[:26], e.count, e.self_device_time_total, str(e.input_shapes)[:90]))
