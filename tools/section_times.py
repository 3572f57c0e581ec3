"""GPU time of the sections of one train step (HIP events on the current stream; no host syncs inside)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import fused_mlp
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
from graspbalance_amd.loss import get_loss
from graspbalance_amd.label_generation import process_grasp_labels

batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0")
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
net = tr.net
acc = {}
for it in range(5):
    marks.clear()
    mark("start")
    ep = dict(batch)
    with fused_mlp.deferred_counters():
        fe = net.view_estimator.FeatureExtraction
        feats, xyz, ep = fe(ep['point_clouds'], ep); mark("backbone fwd")
        ep = net.view_estimator.GraspableClasification(xyz, feats, ep); mark("graspable/view head fwd")
        ep = process_grasp_labels(ep); mark("label matching")
        ep = net.grasp_generator(ep); mark("stage-2 heads fwd (cylinder crops)")
    loss, ep = get_loss(ep); mark("loss")
    loss.backward(); mark("backward")
    tr.grads.reduce(); tr.optimizer.step(); tr.grads.zero_grad(); tr.scheduler.step(); mark("optimizer")
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
tot = sum(acc.values()) / 5
for k, v in acc.items():
    print("%-40s %7.2f ms  %5.1f%%" % (k, v / 5, 100 * v / 5 / tot))
print("%-40s %7.2f ms" % ("total", tot))
