"""Host (Python + launch) time of the sections of one train step, without any synchronisation inside."""
import os, sys, time
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
net = tr.net
acc = {}
def lap(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0) * 1e3; return t1
for it in range(5):
    torch.cuda.synchronize()
    t = time.perf_counter()
    fused_mlp.begin_step(tr.device)
    ep = dict(batch)
    with fused_mlp.deferred_counters():
        feats, xyz, ep = net.view_estimator.FeatureExtraction(ep['point_clouds'], ep); t = lap("backbone fwd", t)
        ep = net.view_estimator.GraspableClasification(xyz, feats, ep); t = lap("graspable/view head fwd", t)
        ep = process_grasp_labels(ep); t = lap("label matching", t)
        ep = net.grasp_generator(ep); t = lap("stage-2 fwd (incl. the row-count sync)", t)
    loss, ep = get_loss(ep); t = lap("loss", t)
    loss.backward(); t = lap("backward", t)
    tr.grads.reduce(); tr.optimizer.step(); tr.grads.zero_grad(); tr.scheduler.step(); t = lap("optimizer", t)
tot = sum(acc.values()) / 5
for k, v in acc.items():
    print("%-44s %7.2f ms  %5.1f%%" % (k, v / 5, 100 * v / 5 / tot))
print("%-44s %7.2f ms" % ("total host", tot))
