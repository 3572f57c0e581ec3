"""Eval forward of the full-size network repeated on the same input: which outputs vary from run to run, and by how much."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd.graspbalance import GraspBalance
from graspbalance_amd.scene import make_batch
from tests.seeded import fill_by_key
DEV = "cuda:0"
net = fill_by_key(GraspBalance(is_training=False), seed=21).eval().to(DEV)
clouds = torch.from_numpy(make_batch([0, 1], 20000)).to(DEV)
outs = []
with torch.no_grad():
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        ep = net({'point_clouds': clouds})
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in ep.items() if torch.is_tensor(v)})
ref = outs[0]
for i, o in enumerate(outs[1:], 1):
    rows = []
    for k in ref:
        if o[k].shape != ref[k].shape:
            rows.append("%s shape" % k); continue
        if o[k].dtype.is_floating_point:
            d = (o[k].double() - ref[k].double())
            n = int((d != 0).sum())
            if n:
                rows.append("%s: %d/%d differ, max %.2e, rel %.2e" % (k, n, d.numel(), float(d.abs().max()), float(d.norm() / ref[k].double().norm())))
        elif not torch.equal(o[k], ref[k]):
            rows.append("%s: %d/%d differ (int)" % (k, int((o[k] != ref[k]).sum()), o[k].numel()))
    print("run", i, "|", "; ".join(rows) if rows else "identical")
