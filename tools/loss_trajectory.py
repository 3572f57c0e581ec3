"""Loss over a few train steps: fused kernels vs the plain torch composition on the same seeds (sanity check that
the reformulations - LocalAggregation without grouping, distinct cylinder rows, closed-form first layers - train
the same model; trajectories separate slowly through max-pool / ReLU routing chaos, not abruptly)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import fused_mlp
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
B, N, steps = 2, 20000, int(sys.argv[1]) if len(sys.argv) > 1 else 8
batch = make_training_batch(range(B), N, device="cuda:0")
out = {}
for name, flag in (("fused", True), ("plain", False)):
    fused_mlp.set_enabled(flag)
    tr = Trainer("cuda:0", seed=7)
    out[name] = [float(tr.train_step(batch)) for _ in range(steps)]
fused_mlp.set_enabled(True)
for i in range(steps):
    print("step %2d  fused %.6f  plain %.6f  rel diff %.2e" % (i, out["fused"][i], out["plain"][i], abs(out["fused"][i] - out["plain"][i]) / abs(out["plain"][i])))
