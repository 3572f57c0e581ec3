"""Which parameter gradients differ between eager runs of the SAME first step (same initial state, same batch)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
DEV = "cuda:0"
# experiment switch: SPREAD_MODE=fwd|bwd -> the split only in the forward / only in the backward products (autograd runs
# the backward on its own thread: the wrapper tells the two apart by the thread)
import threading
from graspbalance_amd import fused_mlp, _lib
_mode = os.environ.get("SPREAD_MODE", "")
if _mode:
    _orig = fused_mlp._opts
    def _opts(dev, st, prec, rows_dev=None):
        main = threading.current_thread() is threading.main_thread()
        if prec in (_lib.PREC_F32, _lib.PREC_F32_SPLIT3):
            prec = _lib.PREC_F32_SPLIT3 if (main == (_mode == "fwd")) else _lib.PREC_F32
        return _orig(dev, st, prec, rows_dev)
    fused_mlp._opts = _opts
batch = make_training_batch([0, 1, 2, 3], num_point=20000, device=DEV)
runs = []
for _ in range(4):
    tr = Trainer(DEV, steps_per_epoch=10, max_epoch=2, graph=False)
    loss = float(tr.train_step(batch, next_batch=batch).detach())
    torch.cuda.synchronize()
    names = [n for n, p in tr.net.named_parameters() if p.requires_grad] if hasattr(tr, "net") else None
    sizes = [p.numel() for p in tr.optimizer._params]
    runs.append((tr.optimizer._flat_g.double().clone(), loss))
if names is None or len(names) != len(sizes):
    names = ["p%d" % i for i in range(len(sizes))]
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
print("losses", [r[1] for r in runs])
print("flat distances to run 0:", ["%.2e" % rel(r[0], runs[0][0]) for r in runs[1:]])
far = max(range(1, 4), key=lambda i: rel(runs[i][0], runs[0][0]))
a, b = runs[far][0].split(sizes), runs[0][0].split(sizes)
rows = sorted(((float((x - y).norm()), rel(x, y), n, s) for x, y, n, s in zip(a, b, names, sizes)), reverse=True)
tot = float((runs[far][0] - runs[0][0]).norm())
for d, r, n, s in rows[:14]:
    print("%-60s n=%8d  |diff| %.3e (%.0f%% of the flat distance)  rel %.2e" % (n, s, d, 100 * d * d / (tot * tot), r))
