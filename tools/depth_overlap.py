"""How much do the 4 depth cylinders of a seed overlap?  unique (seed, point) pairs / total rows, per radius."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import fused_ops
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0")
orig = fused_ops.cylinder_query_multi
def multi(*a, **k):
    out = orig(*a, **k)
    for i, t in enumerate(out):            # t: list over depths? or tensor (D, B, m, ns)
        tt = torch.stack(list(t), 0) if isinstance(t, (list, tuple)) else t
        D, B, m, ns = tt.shape
        ids = tt.permute(1, 2, 0, 3).reshape(B * m, D * ns).long()
        srt, _ = ids.sort(dim=1)
        uniq = 1 + (srt[:, 1:] != srt[:, :-1]).sum(1)
        print("radius #%d: rows per seed %d, unique points per seed %.1f  -> unique fraction %.3f" % (i, D * ns, uniq.float().mean().item(), uniq.float().mean().item() / (D * ns)))
    return out
fused_ops.cylinder_query_multi = multi
tr.train_step(batch)
