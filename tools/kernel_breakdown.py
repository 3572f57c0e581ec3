"""Per-shape breakdown of the channel-last element-wise / pooling kernels of one train step:
HIP-event time per launch and the HBM rate their algorithmic bytes imply."""
import os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer

# name -> bytes(ints)  (ints = the integer arguments of the C call in order)
BYTES = {
    "gb_bn_bwd_apply": lambda a: 3 * 4 * a[0] * a[1],                       # P, C, relu, training
    "gb_bn_bwd_stats": lambda a: 2 * 4 * a[0] * a[1],                       # P, C, relu
    "gb_affine_act": lambda a: 2 * 4 * a[0] * a[1],                         # P, C, relu
    "gb_affine_relu_maxpool": lambda a: 4 * a[0] * a[1] * a[2] + 8 * a[0] * a[2],       # R, ns, C
    "gb_bn_bwd_stats_pool": lambda a: 4 * a[0] * a[2] * 3 + 4 * a[0] * a[2],            # R, ns, C (argmax rows only)
    "gb_bn_bwd_apply_pool": lambda a: 2 * 4 * a[0] * a[1] * a[2] + 12 * a[0] * a[2],    # R, ns, C, training
    "gb_bn_bwd_apply_members": lambda a: 0,   # R, D, C, P_total, training: row count unknown here -> time only
    "gb_affine_relu_maxpool_members": lambda a: 0,
    "gb_bn_bwd_apply_w": lambda a: 3 * 4 * a[0] * a[2],                 # rows, P_total, C, training
    "gb_la_pool_bwd": lambda a: 0,
    "gb_la_point_stats": lambda a: 0,
    "gb_cyl_unique": lambda a: 0,
    "gb_cyl_rows": lambda a: 0,
    "gb_group_concat_cl": lambda a: 4 * a[0] * a[2] * a[3] * (3 + a[4]) * 2,            # b, n, m, ns, c, mode
    "gb_group_concat_cl_grad": lambda a: 4 * a[0] * a[2] * a[3] * (3 + a[4]) + 4 * a[0] * a[1] * a[4],
}
batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0")
for _ in range(3):
    tr.train_step(batch)
with _lib.KernelTimer(list(BYTES)) as kt:
    for _ in range(3):
        tr.train_step(batch)
torch.cuda.synchronize()
acc = defaultdict(lambda: [0, 0.0])
for n in BYTES:
    for a, b, meta in kt.events[n]:
        key = (n[3:],) + tuple(meta["ints"])
        acc[key][0] += 1
        acc[key][1] += a.elapsed_time(b)
print("total %.2f ms/step" % (sum(v[1] for v in acc.values()) / 3))
print("%-24s %-30s %5s %8s %8s %7s" % ("kernel", "ints", "n/st", "ms/step", "avg us", "TB/s"))
for key, (cnt, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    byt = BYTES["gb_" + key[0]](key[1:])
    avg = ms / cnt * 1e3
    print("%-24s %-30s %5.1f %8.3f %8.1f %7.2f" % (key[0], str(key[1:]), cnt / 3, ms / 3, avg, byt / avg / 1e6))
