"""Few-row GEMM launches (P <= 8192) of one train step: total ms/step and per shape.  For tuning the tile / split
heuristics of gemm_cl (run per library build, tools/ab_so.sh style)."""
import os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer

batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0")
for _ in range(3):
    tr.train_step(batch)
names = ["gb_gemm_fwd", "gb_gemm_dgrad", "gb_gemm_wgrad"]
STEPS = 4
with _lib.KernelTimer(names) as kt:
    for _ in range(STEPS):
        tr.train_step(batch)
torch.cuda.synchronize()
acc = defaultdict(lambda: [0, 0.0])
for n in names:
    for a, b, meta in kt.events[n]:
        P, K, N = meta["pkn"]
        if P <= 8192:
            acc[(n[8:], P, K, N)][0] += 1
            acc[(n[8:], P, K, N)][1] += a.elapsed_time(b)
tot = sum(v[1] for v in acc.values()) / STEPS
print("FEWROW total %.3f ms/step" % tot)
if "-v" in sys.argv:
    for key, (cnt, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
        print("  %-6s %6d %5d %5d  n/st %4.1f  avg %6.1f us" % (key + (cnt / STEPS, ms / cnt * 1e3)))
