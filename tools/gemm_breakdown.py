"""Per-shape breakdown of the GEMM launches of one train step: time, TF/s, and the per-shape
roofline max(flop / MFMA peak, algorithmic bytes / HBM peak)."""
import os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer

MFMA, HBM = 157.3e12, 8.0e12
batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0", graph=False)   # (events around single launches: not under graph replay)
for _ in range(3):
    tr.train_step(batch)
names = ["gb_gemm_fwd", "gb_gemm_fwd_w", "gb_gemm_fwd_pool", "gb_gemm_fwd_gen3", "gb_gemm_dgrad", "gb_gemm_dgrad_first", "gb_gemm_dgrad_first_gen3", "gb_gemm_wgrad", "gb_gemm_wgrad_gen3", "gb_gemm_dgrad_wgrad", "gb_gemm_wgrad_group"]
with _lib.KernelTimer(names) as kt:
    for _ in range(3):
        tr.train_step(batch)
torch.cuda.synchronize()
acc = defaultdict(lambda: [0, 0.0])
for n in names:
    for a, b, meta in kt.events[n]:
        if n == "gb_gemm_wgrad_group":   # one grouped launch: P = the rows of all its products, K = N = 0 (FLOP from the meta)
            key = ("wgrad_group(%d)" % meta["products"], int(meta["flop"] / 2e6), 1000, 1000)
        else:
            key = (n[8:] + ("*" if meta["kernel"] == "gemm_rs_kernel" else ""),) + tuple(meta["pkn"])
        acc[key][0] += 1
        acc[key][1] += a.elapsed_time(b)
tot = sum(v[1] for v in acc.values()) / 3
print("total gemm event time %.2f ms/step" % tot)
print("%-6s %9s %5s %5s %5s %9s %8s %7s %7s %6s" % ("kind", "P", "K", "N", "n/st", "ms/step", "avg us", "TF/s", "roof us", "frac"))
for key, (cnt, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    kind, P, K, N = key
    flop = 2.0 * P * K * N
    byt = 4.0 * (P * K + P * N + K * N)
    roof = max(flop / MFMA, byt / HBM) * 1e6
    avg = ms / cnt * 1e3
    print("%-6s %9d %5d %5d %5.1f %9.3f %8.1f %7.1f %7.1f %6.2f" % (kind, P, K, N, cnt / 3, ms / 3, avg, flop / avg / 1e6, roof, roof / avg))
