"""A longer run of the graph-replayed trainer on CHANGING data: N steps over a pool of distinct synthetic batches (each
step announces the next one), loss finite throughout, memory flat, and the same schedule launch by launch for
comparison of the loss curves (they are not expected to be equal step by step: see tests/test_graph_step_gpu.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
pool = [make_training_batch(range(4 * i, 4 * i + 4), 20000, device=dev) for i in range(6)]
for mode in (True, False):
    torch.manual_seed(0)
    tr = Trainer(dev, graph=mode, steps_per_epoch=N)
    losses, mem = [], []
    for i in range(5):     # (capture / lazy initialisation outside the clock)
        tr.train_step(pool[i % len(pool)], next_batch=pool[(i + 1) % len(pool)])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N):
        b, nb = pool[i % len(pool)], pool[(i + 1) % len(pool)]
        loss = tr.train_step(b, next_batch=nb)
        if i % 10 == 9 or i == 0:
            losses.append(float(loss.detach())); mem.append(torch.cuda.memory_allocated() / 2**30)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e3
    assert all(l == l and abs(l) < 1e6 for l in losses), losses
    print("%-16s %d steps, %.2f ms per step (incl. staging of each new batch); loss every 10 steps: %s; allocated GiB first/last %.2f / %.2f"
          % ("graph replay:" if mode else "launch by launch:", N, dt, " ".join("%.3f" % l for l in losses), mem[0], mem[-1]), flush=True)
