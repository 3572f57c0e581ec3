"""configs[4]-style smoke: B clouds of 50 000 points through one full train step (the first-level FPS takes fps_pruned_big_kernel,
everything else as in the bench); prints the step time.  Not a bench line."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 50000
batch = make_training_batch(range(B), N, device="cuda:0")
tr = Trainer("cuda:0")
for _ in range(2):
    loss = tr.train_step(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    loss = tr.train_step(batch)
torch.cuda.synchronize()
print("B=%d N=%d: %.1f ms/step, loss %.4f, finite=%s, peak mem %.1f GB" % (B, N, (time.perf_counter() - t0) / 3 * 1e3, float(loss.detach()), bool(torch.isfinite(loss)), torch.cuda.max_memory_allocated() / 1e9))
