"""configs[4] stress shape: B clouds of 50 000 points through full train steps (first-level FPS on
fps_pruned_big_kernel, everything else as in the bench), in fp32 and in the bf16 MLP mode; prints the step times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import fused_mlp
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 50000
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f32", "bf16"]
batch = make_training_batch(range(B), N, device="cuda:0")
for mode in modes:
    tr = Trainer("cuda:0", mlp_precision=mode)
    for _ in range(2):
        loss = tr.train_step(batch, next_batch=batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        loss = tr.train_step(batch, next_batch=batch)
    torch.cuda.synchronize()
    print("%s B=%d N=%d: %.1f ms/step, loss %.4f, finite=%s, peak mem %.1f GB" % (mode, B, N, (time.perf_counter() - t0) / 4 * 1e3, float(loss.detach()), bool(torch.isfinite(loss)), torch.cuda.max_memory_allocated() / 1e9))
    del tr
