"""rocBLAS / hipBLASLt (through torch.mm) on the few-row GEMM shapes of the train step, for comparison with gemm_cl's
per-shape times (tools/fewrow_breakdown.py).  Plain products only: no affine-on-load, no BatchNorm sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = "cuda:0"
shapes = [("fwd", 4096, 1024, 256), ("fwd", 4096, 256, 1024), ("fwd", 4096, 256, 256), ("fwd", 8192, 512, 128),
          ("fwd", 2048, 1024, 256), ("fwd", 1024, 1024, 256), ("fwd", 8192, 128, 512), ("fwd", 2048, 256, 1024)]
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)   # evict L2 / MALL between... (not used per-iter: hot numbers)
for kind, P, K, N in shapes:
    X = torch.randn(P, K, device=dev); W = torch.randn(N, K, device=dev); dY = torch.randn(P, N, device=dev)
    t_f = bench(lambda: torch.mm(X, W.t()))
    t_d = bench(lambda: torch.mm(dY, W))
    t_w = bench(lambda: torch.mm(dY.t(), X))
    fl = 2.0 * P * K * N
    print("%6d %5d %5d  fwd %6.1f us (%5.1f TF/s)  dgrad %6.1f us  wgrad %6.1f us" % (P, K, N, t_f, fl / t_f / 1e6, t_d, t_w))
