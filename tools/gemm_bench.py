"""Own fp32 MFMA GEMMs vs torch.mm (rocBLAS) on the layer shapes of the GraspBalance step (B=4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib as L
DEV = "cuda:0"
SLOTS = int(os.environ.get("SLOTS", "32"))

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

SHAPES = [("sa1.l0", 524288, 3, 64), ("sa1.l1", 524288, 64, 64), ("sa1.l2", 524288, 64, 128), ("inv1.agg", 524288, 131, 128),
          ("inv1.pw0", 8192, 128, 512), ("inv1.pw1", 8192, 512, 128), ("sa2.l0", 131072, 131, 128), ("sa2.l2", 131072, 128, 256),
          ("inv2.agg", 131072, 259, 256), ("inv2.pw0", 4096, 256, 1024), ("wg.l0", 1048576, 3, 64), ("wg.l1", 1048576, 64, 128),
          ("wg.l2", 1048576, 128, 256)]
if os.environ.get("SMALL"):
    SHAPES = [("inv2.pw0", 4096, 256, 1024), ("inv2.pw1", 4096, 1024, 256), ("inv3.pw0", 2048, 256, 1024),
              ("inv3.pw1", 2048, 1024, 256), ("inv4.pw0", 1024, 256, 1024), ("inv4.pw1", 1024, 1024, 256),
              ("inv1.pw0", 8192, 128, 512), ("inv1.pw1", 8192, 512, 128), ("la2.G", 4096, 256, 256), ("la1.G", 8192, 128, 128)]
lib = L.lib()
tot = {"own": 0.0, "blas": 0.0}
for name, P, K, N in SHAPES:
    X = torch.randn(P, K, device=DEV); W = torch.randn(N, K, device=DEV); dY = torch.randn(P, N, device=DEV)
    Y = torch.empty(P, N, device=DEV); dX = torch.empty(P, K, device=DEV); dW = torch.zeros(N, K, device=DEV)
    Wt = W.t().contiguous(); st = torch.zeros(SLOTS * 2 * N, dtype=torch.float64, device=DEV)
    fl = 2.0 * P * K * N
    r = {}
    r["fwd own"] = timeit(lambda: lib.gb_gemm_fwd(L.ptr(X), L.ptr(W), None, L.ptr(Y), L.ptr(st), SLOTS, P, K, N, None, None))
    r["fwd blas"] = timeit(lambda: torch.mm(X, W.t(), out=Y))
    r["dgrad own"] = timeit(lambda: lib.gb_gemm_dgrad(L.ptr(dY), L.ptr(W), L.ptr(dX), None, None, None, 0, P, K, N, None, None, None, None))
    r["dgrad blas"] = timeit(lambda: torch.mm(dY, W, out=dX))
    r["wgrad own"] = timeit(lambda: lib.gb_gemm_wgrad(L.ptr(dY), L.ptr(X), None, L.ptr(dW), P, K, N, None))
    from graspbalance_amd.fused_mlp import _wgrad
    r["wgrad blas"] = timeit(lambda: _wgrad(dY, X))
    print("%-9s P=%7d K=%4d N=%4d | " % (name, P, K, N) + " | ".join("%s %7.1f us %5.1f TF" % (k, v, fl / v / 1e6) for k, v in r.items()))
    for k, v in r.items(): tot[k.split()[1]] += v
print("total own %.1f us, blas %.1f us" % (tot["own"], tot["blas"]))
