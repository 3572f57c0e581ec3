"""Cost of the fused GEMM variants vs the plain kernel + separate element-wise pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib as L
DEV = "cuda:0"
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
lib = L.lib()
for name, P, K, N in [("sa1.l1", 524288, 64, 64), ("sa1.l2", 524288, 64, 128), ("wg.l1", 1048576, 64, 128), ("wg.l2", 1048576, 128, 256), ("inv1.pw1", 8192, 512, 128)]:
    X = torch.randn(P, K, device=DEV); W = torch.randn(N, K, device=DEV); dY = torch.randn(P, N, device=DEV)
    Y = torch.empty(P, N, device=DEV); dX = torch.empty(P, K, device=DEV); dW = torch.zeros(N, K, device=DEV)
    Z = torch.empty(P, K, device=DEV)
    Wt = W.t().contiguous(); st = torch.zeros(32 * 2 * N, dtype=torch.float64, device=DEV)
    affK = torch.randn(4 * K, device=DEV); dst = torch.zeros(32 * 2 * K, dtype=torch.float64, device=DEV)
    t = {}
    t["fwd"] = timeit(lambda: lib.gb_gemm_fwd(L.ptr(X), L.ptr(W), None, L.ptr(Y), L.ptr(st), 32, P, K, N, None, None))
    t["fwd+aff"] = timeit(lambda: lib.gb_gemm_fwd(L.ptr(X), L.ptr(W), L.ptr(affK), L.ptr(Y), L.ptr(st), 32, P, K, N, None, None))
    t["affine_act(K)"] = timeit(lambda: lib.gb_affine_act(L.ptr(X), L.ptr(affK), None, L.ptr(Z), P, K, 1, None))
    t["dgrad"] = timeit(lambda: lib.gb_gemm_dgrad(L.ptr(dY), L.ptr(W), L.ptr(dX), None, None, None, 0, P, K, N, None, None, None, None))
    t["dgrad+bn"] = timeit(lambda: lib.gb_gemm_dgrad(L.ptr(dY), L.ptr(W), L.ptr(dX), L.ptr(X), L.ptr(affK), L.ptr(dst), 32, P, K, N, None, None, None, None))
    t["bn_bwd_stats(K)"] = timeit(lambda: lib.gb_bn_bwd_stats(L.ptr(dX), L.ptr(X), L.ptr(affK), None, P, K, 1, L.ptr(dst), None, None, None))
    t["wgrad"] = timeit(lambda: lib.gb_gemm_wgrad(L.ptr(dY), L.ptr(X), None, L.ptr(dW), P, K, N, None))
    t["wgrad+aff"] = timeit(lambda: lib.gb_gemm_wgrad(L.ptr(dY), L.ptr(X), L.ptr(affK), L.ptr(dW), P, K, N, None))
    print("%-8s " % name + " | ".join("%s %.0f" % kv for kv in t.items()))
