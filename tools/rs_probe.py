"""The big GEMM launches of the step in isolation (for rocprofv3 --pmc): fwd+stats+affine, dgrad+BN sums, wgrad at 0.4 M rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib as L, fused_mlp
DEV = "cuda:0"
lib = L.lib()
P = int(os.environ.get("P", 400000)); reps = int(os.environ.get("REPS", 5))
for (K, N) in [(128, 256), (64, 128)]:
    X = torch.randn(P, K, device=DEV); W = torch.randn(N, K, device=DEV) / K ** 0.5; dY = torch.randn(P, N, device=DEV)
    Y = torch.empty(P, N, device=DEV); dX = torch.empty(P, K, device=DEV); dW = torch.zeros(N, K, device=DEV)
    aff = torch.cat([torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1]).contiguous()
    ab = torch.cat([torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1, torch.zeros(K, device=DEV), torch.ones(K, device=DEV)]).contiguous()
    st = torch.zeros(32 * 2 * N, dtype=torch.float64, device=DEV); dst = torch.zeros(33 * 2 * K, dtype=torch.float64, device=DEV)
    s = fused_mlp._s(X)
    for i in range(reps):
        L.check(lib.gb_gemm_fwd(L.ptr(X), L.ptr(W), L.ptr(aff), L.ptr(Y), L.ptr(st), 32, P, K, N, None, s), "fwd")
        L.check(lib.gb_gemm_dgrad(L.ptr(dY), L.ptr(W), L.ptr(dX), L.ptr(X), L.ptr(ab), L.ptr(dst), 32, P, K, N, None, None, None, s), "dgrad")
        L.check(lib.gb_gemm_wgrad(L.ptr(dY), L.ptr(X), L.ptr(aff), L.ptr(dW), P, K, N, s), "wgrad")
    torch.cuda.synchronize()
print("done")
