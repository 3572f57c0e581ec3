"""fp32 GEMM accuracy vs fp64: torch CPU (MKL), torch GPU (rocBLAS), the repo's MFMA kernels (gb_gemm_fwd)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib, fused_mlp
torch.manual_seed(0)
dev = "cuda:0"
for (P, K, N) in [(65536, 64, 128), (65536, 128, 256), (16384, 256, 256), (16384, 1024, 256), (4096, 512, 128)]:
    X = torch.randn(P, K); W = torch.randn(N, K)
    truth = X.double() @ W.double().t()
    cpu = X @ W.t()
    Xg, Wg = X.to(dev), W.to(dev)
    gpu = Xg @ Wg.t()
    Y = torch.empty(P, N, device=dev)
    fused_mlp._call("gb_gemm_fwd", Xg.device, _lib.ptr(Xg), _lib.ptr(Wg), None, _lib.ptr(Y), None, 1, P, K, N, None, fused_mlp._s(Xg))
    torch.cuda.synchronize()
    rel = lambda a: float((a.double().cpu() - truth).norm() / truth.norm())
    rs = _lib.lib().gb_gemm_uses_rs(P, K, N, 0, 0, 0)
    print("P=%d K=%d N=%d: cpu %.2e rocblas %.2e own(%s) %.2e" % (P, K, N, rel(cpu), rel(gpu), "rs" if rs else "cl", rel(Y)))
