"""gemm_cl / gemm_rs on the few-row shapes as PLAIN products (no affine, no statistics), hot operands, 50 launches
each: run under rocprofv3 --kernel-trace --stats to read the kernels' own durations next to tools/blas_probe.py's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib, fused_mlp
dev = torch.device("cuda:0")
lib = _lib.lib()
shapes = [(4096, 1024, 256), (4096, 256, 1024), (4096, 256, 256), (8192, 512, 128), (2048, 1024, 256), (1024, 1024, 256),
          (8192, 128, 512), (2048, 256, 1024)]
st = _lib.current_stream(dev)
opts = fused_mlp._opts(dev, st, _lib.PREC_F32)
for P, K, N in shapes:
    X = torch.randn(P, K, device=dev); W = torch.randn(N, K, device=dev); dY = torch.randn(P, N, device=dev)
    Y = torch.empty(P, N, device=dev); dX = torch.empty(P, K, device=dev); dW = torch.zeros(N, K, device=dev)
    for _ in range(50):
        _lib.check(lib.gb_gemm_fwd(_lib.ptr(X), _lib.ptr(W), None, _lib.ptr(Y), None, 1, P, K, N, None, opts, st), "fwd")
    for _ in range(50):
        _lib.check(lib.gb_gemm_dgrad(_lib.ptr(dY), _lib.ptr(W), _lib.ptr(dX), None, None, None, 0, P, K, N, None, None, None,
                                     opts, st), "dgrad")
    for _ in range(50):
        _lib.check(lib.gb_gemm_wgrad(_lib.ptr(dY), _lib.ptr(X), None, _lib.ptr(dW), P, K, N, opts, st), "wgrad")
    torch.cuda.synchronize()
    print("done", P, K, N, flush=True)
