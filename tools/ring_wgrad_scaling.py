"""GPU: where does the ring kernel's wgrad tile saturate?  One product, growing row count (fixed costs vanish), as a single
gb_gemm_wgrad (the planner picks tile and split) and as a one-item gb_gemm_wgrad_group (64 x 64 tiles).  GB_GEMM_NO_DIRECT keeps
the tall ones off the register-direct kernel."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from graspbalance_amd import _lib as L   # noqa: E402

DEV = "cuda:0"
lib = L.lib()
ws = torch.empty(L.GEMM_SCRATCH_BYTES, dtype=torch.uint8, device=DEV)
opts = ctypes.pointer(L.GemmOpts(0, 0, ws.data_ptr(), ws.numel(), None, L.GEMM_NO_DIRECT))


def bench(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for K, N, aff in ((1024, 256, True), (256, 1024, False), (256, 256, False), (128, 128, True)):
    for P in (4096, 16384, 65536, 131072):
        dY, X = torch.randn(P, N, device=DEV), torch.randn(P, K, device=DEV)
        a = torch.cat([torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.3]) if aff else None
        dW = torch.zeros(N, K, device=DEV)
        arr = (L.WgradItem * 1)((dY.data_ptr(), X.data_ptr(), a.data_ptr() if aff else None, dW.data_ptr(), P, K, N, K))
        flop = 2.0 * P * K * N
        ts = bench(lambda: L.check(lib.gb_gemm_wgrad(L.ptr(dY), L.ptr(X), L.ptr(a), L.ptr(dW), P, K, N, opts, None), "w"))
        tg = bench(lambda: L.check(lib.gb_gemm_wgrad_group(ctypes.cast(arr, ctypes.c_void_p), 1, opts, None), "g"))
        print("P %6d K %4d N %4d aff %d: single %7.1f us = %5.1f TF/s | group of one %7.1f us = %5.1f TF/s"
              % (P, K, N, aff, ts, flop / ts / 1e6, tg, flop / tg / 1e6))
