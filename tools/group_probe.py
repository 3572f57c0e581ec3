"""GPU: gb_gemm_wgrad_group on the weight gradients a train step records (B = 4 x 20 000), per stage and all together,
against the same products as single gb_gemm_wgrad calls.  usage: python tools/group_probe.py [prec]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from graspbalance_amd import _lib as L   # noqa: E402

DEV = "cuda:0"
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lib = L.lib()
ws = torch.empty(L.GEMM_SCRATCH_BYTES, dtype=torch.uint8, device=DEV)
opts = ctypes.pointer(L.GemmOpts(prec, 0, ws.data_ptr(), ws.numel()))
STAGES = {
    "stage1 (3 blocks, 8192 rows)": [(8192, 512, 128, True), (8192, 128, 512, False), (8192, 128, 128, False)] * 3,
    "stage2 (6 blocks, 4096 rows)": [(4096, 1024, 256, True), (4096, 256, 1024, False), (4096, 256, 256, False)] * 6,
    "stage3 (3 blocks, 2048 rows)": [(2048, 1024, 256, True), (2048, 256, 1024, False), (2048, 256, 256, False)] * 3,
    "stage4 (3 blocks, 1024 rows)": [(1024, 1024, 256, True), (1024, 256, 1024, False), (1024, 256, 256, False)] * 3,
    "fp + heads": [(16384, 1024, 256, False), (16384, 256, 128, True), (16384, 256, 128, False), (32768, 128, 256, True),
                   (32768, 128, 128, True), (16384, 128, 256, True), (16384, 128, 128, True), (16384, 128, 128, True),
                   (16384, 128, 128, True), (4096, 256, 302, False)],
}
STAGES["all"] = [s for v in list(STAGES.values()) for s in v]


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, shapes in STAGES.items():
    keep, items = [], []
    for P, K, N, aff in shapes:
        dY, X = torch.randn(P, N, device=DEV), torch.randn(P, K, device=DEV)
        a = torch.cat([torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.3]) if aff else None
        dW = torch.zeros(N, K, device=DEV)
        keep.append((dY, X, a, dW))
        items.append((dY.data_ptr(), X.data_ptr(), a.data_ptr() if aff else None, dW.data_ptr(), P, K, N, K))
    arr = (L.WgradItem * len(items))(*items)
    flop = sum(2.0 * P * K * N for P, K, N, _ in shapes)

    def group():
        L.check(lib.gb_gemm_wgrad_group(ctypes.cast(arr, ctypes.c_void_p), len(items), opts, None), "group")

    def singles():
        for (dY, X, a, dW), (P, K, N, _) in zip(keep, shapes):
            L.check(lib.gb_gemm_wgrad(L.ptr(dY), L.ptr(X), L.ptr(a), L.ptr(dW), P, K, N, opts, None), "wgrad")
    tg, ts = bench(group), bench(singles)
    print("%-30s %3d products %6.1f GFLOP: grouped %7.1f us = %5.1f TF/s (%.2f of 157.3) | single calls %7.1f us = %5.1f TF/s"
          % (name, len(shapes), flop / 1e9, tg, flop / tg / 1e6, flop / tg / 1e6 / 157.3, ts, flop / ts / 1e6))
