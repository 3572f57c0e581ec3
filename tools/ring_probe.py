"""Per-shape timing of the few-row GEMM kernels: csrc/gemm_ring.hip (LDS-DMA ring) against csrc/gemm_cl.hip
(GbGemmOpts.flags = GB_GEMM_NO_RING), the step's shapes, HIP events around 20 back-to-back launches (operands hot)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib as L
lib = L.lib()
dev = "cuda:0"
ws = torch.empty(L.GEMM_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
def opts(flags):
    return ctypes.pointer(L.GemmOpts(L.PREC_F32, 0, ws.data_ptr(), ws.numel(), None, flags))
SHAPES = [(4096, 1024, 256), (4096, 256, 1024), (4096, 256, 256), (8192, 512, 128), (8192, 128, 512), (8192, 128, 128),
          (2048, 1024, 256), (2048, 256, 1024), (2048, 256, 256), (1024, 1024, 256), (1024, 256, 1024), (1024, 256, 256),
          (16384, 1024, 256), (16384, 256, 128), (16384, 128, 128), (4096, 256, 304), (32768, 256, 128)]
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("%-22s %9s %9s %9s | %9s %9s %9s   (us: ring / old)" % ("P,K,N", "fwd+st", "dgrad+bn", "wgrad", "fwd", "dgrad", "wgrad"))
tot = [0.0] * 6
for P, K, N in SHAPES:
    X = torch.randn(P, K, device=dev); W = torch.randn(N, K, device=dev); Y = torch.empty(P, N, device=dev)
    dY = torch.randn(P, N, device=dev); dX = torch.empty(P, K, device=dev); dW = torch.zeros(N, K, device=dev)
    aff = torch.randn(2 * K, device=dev); st = torch.zeros(2 * N, dtype=torch.float64, device=dev)
    ab = torch.randn(4 * K, device=dev); dst = torch.zeros(2 * K, dtype=torch.float64, device=dev)
    row = []
    for flags in (0, L.GEMM_NO_RING):
        o = opts(flags)
        row.append(bench(lambda: lib.gb_gemm_fwd(L.ptr(X), L.ptr(W), L.ptr(aff), L.ptr(Y), L.ptr(st), 1, P, K, N, None, o, None)))
        row.append(bench(lambda: lib.gb_gemm_dgrad(L.ptr(dY), L.ptr(W), L.ptr(dX), L.ptr(X), L.ptr(ab), L.ptr(dst), 1, P, K, N, None, None, None, o, None)))
        row.append(bench(lambda: lib.gb_gemm_wgrad(L.ptr(dY), L.ptr(X), L.ptr(aff), L.ptr(dW), P, K, N, o, None)))
    tot = [a + b for a, b in zip(tot, row)]
    fl = 2.0 * P * K * N
    print("%-22s %9.1f %9.1f %9.1f | %9.1f %9.1f %9.1f   TF/s ring %5.1f %5.1f %5.1f" % ((str((P, K, N)),) + tuple(row) + tuple(fl / (t * 1e-6) / 1e12 for t in row[:3])))
print("%-22s %9.1f %9.1f %9.1f | %9.1f %9.1f %9.1f" % (("sum",) + tuple(tot)))
