"""Kernel-only timing of tall wgrads through a given build of the library.  usage: wg_probe.py <lib.so> [iters]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
so = ctypes.CDLL(os.path.abspath(sys.argv[1]))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P_ = ctypes.c_void_p
so.gb_gemm_wgrad.argtypes = [P_, P_, P_, P_, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, P_, P_]
so.gb_gemm_wgrad_gen3.argtypes = [P_, P_, P_, P_, P_, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, P_, P_]
dev = "cuda:0"
import ctypes as _c


class GemmOpts(_c.Structure):
    _fields_ = [("precision", _c.c_int), ("reserved_cus", _c.c_int), ("scratch", _c.c_void_p), ("scratch_bytes", _c.c_ulonglong),
                ("rows_dev", _c.c_void_p), ("flags", _c.c_int)]


flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # 4 = GB_GEMM_NO_DIRECT
prec = int(sys.argv[4]) if len(sys.argv) > 4 else 0        # 2 = GB_PREC_F32_SPLIT3
opts = _c.pointer(GemmOpts(prec, 0, None, 0, None, flags))
shapes = [(400000, 128, 256, "aff"), (400000, 64, 128, "gen3"), (524288, 64, 128, "aff"), (524288, 64, 64, "gen3"),
          (131072, 128, 256, "aff"), (131072, 128, 128, "aff"), (400000, 128, 256, "plain"), (65536, 128, 256, "aff"),
          (32768, 128, 256, "aff"), (32768, 128, 128, "aff"), (16384, 128, 256, "aff"), (32768, 64, 128, "aff")]
for P, K, N, mode in shapes:
    dY = torch.randn(P, N, device=dev)
    X = torch.randn(P, K, device=dev)
    x0 = torch.randn(P, 3, device=dev)
    w1 = torch.randn(K, 3, device=dev)
    aff = torch.randn(2 * K, device=dev)
    dW = torch.zeros(N, K, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        if mode == "gen3":
            rc = so.gb_gemm_wgrad_gen3(dY.data_ptr(), x0.data_ptr(), w1.data_ptr(), aff.data_ptr(), dW.data_ptr(), P, K, N, opts, st)
        else:
            rc = so.gb_gemm_wgrad(dY.data_ptr(), X.data_ptr(), aff.data_ptr() if mode == "aff" else None, dW.data_ptr(), P, K, N, opts, st)
        assert rc == 0, rc
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / iters * 1e3
    print("%-28s %8.1f us  %6.1f TF/s  %.2f of peak" % ("%d x %d x %d %s" % (P, K, N, mode), us, 2.0 * P * K * N / us / 1e6, 2.0 * P * K * N / us / 1e6 / 157.3), flush=True)
