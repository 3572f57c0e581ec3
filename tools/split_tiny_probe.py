"""GPU: the three-way bf16 split (GB_PREC_F32_SPLIT3) against fp32 MFMA and fp64 as the operands' magnitude sinks towards
bf16's denormal range (the matrix cores flush bf16 denormals: the lower slices of operands below ~2^-110 are lost)."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from graspbalance_amd import _lib as L   # noqa: E402

DEV = "cuda:0"
lib = L.lib()
ws = torch.empty(L.GEMM_SCRATCH_BYTES, dtype=torch.uint8, device=DEV)
P, K, N = 70000, 128, 256
g = torch.Generator(device=DEV).manual_seed(1)
X = torch.randn(P, K, device=DEV, generator=g)
W = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
for sx, sw in ((-20, 20), (-60, 60), (-100, 100), (-110, 100), (-115, 100), (-120, 110), (-126, 120)):
    Xt, Wt = X * 2.0 ** sx, W * 2.0 ** sw
    ref = Xt.double() @ Wt.double().t()
    out = []
    for prec in (0, 2):
        Y = torch.empty(P, N, device=DEV)
        o = ctypes.pointer(L.GemmOpts(prec, 0, ws.data_ptr(), ws.numel()))
        L.check(lib.gb_gemm_fwd(L.ptr(Xt), L.ptr(Wt), None, L.ptr(Y), None, 1, P, K, N, None, o, None), "fwd")
        torch.cuda.synchronize()
        out.append((float((Y.double() - ref).abs().max()) / float(ref.abs().max()), int((~torch.isfinite(Y)).sum())))
    print("X x 2^%d, W x 2^%d: fp32 MFMA %.2e (%d non-finite) | split %.2e (%d non-finite)" % (sx, sw, *out[0], *out[1]))
