"""Run each row-streaming product twice on the same inputs (GB_PREC_F32_SPLIT3 and fp32) and compare the outputs bit for bit."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib as L
lib = L.lib()
DEV = "cuda:0"
for prec in (0, 2):
    opts = ctypes.pointer(L.GemmOpts(prec, 0, None, 0, None))
    for P, K, N in [(400000, 128, 256), (131072, 128, 128), (200000, 64, 128), (70001, 256, 128)]:
        torch.manual_seed(P)
        X = torch.randn(P, K, device=DEV); W = torch.randn(N, K, device=DEV) * 0.1
        aff = torch.cat([torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1])
        outs = []
        for rep in range(3):
            Y = torch.full((P, N), 7.0, device=DEV); st = torch.zeros(32 * 2 * N, dtype=torch.float64, device=DEV)
            L.check(lib.gb_gemm_fwd(L.ptr(X), L.ptr(W), L.ptr(aff), L.ptr(Y), L.ptr(st), 32, P, K, N, None, opts, None), "fwd")
            torch.cuda.synchronize()
            outs.append((Y.clone(), st.view(32, -1).sum(0)))
        same = all(torch.equal(outs[0][0], o[0]) for o in outs[1:])
        print("prec %d fwd %7d x %3d -> %3d: Y identical %s, stats rel diff %.1e" % (prec, P, K, N, same,
              float((outs[0][1] - outs[1][1]).abs().max() / outs[0][1].abs().max())))
        dY = torch.randn(P, N, device=DEV); yp = torch.randn(P, K, device=DEV)
        ab = torch.cat([torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1, torch.zeros(K, device=DEV), torch.ones(K, device=DEV)])
        outs = []
        for rep in range(3):
            dX = torch.full((P, K), 7.0, device=DEV); ds = torch.zeros(32 * 2 * K, dtype=torch.float64, device=DEV)
            L.check(lib.gb_gemm_dgrad(L.ptr(dY), L.ptr(W), L.ptr(dX), L.ptr(yp), L.ptr(ab), L.ptr(ds), 32, P, K, N, None, None, None, opts, None), "dgrad")
            torch.cuda.synchronize()
            outs.append((dX.clone(), ds.view(32, -1).sum(0)))
        same = all(torch.equal(outs[0][0], o[0]) for o in outs[1:])
        print("prec %d dgrad %6d x %3d -> %3d: dX identical %s, sums rel diff %.1e" % (prec, P, N, K, same,
              float((outs[0][1] - outs[1][1]).abs().max() / outs[0][1].abs().max())))
