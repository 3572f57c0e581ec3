"""gb_la_pool_bwd in isolation on the four stage shapes of the DRP backbone (valid out / arg from gb_la_pool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib as L, fused_mlp, pointnet2_utils as pu
from graspbalance_amd.scene import make_batch
DEV = "cuda:0"; lib = L.lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
cloud = torch.from_numpy(make_batch(range(4), 20000)).to(DEV)
out_line = []
for (npts, C, radius, ns) in [(2048, 128, 0.08, 64), (1024, 256, 0.2, 32), (512, 256, 0.4, 16), (256, 256, 0.6, 16)]:
    p = cloud[:, torch.randperm(20000, device=DEV)[:npts]].contiguous()
    idx = pu.ball_query(radius, ns, p, p)
    geo = fused_mlp.LocalGeometry(p, p, idx, mode=0)
    rows = 4 * npts
    G = torch.randn(rows, C, device=DEV); Wx = torch.randn(C, 3, device=DEV)
    ab = torch.cat([torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1, torch.zeros(C, device=DEV), torch.ones(C, device=DEV)])
    out = torch.empty(rows, C, device=DEV); arg = torch.empty(rows, C, dtype=torch.int32, device=DEV)
    L.check(lib.gb_la_pool(L.ptr(G), L.ptr(geo.xyz), L.ptr(geo.centres), L.ptr(geo.idx), L.ptr(Wx), L.ptr(ab), L.ptr(out), L.ptr(arg),
                           geo.b, geo.n, geo.m, geo.ns, C, geo.mode, geo.scale, None), "pool")
    dout = torch.randn(rows, C, device=DEV)
    sg = torch.zeros(rows, C, device=DEV); red = torch.zeros(5 * C, dtype=torch.float64, device=DEV)
    def bwd():
        L.check(lib.gb_la_pool_bwd(L.ptr(dout), L.ptr(out), L.ptr(arg), L.ptr(G), L.ptr(geo.xyz), L.ptr(geo.centres), L.ptr(geo.idx),
                                   L.ptr(Wx), L.ptr(ab), L.ptr(sg), L.ptr(red), geo.b, geo.n, geo.m, geo.ns, C, geo.mode, geo.scale, None), "bwd")
    def fwd():
        lib.gb_la_pool(L.ptr(G), L.ptr(geo.xyz), L.ptr(geo.centres), L.ptr(geo.idx), L.ptr(Wx), L.ptr(ab), L.ptr(out), L.ptr(arg),
                       geo.b, geo.n, geo.m, geo.ns, C, geo.mode, geo.scale, None)
    live = float((out > 0).float().mean())
    out_line.append("R=%d C=%d ns=%d live %.2f: bwd %.1f us, fwd %.1f us" % (rows, C, ns, live, timeit(bwd), timeit(fwd)))
print(" | ".join(out_line))
