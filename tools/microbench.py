"""Per-kernel timing of the C-ABI entry points with HIP events (run on the GPU box)."""
import ctypes
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib
from graspbalance_amd.scene import make_batch
from graspbalance_amd.pointnet2 import _ext

dev = "cuda:0"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    xyz = torch.from_numpy(make_batch(range(B), 20000)).to(dev)
    res = {}
    L = _lib.lib()
    st = None
    for (n, m) in [(20000, 2048), (20000, 1024), (2048, 1024), (1024, 512), (512, 256)]:
        x = xyz[:, :n].contiguous()
        idx = torch.zeros(B, m, dtype=torch.int32, device=dev)
        for name, flags in [("tree512+skip", 0x11), ("lowest", 0)]:
            t = timeit(lambda: L.gb_fps(_lib.ptr(x), None, _lib.ptr(idx), B, n, m, flags, st))
            res["fps_%d_%d_%s" % (n, m, name)] = t
            print("fps n=%d m=%d %s: %.1f us  (%.3f us/iter)  stream-model %.2f TB/s" % (n, m, name, t, t / (m - 1), B * (20.0 * n * (m - 1) + 4 * m) / t / 1e6))
    inds = _ext.furthest_point_sampling(xyz, 2048)
    new_xyz = torch.gather(xyz, 1, inds.long()[:, :, None].expand(-1, -1, 3)).contiguous()
    for (m, r, ns) in [(2048, 0.04, 64), (1024, 0.04, 32)]:
        nx = new_xyz[:, :m].contiguous()
        idx = torch.zeros(B, m, ns, dtype=torch.int32, device=dev)
        sc = torch.zeros(B, m, dtype=torch.int32, device=dev)
        L.gb_ball_query(_lib.ptr(nx), _lib.ptr(xyz), _lib.ptr(idx), _lib.ptr(sc), B, 20000, m, r, ns, st)
        P = int(sc.sum().item())
        t = timeit(lambda: L.gb_ball_query(_lib.ptr(nx), _lib.ptr(xyz), _lib.ptr(idx), None, B, 20000, m, r, ns, st))
        bytes_alg = 12.0 * P + 12 * B * m + 4 * B * m * ns
        res["ball_%d_%g_%d" % (m, r, ns)] = t
        print("ball m=%d r=%g ns=%d: %.1f us  scanned pairs %d (%.1f%% of full)  stream-model %.2f TB/s" % (m, r, ns, t, P, 100.0 * P / (B * m * 20000), bytes_alg / t / 1e6))
    # cylinder: 16 separate vs fused
    m, ns = 1024, 64
    nx = new_xyz[:, :m].contiguous()
    rot = torch.eye(3, device=dev).view(1, 1, 9).repeat(B, m, 1).contiguous()
    radii, hmaxs = [0.02, 0.04, 0.06, 0.08], [0.01, 0.02, 0.03, 0.04]
    idx = torch.zeros(16, B, m, ns, dtype=torch.int32, device=dev)

    def sep():
        for ir, r in enumerate(radii):
            for ih, h in enumerate(hmaxs):
                L.gb_cylinder_query(_lib.ptr(nx), _lib.ptr(xyz), _lib.ptr(rot), _lib.ptr(idx[ir * 4 + ih]), None, B, 20000, m, r, -0.02, h, ns, st)
    ra = (ctypes.c_float * 4)(*radii)
    ha = (ctypes.c_float * 4)(*hmaxs)

    def fused():
        L.gb_cylinder_query_multi(_lib.ptr(nx), _lib.ptr(xyz), _lib.ptr(rot), _lib.ptr(idx), B, 20000, m, ctypes.cast(ra, ctypes.c_void_p), 4, -0.02, ctypes.cast(ha, ctypes.c_void_p), 4, ns, st)
    res["cyl16_separate"] = timeit(sep)
    res["cyl16_fused"] = timeit(fused)
    print("cylinder x16 separate: %.1f us, fused: %.1f us" % (res["cyl16_separate"], res["cyl16_fused"]))
    # group fwd/bwd at the biggest instance (B,128,2048,64)
    feats = torch.randn(B, 128, 2048, device=dev)
    gidx = torch.randint(0, 2048, (B, 2048, 64), dtype=torch.int32, device=dev)
    out = torch.empty(B, 128, 2048, 64, device=dev)
    t = timeit(lambda: L.gb_group(_lib.ptr(feats), _lib.ptr(gidx), _lib.ptr(out), B, 128, 2048, 2048, 64, st))
    nbytes = out.numel() * 4 + gidx.numel() * 4
    res["group_fwd"] = t
    print("group fwd (B,128,2048,64): %.1f us  %.2f TB/s written" % (t, nbytes / t / 1e6))
    gin = torch.zeros(B, 128, 2048, device=dev)
    t = timeit(lambda: L.gb_group_grad(_lib.ptr(out), _lib.ptr(gidx), _lib.ptr(gin), B, 128, 2048, 2048, 64, st))
    res["group_bwd"] = t
    print("group bwd: %.1f us  %.2f TB/s read" % (t, nbytes / t / 1e6))
    unknown = xyz
    known = new_xyz[:, :1024].contiguous()
    d2 = torch.empty(B, 20000, 3, device=dev)
    i3 = torch.empty(B, 20000, 3, dtype=torch.int32, device=dev)
    res["three_nn_20000x1024"] = timeit(lambda: L.gb_three_nn(_lib.ptr(unknown), _lib.ptr(known), _lib.ptr(d2), _lib.ptr(i3), B, 20000, 1024, st))
    print("three_nn 20000x1024: %.1f us" % res["three_nn_20000x1024"])
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/microbench_B%d.json" % B, "w"), indent=1)


if __name__ == "__main__":
    main()
