"""First-level FPS (cell-order counting sort + pruned kernel): one sample per selection round vs multi-pick, same outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib
from graspbalance_amd.scene import make_batch
L = _lib.lib()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (B, N, m) in [(4, 20000, 2048), (4, 20000, 1024), (2, 8192, 2048), (8, 16000, 1024)]:
    xyz = torch.from_numpy(make_batch(range(B), N)).cuda()
    perm = torch.empty(B, N, dtype=torch.int32, device="cuda")
    _lib.check(L.gb_fps_cell_order(_lib.ptr(xyz), _lib.ptr(perm), B, N, None), "order")
    res = {}
    for name, extra in (("single", 0), ("multi", _lib.FPS_MULTI_PICK)):
        for tie in (_lib.FPS_TIE_TREE512, _lib.FPS_TIE_LOWEST):
            idx = torch.zeros(B, m, dtype=torch.int32, device="cuda")
            flags = _lib.FPS_SKIP_NEAR_ORIGIN | tie | extra
            t = timeit(lambda: _lib.check(L.gb_fps_pruned(_lib.ptr(xyz), _lib.ptr(perm), None, _lib.ptr(idx), B, N, m, flags, None, None), "fps"))
            res[(name, tie)] = (t, idx.clone())
    for tie in (_lib.FPS_TIE_TREE512, _lib.FPS_TIE_LOWEST):
        same = torch.equal(res[("single", tie)][1], res[("multi", tie)][1])
        print("B=%d N=%d m=%d tie=%#x: single %.0f us, multi %.0f us, identical=%s" % (B, N, m, tie, res[("single", tie)][0], res[("multi", tie)][0], same))
