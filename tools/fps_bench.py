"""FPS 20000 -> 2048 on 4 clouds: gb_fps vs Morton keys + sort + gb_fps_pruned, timed separately."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib
from graspbalance_amd.scene import make_batch
B, N, m = int(os.environ.get('B', 4)), int(os.environ.get('N', 20000)), int(os.environ.get('M', 2048))
scratch = torch.empty(B, N, 4, device='cuda') if N > 20480 else None
xyz = torch.from_numpy(make_batch(range(B), N)).cuda()
idx = torch.zeros(B, m, dtype=torch.int32, device="cuda")
keys = torch.empty(B, N, dtype=torch.int32, device="cuda")
L = _lib.lib()
tmp = torch.empty(B, N, device='cuda')
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
flags = _lib.FPS_SKIP_NEAR_ORIGIN | _lib.FPS_TIE_TREE512
print("gb_fps                 %8.1f us" % timeit(lambda: _lib.check(L.gb_fps(_lib.ptr(xyz), _lib.ptr(tmp.fill_(1e10)), _lib.ptr(idx), B, N, m, flags, None), 'gb_fps')))
ref = idx.clone()
print("morton keys            %8.1f us" % timeit(lambda: L.gb_fps_morton_keys(_lib.ptr(xyz), _lib.ptr(keys), B, N, None)))
print("argsort + int32        %8.1f us" % timeit(lambda: torch.argsort(keys, dim=1).to(torch.int32)))
perm = torch.argsort(keys, dim=1).to(torch.int32)
for lname, layout in _lib.FPS_LAYOUT.items():
    if N > 20480 and lname not in ("auto", "r4"):
        continue
    wsb = torch.empty(B, N, 4, device="cuda")
    for oname, fn in (("cell", L.gb_fps_cell_order), ("rows", lambda x, p, b, n, st: L.gb_fps_row_order_ws(x, p, _lib.ptr(wsb), b, n, st))):
        pc = torch.empty(B, N, dtype=torch.int32, device="cuda")
        fn(_lib.ptr(xyz), _lib.ptr(pc), B, N, None)
        t = timeit(lambda: L.gb_fps_pruned(_lib.ptr(xyz), _lib.ptr(pc), None, _lib.ptr(idx), B, N, m, flags | layout, _lib.ptr(scratch), None))
        assert torch.equal(idx, ref), (lname, oname)
        print("gb_fps_pruned %-5s order %-5s %8.1f us  = %.3f us/iteration" % (lname, oname, t, t / max(m - 1, 1)))
print("cell order             %8.1f us" % timeit(lambda: L.gb_fps_cell_order(_lib.ptr(xyz), _lib.ptr(keys), B, N, None)))
wsb = torch.empty(B, N, 4, device="cuda")
print("row order              %8.1f us" % timeit(lambda: L.gb_fps_row_order_ws(_lib.ptr(xyz), _lib.ptr(keys), _lib.ptr(wsb), B, N, None)))
print("gb_fps_pruned (morton) %8.1f us" % timeit(lambda: L.gb_fps_pruned(_lib.ptr(xyz), _lib.ptr(perm), None, _lib.ptr(idx), B, N, m, flags, _lib.ptr(scratch), None)))
assert torch.equal(idx, ref)
ident = torch.arange(N, device="cuda", dtype=torch.int32).repeat(B, 1).contiguous()
print("gb_fps_pruned (ident.) %8.1f us" % timeit(lambda: L.gb_fps_pruned(_lib.ptr(xyz), _lib.ptr(ident), None, _lib.ptr(idx), B, N, m, flags, _lib.ptr(scratch), None)))
assert torch.equal(idx, ref)
