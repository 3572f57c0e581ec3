import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib as L
from graspbalance_amd.scene import make_batch
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
xyz = torch.from_numpy(make_batch(range(B), 20000)).to(dev)
lib = L.lib()
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (n, m) in [(20000, 2048), (2048, 1024), (1024, 512), (512, 256)]:
    x = xyz[:, :n].contiguous()
    idx = torch.zeros(B, m, dtype=torch.int32, device=dev)
    t = timeit(lambda: lib.gb_fps(L.ptr(x), None, L.ptr(idx), B, n, m, 0x11, None))
    print("fps n=%d m=%d: %.1f us (%.3f us/iter)" % (n, m, t, t / (m - 1)))
