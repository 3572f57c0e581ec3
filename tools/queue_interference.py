"""What does a kernel resident on a second queue cost the main queue: per kernel boundary, or per unit of work?
A stream of MANY SHORT kernels and a stream of FEW LONG kernels of the same total duration, each captured as a linear
graph, replayed alone and beside the first-level sampling (one workgroup per cloud, ~2.1 ms) on a side stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import pointnet2_utils
from graspbalance_amd.scene import make_batch
dev = torch.device("cuda", 0)
clouds = torch.from_numpy(make_batch([0, 1, 2, 3], 20000)).to(dev)
side = torch.cuda.Stream()
small = torch.zeros(4096, device=dev)
big_a = torch.randn(8192, 8192, device=dev)
big_b = torch.randn(8192, 8192, device=dev)
def short_work():
    for _ in range(600):
        small.add_(1.0)
def long_work():
    for _ in range(3):
        torch.mm(big_a, big_b)
def mid_work():
    x = torch.empty(64 << 20, device=dev)
    def fn():
        for _ in range(40):
            x.add_(1.0)
    return fn
def capture(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    return g
def timeit(fn, n=20):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def fps():
    with torch.cuda.stream(side), torch.no_grad():
        pointnet2_utils.furthest_point_sample(clouds, 2048)
for name, work in (("600 short kernels", short_work), ("3 long GEMMs", long_work), ("40 HBM-bound passes of 256 MB", mid_work())):
    g = capture(work)
    alone = timeit(g.replay)
    def beside():
        cur = torch.cuda.current_stream(); fps(); g.replay(); cur.wait_stream(side)
    b = timeit(beside)
    print("%-32s alone %.3f ms, beside the sampling %.3f ms (+%.2f)" % (name, alone, b, b - alone), flush=True)
print("sampling alone %.3f ms" % timeit(lambda: (fps(), torch.cuda.current_stream().wait_stream(side))))
# ---- stream priorities: the work on a HIGH-priority stream, the sampling on a normal / low one
hp = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)
print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a", flush=True)
for name, work in (("3 long GEMMs", long_work), ("40 HBM-bound passes of 256 MB", mid_work())):
    g = capture(work)
    def alone_hp():
        with torch.cuda.stream(hp): g.replay()
    def beside_hp():
        with torch.cuda.stream(lo), torch.no_grad():
            pointnet2_utils.furthest_point_sample(clouds, 2048)
        with torch.cuda.stream(hp):
            g.replay()
            hp.wait_stream(lo)
    a = timeit(alone_hp); b = timeit(beside_hp)
    print("%-32s on a high-priority stream: alone %.3f ms, beside the sampling %.3f ms (+%.2f)" % (name, a, b, b - a), flush=True)
# ---- does it matter WHICH hardware queue the side stream maps to?  (streams are dealt over the queues round-robin)
g = capture(mid_work())
base = timeit(g.replay)
streams = [torch.cuda.Stream() for _ in range(10)]
for i, s_i in enumerate(streams):
    def beside_i():
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(s_i), torch.no_grad():
            pointnet2_utils.furthest_point_sample(clouds, 2048)
        g.replay(); cur.wait_stream(s_i)
    print("side stream #%d: HBM passes beside the sampling %.3f ms (alone %.3f)" % (i, timeit(beside_i, 10), base), flush=True)
# ---- a one-thread sleeping kernel of the same duration instead of the sampling
for cyc in (2000000, 4000000):
    def sl():
        with torch.cuda.stream(side): torch.cuda._sleep(cyc)
        torch.cuda.current_stream().wait_stream(side)
    def beside_sl():
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(side): torch.cuda._sleep(cyc)
        g.replay(); cur.wait_stream(side)
    print("one-thread sleep %d: alone %.3f ms; HBM passes beside it %.3f ms (alone %.3f)" % (cyc, timeit(sl, 10), timeit(beside_sl, 10), base), flush=True)
# ---- which PART of the sampling call slows the main queue?  (pre-allocated buffers, direct C-ABI calls on the side stream)
from graspbalance_amd import _lib as L
xyz = clouds[..., 0:3].contiguous()
b_, n_, m_ = 4, 20000, 2048
perm = torch.empty((b_, n_), dtype=torch.int32, device=dev)
temp = torch.full((b_, n_), 1e10, device=dev)
outi = torch.zeros((b_, m_), dtype=torch.int32, device=dev)
flags = 0
import ctypes
sstream = ctypes.c_void_p(side.cuda_stream)
L.check(L.lib().gb_fps_cell_order(L.ptr(xyz), L.ptr(perm), b_, n_, L.current_stream(dev)), "order")
torch.cuda.synchronize()
def k_pruned():
    L.check(L.lib().gb_fps_pruned(L.ptr(xyz), L.ptr(perm), L.ptr(temp), L.ptr(outi), b_, n_, m_, flags, None, sstream), "pruned")
def k_plain():
    L.check(L.lib().gb_fps(L.ptr(xyz), L.ptr(temp), L.ptr(outi), b_, n_, m_, flags, sstream), "fps")
def k_order():
    for _ in range(20):
        L.check(L.lib().gb_fps_cell_order(L.ptr(xyz), L.ptr(perm), b_, n_, sstream), "order")
def k_fill():
    with torch.cuda.stream(side):
        for _ in range(50):
            temp.fill_(1e10)
for name, k in (("fps_pruned kernel only", k_pruned), ("fps_reg kernel (un-pruned) only", k_plain), ("20 x cell-order kernel", k_order), ("50 x fill", k_fill)):
    def alone_k():
        k(); torch.cuda.current_stream().wait_stream(side)
    def beside_k():
        cur = torch.cuda.current_stream(); k(); g.replay(); cur.wait_stream(side)
    print("%-34s alone %.3f ms; HBM passes beside it %.3f ms (alone %.3f)" % (name, timeit(alone_k, 10), timeit(beside_k, 10), base), flush=True)
# ---- bisect the wrapper
def k_seq():      # cell order + pruned kernel, everything pre-allocated
    L.check(L.lib().gb_fps_cell_order(L.ptr(xyz), L.ptr(perm), b_, n_, sstream), "order")
    k_pruned()
def k_seq_fill():  # + the two fills the wrapper does
    with torch.cuda.stream(side):
        temp.fill_(1e10); outi.zero_()
    k_seq()
def k_libfps():   # _lib.fps: allocates perm on the side stream
    with torch.cuda.stream(side):
        L.check(L.fps(xyz, temp, outi, b_, n_, m_, flags, sstream), "fps")
def k_wrapper():  # the python wrapper: allocates temp / output / perm, fills
    with torch.cuda.stream(side), torch.no_grad():
        pointnet2_utils.furthest_point_sample(xyz, 2048)
def k_wrapper_copy():
    with torch.cuda.stream(side), torch.no_grad():
        outi.copy_(pointnet2_utils.furthest_point_sample(clouds[..., 0:3].contiguous(), 2048))
for name, k in (("order + pruned, pre-allocated", k_seq), ("... + fills", k_seq_fill), ("_lib.fps (allocates perm)", k_libfps),
                ("python wrapper", k_wrapper), ("wrapper + contiguous + copy", k_wrapper_copy)):
    def alone_k():
        k(); torch.cuda.current_stream().wait_stream(side)
    def beside_k():
        cur = torch.cuda.current_stream(); k(); g.replay(); cur.wait_stream(side)
    print("%-34s alone %.3f ms; HBM passes beside it %.3f ms (alone %.3f)" % (name, timeit(alone_k, 10), timeit(beside_k, 10), base), flush=True)
# ---- the un-pruned kernel (every point updated every iteration: no ballot / readlane / LDS key table) at a length that
#      fits under the main work, with a fresh min-distance array each time
out_small = torch.zeros((b_, 1100), dtype=torch.int32, device=dev)
def k_plain_short():
    with torch.cuda.stream(side):
        temp.fill_(1e10)
    L.check(L.lib().gb_fps(L.ptr(xyz), L.ptr(temp), L.ptr(out_small), b_, n_, 1100, flags, sstream), "fps")
def k_pruned_fresh_short():
    with torch.cuda.stream(side):
        temp.fill_(1e10)
    L.check(L.lib().gb_fps_pruned(L.ptr(xyz), L.ptr(perm), L.ptr(temp), L.ptr(out_small), b_, n_, 1100, flags, None, sstream), "pruned")
for name, k in (("un-pruned kernel, 1100 picks, fresh", k_plain_short), ("pruned kernel, 1100 picks, fresh", k_pruned_fresh_short)):
    def alone_k():
        k(); torch.cuda.current_stream().wait_stream(side)
    def beside_k():
        cur = torch.cuda.current_stream(); k(); g.replay(); cur.wait_stream(side)
    print("%-38s alone %.3f ms; HBM passes beside it %.3f ms (alone %.3f)" % (name, timeit(alone_k, 10), timeit(beside_k, 10), base), flush=True)
# ---- where in the sampling does the interference come from?  pruned kernel, fresh array, growing number of picks
for mm in (300, 600, 1100, 1400, 1700, 2048):
    out_m = torch.zeros((b_, mm), dtype=torch.int32, device=dev)
    def k_m():
        with torch.cuda.stream(side):
            temp.fill_(1e10)
        L.check(L.lib().gb_fps_pruned(L.ptr(xyz), L.ptr(perm), L.ptr(temp), L.ptr(out_m), b_, n_, mm, flags, None, sstream), "pruned")
    def alone_k():
        k_m(); torch.cuda.current_stream().wait_stream(side)
    def beside_k():
        cur = torch.cuda.current_stream(); k_m(); g.replay(); cur.wait_stream(side)
    a = timeit(alone_k, 10); bb = timeit(beside_k, 10)
    print("pruned kernel, %4d picks: alone %.3f ms; HBM passes beside it %.3f ms (+%.2f)" % (mm, a, bb, bb - base), flush=True)
# ---- is it the LENGTH of one kernel's residency?  two back-to-back launches of 1024 picks against one of 2048
out_h = torch.zeros((b_, 1024), dtype=torch.int32, device=dev)
def k_two():
    for _ in range(2):
        with torch.cuda.stream(side):
            temp.fill_(1e10)
        L.check(L.lib().gb_fps_pruned(L.ptr(xyz), L.ptr(perm), L.ptr(temp), L.ptr(out_h), b_, n_, 1024, flags, None, sstream), "pruned")
def k_four():
    out_q = out_h[:, :512].contiguous()
    for _ in range(4):
        with torch.cuda.stream(side):
            temp.fill_(1e10)
        L.check(L.lib().gb_fps_pruned(L.ptr(xyz), L.ptr(perm), L.ptr(temp), L.ptr(out_q), b_, n_, 512, flags, None, sstream), "pruned")
for name, k in (("2 x 1024 picks back to back", k_two), ("4 x 512 picks back to back", k_four)):
    def alone_k():
        k(); torch.cuda.current_stream().wait_stream(side)
    def beside_k():
        cur = torch.cuda.current_stream(); k(); g.replay(); cur.wait_stream(side)
    a = timeit(alone_k, 10); bb = timeit(beside_k, 10)
    print("%-30s alone %.3f ms; HBM passes beside it %.3f ms (+%.2f)" % (name, a, bb, bb - base), flush=True)
# ---- does a pause reset it?  half the picks beside the first half of the work, the other half beside the second half
def half_work():
    x = torch.empty(64 << 20, device=dev)
    def fn():
        for _ in range(20):
            x.add_(1.0)
    return fn
g_half = capture(half_work())
base_half = timeit(g_half.replay)
def k_half():
    with torch.cuda.stream(side):
        temp.fill_(1e10)
    L.check(L.lib().gb_fps_pruned(L.ptr(xyz), L.ptr(perm), L.ptr(temp), L.ptr(out_h), b_, n_, 1024, flags, None, sstream), "pruned")
def split_with_pause():
    cur = torch.cuda.current_stream()
    k_half(); g_half.replay()
    side.wait_stream(cur)          # the second half starts when the first half of the work is done
    k_half(); g_half.replay()
    cur.wait_stream(side)
def two_halves_alone():
    g_half.replay(); g_half.replay()
print("two halves of the work alone %.3f ms; each half beside 1024 picks, second half started after a pause: %.3f ms"
      % (timeit(two_halves_alone, 10), timeit(split_with_pause, 10)), flush=True)
