"""How much does the next batch's first-level sampling on the side stream cost the step it runs beside?  The same
captured graph replayed (a) with the sampling beside it (the shipped loop), (b) with no sampling launched at all (the
floor: what the step's own kernels take), (c) with the sampling launched and finished BEFORE the replay (serial)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import pointnet2_utils
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
dev = torch.device("cuda", 0)
tr = Trainer(dev)
batch = tr.resident(make_training_batch(range(4), 20000, device=dev))
for _ in range(3):
    tr.train_step(batch, next_batch=batch)
g = [v for k, v in tr._graphs.items() if k[1]][0]   # (captured with GB_SAMPLE_AT's reserved-CU choice)
st, side = tr._static, tr.prefetch.side
HOST = [0.0]
def timeit(fn, n=20):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    HOST[0] = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def fps():
    with torch.cuda.stream(side), torch.no_grad():
        st.inds_ring[0].copy_(pointnet2_utils.furthest_point_sample(st.next_clouds[..., 0:3].contiguous(), 2048))
def replay():
    g.fwd.replay(); g.bwd.replay()
def at_start():
    cur = torch.cuda.current_stream(); side.wait_stream(cur); fps(); replay(); cur.wait_stream(side)
def between():
    cur = torch.cuda.current_stream(); g.fwd.replay(); side.wait_stream(cur); fps(); g.bwd.replay(); cur.wait_stream(side)
def serial():
    cur = torch.cuda.current_stream(); side.wait_stream(cur); fps(); cur.wait_stream(side); replay()
def only_fps():
    cur = torch.cuda.current_stream(); side.wait_stream(cur); fps(); cur.wait_stream(side)
for name, fn in (("sampling beside the forward", at_start), ("sampling beside the backward", between), ("graphs alone", replay),
                 ("sampling, then graphs", serial), ("sampling alone", only_fps)):
    print("%-30s %.3f ms (host %.2f)" % (name, timeit(fn), HOST[0]), flush=True)
# ---- is it the sampling kernel, or ANY resident kernel on a second queue?  A one-wave spin kernel of the same length:
for cyc in (200000, 400000):
    def spin_only():
        cur = torch.cuda.current_stream(); side.wait_stream(cur)
        with torch.cuda.stream(side): torch.cuda._sleep(cyc)
        cur.wait_stream(side)
    def spin_beside():
        cur = torch.cuda.current_stream(); side.wait_stream(cur)
        with torch.cuda.stream(side): torch.cuda._sleep(cyc)
        replay(); cur.wait_stream(side)
    a = timeit(spin_only); b = timeit(spin_beside)
    print("one-wave spin %7d: alone %.3f ms, graphs beside it %.3f ms (host %.2f)" % (cyc, a, b, HOST[0]), flush=True)
def ev_only():   # the cross-stream waits without any kernel on the side stream
    cur = torch.cuda.current_stream(); side.wait_stream(cur); replay(); cur.wait_stream(side)
print("events only, graphs: %.3f ms (host %.2f)" % (timeit(ev_only), HOST[0]), flush=True)
def ev_a():   # record on the main stream, wait on the side stream
    cur = torch.cuda.current_stream(); side.wait_stream(cur); replay()
def ev_b():   # record on the side stream, wait on the main stream
    cur = torch.cuda.current_stream(); replay(); cur.wait_stream(side)
def ev_rec():  # an event recorded on the main stream, nobody waits
    e = torch.cuda.Event(); e.record(); replay()
print("main->side only: %.3f ms" % timeit(ev_a), flush=True)
print("side->main only: %.3f ms" % timeit(ev_b), flush=True)
print("record only:     %.3f ms" % timeit(ev_rec), flush=True)
print("graphs alone:    %.3f ms" % timeit(replay), flush=True)
# (a device-side gate - a one-wave kernel on the side stream polling a counter the main stream bumps - was measured too
#  and is WORSE: a permanently resident polling wave cost the step 2.2 ms, gates both ways 2.4 ms; kernels not kept)
def no_main_to_side():   # the sampling launched without waiting for the main stream; the main stream waits for it at the end
    cur = torch.cuda.current_stream(); fps(); replay(); cur.wait_stream(side)
print("no main->side wait, sampling: %.3f ms (host %.2f)" % (timeit(no_main_to_side), HOST[0]), flush=True)
print("graphs alone:                 %.3f ms" % timeit(replay), flush=True)
# ---- does the cost scale with how long / on how many CUs the sampling is resident?  (no main->side wait in any of these)
def fps_var(nclouds, npoint):
    def fn():
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(side), torch.no_grad():
            pointnet2_utils.furthest_point_sample(st.next_clouds[:nclouds, :, 0:3].contiguous(), npoint)
        replay(); cur.wait_stream(side)
    return fn
def fps_only(nclouds, npoint):
    def fn():
        with torch.no_grad():
            pointnet2_utils.furthest_point_sample(st.next_clouds[:nclouds, :, 0:3].contiguous(), npoint)
    return fn
for nc, npnt in ((4, 2048), (4, 1024), (4, 512), (4, 128), (2, 2048), (1, 2048)):
    alone = timeit(fps_only(nc, npnt)); t = timeit(fps_var(nc, npnt)); base = timeit(replay)
    print("sampling %d clouds x %4d picks (alone %.2f ms) beside the graphs: %.3f ms, graphs alone %.3f ms -> +%.2f" % (nc, npnt, alone, t, base, t - base), flush=True)
def after():   # launched after both graph launches
    cur = torch.cuda.current_stream(); replay(); fps(); cur.wait_stream(side)
t = timeit(after); base = timeit(replay)
print("sampling launched after both graphs: %.3f ms vs %.3f" % (t, base), flush=True)
# ---- is the cost of "the side stream waits for the main stream" the TIME its wait packet sits unsatisfied?  The host is
#      normally up to three steps ahead; here it is held one step / zero steps ahead
def throttled(fn, ahead):
    evs = []
    def run():
        cur = torch.cuda.current_stream()
        if len(evs) > ahead:
            evs.pop(0).synchronize()
        fn()
        e = torch.cuda.Event(); e.record(cur); evs.append(e)
    return run
for ahead in (3, 1, 0):
    print("main->side wait only, host at most %d step(s) ahead: %.3f ms;   sampling with that wait: %.3f ms;   graphs alone likewise: %.3f ms"
          % (ahead, timeit(throttled(ev_a, ahead)), timeit(throttled(at_start, ahead)), timeit(throttled(replay, ahead))), flush=True)
# ---- is it what is queued BEHIND the sampling kernel?  the bare kernels (pre-allocated outputs, no trailing copy), with and
#      without an event record queued behind them while the graphs run
from graspbalance_amd import _lib as L
import ctypes
xyz_c = st.next_clouds[..., 0:3].contiguous()
b_, n_ = xyz_c.shape[0], xyz_c.shape[1]
perm_ = torch.empty((b_, n_), dtype=torch.int32, device=dev)
out_ = torch.zeros((b_, 2048), dtype=torch.int32, device=dev)
sstream = ctypes.c_void_p(side.cuda_stream)
def bare():
    L.check(L.lib().gb_fps_cell_order(L.ptr(xyz_c), L.ptr(perm_), b_, n_, sstream), "order")
    L.check(L.lib().gb_fps_pruned(L.ptr(xyz_c), L.ptr(perm_), None, L.ptr(out_), b_, n_, 2048, 0, None, sstream), "pruned")
def bare_joined():
    cur = torch.cuda.current_stream(); bare(); replay(); cur.wait_stream(side)
def bare_host_joined():
    bare(); replay(); side.synchronize()
def alone_host():
    replay(); side.synchronize()
print("bare sampling kernels, joined by an event wait of the main stream: %.3f ms" % timeit(bare_joined), flush=True)
print("bare sampling kernels, nothing queued behind them (host-side join): %.3f ms" % timeit(bare_host_joined), flush=True)
print("graphs alone (host-side join of the idle side stream):             %.3f ms" % timeit(alone_host), flush=True)
