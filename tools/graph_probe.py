"""Feasibility probe: HIP-graph capture of the DRP backbone's train-mode forward + backward (no host sync inside),
replay time against the eager loop.  python tools/graph_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graspbalance_amd import _lib, fused_mlp  # noqa: E402
from graspbalance_amd.drp import DRP  # noqa: E402
from graspbalance_amd.scene import make_batch  # noqa: E402

dev = torch.device("cuda", 0)
_lib.lib()
torch.manual_seed(1234)
net = DRP().to(dev).train()
clouds = torch.from_numpy(make_batch([0, 1, 2, 3], 20000)).to(dev)
params = [p for p in net.parameters()]


def step():
    fused_mlp.begin_step(dev)
    feats, xyz, _ = net(clouds, {})
    loss = feats.square().mean()
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    return loss, grads


def timeit(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return th / n * 1e3, (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    step()
host, wall = timeit(step, 10)
print("eager: host %.2f ms, wall %.2f ms per step" % (host, wall), flush=True)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    loss, grads = step()
print("capture took %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
g.replay()
torch.cuda.synchronize()
l_graph = float(loss)
host, wall = timeit(g.replay, 20)
print("graph: host %.2f ms, wall %.2f ms per replay, loss %.6f" % (host, wall, l_graph), flush=True)
l_eager, _ = step()
print("eager loss after: %.6f" % float(l_eager))
