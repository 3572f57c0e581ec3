"""CPU simulation of the pruned FPS (csrc/fps.hip) on a bench cloud: how many 64-point rows / 20-point thread slabs /
waves a new sample forces to update, per phase of the 2048 iterations (numpy; ~1 min)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graspbalance_amd.scene import make_batch
xyz = make_batch([0], 20000)[0].astype(np.float32)
N, m, P = 20000, 2048, 20
lo, hi = xyz.min(0), xyz.max(0)
q = np.clip(((xyz - lo) * (1023.0 / (hi - lo))).astype(np.int64), 0, 1023)
def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
perm = np.argsort(key, kind="stable")
pts = xyz[perm]
pad = 1024 * P - N
ptsp = np.concatenate([pts, np.full((pad, 3), np.nan, np.float32)])
thr = ptsp.reshape(1024, P, 3)
blo, bhi = np.nanmin(thr, 1), np.nanmax(thr, 1)
temp = np.full(N, 1e10, np.float32)
old = int(np.where(perm == 0)[0][0])
act_w, act_t = [], []
for j in range(1, m):
    s = pts[old]
    b = np.maximum(np.maximum(blo - s, s - bhi), 0)
    lb = (b ** 2).sum(1)
    tmax = np.full(1024, -1.0, np.float32)
    tp = np.concatenate([temp, np.full(pad, -1, np.float32)]).reshape(1024, P)
    tmax = tp.max(1)
    need = lb < tmax
    act_t.append(need.sum())
    act_w.append(need.reshape(16, 64).any(1).sum())
    d = ((pts - s) ** 2).sum(1)
    temp = np.minimum(temp, d)
    old = int(temp.argmax())
act_w = np.array(act_w); act_t = np.array(act_t)
for a, b in [(0, 64), (64, 256), (256, 1024), (1024, 2047)]:
    print("iters %4d-%4d: active waves %.1f /16, active threads %.0f /1024" % (a, b, act_w[a:b].mean(), act_t[a:b].mean()))
print("overall active waves %.2f, threads %.1f" % (act_w.mean(), act_t.mean()))

# --- row granularity (64 consecutive sorted points), 20 rows per wave
rows = ptsp.reshape(320, 64, 3)
rlo, rhi = np.nanmin(rows, 1), np.nanmax(rows, 1)
temp = np.full(N, 1e10, np.float32)
old = int(np.where(perm == 0)[0][0])
act, busiest = [], []
for j in range(1, m):
    s = pts[old]
    b = np.maximum(np.maximum(rlo - s, s - rhi), 0)
    lb = (b ** 2).sum(1)
    tp = np.concatenate([temp, np.full(pad, -1, np.float32)]).reshape(320, 64)
    need = lb < tp.max(1)
    act.append(need.sum()); busiest.append(need.reshape(16, 20).sum(1).max())
    d = ((pts - s) ** 2).sum(1)
    temp = np.minimum(temp, d)
    old = int(temp.argmax())
act = np.array(act); busiest = np.array(busiest)
for a, b in [(0, 64), (64, 256), (256, 1024), (1024, 2047)]:
    print("iters %4d-%4d: active rows %.1f /320, busiest wave %.1f rows" % (a, b, act[a:b].mean(), busiest[a:b].mean()))
print("overall active rows %.2f, busiest wave %.2f" % (act.mean(), busiest.mean()))
