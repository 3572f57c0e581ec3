"""Collision check of 1024 grasp candidates against a voxel-down-sampled scene: csrc/collision.hip vs the numpy
restatement of collision_detector.py:16-64 (oracle/data_path.py) on the host."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import data_path
from graspbalance_amd.collision_detector import ModelFreeCollisionDetector
DEV = "cuda:0"
rng = np.random.default_rng(0)
m = 200000
scene = np.stack([rng.uniform(-0.4, 0.4, m), rng.uniform(-0.3, 0.3, m), rng.normal(0, 0.01, m)], 1)
sc = torch.from_numpy(scene).to(DEV)
for _ in range(2):
    det = ModelFreeCollisionDetector(sc, 0.005)
torch.cuda.synchronize(); t0 = time.time()
det = ModelFreeCollisionDetector(sc, 0.005)
torch.cuda.synchronize(); t_down = time.time() - t0
down = det.scene_points.cpu().numpy()
T, R, h, d, w = data_path.synthetic_grasps(1, down, n=1024)
gg = types.SimpleNamespace(**{k: torch.from_numpy(v).to(DEV) for k, v in dict(translations=T, rotation_matrices=R, heights=h, depths=d, widths=w).items()})
for _ in range(3):
    out = det.detect(gg, return_empty_grasp=True, return_ious=True)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10):
    out = det.detect(gg, return_empty_grasp=True, return_ious=True)
torch.cuda.synchronize(); t_hip = (time.time() - t0) / 10
t0 = time.time()
want = data_path.collision_detect(down, T, R, h, d, w)
t_np = time.time() - t0
assert np.array_equal(out[0].cpu().numpy(), want[0])
print("scene %d -> %d points (down-sample %.2f ms on the GPU); detect, 1024 grasps: HIP %.3f ms, numpy %.0f ms (%.0fx); pair tests %.1f G/s"
      % (m, len(down), t_down * 1e3, t_hip * 1e3, t_np * 1e3, t_np / t_hip, 1024 * len(down) / t_hip / 1e9))
