"""GPU: graspbalance_amd/collision_detector.py (csrc/collision.hip) against the numpy oracle and the fixture produced
by the reference's ModelFreeCollisionDetector.detect: voxel means bit-exact in float64, per-volume point counts equal,
masks and IoUs identical."""
import types

import numpy as np
import pytest
import torch

from oracle import data_path
from tests.golden.make_golden_r2 import g19_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _group(T, R, heights, depths, widths, dtype=np.float64, device=None):
    f = (lambda a: torch.from_numpy(np.asarray(a, dtype=dtype)).to(device)) if device else (lambda a: np.asarray(a, dtype=dtype))
    return types.SimpleNamespace(translations=f(T), rotation_matrices=f(R), heights=f(heights), depths=f(depths),
                                 widths=f(widths))


def test_detector_matches_reference_fixture(golden):
    from graspbalance_amd.collision_detector import ModelFreeCollisionDetector
    g = golden.load("g19_collision")
    scene, grasps = g19_inputs()
    det = ModelFreeCollisionDetector(torch.from_numpy(scene).to(DEV), voxel_size=0.005)
    assert np.array_equal(det.scene_points.cpu().numpy(), g["scene_down"])
    for gg in (_group(*grasps), _group(*grasps, device=DEV)):       # numpy (like graspnetAPI) or device arrays
        for name, approach in (("a", 0.03), ("b", 0.005)):
            coll, empty, ious = det.detect(gg, approach_dist=approach, return_empty_grasp=True, return_ious=True)
            assert np.array_equal(coll.cpu().numpy(), g[name + "_collision"])
            assert np.array_equal(empty.cpu().numpy(), g[name + "_empty"])
            assert np.array_equal(torch.stack(ious, 0).cpu().numpy(), g[name + "_ious"])
            only = det.detect(gg, approach_dist=approach)
            assert torch.equal(only, coll)
            _, e2 = det.detect(gg, approach_dist=approach, return_empty_grasp=True)
            assert torch.equal(e2, empty)


@pytest.mark.parametrize("m,n_grasps,voxel", [(50000, 300, 0.004), (700, 5, 0.01), (1, 3, 0.005)])
def test_counts_and_voxel_means_match_oracle(m, n_grasps, voxel):
    from graspbalance_amd.collision_detector import ModelFreeCollisionDetector, voxel_down_sample
    rng = np.random.default_rng(m)
    # a slab of points (a table top) with bumps: dense enough that the finger volumes are populated
    scene = np.stack([rng.uniform(-0.3, 0.3, m), rng.uniform(-0.3, 0.3, m), rng.normal(0, 0.01, m)], 1).astype(np.float32)
    want_down = data_path.voxel_down_sample(scene, voxel)
    got_down = voxel_down_sample(torch.from_numpy(scene).to(DEV), voxel)
    assert np.array_equal(got_down.cpu().numpy(), want_down)
    T, R, heights, depths, widths = data_path.synthetic_grasps(m + 1, want_down, n=n_grasps)
    det = ModelFreeCollisionDetector(torch.from_numpy(scene).to(DEV), voxel_size=voxel)
    counts, _ = det.counts(_group(T, R, heights, depths, widths), approach_dist=0.03)
    coll, empty, ious, want_counts = data_path.collision_detect(want_down, T, R, heights, depths, widths, voxel_size=voxel)
    assert np.array_equal(counts.cpu().numpy(), want_counts)
    got = det.detect(_group(T, R, heights, depths, widths), return_empty_grasp=True, return_ious=True)
    assert np.array_equal(got[0].cpu().numpy(), coll) and np.array_equal(got[1].cpu().numpy(), empty)
    assert np.array_equal(torch.stack(got[2], 0).cpu().numpy(), np.stack(ious, 0))


def test_float32_grasp_arrays_empty_group_and_cpu_rejection():
    from graspbalance_amd.collision_detector import ModelFreeCollisionDetector
    scene, grasps = g19_inputs()
    det = ModelFreeCollisionDetector(torch.from_numpy(scene).to(DEV))
    g32 = _group(*grasps, dtype=np.float32)
    want = data_path.collision_detect(det.scene_points.cpu().numpy(), *[np.asarray(a, dtype=np.float32).astype(np.float64) for a in grasps])
    assert np.array_equal(det.detect(g32).cpu().numpy(), want[0])
    none = _group(np.zeros((0, 3)), np.zeros((0, 3, 3)), np.zeros(0), np.zeros(0), np.zeros(0))
    assert det.detect(none).shape == (0,)
    with pytest.raises(RuntimeError):
        ModelFreeCollisionDetector(torch.from_numpy(scene))
