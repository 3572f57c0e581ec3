"""CPU: the numpy restatement of the collision check (oracle/data_path.py) against a fixture produced by the
reference's own ModelFreeCollisionDetector.detect (tests/golden/make_golden_r2.py g19; the voxel down-sampling it
starts from is open3d's and enters the fixture as an input - see the generator)."""
import numpy as np

from oracle import data_path
from tests.golden.make_golden_r2 import g19_inputs


def test_oracle_collision_detect_matches_reference(golden):
    g = golden.load("g19_collision")
    scene, (T, R, heights, depths, widths) = g19_inputs()
    down = data_path.voxel_down_sample(scene, 0.005)
    assert np.array_equal(down, g["scene_down"])            # the generator's stand-in is this function: a regression pin
    for name, approach in (("a", 0.03), ("b", 0.005)):
        coll, empty, ious, counts = data_path.collision_detect(g["scene_down"], T, R, heights, depths, widths,
                                                                approach_dist=approach)
        assert np.array_equal(coll, g[name + "_collision"]) and np.array_equal(empty, g[name + "_empty"])
        assert np.array_equal(np.stack(ious, 0), g[name + "_ious"])
        assert counts[:, 4].max() > 0 and (counts[:, 4] <= counts[:, :4].sum(axis=1)).all()


def test_voxel_down_sample_properties():
    rng = np.random.default_rng(0)
    pts = rng.uniform(-0.1, 0.1, (5000, 3))
    down = data_path.voxel_down_sample(pts, 0.01)
    lo = pts.min(axis=0) - 0.005
    cells = np.floor((pts - lo) / 0.01).astype(np.int64)
    assert len(down) == len(np.unique(cells, axis=0))
    assert np.allclose(down.mean(axis=0), pts.mean(axis=0), atol=2e-3)
    # every mean lies in its own voxel: the voxels of the output are distinct
    out_cells = np.floor((down - lo) / 0.01).astype(np.int64)
    assert len(np.unique(out_cells, axis=0)) == len(down)
    one = data_path.voxel_down_sample(pts[:1], 0.01)
    assert np.array_equal(one, pts[:1])
