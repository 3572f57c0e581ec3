"""GPU: graspbalance_amd/data_utils.py (csrc/frame.hip) against the numpy oracle and the reference-run fixture:
unprojected cloud bit-exact in float32, workspace / depth masks and the compaction order identical."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import data_path
from tests.seeded import check_summary

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("seed,u16", [(0, True), (1, True), (1, False), (2, True)])
def test_frame_to_cloud_matches_oracle_and_reference(golden, seed, u16):
    from graspbalance_amd import data_utils as du
    depth, seg, cam, trans = data_path.synthetic_frame(seed)
    want = data_path.frame_to_cloud(depth, seg, *cam, trans=trans, outlier=0.02)
    h, w = depth.shape
    d = torch.from_numpy(depth).to(DEV) if u16 else torch.from_numpy(depth.astype(np.float32)).to(DEV)
    camera = du.CameraInfo(w, h, *cam)
    got = du.frame_to_cloud(d, torch.from_numpy(seg).to(DEV), camera, trans=trans, outlier=0.02)
    assert np.array_equal(got["cloud"].cpu().numpy(), want["cloud"].reshape(-1, 3))
    assert np.array_equal(got["workspace_mask"].cpu().numpy(), want["workspace_mask"])
    assert np.array_equal(got["mask"].cpu().numpy(), want["mask"])
    assert np.array_equal(got["index"].cpu().numpy(), np.nonzero(want["mask"].reshape(-1))[0])
    assert np.array_equal(got["cloud_masked"].cpu().numpy(), want["cloud_masked"])
    assert np.array_equal(got["seg_masked"].cpu().numpy(), want["seg_masked"])
    organized = du.create_point_cloud_from_depth_image(d, camera, organized=True)
    assert organized.shape == (h, w, 3) and np.array_equal(organized.cpu().numpy(), want["cloud"])
    if seed < 2:
        g = golden.load("g18_data_path")
        check_summary(g, "f%d/cloud" % seed, got["cloud"].view(h, w, 3), 0.0)
        check_summary(g, "f%d/cloud_masked" % seed, got["cloud_masked"], 0.0)
        assert hashlib.sha256(np.packbits(got["mask"].cpu().numpy()).tobytes()).digest() == g["f%d_mask_sha256" % seed].tobytes()
        gp = got["cloud_masked"].cpu().numpy()[::997][:40].astype(np.float64) + 0.004
        gp[::3] += 0.5
        vis = du.remove_invisible_grasp_points(got["cloud_masked"], torch.from_numpy(gp).to(DEV), torch.eye(4, dtype=torch.float64, device=DEV), th=0.01)
        assert np.array_equal(vis.cpu().numpy(), g["f%d_visible" % seed])


def test_no_outlier_removal_and_sampling_rule():
    from graspbalance_amd import data_utils as du
    depth, seg, cam, trans = data_path.synthetic_frame(3, h=240, w=320)
    camera = du.CameraInfo(320, 240, *cam)
    got = du.frame_to_cloud(torch.from_numpy(depth).to(DEV), torch.from_numpy(seg).to(DEV), camera, remove_outlier=False)
    assert got["workspace_mask"] is None and np.array_equal(got["mask"].cpu().numpy(), depth > 0)
    m = got["cloud_masked"].shape[0]
    idx = du.sample_points(m, 2000, DEV)
    assert idx.shape == (2000,) and idx.unique().numel() == 2000 and int(idx.max()) < m
    idx = du.sample_points(m, m + 500, DEV)
    assert idx.shape == (m + 500,) and torch.equal(idx[:m].cpu(), torch.arange(m)) and int(idx.max()) < m
    with pytest.raises(RuntimeError, match="CPU not supported"):
        du.create_point_cloud_from_depth_image(torch.from_numpy(depth.astype(np.float32)), camera)
