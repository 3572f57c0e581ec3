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


def _dev_labels(grasp_labels, collision_labels):
    gl = {k: tuple(torch.from_numpy(a).to(DEV) for a in v) for k, v in grasp_labels.items()}
    return gl, {k: torch.from_numpy(v).to(DEV) for k, v in collision_labels.items()}


@pytest.mark.parametrize("num_points,remove_outlier", [(20000, True), (400000, False)])
def test_frame_to_sample_equals_dataset_getitem_restatement(num_points, remove_outlier):
    """get_data_label (graspnet_dataset.py:143-237) with the device path's own random choices replayed in numpy:
    fewer / more points than requested, invalid and absent objects, invisible grasp points, collision zeroing."""
    from graspbalance_amd import data_utils as du
    depth, seg, cam, trans = data_path.synthetic_frame(4)
    h, w = depth.shape
    color = np.random.default_rng(4).random((h, w, 3)).astype(np.float32)
    obj_idxs, poses, grasp_labels, collision_labels, valid = data_path.synthetic_frame_labels(4, depth, seg, cam)
    gl, cl = _dev_labels(grasp_labels, collision_labels)
    g = torch.Generator(device=DEV).manual_seed(7)
    got = du.frame_to_sample(torch.from_numpy(depth).to(DEV), torch.from_numpy(color).to(DEV), torch.from_numpy(seg).to(DEV),
                             du.CameraInfo(w, h, *cam), num_points=num_points, trans=trans, remove_outlier=remove_outlier,
                             obj_idxs=obj_idxs, poses=torch.from_numpy(poses).to(DEV), grasp_labels=gl, collision_labels=cl,
                             valid_obj_idxs=valid, generator=g)
    cloud_idxs = got["_cloud_idxs"].cpu().numpy()
    want = data_path.sample_from_frame(depth, color, seg, cam, num_points, cloud_idxs, trans=trans,
                                       remove_outlier=remove_outlier, obj_idxs=obj_idxs, poses=poses,
                                       grasp_labels=grasp_labels, collision_labels=collision_labels, valid_obj_idxs=valid,
                                       grasp_idxs=[p.cpu().numpy() for p in got["_grasp_idxs"]])
    n_masked = int(((depth > 0) & (data_path.frame_to_cloud(depth, seg, *cam, trans=trans)["workspace_mask"] if remove_outlier else True)).sum())
    if n_masked >= num_points:
        assert len(np.unique(cloud_idxs)) == num_points
    else:
        assert np.array_equal(cloud_idxs[:n_masked], np.arange(n_masked)) and cloud_idxs.max() < n_masked
    for k in ("point_clouds", "cloud_colors", "objectness_label"):
        assert got[k].dtype == torch.from_numpy(want[k]).dtype and np.array_equal(got[k].cpu().numpy(), want[k]), k
    assert 2 <= len(want["object_poses_list"]) <= 4       # object 3 invalid, 7 absent
    for k in ("object_poses_list", "grasp_points_list", "grasp_offsets_list", "grasp_labels_list", "grasp_tolerance_list"):
        assert len(got[k]) == len(want[k])
        for a, b in zip(got[k], want[k]):
            assert np.array_equal(a.cpu().numpy(), b), k
    assert any((s == 0).any() for s in want["grasp_labels_list"])
    batch = du.collate_fn([got, got])
    assert batch["point_clouds"].shape == (2, num_points, 3) and len(batch["grasp_points_list"]) == 2
    assert batch["grasp_points_list"][1][0] is got["grasp_points_list"][0]
    plain = du.frame_to_sample(torch.from_numpy(depth).to(DEV), torch.from_numpy(color).to(DEV), torch.from_numpy(seg).to(DEV),
                               du.CameraInfo(w, h, *cam), num_points=num_points)
    assert set(plain) == {"point_clouds", "cloud_colors", "_cloud_idxs"}


def test_augment_data_follows_the_reference_transforms():
    from graspbalance_amd import data_utils as du
    rng = np.random.default_rng(0)
    cloud = rng.normal(size=(1000, 3))
    poses = [rng.normal(size=(3, 4)).astype(np.float32) for _ in range(3)]
    for flip, ang in ((True, 0.3), (False, -0.5)):
        got_c, got_p = du.augment_data(torch.from_numpy(cloud).to(DEV), [torch.from_numpy(p).to(DEV) for p in poses],
                                       flip=flip, rot_angle=ang)
        c, p = cloud, [q for q in poses]
        if flip:                                            # graspnet_dataset.py:71-77
            m = np.array([[-1, 0, 0], [0, 1, 0], [0, 0, 1]])
            c = np.dot(m, c.T).T
            p = [np.dot(m, q).astype(np.float32) for q in p]
        co, si = np.cos(ang), np.sin(ang)                   # :79-86
        m = np.array([[1, 0, 0], [0, co, -si], [0, si, co]])
        c = np.dot(m, c.T).T
        p = [np.dot(m, q).astype(np.float32) for q in p]
        assert np.allclose(got_c.cpu().numpy(), c, rtol=0, atol=1e-14)
        for a, b in zip(got_p, p):
            assert np.allclose(a.cpu().numpy(), b, rtol=0, atol=1e-6)
    g = torch.Generator().manual_seed(0)
    angles = []
    for _ in range(50):
        c2, _ = du.augment_data(torch.tensor([[0., 1., 0.]], dtype=torch.float64, device=DEV), [], generator=g)
        angles.append(float(torch.atan2(c2[0, 2], c2[0, 1])))
    assert max(abs(a) for a in angles) <= np.pi / 6 + 1e-9 and np.std(angles) > 0.1
