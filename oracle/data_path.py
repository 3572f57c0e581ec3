"""numpy restatement of the reference's host data path (data_utils.py:14-72, graspnet_dataset.py:110-136).
TEST INFRASTRUCTURE ONLY (like the rest of oracle/): the checker of graspbalance_amd/data_utils.py."""
import numpy as np


def depth_to_cloud(depth, fx, fy, cx, cy, scale):
    """data_utils.py:14-25: float64 throughout; (H,W,3)."""
    h, w = depth.shape
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    z = depth / scale
    return np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], axis=-1)


def workspace_mask(cloud, seg, trans=None, outlier=0.0):
    """data_utils.py:52-72 (organized): strictly inside the foreground's bounding box widened by `outlier`."""
    h, w, _ = cloud.shape
    pts = cloud.reshape(h * w, 3)
    if trans is not None:
        pts = (trans @ np.concatenate([pts, np.ones((h * w, 1))], axis=1).T).T[:, :3]
    fg = pts[seg.reshape(-1) > 0]
    lo, hi = fg.min(axis=0), fg.max(axis=0)
    inside = np.ones(h * w, dtype=bool)
    for a in range(3):
        inside &= (pts[:, a] > lo[a] - outlier) & (pts[:, a] < hi[a] + outlier)
    return inside.reshape(h, w)


def frame_to_cloud(depth, seg, fx, fy, cx, cy, scale, trans=None, outlier=0.02, remove_outlier=True):
    """graspnet_dataset.py:110-127."""
    cloud = depth_to_cloud(depth, fx, fy, cx, cy, scale)
    mask = depth > 0
    ws = None
    if remove_outlier:
        ws = workspace_mask(cloud, seg, trans, outlier)
        mask = mask & ws
    return {"cloud": cloud.astype(np.float32), "workspace_mask": ws, "mask": mask,
            "cloud_masked": cloud[mask].astype(np.float32), "seg_masked": seg[mask]}


def synthetic_frame(seed, h=360, w=640):
    """A depth frame (uint16 millimetres), a segmentation image and a camera->table transform: a tilted table plane
    with a few box-shaped objects and a sprinkling of invalid (zero) depths."""
    rng = np.random.default_rng(seed)
    fx = fy = 460.0 * w / 640.0
    cx, cy = w / 2.0 - 0.5, h / 2.0 - 0.5
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    z = 0.55 + 0.25 * (v / h) + 0.02 * (u / w)
    seg = np.zeros((h, w), dtype=np.int32)
    for k in range(5):
        x0, y0 = rng.integers(40, w - 120), rng.integers(40, h - 100)
        ww, hh = rng.integers(30, 90), rng.integers(25, 70)
        z[y0:y0 + hh, x0:x0 + ww] -= rng.uniform(0.03, 0.09)
        seg[y0:y0 + hh, x0:x0 + ww] = k + 1
    depth = np.round((z + rng.normal(0, 0.001, z.shape)) * 1000.0).astype(np.uint16)
    depth[rng.random((h, w)) < 0.03] = 0
    ang = np.deg2rad(-35.0)
    trans = np.eye(4)
    trans[:3, :3] = np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]])
    trans[:3, 3] = [0.02, -0.3, 0.6]
    return depth, seg, (float(fx), float(fy), float(cx), float(cy), 1000.0), trans


def voxel_down_sample(points, voxel_size):
    """open3d.geometry.PointCloud.voxel_down_sample as the detector uses it (collision_detector.py:11-14).  open3d is a
    third-party dependency absent from this image and from /root/reference (no pinned version there); this restates its
    published algorithm (PointCloud::VoxelDownSample): voxel index = floor((p - (min_bound - voxel/2)) / voxel) per
    axis, the points of a voxel are accumulated in fp64 in their original order and divided by their number.  open3d
    emits the voxels in hash-map order; this returns them sorted by voxel index (x major) - every consumer here sums
    over the points, so their order does not matter.  PARITY UNPINNED for this function (no open3d to run)."""
    pts = np.asarray(points, dtype=np.float64)
    lo = pts.min(axis=0) - voxel_size * 0.5
    idx = np.floor((pts - lo) / voxel_size).astype(np.int64)
    dims = idx.max(axis=0) + 1
    key = (idx[:, 0] * dims[1] + idx[:, 1]) * dims[2] + idx[:, 2]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    starts = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
    ends = np.concatenate([starts[1:], [len(ks)]])
    out = np.empty((len(starts), 3))
    sp = pts[order]
    for v, (b, e) in enumerate(zip(starts, ends)):
        acc = np.zeros(3)
        for p in range(b, e):
            acc += sp[p]
        out[v] = acc / float(e - b)
    return out


def collision_detect(scene_points, translations, rotations, heights, depths, widths, voxel_size=0.005,
                     approach_dist=0.03, collision_thresh=0.05, empty_thresh=0.01, finger_width=0.01,
                     finger_length=0.06):
    """ModelFreeCollisionDetector.detect (collision_detector.py:16-64) on an already down-sampled scene, every
    return value at once: (collision_mask, empty_mask, [global, left, right, bottom, shifting] ious, counts (G,6))."""
    approach_dist = max(approach_dist, finger_width)                                     # :17
    T, R = np.asarray(translations), np.asarray(rotations)
    heights, depths, widths = (np.asarray(a)[:, None] for a in (heights, depths, widths))  # :20-22
    targets = np.matmul(scene_points[None, :, :] - T[:, None, :], R)                       # :23-24
    x, y, z = targets[:, :, 0], targets[:, :, 1], targets[:, :, 2]
    m1 = (z > -heights / 2) & (z < heights / 2)                                            # :26
    m2 = (x > depths - finger_length) & (x < depths)                                       # :27
    m3 = y > -(widths / 2 + finger_width)
    m4 = y < -widths / 2
    m5 = y < (widths / 2 + finger_width)
    m6 = y > widths / 2
    m7 = (x <= depths - finger_length) & (x > depths - finger_length - finger_width)       # :32-33
    m8 = (x <= depths - finger_length - finger_width) & (x > depths - finger_length - finger_width - approach_dist)
    left, right = m1 & m2 & m3 & m4, m1 & m2 & m5 & m6
    bottom, shifting = m1 & m3 & m5 & m7, m1 & m3 & m5 & m8
    glob = left | right | bottom | shifting
    lr_vol = (heights * finger_length * finger_width / (voxel_size ** 3)).reshape(-1)      # :43-45
    bottom_vol = (heights * (widths + 2 * finger_width) * finger_width / (voxel_size ** 3)).reshape(-1)
    shift_vol = (heights * (widths + 2 * finger_width) * approach_dist / (voxel_size ** 3)).reshape(-1)
    volume = lr_vol * 2 + bottom_vol + shift_vol
    global_iou = glob.sum(axis=1) / (volume + 1e-6)
    inner = m1 & m2 & (~m4) & (~m6)                                                        # :50-52
    inner_vol = (heights * finger_length * widths / (voxel_size ** 3)).reshape(-1)
    empty = inner.sum(axis=-1) / inner_vol < empty_thresh
    ious = [global_iou, left.sum(axis=1) / (lr_vol + 1e-6), right.sum(axis=1) / (lr_vol + 1e-6),
            bottom.sum(axis=1) / (bottom_vol + 1e-6), shifting.sum(axis=1) / (shift_vol + 1e-6)]
    counts = np.stack([m.sum(axis=1) for m in (left, right, bottom, shifting, glob, inner)], axis=1).astype(np.int32)
    return global_iou > collision_thresh, empty, ious, counts


def synthetic_grasps(seed, scene, n=64):
    """Grasp candidates near the scene's points: (translations, rotation matrices, heights, depths, widths), float64."""
    rng = np.random.default_rng(seed)
    T = scene[rng.integers(0, len(scene), n)] + rng.normal(0, 0.004, (n, 3))
    T[::5] += rng.normal(0, 0.05, T[::5].shape)          # some grasps in free space
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1).reshape(n, 3, 3)
    heights = np.full(n, 0.02)
    depths = rng.choice([0.01, 0.02, 0.03, 0.04], n)
    widths = rng.uniform(0.02, 0.1, n)
    return T, R, heights, depths, widths


def transform_point_cloud(cloud, transform):
    """data_utils.py:28-39 ('4x4' with a 3x4 or 4x4 matrix)."""
    ones = np.ones(cloud.shape[0], dtype=cloud.dtype)[:, None]
    return np.dot(transform, np.concatenate([cloud, ones], axis=1).T).T[:, :3]


def sample_from_frame(depth, color, seg, cam, num_points, cloud_idxs, trans=None, remove_outlier=False, obj_idxs=None,
                      poses=None, grasp_labels=None, collision_labels=None, valid_obj_idxs=None, grasp_idxs=None,
                      remove_invisible=True):
    """GraspPoseDataset.get_data_label (graspnet_dataset.py:143-237) after its file reads, with the random choices
    (`cloud_idxs` :175-180, `grasp_idxs` per kept object :209) handed in; no augmentation."""
    f = frame_to_cloud(depth, seg, *cam, trans=trans, outlier=0.02, remove_outlier=remove_outlier)
    cloud = depth_to_cloud(depth, *cam)
    mask = f["mask"]
    cloud_sampled, color_sampled, seg_sampled = cloud[mask][cloud_idxs], color[mask][cloud_idxs], seg[mask][cloud_idxs]
    ret = {"point_clouds": cloud_sampled.astype(np.float32), "cloud_colors": color_sampled.astype(np.float32)}
    if obj_idxs is None:
        return ret
    objectness = seg_sampled.copy()
    objectness[objectness > 1] = 1
    ret["objectness_label"] = objectness.astype(np.int64)
    for k in ("object_poses_list", "grasp_points_list", "grasp_offsets_list", "grasp_labels_list", "grasp_tolerance_list"):
        ret[k] = []
    kept = 0
    for i, obj_idx in enumerate(obj_idxs):
        if obj_idx not in valid_obj_idxs or (seg_sampled == obj_idx).sum() < 50:
            continue
        points, offsets, scores, tolerance = grasp_labels[obj_idx]
        collision = collision_labels[i]
        if remove_invisible:
            pts = transform_point_cloud(points, poses[:, :, i])
            d = np.linalg.norm(pts[:, None, :] - cloud_sampled[seg_sampled == obj_idx][None, :, :], axis=-1).min(axis=1)
            vis = d < 0.01
            points, offsets, scores, tolerance, collision = (a[vis] for a in (points, offsets, scores, tolerance, collision))
        idxs = grasp_idxs[kept]
        kept += 1
        assert len(idxs) == min(max(int(len(points) / 4), 300), len(points)) and len(np.unique(idxs)) == len(idxs)
        collision = collision[idxs].copy()
        scores, tolerance = scores[idxs].copy(), tolerance[idxs].copy()
        scores[collision] = 0
        tolerance[collision] = 0
        ret["object_poses_list"].append(poses[:, :, i])
        ret["grasp_points_list"].append(points[idxs])
        ret["grasp_offsets_list"].append(offsets[idxs])
        ret["grasp_labels_list"].append(scores)
        ret["grasp_tolerance_list"].append(tolerance)
    assert kept == len(grasp_idxs)
    return ret


def synthetic_frame_labels(seed, depth, seg, cam, n_points=1200, V=6, A=4, D=2):
    """Per-object grasp labels for a synthetic frame: object k+1's grasp points lie on its visible surface (half of
    them pushed 5 cm away: invisible).  -> obj_idxs, poses (3,4,K) identity, grasp_labels, collision_labels, valid ids
    (object 3 is not valid, object ids without 50 sampled points drop out by themselves)."""
    rng = np.random.default_rng(seed)
    cloud = depth_to_cloud(depth, *cam)
    obj_idxs = np.array([1, 2, 3, 4, 5, 7], dtype=np.int32)      # 7 does not occur in seg
    poses = np.repeat(np.eye(4, dtype=np.float32)[:3, :, None], len(obj_idxs), axis=2)
    grasp_labels, collision_labels = {}, {}
    for i, o in enumerate(obj_idxs):
        surf = cloud[(seg == o) & (depth > 0)]
        if len(surf) == 0:
            surf = np.zeros((1, 3))
        pts = surf[rng.integers(0, len(surf), n_points)].astype(np.float32)
        pts[::2] += np.float32(0.05)
        grasp_labels[int(o)] = (pts, rng.uniform(0, 1, (n_points, V, A, D, 3)).astype(np.float32),
                                rng.uniform(0, 1, (n_points, V, A, D)).astype(np.float32),
                                rng.uniform(0, 0.05, (n_points, V, A, D)).astype(np.float32))
        collision_labels[i] = rng.random((n_points, V, A, D)) < 0.3
    return obj_idxs, poses, grasp_labels, collision_labels, [1, 2, 4, 5, 7]
