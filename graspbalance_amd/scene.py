"""Synthetic GraspNet-like scenes: the inputs of every bench / parity config.

The real dataset (DataProcessing/graspnet_dataset.py) is absent, so clouds are generated:
a table plane plus K objects resting on it, seen from a tilted camera, randomly permuted and —
like ``graspnet_dataset.py:129-133`` — padded by sampling WITH replacement when fewer than
``num_point`` unique points exist, so exact duplicates (FPS ties) occur as in the real data.

Everything is numpy + a seeded ``np.random.Generator`` (PCG64, stable across numpy versions), so a
(seed, N) pair names the same cloud here, in the golden fixtures and on the GPU box.
"""
import numpy as np


def _sample_box(rng, n, size):
    sx, sy, sz = size
    areas = np.array([sy * sz, sy * sz, sx * sz, sx * sz, sx * sy])  # 4 sides + top (bottom hidden)
    face = rng.choice(5, size=n, p=areas / areas.sum())
    u = rng.uniform(-0.5, 0.5, size=(n, 2))
    pts = np.zeros((n, 3))
    for f in range(5):
        sel = face == f
        k = int(sel.sum())
        if k == 0:
            continue
        a, b = u[sel, 0], u[sel, 1]
        if f == 0:
            pts[sel] = np.stack([np.full(k, 0.5 * sx), a * sy, b * sz], 1)
        elif f == 1:
            pts[sel] = np.stack([np.full(k, -0.5 * sx), a * sy, b * sz], 1)
        elif f == 2:
            pts[sel] = np.stack([a * sx, np.full(k, 0.5 * sy), b * sz], 1)
        elif f == 3:
            pts[sel] = np.stack([a * sx, np.full(k, -0.5 * sy), b * sz], 1)
        else:
            pts[sel] = np.stack([a * sx, b * sy, np.full(k, 0.5 * sz)], 1)
    pts[:, 2] += 0.5 * sz  # rest on z = 0
    return pts


def _sample_sphere(rng, n, size):
    r = 0.5 * size[0]
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    pts = v * r
    pts[:, 2] += r
    return pts


def _sample_cylinder(rng, n, size):
    r, h = 0.5 * size[0], size[2]
    side = 2 * np.pi * r * h
    top = np.pi * r * r
    on_top = rng.uniform(size=n) < top / (side + top)
    ang = rng.uniform(0, 2 * np.pi, size=n)
    rad = np.where(on_top, r * np.sqrt(rng.uniform(size=n)), r)
    z = np.where(on_top, h, rng.uniform(0, h, size=n))
    return np.stack([rad * np.cos(ang), rad * np.sin(ang), z], 1)


def make_scene(seed, num_point=20000, num_objects=8, dup_fraction=0.03, return_labels=False):
    """One cloud ``(num_point, 3) float32`` in the camera frame (z forward, metres).

    60 % table points ``z_table = 0.002*N(0,1)`` over ``x,y in U(-0.35,0.35)``, 40 % split over
    ``num_objects`` boxes / spheres / cylinders of size U(0.03,0.12) standing on the table; the
    table sits 0.75 m from the camera, tilted U(-30,30) degrees about the camera x axis.
    ``dup_fraction`` of the points are exact copies of other points (with-replacement padding).
    With ``return_labels`` also returns ``seg (num_point,) int64`` (0 table, 1..K object id) and the
    per-object poses ``(K,3,4) float32`` (object frame -> camera frame).
    """
    rng = np.random.default_rng(int(seed))
    n_unique = max(1, int(round(num_point * (1.0 - dup_fraction))))
    n_table = int(round(0.6 * n_unique))
    n_obj_total = n_unique - n_table
    table = np.stack([rng.uniform(-0.35, 0.35, n_table), rng.uniform(-0.35, 0.35, n_table),
                      0.002 * rng.normal(size=n_table)], 1)
    pts = [table]
    seg = [np.zeros(n_table, dtype=np.int64)]
    poses_world = []
    counts = np.full(num_objects, n_obj_total // max(num_objects, 1))
    if num_objects > 0:
        counts[: n_obj_total - counts.sum()] += 1
    for k in range(num_objects):
        kind = int(rng.integers(0, 3))
        size = rng.uniform(0.03, 0.12, size=3)
        centre = rng.uniform(-0.28, 0.28, size=2)
        yaw = rng.uniform(0, 2 * np.pi)
        sampler = (_sample_box, _sample_sphere, _sample_cylinder)[kind]
        local = sampler(rng, int(counts[k]), size)
        c, s = np.cos(yaw), np.sin(yaw)
        rot = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        world = local @ rot.T
        world[:, 0] += centre[0]
        world[:, 1] += centre[1]
        pts.append(world)
        seg.append(np.full(int(counts[k]), k + 1, dtype=np.int64))
        pose = np.zeros((3, 4))
        pose[:, :3] = rot
        pose[:2, 3] = centre
        poses_world.append(pose)
    pts = np.concatenate(pts, 0)
    seg = np.concatenate(seg, 0)
    # world (table frame, z up) -> camera frame: flip so that +z looks at the table from above,
    # tilt about x, push 0.75 m along the optical axis.
    tilt = np.deg2rad(rng.uniform(-30.0, 30.0))
    ct, st = np.cos(tilt), np.sin(tilt)
    flip = np.array([[1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0]])
    rx = np.array([[1.0, 0.0, 0.0], [0.0, ct, -st], [0.0, st, ct]])
    cam_r = rx @ flip
    cam_t = np.array([0.0, 0.0, 0.75])
    cam = pts @ cam_r.T + cam_t
    perm = rng.permutation(n_unique)
    cam, seg = cam[perm], seg[perm]
    if num_point > n_unique:  # with-replacement padding -> exact duplicates
        extra = rng.integers(0, n_unique, size=num_point - n_unique)
        cam = np.concatenate([cam, cam[extra]], 0)
        seg = np.concatenate([seg, seg[extra]], 0)
        perm2 = rng.permutation(num_point)
        cam, seg = cam[perm2], seg[perm2]
    cloud = np.ascontiguousarray(cam.astype(np.float32))
    if not return_labels:
        return cloud
    poses = []
    for pose in poses_world:
        p = np.zeros((3, 4))
        p[:, :3] = cam_r @ pose[:, :3]
        p[:, 3] = cam_r @ pose[:, 3] + cam_t
        poses.append(p)
    poses = np.asarray(poses, dtype=np.float32).reshape(-1, 3, 4)
    return cloud, seg, poses


def make_batch(seeds, num_point=20000, **kw):
    """Stack ``make_scene(seed)`` for every seed -> (B, num_point, 3) float32 ndarray."""
    return np.stack([make_scene(s, num_point, **kw) for s in seeds], 0)
