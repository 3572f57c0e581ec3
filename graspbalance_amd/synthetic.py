"""Synthetic training batches shaped like the GraspNet-1B loader's output
(DataProcessing/graspnet_dataset.py:222-232, collate_fn :255-261): the dict keys consumed by
label_generation.py:19-43 and loss.py:56-58.  Clouds come from ``scene.make_scene``; grasp labels are
seeded uniforms (there is no dataset in this environment — bench.py says so in its ``data`` field).
"""
import numpy as np
import torch

from .scene import make_scene


def make_training_batch(seeds, num_point=20000, num_objects=8, grasp_points_per_object=300, num_view=300,
                        num_angle=12, num_depth=4, device="cpu"):
    """One collated batch.  ``*_list`` entries are list[B] of list[objects] of tensors."""
    device = torch.device(device)
    clouds, objectness = [], []
    lists = {k: [] for k in ('object_poses_list', 'grasp_points_list', 'grasp_offsets_list',
                             'grasp_labels_list', 'grasp_tolerance_list')}
    V, A, D = num_view, num_angle, num_depth
    for seed in seeds:
        cloud, seg, poses = make_scene(seed, num_point, num_objects, return_labels=True)
        clouds.append(torch.from_numpy(cloud))
        objectness.append(torch.from_numpy((seg > 0).astype(np.int64)))
        gen = torch.Generator(device=device).manual_seed(int(seed) * 7919 + 17)
        per = {k: [] for k in lists}
        cloud_t = torch.from_numpy(cloud)
        for k in range(poses.shape[0]):
            pose = torch.from_numpy(poses[k])
            obj_pts = cloud_t[torch.from_numpy(seg == (k + 1))]
            pick = torch.from_numpy(np.random.default_rng(int(seed) * 131 + k).integers(
                0, obj_pts.shape[0], size=grasp_points_per_object))
            local = (obj_pts[pick] - pose[:, 3]) @ pose[:, :3]  # R^T (p - t): camera -> object frame
            Np = local.shape[0]
            scores = torch.rand((Np, V, A, D), generator=gen, device=device) * 1.1 + 0.1
            scores = scores * (torch.rand((Np, V, A, D), generator=gen, device=device) > 0.5)
            offsets = torch.rand((Np, V, A, D, 3), generator=gen, device=device)
            offsets[..., 2] *= 0.12                       # width (m); some exceed GRASP_MAX_WIDTH
            offsets[..., 1] = offsets[..., 1] * 0.04      # depth
            offsets[..., 0] = offsets[..., 0] * np.pi     # in-plane angle
            tolerance = torch.rand((Np, V, A, D), generator=gen, device=device) * 0.05
            per['object_poses_list'].append(pose.to(device))
            per['grasp_points_list'].append(local.float().to(device))
            per['grasp_offsets_list'].append(offsets)
            per['grasp_labels_list'].append(scores)
            per['grasp_tolerance_list'].append(tolerance)
        for k in lists:
            lists[k].append(per[k])
    batch = {'point_clouds': torch.stack(clouds, 0).to(device),
             'objectness_label': torch.stack(objectness, 0).to(device)}
    batch.update(lists)
    return batch


def clone_batch(batch):
    """Shallow copy whose label tensors are fresh clones (process_grasp_labels edits labels in place
    only on its own stacked copies, but end_points is mutated, so every step needs its own dict)."""
    out = {}
    for k, v in batch.items():
        out[k] = [list(inner) for inner in v] if isinstance(v, list) else v
    return out
