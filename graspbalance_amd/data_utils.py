"""The reference's host data path (data_utils.py:14-72, DataProcessing/graspnet_dataset.py:110-136) on the GPU:
a depth frame becomes the network's input cloud without a round trip through numpy - unprojection, workspace mask,
ordered compaction (csrc/frame.hip) and the point sampling rule.  Names follow the reference; tensors are torch CUDA
tensors where the reference takes numpy arrays.  SURVEY.md section 8 row f4.
"""
import ctypes

import torch

from . import _lib


class CameraInfo:
    """data_utils.py:3-11."""

    def __init__(self, width, height, fx, fy, cx, cy, scale):
        self.width, self.height = width, height
        self.fx, self.fy, self.cx, self.cy, self.scale = fx, fy, cx, cy, scale

    def _cam5(self):
        vals = [float(torch.as_tensor(v).reshape(-1)[0]) for v in (self.fx, self.fy, self.cx, self.cy, self.scale)]
        return (ctypes.c_double * 5)(*vals)


def _depth_arg(depth):
    if not depth.is_cuda:
        raise RuntimeError("CPU not supported")
    if depth.dtype == torch.uint16:
        return depth.contiguous(), 1
    if depth.dtype in (torch.int16, torch.int32, torch.int64, torch.float64):
        depth = depth.to(torch.float32)   # exact for depth values below 2^24
    if depth.dtype != torch.float32:
        raise RuntimeError("depth must be uint16 or float32")
    return depth.contiguous(), 0


def _trans12(trans):
    if trans is None:
        return None
    t = torch.as_tensor(trans, dtype=torch.float64).cpu().reshape(-1, 4)[:3].contiguous().reshape(-1).tolist()
    return (ctypes.c_double * 12)(*t)


def create_point_cloud_from_depth_image(depth, camera, organized=True):
    """depth (H,W) -> cloud (H,W,3) float32 [(H*W,3) if not organized]: z = depth / scale, x = (u - cx) z / fx,
    y = (v - cy) z / fy in float64, rounded to float32 (data_utils.py:14-25 followed by graspnet_dataset.py:136)."""
    depth, is_u16 = _depth_arg(depth)
    H, W = depth.shape
    assert H == camera.height and W == camera.width
    cloud = torch.empty((H, W, 3), dtype=torch.float32, device=depth.device)
    with _lib.device_ctx(depth.device):
        _lib.check(_lib.lib().gb_frame_cloud(_lib.ptr(depth), is_u16, None, ctypes.cast(camera._cam5(), ctypes.c_void_p),
                                             None, H, W, _lib.ptr(cloud), None, _lib.current_stream(depth.device)),
                   "gb_frame_cloud")
    return cloud if organized else cloud.view(-1, 3)


def frame_to_cloud(depth, seg, camera, trans=None, outlier=0.02, remove_outlier=True):
    """graspnet_dataset.py:110-127 for one frame: -> dict(cloud (H*W,3) f32 of every pixel, workspace_mask (H,W) bool or
    None, mask (H,W) bool = depth > 0 [& workspace mask], index (M,) int32 = np.nonzero(mask) in pixel order,
    cloud_masked (M,3), seg_masked (M,)).  One device->host read (M, the output size)."""
    depth, is_u16 = _depth_arg(depth)
    H, W = depth.shape
    dev = depth.device
    seg32 = seg.to(torch.int32).contiguous()
    cam = ctypes.cast(camera._cam5(), ctypes.c_void_p)
    t12 = _trans12(trans) if remove_outlier else None
    tp = ctypes.cast(t12, ctypes.c_void_p) if t12 is not None else None
    cloud = torch.empty((H * W, 3), dtype=torch.float32, device=dev)
    box = None
    if remove_outlier:
        box = torch.tensor([-1, -1, -1, 0, 0, 0], dtype=torch.int64, device=dev)  # {~0 x3, 0 x3} as uint64 bits
    nwg = (H * W + 255) // 256
    counts = torch.empty(nwg, dtype=torch.int32, device=dev)
    wmask = torch.empty(H * W, dtype=torch.uint8, device=dev)
    with _lib.device_ctx(dev):
        st = _lib.current_stream(dev)
        L = _lib.lib()
        _lib.check(L.gb_frame_cloud(_lib.ptr(depth), is_u16, _lib.ptr(seg32), cam, tp, H, W, _lib.ptr(cloud), _lib.ptr(box),
                                    st), "gb_frame_cloud")
        _lib.check(L.gb_frame_mask(_lib.ptr(depth), is_u16, cam, tp, H, W, _lib.ptr(box), float(outlier), _lib.ptr(wmask),
                                   _lib.ptr(counts), st), "gb_frame_mask")
        ends = torch.cumsum(counts, 0, dtype=torch.int64)
        offsets = (ends - counts).contiguous()
        M = int(ends[-1])  # the one synchronisation: the output size
        index = torch.empty(M, dtype=torch.int32, device=dev)
        _lib.check(L.gb_frame_compact(_lib.ptr(depth), is_u16, cam, tp, H, W, _lib.ptr(box), float(outlier),
                                      _lib.ptr(offsets), _lib.ptr(index), st), "gb_frame_compact")
    idx = index.long()
    mask = torch.zeros(H * W, dtype=torch.bool, device=dev)
    mask[idx] = True
    return {"cloud": cloud, "workspace_mask": wmask.view(H, W).bool() if remove_outlier else None, "mask": mask.view(H, W),
            "index": index, "cloud_masked": cloud.index_select(0, idx), "seg_masked": seg.reshape(-1).index_select(0, idx)}


def sample_points(num_masked, num_points, device, generator=None):
    """The sampling rule of graspnet_dataset.py:128-133 as device indices: `num_points` distinct points when there are
    enough, else every point once and the remainder drawn with replacement."""
    if num_masked >= num_points:
        return torch.randperm(num_masked, device=device, generator=generator)[:num_points]
    extra = torch.randint(0, num_masked, (num_points - num_masked,), device=device, generator=generator)
    return torch.cat([torch.arange(num_masked, device=device), extra], 0)


def transform_point_cloud(cloud, transform, format='4x4'):
    """data_utils.py:28-39 on tensors."""
    if format not in ('3x3', '4x4', '3x4'):
        raise ValueError('Unknown transformation format, only support \'3x3\' or \'4x4\' or \'3x4\'.')
    transform = torch.as_tensor(transform, dtype=cloud.dtype, device=cloud.device)
    if format == '3x3':
        return torch.matmul(transform, cloud.T).T
    ones = cloud.new_ones(cloud.shape[0], 1)
    return torch.matmul(transform, torch.cat([cloud, ones], 1).T).T[:, :3]


def compute_point_dists(A, B):
    """data_utils.py:41-45: (Na,3),(Nb,3) -> (Na,Nb) Euclidean distances."""
    return (A[:, None, :] - B[None, :, :]).norm(dim=-1)


def get_workspace_mask(cloud, seg, trans=None, organized=True, outlier=0):
    """data_utils.py:56-72 on tensors, in the cloud's dtype: strictly inside the bounding box of the points with
    seg > 0 (after `trans`), widened by `outlier`.  frame_to_cloud computes the same mask from the depth frame
    without materialising the float64 cloud."""
    if organized:
        h, w, _ = cloud.shape
        cloud = cloud.reshape(h * w, 3)
        seg = seg.reshape(h * w)
    if trans is not None:
        cloud = transform_point_cloud(cloud, trans)
    foreground = cloud[seg > 0]
    lo, hi = foreground.min(dim=0).values, foreground.max(dim=0).values
    mask = ((cloud > lo - outlier) & (cloud < hi + outlier)).all(dim=1)
    return mask.reshape(h, w) if organized else mask


def remove_invisible_grasp_points(cloud, grasp_points, pose, th=0.01):
    """data_utils.py:47-52: a grasp point is visible when some cloud point lies within `th` of it (1-NN on the HIP
    kNN kernel instead of the reference's dense (Np, N) distance matrix)."""
    from .knn_modules import myknn
    pts = transform_point_cloud(grasp_points, pose, '4x4' if torch.as_tensor(pose).shape[0] == 4 else '3x4').float()
    ref = cloud.float().T.contiguous().unsqueeze(0)
    nn = myknn(ref, pts.T.contiguous().unsqueeze(0), k=1).view(-1) - 1
    return (pts - cloud.float().index_select(0, nn)).norm(dim=1) < th


# ---- a frame's training sample: DataProcessing/graspnet_dataset.py:70-88, 98-237, 252-259 after its file reads ------
def augment_data(point_clouds, object_poses_list, flip=None, rot_angle=None, generator=None):
    """graspnet_dataset.py:70-88: mirror along the YZ plane with probability 1/2, then rotate about X by an angle in
    [-30, 30) degrees; the object poses (3,4) follow.  `flip` / `rot_angle` override the random draws."""
    dev = point_clouds.device
    if flip is None:
        flip = bool(torch.rand((), generator=generator) > 0.5)
    if rot_angle is None:
        rot_angle = float(torch.rand((), generator=generator, dtype=torch.float64)) * (torch.pi / 3) - torch.pi / 6
    mats = []
    if flip:
        mats.append(torch.tensor([[-1., 0, 0], [0, 1, 0], [0, 0, 1]], dtype=torch.float64))
    c, s = torch.cos(torch.tensor(rot_angle, dtype=torch.float64)), torch.sin(torch.tensor(rot_angle, dtype=torch.float64))
    mats.append(torch.tensor([[1, 0, 0], [0, float(c), -float(s)], [0, float(s), float(c)]], dtype=torch.float64))
    object_poses_list = list(object_poses_list)
    for m in mats:
        point_clouds = transform_point_cloud(point_clouds, m.to(dev), '3x3')
        for i in range(len(object_poses_list)):
            pose = object_poses_list[i]
            object_poses_list[i] = torch.matmul(m.to(pose.device), pose.to(torch.float64)).to(torch.float32)
    return point_clouds, object_poses_list


def frame_to_sample(depth, color, seg, camera, num_points=20000, trans=None, remove_outlier=False, obj_idxs=None,
                    poses=None, grasp_labels=None, collision_labels=None, valid_obj_idxs=None, remove_invisible=True,
                    augment=False, generator=None):
    """GraspPoseDataset.get_data / get_data_label (graspnet_dataset.py:98-237) from the arrays its file reads produce,
    on the device: depth (H,W), color (H,W,3) float32 in [0,1], seg (H,W); with labels: obj_idxs (K,), poses (3,4,K),
    grasp_labels {obj_idx: (points (Np,3), offsets (Np,V,A,D,3), scores (Np,V,A,D), tolerance (Np,V,A,D))},
    collision_labels {i: bool (Np,V,A,D)} of this scene, valid_obj_idxs.  -> the reference's ret_dict (tensors instead
    of numpy arrays) plus '_cloud_idxs' / '_grasp_idxs' (the random choices made, for reproduction)."""
    dev = depth.device
    f = frame_to_cloud(depth, seg, camera, trans=trans, outlier=0.02, remove_outlier=remove_outlier)
    pix = f["index"].long()
    n_masked = pix.numel()
    idxs = sample_points(n_masked, num_points, dev, generator)                               # :128-133 / :175-180
    sel = pix.index_select(0, idxs)
    color_sampled = color.reshape(-1, color.shape[-1]).index_select(0, sel)
    seg_sampled = seg.reshape(-1).index_select(0, sel)
    # the reference carries the cloud in float64 up to the final astype(float32): only augmentation can tell
    if augment:
        fx, fy, cx, cy, scale = (torch.tensor(float(v), dtype=torch.float64, device=dev) for v in camera._cam5())
        z = depth.reshape(-1).index_select(0, sel).to(torch.float64) / scale
        u, v = (sel % depth.shape[1]).to(torch.float64), (sel // depth.shape[1]).to(torch.float64)
        cloud_sampled = torch.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], dim=1)
    else:
        cloud_sampled = f["cloud"].index_select(0, sel)
    ret = {"_cloud_idxs": idxs}
    if obj_idxs is None:
        ret.update(point_clouds=cloud_sampled.to(torch.float32), cloud_colors=color_sampled.to(torch.float32))
        return ret
    objectness_label = seg_sampled.clamp(max=1)                                               # :184-185
    obj_list = [int(o) for o in obj_idxs]
    per_obj = (seg_sampled.view(1, -1) == torch.tensor(obj_list, device=dev).view(-1, 1)).sum(dim=1).tolist()
    out = {k: [] for k in ("object_poses_list", "grasp_points_list", "grasp_offsets_list", "grasp_labels_list",
                           "grasp_tolerance_list", "_grasp_idxs")}
    cloud32 = cloud_sampled.to(torch.float32)
    for i, obj_idx in enumerate(obj_list):
        if obj_idx not in valid_obj_idxs or per_obj[i] < 50:                                  # :193-196
            continue
        pose = torch.as_tensor(poses[:, :, i], device=dev)
        points, offsets, scores, tolerance = grasp_labels[obj_idx]
        collision = collision_labels[i]
        if remove_invisible:                                                                  # :201-207
            visible = remove_invisible_grasp_points(cloud32[seg_sampled == obj_idx], points, pose, th=0.01)
            points, offsets, scores, tolerance, collision = (t[visible] for t in (points, offsets, scores, tolerance,
                                                                                   collision))
        n = points.shape[0]
        pick = torch.randperm(n, device=dev, generator=generator)[:min(max(int(n / 4), 300), n)]   # :209
        collision = collision.index_select(0, pick)
        zero = torch.zeros((), dtype=scores.dtype, device=dev)
        out["object_poses_list"].append(pose)
        out["grasp_points_list"].append(points.index_select(0, pick))
        out["grasp_offsets_list"].append(offsets.index_select(0, pick))
        out["grasp_labels_list"].append(torch.where(collision, zero, scores.index_select(0, pick)))         # :212-215
        out["grasp_tolerance_list"].append(torch.where(collision, zero.to(tolerance.dtype), tolerance.index_select(0, pick)))
        out["_grasp_idxs"].append(pick)
    if augment:
        cloud_sampled, out["object_poses_list"] = augment_data(cloud_sampled, out["object_poses_list"], generator=generator)
    ret.update(point_clouds=cloud_sampled.to(torch.float32), cloud_colors=color_sampled.to(torch.float32),
               objectness_label=objectness_label.to(torch.int64), **out)
    return ret


def collate_fn(batch):
    """graspnet_dataset.py:252-259 for samples made of tensors: stack arrays, recurse into dicts, keep lists nested."""
    from collections.abc import Mapping, Sequence
    first = batch[0]
    if torch.is_tensor(first):
        return torch.stack(batch, 0)
    if isinstance(first, Mapping):
        return {key: collate_fn([d[key] for d in batch]) for key in first}
    if isinstance(first, Sequence):
        return [list(b) for b in batch]
    raise TypeError("batch must contain tensors, dicts or lists; found {}".format(type(first)))
