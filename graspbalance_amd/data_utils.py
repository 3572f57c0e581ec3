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


def remove_invisible_grasp_points(cloud, grasp_points, pose, th=0.01):
    """data_utils.py:47-52: a grasp point is visible when some cloud point lies within `th` of it (1-NN on the HIP
    kNN kernel instead of the reference's dense (Np, N) distance matrix)."""
    from .knn_modules import myknn
    pts = transform_point_cloud(grasp_points, pose, '4x4' if torch.as_tensor(pose).shape[0] == 4 else '3x4').float()
    ref = cloud.float().T.contiguous().unsqueeze(0)
    nn = myknn(ref, pts.T.contiguous().unsqueeze(0), k=1).view(-1) - 1
    return (pts - cloud.float().index_select(0, nn)).norm(dim=1) < th
