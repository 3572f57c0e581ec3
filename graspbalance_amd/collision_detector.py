"""ModelFreeCollisionDetector (collision_detector.py:6-64) on the GPU: the scene is voxel-down-sampled and every
grasp candidate's finger / bottom / approach / inner boxes are counted against it by csrc/collision.hip instead of the
reference's dense (G, M, 3) float64 tensor and eight (G, M) masks.  Same constructor and `detect` signature; arrays are
torch CUDA tensors where the reference takes and returns numpy arrays.  SURVEY.md section 8 row f4.
"""
import torch

from . import _lib


def _f64(a, device):
    return torch.as_tensor(a, device=device).to(torch.float64).contiguous()


def voxel_down_sample(points, voxel_size):
    """open3d's PointCloud.voxel_down_sample(voxel_size) as the detector uses it (:11-14): voxel index =
    floor((p - (min_bound - voxel/2)) / voxel); a voxel's points are summed in fp64 in their original order and divided
    by their number (gb_voxel_mean).  -> (V,3) float64, voxels in index order (open3d: hash-map order)."""
    if not points.is_cuda:
        raise RuntimeError("CPU not supported")
    pts = points.to(torch.float64).contiguous()
    if pts.shape[0] == 0:
        return pts
    lo = pts.min(dim=0).values - voxel_size * 0.5
    idx = torch.floor((pts - lo) / voxel_size).to(torch.int64)
    dims = idx.max(dim=0).values + 1
    key = (idx[:, 0] * dims[1] + idx[:, 1]) * dims[2] + idx[:, 2]
    ks, order = torch.sort(key, stable=True)
    _, counts = torch.unique_consecutive(ks, return_counts=True)
    seg = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=pts.device)
    torch.cumsum(counts, 0, out=seg[1:])
    sp = pts.index_select(0, order).contiguous()
    out = torch.empty((counts.numel(), 3), dtype=torch.float64, device=pts.device)
    with _lib.device_ctx(pts.device):
        _lib.check(_lib.lib().gb_voxel_mean(_lib.ptr(sp), _lib.ptr(seg), _lib.ptr(out), counts.numel(),
                                            _lib.current_stream(pts.device)), "gb_voxel_mean")
    return out


class ModelFreeCollisionDetector:
    def __init__(self, scene_points, voxel_size=0.005):
        self.finger_width = 0.01
        self.finger_length = 0.06
        self.voxel_size = voxel_size
        self.scene_points = voxel_down_sample(scene_points, voxel_size)

    def counts(self, grasp_group, approach_dist=0.03):
        """(G,6) int32: scene points inside the {left finger, right finger, bottom, approach, any of the four, inner}
        volume of every grasp - the row sums of the reference's masks (:37-41, :50)."""
        dev = self.scene_points.device
        approach_dist = max(approach_dist, self.finger_width)
        T = _f64(grasp_group.translations, dev).view(-1, 3)
        R = _f64(grasp_group.rotation_matrices, dev).view(-1, 9)
        h = _f64(grasp_group.heights, dev).view(-1)
        d = _f64(grasp_group.depths, dev).view(-1)
        w = _f64(grasp_group.widths, dev).view(-1)
        G = T.shape[0]
        fl, fw = self.finger_length, self.finger_width
        # the per-grasp scalars of :26-35, each in the reference's evaluation order
        thr = torch.stack([h / 2, d, d - fl, -(w / 2 + fw), -w / 2, w / 2 + fw, w / 2, d - fl - fw,
                           d - fl - fw - approach_dist], dim=1).contiguous()
        out = torch.empty((G, 6), dtype=torch.int32, device=dev)
        with _lib.device_ctx(dev):
            _lib.check(_lib.lib().gb_collision_counts(_lib.ptr(self.scene_points), _lib.ptr(T), _lib.ptr(R),
                                                      _lib.ptr(thr), _lib.ptr(out), G, self.scene_points.shape[0],
                                                      _lib.current_stream(dev)), "gb_collision_counts")
        return out, (h, d, w, approach_dist)

    def detect(self, grasp_group, approach_dist=0.03, collision_thresh=0.05, return_empty_grasp=False,
               empty_thresh=0.01, return_ious=False):
        """collision_detector.py:16-64; `grasp_group` needs .translations (G,3), .rotation_matrices (G,3,3), .heights,
        .depths, .widths (G,) like graspnetAPI's GraspGroup."""
        counts, (h, d, w, approach_dist) = self.counts(grasp_group, approach_dist)
        c = counts.to(torch.float64)
        fl, fw = self.finger_length, self.finger_width
        # a tensor divisor: torch turns `x / python_scalar` into a multiplication by the reciprocal on the device,
        # one ulp away from numpy's division
        vox3 = torch.full_like(h, self.voxel_size ** 3)
        left_right_volume = h * fl * fw / vox3
        bottom_volume = h * (w + 2 * fw) * fw / vox3
        shifting_volume = h * (w + 2 * fw) * approach_dist / vox3
        volume = left_right_volume * 2 + bottom_volume + shifting_volume
        global_iou = c[:, 4] / (volume + 1e-6)
        collision_mask = global_iou > collision_thresh
        if not (return_empty_grasp or return_ious):
            return collision_mask
        ret_value = [collision_mask]
        if return_empty_grasp:
            inner_volume = h * fl * w / vox3
            ret_value.append(c[:, 5] / inner_volume < empty_thresh)
        if return_ious:
            ret_value.append([global_iou, c[:, 0] / (left_right_volume + 1e-6), c[:, 1] / (left_right_volume + 1e-6),
                              c[:, 2] / (bottom_volume + 1e-6), c[:, 3] / (shifting_volume + 1e-6)])
        return ret_value
