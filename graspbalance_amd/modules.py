"""Grasp heads and seed re-sampling with the names / parameters of the reference's
TrainModel/modules.py (ForegroundSampling :19, GraspableDetection :49, GraspWidthGrouping :89,
GraspPoseParametersDetection :127, ToleranceNet :155, ObjectBalanceSampling :178)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused_mlp
from . import pytorch_utils as pt_utils
from .loss_utils import batch_viewpoint_params_to_matrix, grasp_view_rotations_on, grasp_views_on
from .pointnet2_utils import CylinderQueryAndGroup, furthest_point_sample


def _resample_seeds(end_points, per_cloud_indices):
    """Replace the FPS seeds by the given per-cloud index sets (list of (1024,) int64 tensors)."""
    points = end_points['point_clouds']
    features = end_points['up_sample_features'].permute(0, 2, 1)  # (B,N,256)
    fp2_inds = torch.stack(per_cloud_indices, 0)
    end_points['fp2_inds_fps'] = end_points['fp2_inds']
    end_points['fp2_inds'] = fp2_inds.int()
    end_points['fp2_xyz'] = torch.gather(points, 1, fp2_inds.unsqueeze(-1).expand(-1, -1, 3))
    end_points['fp2_features'] = torch.gather(
        features, 1, fp2_inds.unsqueeze(-1).expand(-1, -1, features.size(-1))).permute(0, 2, 1)
    return end_points


def _fps_within(points, mask_inds, count):
    """FPS of `count` points among points[mask_inds]; returns indices into `points`."""
    picked = furthest_point_sample(points[mask_inds].unsqueeze(0).contiguous(), count)[0].long()
    return torch.gather(mask_inds, 0, picked)


def ForegroundSampling(end_points):
    """1024 seeds by FPS over the points labelled foreground (seed_cluster == 1)."""
    seg = end_points["seed_cluster"]
    picks = [_fps_within(end_points['point_clouds'][i], torch.where(seg[i] == 1)[0], 1024)
             for i in range(seg.shape[0])]
    return _resample_seeds(end_points, picks)


def _object_shares(num_labels_with_background, num_objects):
    """Seeds per object as the reference computes them (modules.py:190-192): 1024 // (number of distinct labels - 1),
    remainder to the last - i.e. label 0 is assumed present; without it the reference runs out of shares."""
    count = num_labels_with_background - 1
    share = [1024 // count for _ in range(count)]  # ZeroDivisionError for a background-only cloud, like the reference
    share[-1] += 1024 % count
    if num_objects > count:
        raise IndexError("list index out of range")  # the reference's points_per_object[t]
    return share


def _object_balance_sampling_loop(end_points):
    """The reference's composition: one furthest_point_sample call per object per cloud."""
    seg = end_points["seed_cluster"]
    picks = []
    for i in range(seg.shape[0]):
        uniq = torch.unique(seg[i])
        labels = [j for j in uniq if j != 0]
        share = _object_shares(len(uniq), len(labels))
        picks.append(torch.cat([_fps_within(end_points['point_clouds'][i], torch.where(seg[i] == j)[0], n)
                                for j, n in zip(labels, share)], 0))
    return _resample_seeds(end_points, picks)


def ObjectBalanceSampling(end_points):
    """1024 seeds split evenly over the segmented objects (label 0 = background), FPS inside each.  On the GPU the
    per-object FPS calls of all clouds run as ONE segmented launch (fused_ops.fps_segments): a stable sort by label
    lists every object's points in index order - what `torch.where(seg == j)` yields - and one host read of the
    label counts replaces the unique / where synchronisations per object.  Same picks, bit for bit."""
    seg = end_points["seed_cluster"]
    points = end_points['point_clouds']
    if not (points.is_cuda and points.dtype == torch.float32 and points.is_contiguous()):
        return _object_balance_sampling_loop(end_points)
    from . import fused_ops
    B, N = seg.shape
    sorted_labels, order = torch.sort(seg, dim=1, stable=True)
    per_cloud = [torch.unique_consecutive(sorted_labels[i], return_counts=True) for i in range(B)]
    host = [(lab.tolist(), cnt.tolist()) for lab, cnt in per_cloud]
    member_parts, seg_sizes, sample_counts = [], [], []
    for i, (labs, cnts) in enumerate(host):
        objects = [k for k, lab in enumerate(labs) if lab != 0]
        share = _object_shares(len(labs), len(objects))
        start = 0
        starts = []
        for c in cnts:
            starts.append(start)
            start += c
        for k, m in zip(objects, share):
            member_parts.append(order[i, starts[k]:starts[k] + cnts[k]] + i * N)
            seg_sizes.append(cnts[k])
            sample_counts.append(m)
    members = torch.cat(member_parts, 0)                                  # flat (cloud * N + point) ids
    packed = torch.index_select(points.view(B * N, 3), 0, members)
    picked = fused_ops.fps_segments(packed, seg_sizes, sample_counts)     # indices into `members`
    flat = torch.index_select(members, 0, picked).view(B, 1024)
    picks = [flat[i] - i * N for i in range(B)]
    return _resample_seeds(end_points, picks)


# Which grasp heads run on the fused channel-last stack (own MFMA GEMMs) instead of Conv1d / BatchNorm1d through torch
# (MIOpen / rocBLAS): "1" = all of them (default since round 4: GraspableDetection and the stage-2 tail - scale fusion,
# gate, the two depth heads), "gd" = GraspableDetection only, "0" = none.  History (ms per step, one box each): rounds 2-3
# none 22.6, gd 22.8-23.2, all 23.6-24.7 - the 4 096 / 16 384-row products of the heads were too short for the
# row-streaming GEMM and too slow on the register-staged tiles; with the LDS-DMA ring kernel (csrc/gemm_ring.hip) all
# 19.13, gd 19.22, none 19.31.  The tail on own kernels moves the grasp scores from 7e-6 to 9e-6..1.5e-5 of the fp64
# truth on the by-key random network (tests/test_parity_f64_gpu.py).
_HEADS_MODE = os.environ.get("GB_HEADS_FUSED", "1")
_HEADS_FUSED = _HEADS_MODE == "1"
_GD_FUSED = _HEADS_MODE in ("1", "gd")


def _bn_fusable(*bns):
    return all(not (bn.momentum is None and bn.track_running_stats) for bn in bns)


class RowsView:
    """The (B, C, num_seed, num_depth) view-point features kept as channel-last rows (B*num_seed*num_depth, C): what the
    fused stage-2 tail hands to the fused depth heads, so the tensor is never laid out channel-major in between."""

    def __init__(self, rows, B, num_seed, num_depth):
        self.rows, self._size = rows, (B, rows.shape[1], num_seed, num_depth)

    def size(self):
        return self._size


class GraspableDetection(nn.Module):
    """Per-seed objectness (2) + approach-view scores (num_view); picks the top view's rotation."""

    def __init__(self, num_view, seed_feature_dim):
        super().__init__()
        self.num_view = num_view
        self.in_dim = seed_feature_dim
        self.conv1 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv2 = nn.Conv1d(self.in_dim, 2 + self.num_view, 1)
        self.conv3 = nn.Conv1d(2 + self.num_view, 2 + self.num_view, 1)
        self.bn1 = nn.BatchNorm1d(self.in_dim)
        self.bn2 = nn.BatchNorm1d(2 + self.num_view)

    @staticmethod
    def _top_view(view_score):
        """(scores, indices) of every seed's best approach view (modules.py:74: torch.max over the views)."""
        return torch.max(view_score, dim=2)

    def _fused_ok(self, x):
        return _GD_FUSED and fused_mlp.enabled(x) and x.dtype == torch.float32 and _bn_fusable(self.bn1, self.bn2)

    def forward(self, seed_xyz, seed_features, end_points, record=True):
        B, num_seed, _ = seed_xyz.size()
        if self._fused_ok(seed_features):
            # channel-last rows (b, seed): conv1+bn1+ReLU, conv2+bn2+ReLU as one fused stack, conv3 as GEMM + bias
            rows = seed_features.transpose(1, 2).reshape(B * num_seed, -1)
            rows = fused_mlp.conv_bn_act_chain(rows, [(self.conv1, self.bn1), (self.conv2, self.bn2)])
            rows = fused_mlp.linear_bias(rows, self.conv3).view(B, num_seed, -1)
            if record == False:  # noqa: E712
                return rows.transpose(1, 2).contiguous()
            view_score = rows[:, :, 2:2 + self.num_view].contiguous()       # == features[:, 2:].transpose(1, 2)
            objectness = rows[:, :, :2].transpose(1, 2).contiguous()
            return self._record(end_points, rows.device, B, num_seed, objectness, view_score)
        features = F.relu(self.bn1(self.conv1(seed_features)), inplace=True)
        features = F.relu(self.bn2(self.conv2(features)), inplace=True)
        features = self.conv3(features)
        if record == False:  # noqa: E712  (the reference tests equality with False)
            return features
        view_score = features[:, 2:2 + self.num_view, :].transpose(1, 2).contiguous()
        return self._record(end_points, features.device, B, num_seed, features[:, :2, :], view_score)

    def _record(self, end_points, device, B, num_seed, objectness, view_score):
        end_points['objectness_score'] = objectness
        end_points['view_score'] = view_score
        top_view_scores, top_view_inds = self._top_view(view_score)
        template_views = grasp_views_on(device, self.num_view)  # (V,3)
        vp_xyz = template_views[top_view_inds]  # (B,num_seed,3) == gather of the expanded templates
        # batch_viewpoint_params_to_matrix(-vp_xyz, 0) per seed (modules.py:76-79) is row-wise and its argument one of V
        # constants: a gather from the V template rotations, computed once (loss_utils.grasp_view_rotations_on)
        vp_rot = grasp_view_rotations_on(device, self.num_view)[top_view_inds]          # (B,num_seed,3,3)
        end_points['grasp_top_view_inds'] = top_view_inds
        end_points['grasp_top_view_score'] = top_view_scores
        end_points['grasp_top_view_xyz'] = vp_xyz
        end_points['grasp_top_view_rot'] = vp_rot
        return end_points


class GraspWidthGrouping(nn.Module):
    """Cylinder grouping at `len(hmax_list)` depths around every seed -> SharedMLP -> max over samples."""

    def __init__(self, nsample, seed_feature_dim, cylinder_radius=0.05, hmin=-0.02,
                 hmax_list=[0.01, 0.02, 0.03, 0.04]):
        super().__init__()
        self.nsample = nsample
        self.in_dim = seed_feature_dim
        self.cylinder_radius = cylinder_radius
        self.hmin = hmin
        self.hmax_list = list(hmax_list)
        # a plain python list (not ModuleList) as in the reference: groupers add no state_dict keys
        self.groupers = [CylinderQueryAndGroup(cylinder_radius, hmin, hmax, nsample, use_xyz=True)
                         for hmax in hmax_list]
        self.mlps = pt_utils.SharedMLP([self.in_dim, 64, 128, 256], bn=True)

    def _cl_ok(self, pointcloud):
        return fused_mlp.enabled(pointcloud) and fused_mlp.supports(self.mlps) \
            and all(g.use_xyz and g.rotate_xyz and not g.normalize_xyz for g in self.groupers)

    def forward(self, seed_xyz, pointcloud, vp_rot, idx=None, rows=None, channel_last=False):
        """idx: optional precomputed neighbour indices (num_depth,B,num_seed,nsample) from the fused
        multi-query kernel; None runs one cylinder query per depth like the reference.
        rows: optional (x0, RowSet) from fused_mlp.cylinder_rows - the distinct (seed, point) rows of the crops.
        channel_last (fused paths only): return the pooled rows (B*num_seed*num_depth, C) as they are instead of the
        reference's (B, C, num_seed, num_depth) layout."""
        B, num_seed, _, _ = vp_rot.size()
        num_depth = len(self.groupers)
        if rows is not None and self._cl_ok(pointcloud):
            x0, rowset = rows
            out = fused_mlp.shared_mlp_cl(x0, self.mlps, rows=rowset)  # (B*seed*depth, 256), rows (b, seed, depth)
            if channel_last:
                return out
            return out.view(B, num_seed, num_depth, -1).permute(0, 3, 1, 2).contiguous()
        if idx is not None and self._cl_ok(pointcloud):
            # channel-last: rows ordered (b, seed, depth, sample) exactly like the reference's stacked view
            rows = [fused_mlp.group_concat_cl(pointcloud, seed_xyz, idx[d], None, mode=2, rot=vp_rot)
                    .view(B, num_seed, self.nsample, 3) for d in range(num_depth)]
            x0 = torch.stack(rows, dim=2).view(-1, 3)
            out = fused_mlp.shared_mlp_cl(x0, self.mlps, pool_ns=self.nsample)  # (B*seed*depth, 256)
            if channel_last:
                return out
            return out.view(B, num_seed, num_depth, -1).permute(0, 3, 1, 2).contiguous()
        if idx is None:
            grouped = [g(pointcloud, seed_xyz, vp_rot) for g in self.groupers]
        else:
            grouped = [g.group(pointcloud, seed_xyz, vp_rot, idx[d])[1] for d, g in enumerate(self.groupers)]
        grouped_features = torch.stack(grouped, dim=3).view(B, -1, num_seed * num_depth, self.nsample)
        vp_features = self.mlps(grouped_features)
        vp_features = pt_utils.max_over_samples(vp_features, keepdim=True)  # == max_pool2d([1,nsample])
        return vp_features.view(B, -1, num_seed, num_depth)


class _DepthHead(nn.Module):
    """Conv1d 256 -> 128 -> 128 -> out over the (seed, depth) grid."""

    def __init__(self, out_channels):
        super().__init__()
        self.conv1 = nn.Conv1d(256, 128, 1)
        self.conv2 = nn.Conv1d(128, 128, 1)
        self.conv3 = nn.Conv1d(128, out_channels, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.bn2 = nn.BatchNorm1d(128)

    def _run_rows(self, rows, B, num_seed, num_depth):
        """rows (B*num_seed*num_depth, 256) channel-last -> (B, out, num_seed, num_depth) like _run."""
        x = fused_mlp.conv_bn_act_chain(rows, [(self.conv1, self.bn1), (self.conv2, self.bn2)])
        x = fused_mlp.linear_bias(x, self.conv3)
        return x.view(B, num_seed, num_depth, -1).permute(0, 3, 1, 2).contiguous()

    def _run(self, vp_features):
        B, _, num_seed, num_depth = vp_features.size()
        if isinstance(vp_features, RowsView):
            return self._run_rows(vp_features.rows, B, num_seed, num_depth)
        if (_HEADS_FUSED and fused_mlp.enabled(vp_features) and vp_features.dtype == torch.float32
                and _bn_fusable(self.bn1, self.bn2)):
            rows = vp_features.permute(0, 2, 3, 1).reshape(B * num_seed * num_depth, -1)
            return self._run_rows(rows, B, num_seed, num_depth)
        x = vp_features.view(B, -1, num_seed * num_depth)
        x = F.relu(self.bn1(self.conv1(x)), inplace=True)
        x = F.relu(self.bn2(self.conv2(x)), inplace=True)
        return self.conv3(x).view(B, -1, num_seed, num_depth)


class GraspPoseParametersDetection(_DepthHead):
    def __init__(self, num_angle, num_depth):
        super().__init__(3 * num_angle)
        self.num_angle = num_angle
        self.num_depth = num_depth

    def forward(self, vp_features, end_points, record=True):
        out = self._run(vp_features)
        if not record:
            return out
        A = self.num_angle
        end_points['grasp_score_pred'] = out[:, 0:A]
        end_points['grasp_angle_cls_pred'] = out[:, A:2 * A]
        end_points['grasp_width_pred'] = out[:, 2 * A:3 * A]
        return end_points


class ToleranceNet(_DepthHead):
    def __init__(self, num_angle, num_depth):
        super().__init__(num_angle)

    def forward(self, vp_features, end_points, record=True):
        out = self._run(vp_features)
        if not record:
            return out
        end_points['grasp_tolerance_pred'] = out
        return end_points
