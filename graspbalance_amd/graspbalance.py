"""The GraspBalance network with the module tree of the reference's TrainModel/graspbalance.py
(GraspPoseStage1 :24, GraspPoseStage2 :56, GraspPoseStage2_seed_features_multi_scale :77,
GraspBalance :122, pred_decode :139): 9 052 396 parameters, 490 state_dict entries, same key names.

Differences that do not change results: the 16 cylinder queries of stage 2 run as ONE fused HIP
pass (``fused_ops.cylinder_query_multi``, bit-identical indices); ``fused_cylinder=False`` runs
them one by one like the reference.
"""
import numpy as np
import torch
import torch.nn as nn

from . import fused_mlp, fused_ops, modules
from .drp import DRP
from .label_generation import match_grasp_view_and_label, process_grasp_labels
from .loss_utils import GRASP_MAX_TOLERANCE, GRASP_MAX_WIDTH, batch_viewpoint_params_to_matrix
from .modules import (GraspableDetection, GraspPoseParametersDetection, GraspWidthGrouping,
                      ObjectBalanceSampling, ToleranceNet)
from .pointnet2_utils import three_interpolate, three_nn


class GraspPoseStage1(nn.Module):
    """Backbone + graspable / approach-view head."""

    def __init__(self, input_feature_dim=0, num_view=300, obs=False, is_training=True):
        super().__init__()
        self.FeatureExtraction = DRP()
        self.GraspableClasification = GraspableDetection(num_view, 256)  # spelling = reference key name
        self.obs = obs
        self.is_training = is_training

    def forward(self, end_points):
        pointcloud = end_points['point_clouds']
        seed_features, seed_xyz, end_points = self.FeatureExtraction(pointcloud, end_points)
        if self.obs and (not self.is_training):
            dist, idx = three_nn(pointcloud, seed_xyz)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            end_points['up_sample_features'] = three_interpolate(seed_features, idx, weight)
            end_points = ObjectBalanceSampling(end_points)
            seed_xyz = end_points['fp2_xyz']
            seed_features = end_points["fp2_features"]
        return self.GraspableClasification(seed_xyz, seed_features, end_points)


def _stage2_inputs(end_points, is_training):
    if is_training:
        grasp_top_views_rot, _, _, _, end_points = match_grasp_view_and_label(end_points)
        return end_points['batch_grasp_point'], grasp_top_views_rot, end_points
    return end_points['fp2_xyz'], end_points['grasp_top_view_rot'], end_points


class GraspPoseStage2(nn.Module):
    """Single-scale stage 2 (kept for API parity; GraspBalance uses the multi-scale variant)."""

    def __init__(self, num_angle=12, num_depth=4, cylinder_radius=0.05, hmin=-0.02,
                 hmax_list=[0.01, 0.02, 0.03, 0.04], is_training=True):
        super().__init__()
        self.num_angle = num_angle
        self.num_depth = num_depth
        self.is_training = is_training
        self.WidthGroup = GraspWidthGrouping(64, 3, cylinder_radius, hmin, hmax_list)
        self.GraspParameters = GraspPoseParametersDetection(num_angle, num_depth)
        self.tolerance = ToleranceNet(num_angle, num_depth)

    def forward(self, end_points):
        pointcloud = end_points['input_xyz']
        seed_xyz, rot, end_points = _stage2_inputs(end_points, self.is_training)
        vp_features = self.WidthGroup(seed_xyz, pointcloud, rot)
        end_points = self.GraspParameters(vp_features, end_points)
        return self.tolerance(vp_features, end_points)


class GraspPoseStage2_seed_features_multi_scale(nn.Module):
    """Four cylinder radii x four depths, fused by a 1x1 conv, plus sigmoid-gated seed features."""

    def __init__(self, num_angle=12, num_depth=4, cylinder_radius=0.05, hmin=-0.02,
                 hmax_list=[0.01, 0.02, 0.03, 0.04], is_training=True, fused_cylinder=True):
        super().__init__()
        self.num_angle = num_angle
        self.num_depth = num_depth
        self.is_training = is_training
        self.fused_cylinder = fused_cylinder
        self.WidthGroup1 = GraspWidthGrouping(64, 3, cylinder_radius * 0.25, hmin, hmax_list)
        self.WidthGroup2 = GraspWidthGrouping(64, 3, cylinder_radius * 0.5, hmin, hmax_list)
        self.WidthGroup3 = GraspWidthGrouping(64, 3, cylinder_radius * 0.75, hmin, hmax_list)
        self.WidthGroup4 = GraspWidthGrouping(64, 3, cylinder_radius, hmin, hmax_list)
        self.GraspParameters = GraspPoseParametersDetection(num_angle, num_depth)
        self.tolerance = ToleranceNet(num_angle, num_depth)
        self.fuse_multi_scale = nn.Conv1d(256 * 4, 256, 1)
        self.gate_fusion = nn.Sequential(nn.Conv1d(256, 256, 1), nn.Sigmoid())

    def forward(self, end_points):
        pointcloud = end_points['input_xyz']
        seed_xyz, rot, end_points = _stage2_inputs(end_points, self.is_training)
        groups = [self.WidthGroup1, self.WidthGroup2, self.WidthGroup3, self.WidthGroup4]
        if self.fused_cylinder:
            seed_xyz = seed_xyz.contiguous()
            rot = rot.contiguous()
            g0 = groups[0]
            idx = fused_ops.cylinder_query_multi(pointcloud, seed_xyz, rot, [g.cylinder_radius for g in groups],
                                                 g0.hmin, g0.hmax_list, g0.nsample)
            cl = all(g._cl_ok(pointcloud) for g in groups)  # every scale returns channel-last pooled rows
            if (fused_mlp.cyl_dedup_enabled() and cl
                    and len(g0.hmax_list) in (1, 2, 4) and len(g0.hmax_list) * g0.nsample <= 256):
                # the crops of a seed are nested: run each MLP on the distinct (seed, point) rows only
                # (no read-back of the row counts where every scale's stack can take them from the device)
                cap = seed_xyz.size(0) * seed_xyz.size(1) * len(g0.hmax_list) * g0.nsample
                static = all(fused_mlp.crop_static_ok(cap, fused_mlp.shared_mlp_widths(g.mlps), len(g.hmax_list))
                             for g in groups)
                rows = fused_mlp.cylinder_rows(idx, pointcloud, seed_xyz, rot, static=static)
                scales = [g(seed_xyz, pointcloud, rot, rows=rows[i], channel_last=True) for i, g in enumerate(groups)]
            else:
                scales = [g(seed_xyz, pointcloud, rot, idx=idx[i], channel_last=cl) for i, g in enumerate(groups)]
        else:
            cl = False
            scales = [g(seed_xyz, pointcloud, rot) for g in groups]
        seed_features = end_points['fp2_features']
        if cl and modules._HEADS_FUSED and seed_features.dtype == torch.float32:
            # the whole tail on channel-last rows and the hand-written GEMMs: fuse convolution over the concatenated
            # scales, sigmoid gate of the seed features, broadcast add over the depths - the depth heads then take the
            # rows as they are (modules.RowsView), nothing is laid out channel-major in between
            B, num_seed, num_depth = seed_xyz.size(0), seed_xyz.size(1), len(g0.hmax_list)
            fused = fused_mlp.linear_bias(torch.cat(scales, dim=1), self.fuse_multi_scale)          # (B*Ns*D, 256)
            seed_rows = seed_features.transpose(1, 2).reshape(B * num_seed, -1)
            gated = torch.sigmoid(fused_mlp.linear_bias(seed_rows, self.gate_fusion[0])) * seed_rows   # (B*Ns, 256)
            rows = (fused.view(B * num_seed, num_depth, -1) + gated.unsqueeze(1)).view(B * num_seed * num_depth, -1)
            vp_features = modules.RowsView(rows, B, num_seed, num_depth)
            end_points = self.GraspParameters(vp_features, end_points)
            return self.tolerance(vp_features, end_points)
        if cl:
            # the 1x1 fuse convolution on the channel-last rows (b, seed, depth): cat along the channels + one linear,
            # then ONE transpose to the reference layout instead of four (one per scale) before the cat
            B, num_seed, num_depth = seed_xyz.size(0), seed_xyz.size(1), len(g0.hmax_list)
            w = self.fuse_multi_scale.weight
            fused = nn.functional.linear(torch.cat(scales, dim=1), w.view(w.size(0), w.size(1)), self.fuse_multi_scale.bias)
            fused = fused.view(B, num_seed, num_depth, -1).permute(0, 3, 1, 2).contiguous()
        else:
            B, _, num_seed, num_depth = scales[0].size()
            fused = self.fuse_multi_scale(torch.cat(scales, dim=1).view(B, -1, num_seed * num_depth))
            fused = fused.view(B, -1, num_seed, num_depth)
        gated = self.gate_fusion(seed_features) * seed_features
        vp_features = fused + gated.unsqueeze(3).repeat(1, 1, 1, 4)
        end_points = self.GraspParameters(vp_features, end_points)
        return self.tolerance(vp_features, end_points)


class GraspBalance(nn.Module):
    def __init__(self, input_feature_dim=0, num_view=300, num_angle=12, num_depth=4, cylinder_radius=0.08,
                 hmin=-0.02, hmax_list=[0.01, 0.02, 0.03, 0.04], is_training=True, obs=False,
                 fused_cylinder=True):
        super().__init__()
        self.is_training = is_training
        self.view_estimator = GraspPoseStage1(input_feature_dim, num_view, is_training=is_training, obs=obs)
        self.grasp_generator = GraspPoseStage2_seed_features_multi_scale(
            num_angle, num_depth, cylinder_radius, hmin, hmax_list, is_training, fused_cylinder=fused_cylinder)

    def forward(self, end_points):
        with fused_mlp.deferred_counters():  # one multi-tensor BN batch-counter update for the whole pass
            end_points = self.view_estimator(end_points)
            if self.is_training:
                end_points = process_grasp_labels(end_points)
            return self.grasp_generator(end_points)


def pred_decode(end_points):
    """Per cloud (Ng,17) grasps: score, width, height, depth, rotation (9), centre (3), object id
    (graspbalance.py:139-192).  Same element-wise arithmetic as the reference's per-cloud loop, on the whole batch at
    once; the objectness mask is read back ONCE (B x Ns booleans) and the kept seeds of all clouds are gathered by index -
    the loop's 7 x B masked selections were 28 synchronisations and ~140 launches of an inference call."""
    objectness_score = end_points['objectness_score'].float()          # (B,2,Ns)
    grasp_score = end_points['grasp_score_pred'].float()               # (B,A,Ns,D)
    B, Ns = objectness_score.shape[0], objectness_score.shape[2]
    grasp_center = end_points['fp2_xyz'].float()
    approaching = -end_points['grasp_top_view_xyz'].float()
    grasp_width = torch.clamp(1.2 * end_points['grasp_width_pred'], min=0, max=GRASP_MAX_WIDTH)
    grasp_tolerance = end_points['grasp_tolerance_pred']
    # best in-plane angle per (seed, depth), then best depth per seed
    angle_cls = torch.argmax(end_points['grasp_angle_cls_pred'], 1)    # (B,Ns,D)
    grasp_angle = angle_cls.float() / 12 * np.pi
    pick_a = angle_cls.unsqueeze(1)
    grasp_score = torch.gather(grasp_score, 1, pick_a).squeeze(1)
    grasp_width = torch.gather(grasp_width, 1, pick_a).squeeze(1)
    grasp_tolerance = torch.gather(grasp_tolerance, 1, pick_a).squeeze(1)
    pick_d = torch.argmax(grasp_score, 2, keepdims=True)               # (B,Ns,1)
    grasp_depth = (pick_d.float() + 1) * 0.01
    grasp_score = torch.gather(grasp_score, 2, pick_d)
    grasp_angle = torch.gather(grasp_angle, 2, pick_d)
    grasp_width = torch.gather(grasp_width, 2, pick_d)
    grasp_tolerance = torch.gather(grasp_tolerance, 2, pick_d)
    keep = torch.argmax(objectness_score, 1) == 1                      # (B,Ns)
    graspable_confident = torch.softmax(objectness_score, dim=1)[:, 1, :].unsqueeze(2)
    grasp_score = grasp_score * graspable_confident
    keep_host = keep.cpu()                                             # the one synchronisation
    counts = keep_host.sum(1).tolist()
    idx = keep_host.reshape(-1).nonzero().squeeze(1).to(keep.device, non_blocking=True)

    def sel(t):
        return t.reshape(B * Ns, -1).index_select(0, idx)
    grasp_score, grasp_width, grasp_depth = sel(grasp_score), sel(grasp_width), sel(grasp_depth)
    approaching, grasp_angle = sel(approaching), sel(grasp_angle)
    grasp_center, grasp_tolerance = sel(grasp_center), sel(grasp_tolerance)
    grasp_score = grasp_score * grasp_tolerance / GRASP_MAX_TOLERANCE
    Ng = grasp_angle.size(0)
    rotation_matrix = batch_viewpoint_params_to_matrix(approaching.view(Ng, 3), grasp_angle.view(Ng)).view(Ng, 9)
    grasp_height = 0.02 * torch.ones_like(grasp_score)
    obj_ids = -1 * torch.ones_like(grasp_score)
    preds = torch.cat([grasp_score, grasp_width, grasp_height, grasp_depth, rotation_matrix, grasp_center, obj_ids],
                      axis=-1)
    return list(torch.split(preds, counts, 0))


def _pred_decode_loop(end_points):
    """The reference's composition (graspbalance.py:139-192), cloud by cloud: seven boolean-mask selections per cloud, each a
    device -> host synchronisation.  Kept as the checker of pred_decode."""
    grasp_preds = []
    for i in range(len(end_points['point_clouds'])):
        objectness_score = end_points['objectness_score'][i].float()
        grasp_score = end_points['grasp_score_pred'][i].float()
        grasp_center = end_points['fp2_xyz'][i].float()
        approaching = -end_points['grasp_top_view_xyz'][i].float()
        grasp_width = torch.clamp(1.2 * end_points['grasp_width_pred'][i], min=0, max=GRASP_MAX_WIDTH)
        grasp_tolerance = end_points['grasp_tolerance_pred'][i]
        # best in-plane angle per (seed, depth), then best depth per seed
        angle_cls = torch.argmax(end_points['grasp_angle_cls_pred'][i], 0)
        grasp_angle = angle_cls.float() / 12 * np.pi
        pick_a = angle_cls.unsqueeze(0)
        grasp_score = torch.gather(grasp_score, 0, pick_a).squeeze(0)
        grasp_width = torch.gather(grasp_width, 0, pick_a).squeeze(0)
        grasp_tolerance = torch.gather(grasp_tolerance, 0, pick_a).squeeze(0)
        pick_d = torch.argmax(grasp_score, 1, keepdims=True)
        grasp_depth = (pick_d.float() + 1) * 0.01
        grasp_score = torch.gather(grasp_score, 1, pick_d)
        grasp_angle = torch.gather(grasp_angle, 1, pick_d)
        grasp_width = torch.gather(grasp_width, 1, pick_d)
        grasp_tolerance = torch.gather(grasp_tolerance, 1, pick_d)
        keep = torch.argmax(objectness_score, 0) == 1
        graspable_confident = torch.softmax(objectness_score, dim=0)[1, :].unsqueeze(1)
        grasp_score = (grasp_score * graspable_confident)[keep]
        grasp_width, grasp_depth, approaching = grasp_width[keep], grasp_depth[keep], approaching[keep]
        grasp_angle, grasp_center, grasp_tolerance = grasp_angle[keep], grasp_center[keep], grasp_tolerance[keep]
        grasp_score = grasp_score * grasp_tolerance / GRASP_MAX_TOLERANCE
        Ns = grasp_angle.size(0)
        rotation_matrix = batch_viewpoint_params_to_matrix(approaching.view(Ns, 3), grasp_angle.view(Ns)).view(Ns, 9)
        grasp_height = 0.02 * torch.ones_like(grasp_score)
        obj_ids = -1 * torch.ones_like(grasp_score)
        grasp_preds.append(torch.cat([grasp_score, grasp_width, grasp_height, grasp_depth, rotation_matrix,
                                      grasp_center, obj_ids], axis=-1))
    return grasp_preds
