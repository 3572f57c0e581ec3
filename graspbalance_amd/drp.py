"""DRP backbone (the one GraspBalance uses): the PointNet++ SA/FP skeleton of backbone.py with
3/6/3/3 inverted-residual MLP blocks after SA1-4 (reference TrainModel/drp.py: get_reduction_fn :20,
LocalAggregation :32, InvResMLP :70, ResBlock :120, DRP :150).  Module / parameter names follow the
reference so checkpoints load: e.g. ``InvResMLP_blocks1.2.pwconv.0.0.weight`` is (512,128,1)."""
import logging
from typing import List

import torch
import torch.nn as nn

from .backbone import break_up_pc, make_sa
from .modified_net_tools.activation import CHANNEL_MAP, create_act
from .modified_net_tools.conv import create_convblock1d, create_convblock2d
from . import fused_mlp
from . import pytorch_utils as pt_utils
from .modified_net_tools.group import QueryAndGroup, ball_query, create_grouper, get_aggregation_feautres
from .pointnet2_modules import PointnetFPModule


def get_reduction_fn(reduction):
    reduction = 'mean' if reduction.lower() == 'avg' else reduction
    assert reduction in ['sum', 'max', 'mean']
    if reduction == 'max':
        return lambda x: pt_utils.max_over_samples(x)
    if reduction == 'mean':
        return lambda x: torch.mean(x, dim=-1, keepdim=False)
    return lambda x: torch.sum(x, dim=-1, keepdim=False)


class LocalAggregation(nn.Module):
    """ball-query group -> [dp, fj] -> 1x1 conv stack -> pool over the neighbours."""

    def __init__(self, channels: List[int], norm_args={'norm': 'bn1d'}, act_args={'act': 'relu'},
                 group_args={'NAME': 'ballquery', 'radius': 0.1, 'nsample': 16}, conv_args=None,
                 feature_type='dp_fj', reduction='max', last_act=True, **kwargs):
        super().__init__()
        if kwargs:
            logging.warning(f"kwargs: {kwargs} are not used in {__class__.__name__}")
        channels[0] = CHANNEL_MAP[feature_type](channels[0])
        last = len(channels) - 2
        self.convs = nn.Sequential(*[
            create_convblock2d(channels[i], channels[i + 1], norm_args=norm_args,
                               act_args=None if (i == last and not last_act) else act_args, **conv_args)
            for i in range(len(channels) - 1)])
        self.grouper = create_grouper(group_args)
        self.reduction = reduction.lower()
        self.pool = get_reduction_fn(self.reduction)
        self.feature_type = feature_type

    def forward(self, pf) -> torch.Tensor:
        p, f = pf
        dp, fj = self.grouper(p, p, f)
        fj = get_aggregation_feautres(p, dp, f, fj, self.feature_type)
        return self.pool(self.convs(fj))


def _pointwise(channels, norm_args, act_args, conv_args, less_act):
    last = len(channels) - 2
    return nn.Sequential(*[
        create_convblock1d(channels[i], channels[i + 1], norm_args=norm_args,
                           act_args=act_args if (i != last and not less_act) else None, **conv_args)
        for i in range(len(channels) - 1)])


class InvResMLP(nn.Module):
    """LocalAggregation(C->C) -> pointwise C->expansion*C->C -> residual add -> act."""

    def __init__(self, in_channels, norm_args=None, act_args=None,
                 aggr_args={'feature_type': 'dp_fj', "reduction": 'max'}, group_args={'NAME': 'ballquery'},
                 conv_args=None, expansion=1, use_res=True, num_posconvs=2, less_act=False, **kwargs):
        super().__init__()
        self.use_res = use_res
        mid_channels = int(in_channels * expansion)
        self.convs = LocalAggregation([in_channels, in_channels], norm_args=norm_args,
                                      act_args=act_args if num_posconvs > 0 else None, group_args=group_args,
                                      conv_args=conv_args, **aggr_args, **kwargs)
        if num_posconvs < 1:
            channels = []
        elif num_posconvs == 1:
            channels = [in_channels, in_channels]
        else:
            channels = [in_channels, mid_channels, in_channels]
        self.pwconv = _pointwise(channels, norm_args, act_args, conv_args, less_act)
        self.act = create_act(act_args)

    def fusable(self):
        """The configuration DRP builds: dp_fj features, max reduction, plain ball-query grouper,
        one conv+BN+ReLU aggregation layer, conv+BN+ReLU -> conv+BN pointwise pair, residual, ReLU."""
        la = self.convs
        g = la.grouper
        def is_block(seq, with_act):
            mods = list(seq.children())
            want = 3 if with_act else 2
            return (len(mods) == want and isinstance(mods[0], (nn.Conv1d, nn.Conv2d)) and mods[0].bias is None
                    and isinstance(mods[1], (nn.BatchNorm1d, nn.BatchNorm2d))
                    and not (mods[1].momentum is None and mods[1].track_running_stats)
                    and (not with_act or isinstance(mods[2], nn.ReLU)))
        return (la.feature_type == 'dp_fj' and la.reduction == 'max' and isinstance(g, QueryAndGroup)
                and g.relative_xyz and not g.normalize_dp and not g.return_only_idx
                and len(la.convs) == 1 and is_block(la.convs[0], True) and len(self.pwconv) == 2
                and is_block(self.pwconv[0], True) and is_block(self.pwconv[1], False) and self.use_res
                and isinstance(self.act, nn.ReLU))

    def forward_cl(self, p, f_cl, idx=None, geo=None):
        """Channel-last execution: f_cl (B,N,C) -> (B,N,C).  `idx` (B,N,ns) and the grouping summary `geo`
        (fused_mlp.LocalGeometry) may be shared by all blocks of a stage (same points, same radius)."""
        B, N, C = f_cl.shape
        g = self.convs.grouper
        if idx is None:
            idx = ball_query(g.radius, g.nsample, p, p)
        agg_conv, agg_bn = self.convs.convs[0][0], self.convs.convs[0][1]
        if fused_mlp.local_agg_supported(agg_conv.weight.shape[0], g.nsample) and fused_mlp.local_agg_enabled():
            # the 1x1 conv commutes with the gather: no grouped (B*N*ns, 3+C) tensor at all
            if geo is None:
                geo = fused_mlp.LocalGeometry(p, p, idx, mode=0)
            agg = fused_mlp.local_agg_pool(f_cl.reshape(B * N, C), agg_conv, agg_bn, geo)            # (B*N, C)
        else:
            x0 = fused_mlp.group_concat_cl(p, p, idx, f_cl, mode=0)               # [dp, fj] rows
            agg = fused_mlp.conv_bn_act(x0, agg_conv, agg_bn, relu=True, pool_ns=g.nsample)   # (B*N, C)
        # C -> 4C -> C pointwise pair as one fused stack; act(bn(.) + identity) at the end
        out = fused_mlp.conv_bn_act_chain(agg, [(self.pwconv[0][0], self.pwconv[0][1]),
                                                (self.pwconv[1][0], self.pwconv[1][1])],
                                          residual=f_cl.reshape(B * N, C), relu_last=True)
        return out.view(B, N, C)

    def forward(self, pf):
        p, f = pf
        if fused_mlp.enabled(f) and self.fusable():
            out = self.forward_cl(p, f.transpose(1, 2).contiguous())
            return [p, out.transpose(1, 2)]
        identity = f
        f = self.pwconv(self.convs([p, f]))
        if f.shape[-1] == identity.shape[-1] and self.use_res:
            f += identity
        return [p, self.act(f)]


def run_stage(blocks, p, f):
    """A Sequential of InvResMLP blocks over the same points: on the fused path the features stay
    channel-last across the blocks and the ball query (same p, same radius) is done once."""
    blocks = list(blocks)
    if not (fused_mlp.enabled(f) and blocks and all(isinstance(b, InvResMLP) and b.fusable() for b in blocks)):
        for blk in blocks:
            p, f = blk([p, f])
        return p, f
    g0 = blocks[0].convs.grouper
    same = all(b.convs.grouper.radius == g0.radius and b.convs.grouper.nsample == g0.nsample for b in blocks)
    idx = ball_query(g0.radius, g0.nsample, p, p) if same else None
    geo = None
    if same and fused_mlp.local_agg_enabled() and all(
            fused_mlp.local_agg_supported(b.convs.convs[0][0].weight.shape[0], g0.nsample) for b in blocks):
        geo = fused_mlp.LocalGeometry(p, p, idx, mode=0)
    f_cl = f.transpose(1, 2).contiguous()
    for blk in blocks:
        f_cl = blk.forward_cl(p, f_cl, idx, geo)
    return p, f_cl.transpose(1, 2)   # a view: the next consumer's transpose(1, 2).contiguous() costs nothing


class ResBlock(nn.Module):
    def __init__(self, in_channels, norm_args=None, act_args=None,
                 aggr_args={'feature_type': 'dp_fj', "reduction": 'max'}, group_args={'NAME': 'ballquery'},
                 conv_args=None, expansion=1, use_res=True, **kwargs):
        super().__init__()
        self.use_res = use_res
        mid_channels = in_channels * expansion
        self.convs = LocalAggregation([in_channels, in_channels, mid_channels, in_channels], norm_args=norm_args,
                                      act_args=None, group_args=group_args, conv_args=conv_args, **aggr_args,
                                      **kwargs)
        self.act = create_act(act_args)

    def forward(self, pf):
        p, f = pf
        identity = f
        f = self.convs([p, f])
        if f.shape[-1] == identity.shape[-1] and self.use_res:
            f += identity
        return [p, self.act(f)]


import os
_PREFETCH_LEVEL = int(os.environ.get("GB_PREFETCH_LEVEL", "1"))  # A/B switch: after which SA level the sampling of the next batch starts

# the gradient cut of the data-parallel graph step (see DRP.forward): end_points flag / the two tensors it leaves
GRAD_CUT, GRAD_CUT_OUT, GRAD_CUT_IN, GRAD_CUT_LEVEL = '_grad_cut', '_grad_cut_out', '_grad_cut_in', 2


def grad_cut_param_index(net):
    """Index (in net.parameters() order, trainable ones) of the first parameter BEHIND the gradient cut, or None when the
    network has no DRP backbone: parameters [index, end) receive their gradients in the first part of the backward."""
    drp = next((m for m in net.modules() if isinstance(m, DRP)), None)
    if drp is None:
        return None
    first = next(iter(getattr(drp, 'InvResMLP_blocks%d' % GRAD_CUT_LEVEL).parameters()), None)
    params = [q for q in net.parameters() if q.requires_grad]
    for i, q in enumerate(params):
        if q is first:
            # the slices must be exactly "registered before the cut" / "registered after": true for DRP (sa1, blocks1,
            # sa2 | blocks2 ... fp2) when the backbone is the first thing the network registers
            shallow = [t for lvl in range(1, GRAD_CUT_LEVEL + 1) for t in getattr(drp, 'sa%d' % lvl).parameters()]
            shallow += [t for lvl in range(1, GRAD_CUT_LEVEL) for t in getattr(drp, 'InvResMLP_blocks%d' % lvl).parameters()]
            same = {id(t) for t in params[:i]} == {id(t) for t in shallow if t.requires_grad}
            return i if same else None
    return None


# (channels, ball radius, nsample, number of InvResMLP blocks) after SA1..SA4 (drp.py:167-262)
STAGE_SPECS = ((128, 0.08, 64, 3), (256, 0.2, 32, 6), (256, 0.4, 16, 3), (256, 0.6, 16, 3))


class DRP(nn.Module):
    def __init__(self, input_feature_dim=0):
        super().__init__()
        self.aggr_args = {'feature_type': 'dp_fj', "reduction": 'max'}
        self.norm_args = {'norm': 'bn'}
        self.act_args = {'act': 'relu'}
        self.conv_args = {'order': 'conv-norm-act'}
        self.use_res = True
        self.expansion = 4
        for level, (channels, radius, nsample, depth) in enumerate(STAGE_SPECS):
            setattr(self, 'sa%d' % (level + 1), make_sa(level, input_feature_dim))
            group_args = {'NAME': 'ballquery', 'radius': radius, 'nsample': nsample}
            blocks = [InvResMLP(in_channels=channels, aggr_args=self.aggr_args, norm_args=self.norm_args,
                                act_args=self.act_args, group_args=group_args, conv_args=self.conv_args,
                                expansion=self.expansion, use_res=self.use_res) for _ in range(depth)]
            setattr(self, 'InvResMLP_blocks%d' % (level + 1), nn.Sequential(*blocks))
        self.fp1 = PointnetFPModule(mlp=[256 + 256, 256, 256])
        self.fp2 = PointnetFPModule(mlp=[256 + 256, 256, 256])

    def _break_up_pc(self, pc):
        return break_up_pc(pc)

    def forward(self, pointcloud: torch.Tensor, end_points=None):
        if not end_points:
            end_points = {}
        xyz, features = break_up_pc(pointcloud)
        end_points['input_xyz'] = xyz
        end_points['input_features'] = features
        pre = end_points.pop('_sa1_inds_prefetched', None)  # sampled one step ahead on a side stream (prefetch.py)
        for level in (1, 2, 3, 4):
            inds = pre if (level == 1 and pre is not None and pre.shape == (xyz.shape[0], self.sa1.npoint)) else None
            xyz, features, fps_inds = getattr(self, 'sa%d' % level)(xyz, features, inds)
            if level == _PREFETCH_LEVEL:
                hook = end_points.pop('_after_sa1', None)
                if hook is not None:
                    hook()
            hooks = end_points.get('_after_level')     # {level: callable}: run once that set-abstraction level is enqueued
            if hooks and level in hooks:
                hooks.pop(level)()
            if level == GRAD_CUT_LEVEL and end_points.get(GRAD_CUT) and features is not None and features.requires_grad:
                # data-parallel graph execution (train.Trainer): the backward is cut HERE into "everything behind" (the
                # heads, the feature propagation, levels 3-4 and this level's blocks: 94 % of the parameters, reached
                # first) and "the rest" (levels 1-2), so that the first slice's gradient all-reduce runs under the
                # second part.  The deep part consumes a detached alias; the caller backpropagates loss -> cut_in,
                # then cut_out with cut_in.grad.  Same values as one backward.
                end_points[GRAD_CUT_OUT] = features
                features = features.detach().requires_grad_()
                end_points[GRAD_CUT_IN] = features
            xyz, features = run_stage(getattr(self, 'InvResMLP_blocks%d' % level), xyz, features)
            if level <= 2:
                end_points['sa%d_inds' % level] = fps_inds
            end_points['sa%d_xyz' % level] = xyz
            end_points['sa%d_features' % level] = features
        end_points.pop('_after_level', None)
        features = self.fp1(end_points['sa3_xyz'], end_points['sa4_xyz'], end_points['sa3_features'],
                            end_points['sa4_features'])
        features = self.fp2(end_points['sa2_xyz'], end_points['sa3_xyz'], end_points['sa2_features'], features)
        end_points['fp2_features'] = features
        end_points['fp2_xyz'] = end_points['sa2_xyz']
        num_seed = end_points['fp2_xyz'].shape[1]
        end_points['fp2_inds'] = end_points['sa1_inds'][:, 0:num_seed]
        return features, end_points['fp2_xyz'], end_points
