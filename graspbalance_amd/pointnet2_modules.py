"""Set-abstraction / feature-propagation modules with the constructor signatures, return values
and ``state_dict`` keys of the reference's PointNet/pointnet2_modules.py (_PointnetSAModuleBase :15,
PointnetSAModuleMSG :51, PointnetSAModule :84, PointnetSAModuleVotes :105, PointnetSAModuleVotesShift
:190, PointnetSAModuleVotes_WOMLP :267, PointnetSAModuleMSGVotes :342, PointnetFPModule :402,
PointnetLFPModuleMSG :437).  Geometry runs on the HIP kernels through ``pointnet2_utils``.

Own structure: the sampling, pooling and grouper-construction steps the reference repeats in every
class are factored into helpers.
"""
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from . import fused_mlp
from . import pointnet2_utils
from . import pytorch_utils as pt_utils


_FP_FUSED = os.environ.get("GB_FP_FUSED", "1") != "0"  # A/B switch: feature-propagation MLPs on the fused stack
_FP_ROWS = os.environ.get("GB_FP_ROWS", "1") != "0"    # A/B switch: interpolation + concatenation written as channel-last rows (one launch)


def _sample_centres(xyz, npoint, inds=None):
    """FPS (unless ``inds`` is given) + gather of the centres: returns (new_xyz (B,npoint,3), inds)."""
    if inds is None:
        inds = pointnet2_utils.furthest_point_sample(xyz, npoint)
    else:
        assert inds.shape[1] == npoint
    xyz_flipped = xyz.transpose(1, 2).contiguous()
    new_xyz = pointnet2_utils.gather_operation(xyz_flipped, inds).transpose(1, 2).contiguous()
    return new_xyz, inds


def _max_over_samples(x):
    """(B,C,npoint,nsample) -> (B,C,npoint,1): F.max_pool2d(kernel=[1,nsample]) of the reference."""
    return pt_utils.max_over_samples(x, keepdim=True)


def _pool(features, grouped_xyz, pooling, sigma, nsample):
    """'max' | 'avg' | 'rbf' pooling over the nsample axis -> (B,C,npoint)."""
    if pooling == 'max':
        out = _max_over_samples(features)
    elif pooling == 'avg':
        out = F.avg_pool2d(features, kernel_size=[1, features.size(3)])
    elif pooling == 'rbf':
        rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1, keepdim=False) / (sigma ** 2) / 2)
        out = torch.sum(features * rbf.unsqueeze(1), -1, keepdim=True) / float(nsample)
    else:  # the reference silently skips pooling for unknown names
        out = features
    return out.squeeze(-1)


def _make_groupers_and_mlps(npoint, radii, nsamples, mlps, bn, use_xyz, sample_uniformly):
    groupers, nets = nn.ModuleList(), nn.ModuleList()
    for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
        groupers.append(
            pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz, sample_uniformly=sample_uniformly)
            if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
        if use_xyz:
            mlp_spec[0] += 3  # in place on the caller's list, like the reference (:79-80)
        nets.append(pt_utils.SharedMLP(mlp_spec, bn=bn))
    return groupers, nets


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def _multi_scale(self, xyz, new_xyz, features):
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            outs.append(_max_over_samples(mlp(grouper(xyz, new_xyz, features))).squeeze(-1))
        return torch.cat(outs, dim=1)

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None):
        new_xyz = _sample_centres(xyz, self.npoint)[0] if self.npoint is not None else None
        return new_xyz, self._multi_scale(xyz, new_xyz, features)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 bn: bool = True, use_xyz: bool = True, sample_uniformly: bool = False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers, self.mlps = _make_groupers_and_mlps(npoint, radii, nsamples, mlps, bn, use_xyz,
                                                          sample_uniformly)


class PointnetSAModule(PointnetSAModuleMSG):
    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn,
                         use_xyz=use_xyz)


class _VotesBase(nn.Module):
    """Shared constructor of the three *Votes* variants (one grouper returning grouped_xyz)."""

    def _setup(self, npoint, radius, nsample, use_xyz, pooling, sigma, normalize_xyz, sample_uniformly,
               ret_unique_cnt):
        self.npoint = npoint
        self.radius = radius
        self.nsample = nsample
        self.pooling = pooling
        self.mlp_module = None
        self.use_xyz = use_xyz
        self.sigma = sigma if sigma is not None else self.radius / 2
        self.normalize_xyz = normalize_xyz
        self.ret_unique_cnt = ret_unique_cnt
        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(
                radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True, normalize_xyz=normalize_xyz,
                sample_uniformly=sample_uniformly, ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)

    def _group(self, xyz, new_xyz, features):
        out = self.grouper(xyz, new_xyz, features)
        if self.ret_unique_cnt:
            return out  # (grouped_features, grouped_xyz, unique_cnt)
        return out[0], out[1], None


class PointnetSAModuleVotes(_VotesBase):
    """FPS -> ball query -> group -> SharedMLP -> pool.  Returns (new_xyz, new_features, inds)."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pooling: str = 'max', sigma: float = None,
                 normalize_xyz: bool = False, sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        self._setup(npoint, radius, nsample, use_xyz, pooling, sigma, normalize_xyz, sample_uniformly,
                    ret_unique_cnt)
        mlp_spec = mlp
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3
        self.mlp_module = pt_utils.SharedMLP(mlp_spec, bn=bn)

    def _fused_ok(self, xyz):
        g = self.grouper
        return (fused_mlp.enabled(xyz) and self.npoint is not None and self.pooling == 'max'
                and self.use_xyz and not self.ret_unique_cnt and not getattr(g, "sample_uniformly", True)
                and fused_mlp.supports(self.mlp_module))

    def _forward_fused(self, xyz, new_xyz, features):
        """Same values as grouper -> SharedMLP -> max_pool2d, executed channel-last: the grouped
        (B,3+C,m,ns) tensor, its BN / ReLU copies and the pooling pass are never materialised."""
        B, m = new_xyz.shape[0], new_xyz.shape[1]
        idx = pointnet2_utils.ball_query(self.radius, self.nsample, xyz, new_xyz)
        feat_cl = features.transpose(1, 2).contiguous() if features is not None else None
        # torch evaluates `grouped_xyz /= radius` on the GPU as a multiply by the fp32 reciprocal
        scale = float(np.float32(1.0) / np.float32(self.radius)) if self.normalize_xyz else 1.0
        x0 = fused_mlp.group_concat_cl(xyz, new_xyz, idx, feat_cl, mode=1 if self.normalize_xyz else 0,
                                       scale=scale)
        out = fused_mlp.shared_mlp_cl(x0, self.mlp_module, pool_ns=self.nsample)  # (B*m, C_out)
        # (B, C_out, m) as a VIEW of the channel-last result: the next fused consumer transposes it back for free; code that
        # needs the reference's memory layout calls .contiguous() (values, shape and dtype are the reference's)
        return out.view(B, m, -1).transpose(1, 2)

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, inds: torch.Tensor = None):
        if self.npoint is not None:
            new_xyz, inds = _sample_centres(xyz, self.npoint, inds)
        else:
            new_xyz = None
            if inds is None:
                inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        if self._fused_ok(xyz):
            return new_xyz, self._forward_fused(xyz, new_xyz, features), inds
        grouped_features, grouped_xyz, unique_cnt = self._group(xyz, new_xyz, features)
        new_features = _pool(self.mlp_module(grouped_features), grouped_xyz, self.pooling, self.sigma,
                             self.nsample)
        if self.ret_unique_cnt:
            return new_xyz, new_features, inds, unique_cnt
        return new_xyz, new_features, inds


class PointnetSAModuleVotesShift(_VotesBase):
    """Votes variant whose centres are supplied by the caller: forward(new_xyz, xyz, features)."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pooling: str = 'max', sigma: float = None,
                 normalize_xyz: bool = False, sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        self._setup(npoint, radius, nsample, use_xyz, pooling, sigma, normalize_xyz, sample_uniformly,
                    ret_unique_cnt)
        mlp_spec = mlp
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3
        self.mlp_module = pt_utils.SharedMLP(mlp_spec, bn=bn)

    def forward(self, new_xyz: torch.Tensor, xyz: torch.Tensor, features: torch.Tensor = None):
        grouped_features, grouped_xyz, unique_cnt = self._group(xyz, new_xyz, features)
        new_features = _pool(self.mlp_module(grouped_features), grouped_xyz, self.pooling, self.sigma,
                             self.nsample)
        return (new_features, unique_cnt) if self.ret_unique_cnt else new_features


class PointnetSAModuleVotes_WOMLP(_VotesBase):
    """Votes variant without an MLP: pools the grouped features directly."""

    def __init__(self, *, npoint: int = None, radius: float = None, nsample: int = None,
                 use_xyz: bool = True, pooling: str = 'max', sigma: float = None,
                 normalize_xyz: bool = False, sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        self._setup(npoint, radius, nsample, use_xyz, pooling, sigma, normalize_xyz, sample_uniformly,
                    ret_unique_cnt)

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, inds: torch.Tensor = None):
        if self.npoint is not None:
            new_xyz, inds = _sample_centres(xyz, self.npoint, inds)
        else:
            new_xyz = None
            if inds is None:
                inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        grouped_features, grouped_xyz, unique_cnt = self._group(xyz, new_xyz, features)
        new_features = _pool(grouped_features, grouped_xyz, self.pooling, self.sigma, self.nsample)
        if self.ret_unique_cnt:
            return new_xyz, new_features, inds, unique_cnt
        return new_xyz, new_features, inds


class PointnetSAModuleMSGVotes(_PointnetSAModuleBase):
    """Multi-scale grouping that also returns the FPS indices."""

    def __init__(self, *, mlps: List[List[int]], npoint: int, radii: List[float], nsamples: List[int],
                 bn: bool = True, use_xyz: bool = True, sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.npoint = npoint
        self.groupers, self.mlps = _make_groupers_and_mlps(npoint, radii, nsamples, mlps, bn, use_xyz,
                                                          sample_uniformly)

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, inds: torch.Tensor = None):
        if inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        if self.npoint is not None:
            xyz_flipped = xyz.transpose(1, 2).contiguous()
            new_xyz = pointnet2_utils.gather_operation(xyz_flipped, inds).transpose(1, 2).contiguous()
        else:
            new_xyz = None
        return new_xyz, self._multi_scale(xyz, new_xyz, features), inds


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance 3-NN interpolation of `known_feats` onto `unknown`,
    concatenation with the skip features, SharedMLP."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor) -> torch.Tensor:
        if (_FP_FUSED and _FP_ROWS and known is not None and unknown.is_cuda and unknown.dtype == torch.float32
                and known.dtype == torch.float32 and not (unknown.requires_grad or known.requires_grad)
                and fused_mlp.enabled(known_feats) and fused_mlp.supports(self.mlp)):
            # interpolation + concatenation written once as the channel-last rows the fused stack reads (the (B,C,n) inputs
            # are transposed views of channel-last buffers on this path: .transpose(1, 2) costs nothing); same values as
            # three_interpolate -> cat -> transpose
            weight, idx = pointnet2_utils.three_nn_weights(unknown, known)
            rows = fused_mlp.interp_concat_cl(known_feats.transpose(1, 2), idx, weight,
                                              None if unknow_feats is None else unknow_feats.transpose(1, 2))
            out = fused_mlp.shared_mlp_cl(rows, self.mlp)
            return out.view(unknown.shape[0], unknown.shape[1], -1).transpose(1, 2)
        if known is not None:
            if (unknown.is_cuda and unknown.dtype == torch.float32 and known.dtype == torch.float32
                    and not (unknown.requires_grad or known.requires_grad)):
                # three_nn + the five element-wise launches of the weights as two kernels (same operations, same order)
                weight, idx = pointnet2_utils.three_nn_weights(unknown, known)
            else:
                dist, idx = pointnet2_utils.three_nn(unknown, known)
                dist_recip = 1.0 / (dist + 1e-8)
                norm = torch.sum(dist_recip, dim=2, keepdim=True)
                weight = dist_recip / norm
            interpolated_feats = pointnet2_utils.three_interpolate(known_feats.contiguous(), idx, weight)
        else:
            interpolated_feats = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        if unknow_feats is not None:
            new_features = torch.cat([interpolated_feats, unknow_feats], dim=1)  # (B, C2 + C1, n)
        else:
            new_features = interpolated_feats
        if _FP_FUSED and fused_mlp.enabled(new_features) and fused_mlp.supports(self.mlp):
            # the SharedMLP as one fused stack on channel-last rows (own MFMA GEMMs, BatchNorm statistics out of
            # their epilogues) instead of Conv2d / BatchNorm2d / ReLU launches on (B,C,n,1)
            B, C, n = new_features.shape
            rows = new_features.transpose(1, 2).reshape(B * n, C)
            out = fused_mlp.shared_mlp_cl(rows, self.mlp)
            return out.view(B, n, -1).transpose(1, 2)   # (a view of the channel-last rows: see PointnetSAModuleVotes)
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)


class PointnetLFPModuleMSG(nn.Module):
    """Learnable feature propagation with multi-scale grouping."""

    def __init__(self, *, mlps: List[List[int]], radii: List[float], nsamples: List[int],
                 post_mlp: List[int], bn: bool = True, use_xyz: bool = True, sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.post_mlp = pt_utils.SharedMLP(post_mlp, bn=bn)
        self.groupers, self.mlps = _make_groupers_and_mlps(0, radii, nsamples, mlps, bn, use_xyz,
                                                          sample_uniformly)

    def forward(self, xyz2: torch.Tensor, xyz1: torch.Tensor, features2: torch.Tensor,
                features1: torch.Tensor) -> torch.Tensor:
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            new_features = _max_over_samples(mlp(grouper(xyz1, xyz2, features1))).squeeze(-1)
            if features2 is not None:
                new_features = torch.cat([new_features, features2], dim=1)
            outs.append(self.post_mlp(new_features.unsqueeze(-1)))
        return torch.cat(outs, dim=1).squeeze(-1)
