"""Grouping ops over the PB-ext surface (reference ModifiedNetTools/group.py:15-252): outputs are
allocated by the caller-side wrappers here and written by ``pointnet2_batch_cuda``."""
import copy
import logging
from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import pointnet2_batch_cuda as pointnet2_cuda


class KNN(nn.Module):
    def __init__(self, neighbors, transpose_mode=True):
        super().__init__()
        self.neighbors = neighbors

    @torch.no_grad()
    def forward(self, support, query):
        """support (B,N,3), query (B,M,3) -> (dist (B,K,M), idx (B,M,K) int32)."""
        dist = torch.cdist(support, query)
        k_dist = dist.topk(k=self.neighbors, dim=1, largest=False)
        return k_dist.values, k_dist.indices.transpose(1, 2).contiguous().int()


class DenseDilated(nn.Module):
    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.dilation, self.stochastic, self.epsilon, self.k = dilation, stochastic, epsilon, k

    def forward(self, edge_index):
        if self.stochastic and torch.rand(1) < self.epsilon and self.training:
            pick = torch.randperm(self.k * self.dilation)[:self.k]
            return edge_index[:, :, pick].contiguous()
        return edge_index[:, :, ::self.dilation].contiguous()


class DilatedKNN(nn.Module):
    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.dilation, self.stochastic, self.epsilon, self.k = dilation, stochastic, epsilon, k
        self._dilated = DenseDilated(k, dilation, stochastic, epsilon)
        self.knn = KNN(k * self.dilation, transpose_mode=True)

    def forward(self, query):
        _, idx = self.knn(query, query)
        return self._dilated(idx)


class GroupingOperation(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint,nsample) int32 -> (B,C,npoint,nsample)."""
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = torch.empty((B, C, nfeatures, nsample), dtype=torch.float32, device=features.device)
        pointnet2_cuda.group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.zeros([B, C, N], dtype=torch.float, device=grad_out.device)
        pointnet2_cuda.group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


def torch_grouping_operation(features, idx):
    """Pure-torch equivalent of ``grouping_operation`` (B,C,N),(B,m,ns) -> (B,C,m,ns)."""
    flat = idx.reshape(idx.shape[0], 1, -1).expand(-1, features.shape[1], -1).long()
    return features.gather(2, flat).reshape(idx.shape[0], features.shape[1], idx.shape[1], idx.shape[2])


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint) int32 -> (B,C,npoint)."""
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, npoint = idx.size()
        _, C, N = features.size()
        output = torch.empty((B, C, npoint), dtype=torch.float32, device=features.device)
        pointnet2_cuda.gather_points_wrapper(B, C, N, npoint, features, idx, output)
        ctx.for_backwards = (idx, C, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = torch.zeros([B, C, N], dtype=torch.float, device=grad_out.device)
        pointnet2_cuda.gather_points_grad_wrapper(B, C, N, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        """xyz (B,N,3) support, new_xyz (B,npoint,3) centres -> (B,npoint,nsample) int32."""
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        # (group.py:116 zero-fills; gb_ball_query writes every element, so the fill is not needed)
        idx = torch.empty((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2_cuda.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """forward(query_xyz, support_xyz, features) -> (grouped_xyz (B,3,M,ns) relative to the query,
    grouped_features (B,C,M,ns) or None)."""

    def __init__(self, radius: float, nsample: int, relative_xyz=True, normalize_dp=False,
                 normalize_by_std=False, normalize_by_allstd=False, normalize_by_allstd2=False,
                 return_only_idx=False, **kwargs):
        super().__init__()
        self.radius, self.nsample = radius, nsample
        self.normalize_dp = normalize_dp
        self.normalize_by_std = normalize_by_std
        self.normalize_by_allstd = normalize_by_allstd
        self.normalize_by_allstd2 = normalize_by_allstd2
        assert self.normalize_dp + self.normalize_by_std + self.normalize_by_allstd < 2
        self.relative_xyz = relative_xyz
        self.return_only_idx = return_only_idx

    def forward(self, query_xyz, support_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, support_xyz, query_xyz)
        if self.return_only_idx:
            return idx
        grouped_xyz = grouping_operation(support_xyz.transpose(1, 2).contiguous(), idx)
        if self.relative_xyz:
            grouped_xyz = grouped_xyz - query_xyz.transpose(1, 2).unsqueeze(-1)
            if self.normalize_dp:
                grouped_xyz /= self.radius
        grouped_features = grouping_operation(features, idx) if features is not None else None
        return grouped_xyz, grouped_features


class GroupAll(nn.Module):
    def forward(self, new_xyz, xyz, features=None):
        grouped_features = features.unsqueeze(2) if features is not None else None
        return xyz.transpose(1, 2).unsqueeze(2), grouped_features


class KNNGroup(nn.Module):
    def __init__(self, nsample: int, relative_xyz=True, normalize_dp=False, return_only_idx=False, **kwargs):
        super().__init__()
        self.nsample = nsample
        self.knn = KNN(nsample, transpose_mode=True)
        self.relative_xyz = relative_xyz
        self.normalize_dp = normalize_dp
        self.return_only_idx = return_only_idx

    def forward(self, query_xyz, support_xyz, features=None):
        _, idx = self.knn(support_xyz, query_xyz)
        if self.return_only_idx:
            return idx
        idx = idx.int()
        grouped_xyz = grouping_operation(support_xyz.transpose(1, 2).contiguous(), idx)
        if self.relative_xyz:
            grouped_xyz -= query_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_dp:
            grouped_xyz /= torch.amax(torch.sqrt(torch.sum(grouped_xyz ** 2, dim=1)), dim=(1, 2)).view(-1, 1, 1, 1)
        grouped_features = grouping_operation(features, idx) if features is not None else None
        return grouped_xyz, grouped_features


def get_aggregation_feautres(p, dp, f, fj, feature_type='dp_fj'):
    """Assemble the aggregation input from relative positions dp, neighbour features fj, centre
    features f and centre positions p (spelling of the reference's name kept)."""
    if feature_type == 'dp_fj':
        return torch.cat([dp, fj], 1)
    if feature_type == 'dp_fj_df':
        return torch.cat([dp, fj, fj - f.unsqueeze(-1)], 1)
    if feature_type == 'pi_dp_fj_df':
        df = fj - f.unsqueeze(-1)
        pi = p.transpose(1, 2).unsqueeze(-1).expand(-1, -1, -1, df.shape[-1])
        return torch.cat([pi, dp, fj, df], 1)
    if feature_type == 'dp_df':
        return torch.cat([dp, fj - f.unsqueeze(-1)], 1)
    return fj


def create_grouper(group_args):
    args = copy.deepcopy(group_args)
    method = args.pop('NAME', 'ballquery')
    radius = args.pop('radius', 0.1)
    nsample = args.pop('nsample', 20)
    logging.info(group_args)
    if nsample is None:
        return GroupAll()
    if method == 'ballquery':
        return QueryAndGroup(radius, nsample, **args)
    if method == 'knn':
        return KNNGroup(nsample, **args)
    raise NotImplementedError(method)
