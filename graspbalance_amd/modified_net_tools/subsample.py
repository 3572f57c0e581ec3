"""Samplers over the PB-ext surface (reference ModifiedNetTools/subsample.py:15-125)."""
import math
from abc import ABC, abstractmethod

import torch
from torch.autograd import Function

from .. import pointnet2_batch_cuda as pointnet2_cuda
from .group import GatherOperation, gather_operation  # noqa: F401  (same op; the reference defines it twice)


class BaseSampler(ABC):
    """Holds exactly one of ratio / num_to_sample / subsampling_param."""

    def __init__(self, ratio=None, num_to_sample=None, subsampling_param=None):
        if num_to_sample is not None:
            if (ratio is not None) or (subsampling_param is not None):
                raise ValueError("Can only specify ratio or num_to_sample or subsampling_param, not several !")
            self._num_to_sample = num_to_sample
        elif ratio is not None:
            self._ratio = ratio
        elif subsampling_param is not None:
            self._subsampling_param = subsampling_param
        else:
            raise Exception('At least ["ratio, num_to_sample, subsampling_param"] should be defined')

    def __call__(self, xyz):
        return self.sample(xyz)

    def _get_num_to_sample(self, npoints) -> int:
        return self._num_to_sample if hasattr(self, "_num_to_sample") else math.floor(npoints * self._ratio)

    def _get_ratio_to_sample(self, batch_size) -> float:
        return self._ratio if hasattr(self, "_ratio") else self._num_to_sample / float(batch_size)

    @abstractmethod
    def sample(self, xyz, feature=None, batch=None):
        pass


class RandomSample(BaseSampler):
    def sample(self, xyz, **kwargs):
        if len(xyz.shape) != 3:
            raise ValueError(" Expects the xyz tensor to be of dimension 3")
        B, N, _ = xyz.shape
        idx = torch.randint(0, N, (B, self._get_num_to_sample(N)), device=xyz.device)
        return torch.gather(xyz, 1, idx.unsqueeze(-1).expand(-1, -1, 3)), idx


def random_sample(xyz, npoint):
    B, N, _ = xyz.shape
    return torch.randint(0, N, (B, npoint), device=xyz.device)


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        """xyz (B,N,3) -> (B,npoint) int32 (PB-ext rules: no near-origin skip, 1024-thread tree ties)."""
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2_cuda.furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


def fps(data, number):
    """data (B,N,3+C) -> the `number` FPS-selected rows (B,number,3+C)."""
    fps_idx = furthest_point_sample(data[:, :, :3].contiguous(), number)
    return torch.gather(data, 1, fps_idx.unsqueeze(-1).long().expand(-1, -1, data.shape[-1]))
