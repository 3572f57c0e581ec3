"""Normalisation factory (reference ModifiedNetTools/norm.py:10-84)."""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F


class LayerNorm2d(nn.LayerNorm):
    def __init__(self, num_channels, **kwargs):
        super().__init__(num_channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps)
        return y.permute(0, 3, 1, 2).contiguous()


class LayerNorm1d(nn.LayerNorm):
    def __init__(self, num_channels, **kwargs):
        super().__init__(num_channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = F.layer_norm(x.permute(0, 2, 1), self.normalized_shape, self.weight, self.bias, self.eps)
        return y.permute(0, 2, 1).contiguous()


class FastBatchNorm1d(nn.Module):
    def __init__(self, num_features, **kwargs):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, **kwargs)

    def forward(self, x):
        if x.dim() == 2:
            return self.bn(x)
        if x.dim() == 3:
            return self.bn(x.transpose(1, 2)).transpose(2, 1)
        raise ValueError("Non supported number of dimensions {}".format(x.dim()))


_NORM_LAYER = dict(
    bn1d=nn.BatchNorm1d, bn2d=nn.BatchNorm2d, bn=nn.BatchNorm2d, in2d=nn.InstanceNorm2d,
    in1d=nn.InstanceNorm1d, gn=nn.GroupNorm, syncbn=nn.SyncBatchNorm, ln=nn.LayerNorm,
    ln1d=LayerNorm1d, ln2d=LayerNorm2d, fastbn1d=FastBatchNorm1d, fastbn2d=FastBatchNorm1d,
    fastbn=FastBatchNorm1d,
)


def create_norm(norm_args, channels, dimension=None):
    """``{'norm': 'bn', **kwargs}`` or a name; ``dimension`` ('1d'/'2d') is appended to the name when
    it is not already part of it, so 'bn' becomes BatchNorm1d / BatchNorm2d."""
    if norm_args is None:
        return None
    if isinstance(norm_args, dict):
        norm_args = copy.deepcopy(norm_args)
        norm = norm_args.pop('norm', None)
    else:
        norm, norm_args = norm_args, {}
    if norm is None:
        return None
    if isinstance(norm, str):
        norm = norm.lower()
        if dimension is not None:
            dimension = str(dimension).lower()
            if dimension not in norm:
                norm += dimension
        assert norm in _NORM_LAYER.keys(), f"input {norm} is not supported"
        norm = _NORM_LAYER[norm]
    return norm(channels, **norm_args)
