"""autograd wrappers and groupers with the API of the reference's PointNet/pointnet2_utils.py
(FurthestPointSampling :46, GatherOperation :59, ThreeNN :79, ThreeInterpolate :94,
GroupingOperation :119, BallQuery :140, QueryAndGroup :152, GroupAll :210, CylinderQuery :235,
CylinderQueryAndGroup :247) over the HIP-backed ``pointnet2._ext``.

``_ext`` is a module attribute exactly as in the reference, so the same functions run on whatever
extension object is bound there (the HIP library in the product).
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from .pointnet2 import _ext  # HIP-backed; raises on CPU tensors like the reference extension


class RandomDropout(nn.Module):
    """API parity with pointnet2_utils.py:35-43.  The reference's forward calls
    ``pt_utils.feature_dropout_no_scaling`` - a function its own pytorch_utils.py does not define (and hands it the
    bound method ``self.train``), so calling the reference's module raises AttributeError; nothing in the reference
    instantiates it.  Same constructor and state (none); forward draws theta ~ U(0, p) like the reference and then applies
    what the missing helper does upstream (Pointnet2_PyTorch): feature dropout with rate theta WITHOUT rescaling, while
    training."""

    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.p = p
        self.inplace = inplace

    def forward(self, X):
        theta = torch.Tensor(1).uniform_(0, self.p)[0]
        if not self.training:
            return X
        keep = (torch.rand(X.shape[:2] + (1,) * (X.dim() - 2), device=X.device) >= theta).to(X.dtype)
        return X.mul_(keep) if self.inplace else X * keep


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) -> (B,npoint) int32 indices; not differentiable."""
        out = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


# The autograd wrappers hand the extension CONTIGUOUS features (a no-op for the reference's own tensors): this package's
# fused modules return (B,C,N) feature tensors as transposed views of their channel-last results.  The extension shims
# themselves (pointnet2/_ext.py) keep the reference's strict contiguity check.
class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) int32 -> (B,C,npoint)."""
        ctx.for_backwards = (idx, features.size(1), features.size(2))
        return _ext.gather_points(features.contiguous(), idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, _, n = ctx.for_backwards
        return _ext.gather_points_grad(grad_out.contiguous(), idx, n), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) l2 distances, idx (B,n,3) int32)."""
        dist2, idx = _ext.three_nn(unknown, known)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


def three_nn_weights(unknown, known):
    """(weight (B,n,3), idx (B,n,3) int32): the three nearest known points of every unknown point and the normalised
    inverse-distance weights PointnetFPModule forms from them (pointnet2_modules.py:260-263), as two launches."""
    from . import _lib
    with torch.no_grad():
        dist2, idx = _ext.three_nn(unknown.contiguous(), known.contiguous())
        weight = torch.empty_like(dist2)
        with _lib.device_ctx(dist2.device):
            _lib.check(_lib.lib().gb_interp_weights(_lib.ptr(dist2), _lib.ptr(weight), dist2.numel() // 3,
                                                    _lib.current_stream(dist2.device)), "interp_weights")
    return weight, idx


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B,c,m), idx (B,n,3), weight (B,n,3) -> (B,c,n)."""
        ctx.three_interpolate_for_backward = (idx, weight, features.size(2))
        return _ext.three_interpolate(features.contiguous(), idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        return _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, m), None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) int32 -> (B,C,npoint,nsample)."""
        ctx.for_backwards = (idx, features.size(2))
        return _ext.group_points(features.contiguous(), idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, n = ctx.for_backwards
        return _ext.group_points_grad(grad_out.contiguous(), idx, n), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """xyz (B,N,3), new_xyz (B,npoint,3) -> (B,npoint,nsample) int32."""
        out = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class CylinderQuery(Function):
    @staticmethod
    def forward(ctx, radius, hmin, hmax, nsample, xyz, new_xyz, rot):
        """xyz (B,N,3), new_xyz (B,npoint,3), rot (B,npoint,9) -> (B,npoint,nsample) int32."""
        out = _ext.cylinder_query(new_xyz, xyz, rot, radius, hmin, hmax, nsample)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None, None, None, None


cylinder_query = CylinderQuery.apply


def _resample_uniformly(idx, nsample):
    """``sample_uniformly`` option (pointnet2_utils.py:167-176): every row keeps its unique
    neighbours and is topped up with random repeats of them; returns the unique counts."""
    unique_cnt = torch.zeros((idx.shape[0], idx.shape[1]))
    for b in range(idx.shape[0]):
        for j in range(idx.shape[1]):
            uniq = torch.unique(idx[b, j, :])
            k = uniq.shape[0]
            unique_cnt[b, j] = k
            fill = torch.randint(0, k, (nsample - k,), dtype=torch.long)
            idx[b, j, :] = torch.cat((uniq, uniq[fill]))
    return unique_cnt


def _assemble(grouped_xyz, idx, features, use_xyz):
    if features is None:
        assert use_xyz, "Cannot have not features and not use xyz as a feature!"
        return grouped_xyz
    grouped_features = grouping_operation(features, idx)
    if use_xyz:
        return torch.cat([grouped_xyz, grouped_features], dim=1)  # (B, 3 + C, npoint, nsample)
    return grouped_features


def _pack(new_features, grouped_xyz, unique_cnt, ret_grouped_xyz, ret_unique_cnt):
    ret = [new_features]
    if ret_grouped_xyz:
        ret.append(grouped_xyz)
    if ret_unique_cnt:
        ret.append(unique_cnt)
    return ret[0] if len(ret) == 1 else tuple(ret)


class QueryAndGroup(nn.Module):
    """Ball query + grouping: (B,3+C,npoint,nsample) with centred (optionally /radius) xyz first."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if self.ret_unique_cnt:
            assert self.sample_uniformly

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        unique_cnt = _resample_uniformly(idx, self.nsample) if self.sample_uniformly else None
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)  # (B,3,npoint,nsample)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz /= self.radius  # a true division, as in the reference (bit parity)
        new_features = _assemble(grouped_xyz, idx, features, self.use_xyz)
        return _pack(new_features, grouped_xyz, unique_cnt, self.ret_grouped_xyz, self.ret_unique_cnt)


class GroupAll(nn.Module):
    """Groups the whole cloud into one region: (B,3+C,1,N)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz  # the reference forgets to store this (:210-231)

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            new_features = grouped_xyz
        elif self.use_xyz:
            new_features = torch.cat([grouped_xyz, features.unsqueeze(2)], dim=1)
        else:
            new_features = features.unsqueeze(2)
        return (new_features, grouped_xyz) if self.ret_grouped_xyz else new_features


class CylinderQueryAndGroup(nn.Module):
    """Cylinder query + grouping; grouped xyz is centred and rotated into the gripper frame."""

    def __init__(self, radius, hmin, hmax, nsample, use_xyz=True, ret_grouped_xyz=False,
                 normalize_xyz=False, rotate_xyz=True, sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.hmin, self.hmax = radius, nsample, hmin, hmax
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.rotate_xyz = rotate_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if self.ret_unique_cnt:
            assert self.sample_uniformly

    def group(self, xyz, new_xyz, rot, idx, features=None):
        """Everything after the query (shared with the fused multi-query path of the grasp head)."""
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz /= self.radius
        if self.rotate_xyz:
            rows = grouped_xyz.permute(0, 2, 3, 1).contiguous()  # (B,npoint,nsample,3)
            grouped_xyz = torch.matmul(rows, rot).permute(0, 3, 1, 2).contiguous()
        return grouped_xyz, _assemble(grouped_xyz, idx, features, self.use_xyz)

    def forward(self, xyz, new_xyz, rot, features=None):
        B, npoint, _ = new_xyz.size()
        idx = cylinder_query(self.radius, self.hmin, self.hmax, self.nsample, xyz, new_xyz,
                             rot.view(B, npoint, 9))
        unique_cnt = _resample_uniformly(idx, self.nsample) if self.sample_uniformly else None
        grouped_xyz, new_features = self.group(xyz, new_xyz, rot, idx, features)
        return _pack(new_features, grouped_xyz, unique_cnt, self.ret_grouped_xyz, self.ret_unique_cnt)
