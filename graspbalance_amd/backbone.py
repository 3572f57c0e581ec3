"""Plain PointNet++ backbone with the layer names / hyper-parameters of the reference's
TrainModel/backbone.py:14-98 (SA1-4: 2048/1024/512/256 centres, FP1-2 -> (B,256,1024) seed features)."""
import torch
import torch.nn as nn

from .pointnet2_modules import PointnetSAModuleVotes, PointnetFPModule

# (npoint, radius, nsample, mlp) of the four set-abstraction levels (backbone.py:18-52, drp.py:161-236)
SA_SPECS = (
    (2048, 0.04, 64, (None, 64, 64, 128)),
    (1024, 0.1, 32, (128, 128, 128, 256)),
    (512, 0.2, 16, (256, 128, 128, 256)),
    (256, 0.3, 16, (256, 128, 128, 256)),
)


def make_sa(level, input_feature_dim=0):
    npoint, radius, nsample, mlp = SA_SPECS[level]
    mlp = [input_feature_dim if c is None else c for c in mlp]
    return PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=nsample, mlp=mlp, use_xyz=True,
                                 normalize_xyz=True)


def break_up_pc(pc):
    """(B,N,3+C) -> xyz (B,N,3) contiguous, features (B,C,N) or None."""
    xyz = pc[..., 0:3].contiguous()
    features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
    return xyz, features


class Pointnet2Backbone(nn.Module):
    def __init__(self, input_feature_dim=0):
        super().__init__()
        self.sa1 = make_sa(0, input_feature_dim)
        self.sa2 = make_sa(1)
        self.sa3 = make_sa(2)
        self.sa4 = make_sa(3)
        self.fp1 = PointnetFPModule(mlp=[256 + 256, 256, 256])
        self.fp2 = PointnetFPModule(mlp=[256 + 256, 256, 256])

    def _break_up_pc(self, pc):
        return break_up_pc(pc)

    def forward(self, pointcloud: torch.Tensor, end_points=None):
        if not end_points:
            end_points = {}
        xyz, features = break_up_pc(pointcloud)
        end_points['input_xyz'] = xyz
        sa1_xyz, sa1_features, sa1_inds = self.sa1(xyz, features)
        sa2_xyz, sa2_features, _ = self.sa2(sa1_xyz, sa1_features)  # FPS of an FPS prefix: 0..1023
        sa3_xyz, sa3_features, _ = self.sa3(sa2_xyz, sa2_features)
        sa4_xyz, sa4_features, _ = self.sa4(sa3_xyz, sa3_features)
        features = self.fp1(sa3_xyz, sa4_xyz, sa3_features, sa4_features)
        features = self.fp2(sa2_xyz, sa3_xyz, sa2_features, features)
        end_points['fp2_features'] = features
        end_points['fp2_xyz'] = sa2_xyz
        num_seed = sa2_xyz.shape[1]
        end_points['fp2_inds'] = sa1_inds[:, 0:num_seed]  # indices into the input cloud
        return features, end_points['fp2_xyz'], end_points
