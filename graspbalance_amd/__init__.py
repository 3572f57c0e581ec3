"""graspbalance_amd — MI355X-native point-cloud hot path of GraspBalance.

Hand-written HIP kernels (gfx950) behind a C-ABI (include/graspbal.h), exposed through the
reference's own python surfaces:

    graspbalance_amd.pointnet2._ext          <->  pointnet2._ext            (PointNet/_ext_src)
    graspbalance_amd.pointnet2_batch_cuda    <->  pointnet2_batch_cuda      (pointnet2_batch/src)
    graspbalance_amd.pointnet2_utils / pointnet2_modules / pytorch_utils  (PointNet/*.py)

``graspbalance_amd.compat.install()`` registers those names in ``sys.modules`` so code written
against the reference imports them unchanged.
"""
__version__ = "0.1.0"
