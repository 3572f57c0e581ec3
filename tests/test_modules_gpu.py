"""GPU: the module-level fixtures g6 (SharedMLP) and g11 (PointnetSAModuleVotes, QueryAndGroup, PointnetFPModule,
CylinderQueryAndGroup) - produced by running the REFERENCE's python (tests/golden/make_golden.py) - through the HIP
path as shipped (C-ABI kernels, fused channel-last stacks).  The CPU twin is tests/test_modules_cpu.py."""
import pytest
import torch

from tests import test_modules_cpu as cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_g6_sharedmlp_matches_reference_gpu(golden):
    """a9: pytorch_utils.SharedMLP([3,64,128]) train / eval outputs and running statistics, 1e-6 absolute on O(1) values
    (the reference's run is torch CPU; here the convolution runs on the GPU)."""
    cases.run_g6_case(golden, DEV, atol=2e-6)


def test_g11_modules_match_reference_gpu(golden):
    """a8 / a10 / a11 / a4: FPS indices and new_xyz bit-exact, SA and FP outputs 1e-5, grouped tensors of QueryAndGroup
    (centre subtraction + division by the radius) the same BYTES as the reference's CPU run, CylinderQueryAndGroup's
    rotated crops 1e-6 (and its indices, through the values: a wrong neighbour is off by centimetres)."""
    from graspbalance_amd import pointnet2_utils
    cyl = cases.run_g11_case(pointnet2_utils, golden, DEV, exact_groups=False)
    assert cyl.is_cuda and bool(torch.isfinite(cyl).all())
