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


@pytest.mark.parametrize("C2,C1,n,m", [(256, 128, 1024, 512), (256, 256, 512, 256), (64, 0, 300, 77)])
def test_feature_propagation_rows_written_once_equal_interpolate_cat_transpose(C2, C1, n, m):
    """PointnetFPModule on the fused path: gb_interp_concat_cl writes the MLP's input rows in one launch - the SAME values
    as three_interpolate -> cat -> transpose (bit for bit: the kernel uses three_interpolate's expression) - and its
    backward adds dense rows into the coarse level's gradient; module output and every gradient against the three-step
    path (GB_FP_ROWS off)."""
    import copy
    from graspbalance_amd import pointnet2_modules as pm
    torch.manual_seed(C2 + n)
    B = 3
    fp = pm.PointnetFPModule(mlp=[C2 + C1, 256, 128]).to("cuda:0").train()
    unknown, known = torch.rand(B, n, 3, device="cuda:0"), torch.rand(B, m, 3, device="cuda:0")
    kf0 = torch.randn(B, m, C2, device="cuda:0").transpose(1, 2)                      # (B,C2,m) view of channel-last, as on the path
    uf0 = torch.randn(B, n, C1, device="cuda:0").transpose(1, 2) if C1 else None
    res = {}
    for flag in (True, False):
        pm._FP_ROWS = flag
        try:
            mod = copy.deepcopy(fp)
            kf = kf0.clone().requires_grad_(True)
            uf = uf0.clone().requires_grad_(True) if C1 else None
            out = mod(unknown, known, uf, kf)
            torch.manual_seed(3)
            (out * torch.randn(out.shape, device=out.device)).sum().backward()
            res[flag] = (out.detach().clone(), kf.grad.clone(), uf.grad.clone() if C1 else None,
                         [p.grad.clone() for p in mod.parameters()])
        finally:
            pm._FP_ROWS = True
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0])                                   # same rows into the same stack: the same output bits
    rel = lambda x, y: float((x - y).abs().max()) / (float(y.abs().max()) + 1e-30)
    assert rel(a[1], b[1]) < 1e-5 and (not C1 or torch.equal(a[2], b[2]))
    for ga, gb_ in zip(a[3], b[3]):
        assert rel(ga, gb_) < 1e-4
