"""Host-side python layer (pytorch_utils / pointnet2_utils / pointnet2_modules) against goldens made
by running the REFERENCE's python files (tests/golden/make_golden.py).  CPU-only: the extension
object is swapped for the oracle-backed one inside the tests (the product itself has no CPU path).
"""
import hashlib

import numpy as np
import pytest
import torch


def sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


@pytest.fixture()
def cpu_ext(monkeypatch, orc):
    from graspbalance_amd import pointnet2_utils
    monkeypatch.setattr(pointnet2_utils, "_ext", orc.ExtBackend())
    return pointnet2_utils


def run_g6_case(golden, device="cpu", atol=1e-6):
    """SharedMLP([3,64,128]) (pytorch_utils.py:61-113) against the reference's own run: train and eval outputs, running
    statistics.  Shared with tests/test_modules_gpu.py (the same module on the HIP path)."""
    from graspbalance_amd import pytorch_utils as pt
    g = golden.load("g6_sharedmlp")
    torch.manual_seed(6)
    mlp = pt.SharedMLP([3, 64, 128], bn=True)  # same construction order -> same seeded init
    assert list(mlp.state_dict().keys()) == golden.manifest["G6_state_dict_keys"]
    x = torch.randn(2, 3, 16, 8)
    mlp = mlp.to(device)
    x = x.to(device)
    mlp.train()
    y = mlp(x.clone())
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["y_train"], rtol=0, atol=atol)
    for k, v in mlp.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), g[k.replace(".", "__")], rtol=0, atol=atol)
    mlp.eval()
    np.testing.assert_allclose(mlp(x.clone()).detach().cpu().numpy(), g["y_eval"], rtol=0, atol=atol)


def test_g6_sharedmlp_matches_reference(golden):
    run_g6_case(golden)


def test_sharedmlp_variants_and_scheduler():
    from graspbalance_amd import pytorch_utils as pt
    m = pt.SharedMLP([4, 8, 8], bn=True, preact=True, first=True)
    keys = list(m.state_dict().keys())
    assert keys[0] == "layer0.conv.weight" and "layer0.bn.bn.weight" not in keys and "layer1.bn.bn.weight" in keys
    assert list(dict(m.layer1.named_children())) == ["bn", "activation", "conv"]
    c = pt.Conv1d(4, 6, bn=False)
    assert c.conv.bias is not None and float(c.conv.bias.detach().abs().sum()) == 0.0
    assert pt.Conv2d(4, 6, bn=True).conv.bias is None
    fc = pt.FC(4, 5, bn=True)
    assert list(dict(fc.named_children())) == ["fc", "bn", "activation"] and fc.fc.bias is None
    net = torch.nn.Sequential(pt.Conv1d(3, 4, bn=True), pt.Conv3d(4, 4, bn=True))
    sched = pt.BNMomentumScheduler(net, bn_lambda=lambda it: max(0.5 * 0.5 ** (it // 2), 0.001), last_epoch=-1)
    assert net[0].bn.bn.momentum == 0.5
    sched.step(); sched.step(); sched.step()  # epochs 0, 1, 2 (the constructor pre-steps to 0, keeps last_epoch -1)
    assert net[1].bn.bn.momentum == 0.25 and sched.last_epoch == 2
    with pytest.raises(RuntimeError):
        pt.BNMomentumScheduler(object(), bn_lambda=lambda it: 0.1)


def run_g11_case(pu, golden, device="cpu", exact_groups=True):
    """PointnetSAModuleVotes / QueryAndGroup / PointnetFPModule / CylinderQueryAndGroup (pointnet2_modules.py:105-188,
    402-435; pointnet2_utils.py:152-308) against the reference's own composition.  `pu` = the pointnet2_utils module
    (CPU: oracle-backed; GPU: the product as shipped).  exact_groups: grouped tensors must have the fixture's hash
    (pure copies / subtractions / one division or one 3x3 rotation per element)."""
    from graspbalance_amd import pointnet2_modules as pm
    g = golden.load("g11_modules")
    torch.manual_seed(11)
    xyz = (torch.rand(2, 2048, 3) * 0.5).to(device)
    feat = torch.randn(2, 8, 2048).to(device)
    mlp = [8, 16, 32]
    sa = pm.PointnetSAModuleVotes(npoint=256, radius=0.1, nsample=16, mlp=mlp, use_xyz=True, normalize_xyz=True)
    assert mlp[0] == 11  # the caller's list is mutated like in the reference
    assert list(sa.state_dict().keys()) == golden.manifest["G11_sa_state_dict_keys"]
    sa.load_state_dict({k[4:].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("sa__")})
    sa = sa.to(device).train()
    new_xyz, new_feat, inds = sa(xyz, feat)
    assert torch.equal(inds.cpu(), torch.from_numpy(g["inds"]))
    assert torch.equal(new_xyz.cpu(), torch.from_numpy(g["new_xyz"]))
    np.testing.assert_allclose(new_feat.detach().cpu().numpy(), g["sa_out"], rtol=0, atol=1e-5)
    qg = pu.QueryAndGroup(0.1, 16, use_xyz=True, ret_grouped_xyz=True, normalize_xyz=True)
    grouped_feat, grouped_xyz = qg(xyz, new_xyz, feat)
    assert torch.equal(grouped_xyz[:, :, :8].cpu(), torch.from_numpy(g["grouped_xyz_head"]))
    assert sha(grouped_feat) == bytes(g["grouped_feat_sha256"]).hex()
    fp = pm.PointnetFPModule(mlp=[32 + 8, 16])
    assert list(fp.state_dict().keys()) == golden.manifest["G11_fp_state_dict_keys"]
    fp.load_state_dict({k[4:].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("fp__")})
    fp = fp.to(device).train()
    out = fp(xyz, new_xyz, feat, torch.from_numpy(g["sa_out"]).to(device))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["fp_out"], rtol=0, atol=1e-5)
    rot = torch.from_numpy(golden.load("g9_views")["rot"])[:256].view(1, 256, 3, 3).repeat(2, 1, 1, 1).contiguous()
    cq = pu.CylinderQueryAndGroup(0.05, -0.02, 0.04, 16, use_xyz=True)
    cyl = cq(xyz, new_xyz, rot.to(device))
    if exact_groups:
        assert torch.equal(cyl[:, :, :32].cpu(), torch.from_numpy(g["cyl_head"]))
        assert sha(cyl) == bytes(g["cyl_sha256"]).hex()
    else:  # the rotation is a torch.matmul: rocBLAS / CPU BLAS may order the three products differently
        np.testing.assert_allclose(cyl[:, :, :32].cpu().numpy(), g["cyl_head"], rtol=0, atol=1e-6)
    return cyl


def test_g11_sa_fp_cylinder_match_reference_composition(cpu_ext, golden):
    run_g11_case(cpu_ext, golden)


def test_sa_module_backward_and_variants(cpu_ext):
    from graspbalance_amd import pointnet2_modules as pm
    torch.manual_seed(1)
    xyz = torch.randn(2, 9, 3, requires_grad=True)
    feats = torch.randn(2, 9, 6).transpose(1, 2).contiguous().requires_grad_(True)
    # the reference's own __main__ smoke configuration (pointnet2_modules.py:498-517)
    msg = pm.PointnetSAModuleMSG(npoint=2, radii=[5.0, 10.0], nsamples=[6, 3], mlps=[[6, 3], [6, 6]])
    new_xyz, new_features = msg(xyz, feats)
    assert new_xyz.shape == (2, 2, 3) and new_features.shape == (2, 9, 2)
    new_features.backward(torch.ones_like(new_features))
    assert feats.grad is not None and bool(torch.isfinite(feats.grad).all())
    votes = pm.PointnetSAModuleMSGVotes(npoint=2, radii=[5.0], nsamples=[4], mlps=[[6, 5]])
    nx, nf, inds = votes(xyz.detach(), feats.detach())
    assert nf.shape == (2, 5, 2) and inds.dtype == torch.int32
    for pooling in ("max", "avg", "rbf"):
        sa = pm.PointnetSAModuleVotes(npoint=3, radius=4.0, nsample=5, mlp=[6, 7], pooling=pooling)
        a, b, c = sa(xyz.detach(), feats.detach())
        assert b.shape == (2, 7, 3)
        a2, b2, c2 = sa(xyz.detach(), feats.detach(), inds=c)
        assert torch.equal(a, a2)
    wo = pm.PointnetSAModuleVotes_WOMLP(npoint=3, radius=4.0, nsample=5)
    assert wo(xyz.detach(), feats.detach())[1].shape == (2, 9, 3)
    shift = pm.PointnetSAModuleVotesShift(npoint=3, radius=4.0, nsample=5, mlp=[6, 7])
    assert shift(xyz.detach()[:, :3].contiguous(), xyz.detach(), feats.detach()).shape == (2, 7, 3)
    single = pm.PointnetSAModule(mlp=[6, 4], npoint=2, radius=3.0, nsample=4)
    assert single(xyz.detach(), feats.detach())[1].shape == (2, 4, 2)
    group_all = pm.PointnetSAModule(mlp=[6, 4])
    assert group_all(xyz.detach(), feats.detach())[1].shape == (2, 4, 1)
    lfp = pm.PointnetLFPModuleMSG(mlps=[[6, 8]], radii=[5.0], nsamples=[4], post_mlp=[8 + 2, 5])
    out = lfp(xyz.detach()[:, :4].contiguous(), xyz.detach(), torch.randn(2, 2, 4), feats.detach())
    assert out.shape == (2, 5, 4)
    fp = pm.PointnetFPModule(mlp=[6, 4])
    assert fp(xyz.detach(), None, None, torch.randn(2, 6, 1)).shape == (2, 4, 9)


def test_random_dropout_mirror():
    """pointnet2_utils.RandomDropout (reference :35-43; dead code there: its helper does not exist): constructor parity,
    identity in eval mode, whole channels dropped without rescaling in training mode."""
    from graspbalance_amd import pointnet2_utils as pu
    m = pu.RandomDropout(p=0.9)
    assert m.p == 0.9 and m.inplace is False and list(m.state_dict()) == []
    x = torch.rand(4, 64, 10) + 0.5
    assert torch.equal(m.eval()(x), x)
    torch.manual_seed(0)
    y = m.train()(x)
    kept = (y != 0).float().mean(dim=2)
    assert bool(((kept == 0) | (kept == 1)).all()) and torch.equal(y[y != 0], x[y != 0])
