"""Closed-form first-layer backward (gb_gemm_dgrad_first) vs layer-by-layer vs fp64 truth on the SA1 module of the g16 case."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import fused_mlp, pointnet2_modules as pm
from tests import f64_truth
from tests.seeded import fill_by_key
from tests.golden import make_golden_r2 as mk
DEV = "cuda:0"
def run(sa, xyz, fuse, enabled=True, dtype=torch.float32):
    sa = copy.deepcopy(sa).to(dtype); fused_mlp._FIRST_FUSE = fuse; fused_mlp.set_enabled(enabled)
    try:
        if dtype == torch.float64:
            with f64_truth.torch_geometry():
                _, f, _ = sa(xyz.double())
        else:
            _, f, _ = sa(xyz)
        torch.manual_seed(1); (f * torch.randn(f.shape, device=DEV).to(dtype)).sum().backward()
    finally:
        fused_mlp._FIRST_FUSE = True; fused_mlp.set_enabled(True)
    out = {k: p.grad.double() for k, p in sa.named_parameters()}
    out["FORWARD"] = f.detach().double()
    return out
for mlp, nrm in (([0, 64, 64, 128], True), ([0, 64, 128], False), ([0, 64, 64, 128], False)):
    sa = fill_by_key(pm.PointnetSAModuleVotes(npoint=2048, radius=0.04, nsample=64, mlp=list(mlp), use_xyz=True, normalize_xyz=nrm), seed=16).to(DEV).train()
    xyz = mk.g16_cloud(DEV)
    a, b, p, t = run(sa, xyz, True), run(sa, xyz, False), run(sa, xyz, False, False), run(sa, xyz, False, False, torch.float64)
    print(mlp, nrm)
    for k in t:
        r = lambda x: float((x - t[k]).norm() / t[k].norm())
        print("  %-28s closed %.2e layerwise %.2e plain %.2e   |g|=%.3e" % (k, r(a[k]), r(b[k]), r(p[k]), float(t[k].norm())))
