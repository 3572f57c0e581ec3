"""Is a fused-vs-plain gradient gap a bug or routing chaos?  Baseline: plain vs plain with weights
perturbed by 1e-7 relative (one fp32 ulp)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_model_cpu import _tiny_net
from graspbalance_amd import fused_mlp
from graspbalance_amd.scene import make_batch
DEV = "cuda:0"
clouds = torch.from_numpy(make_batch([0, 1], 3000)).to(DEV)

def run(flag, perturb=0.0, own=True):
    drp = _tiny_net().view_estimator.FeatureExtraction.to(DEV).train()
    if perturb:
        torch.manual_seed(123)
        with torch.no_grad():
            for p in drp.parameters():
                p.mul_(1.0 + perturb * torch.randn_like(p))
    fused_mlp.set_enabled(flag); fused_mlp.set_own_gemm(own)
    feats, _, _ = drp(clouds)
    torch.manual_seed(7)
    (feats * torch.randn_like(feats)).sum().backward()
    fused_mlp.set_enabled(True); fused_mlp.set_own_gemm(True)
    return feats.detach().clone(), {k: v.grad.clone() for k, v in drp.named_parameters()}

def gap(a, b):
    num = sum(float((a[1][k] - b[1][k]).norm()) ** 2 for k in b[1]) ** 0.5
    den = sum(float(b[1][k].norm()) ** 2 for k in b[1]) ** 0.5
    return float((a[0] - b[0]).norm() / b[0].norm()), num / den

plain = run(False)
print("plain vs plain(again):        feat %.2e grad %.2e" % gap(run(False), plain))
print("plain vs plain(perturb 1e-7): feat %.2e grad %.2e" % gap(run(False, 1e-7), plain))
print("plain vs plain(perturb 1e-6): feat %.2e grad %.2e" % gap(run(False, 1e-6), plain))
print("fused(own gemm) vs plain:     feat %.2e grad %.2e" % gap(run(True), plain))
print("fused(rocblas)  vs plain:     feat %.2e grad %.2e" % gap(run(True, own=False), plain))
