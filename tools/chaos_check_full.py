"""Full-size counterpart of chaos_check.py: the real DRP backbone (4 SA levels, 15 InvResMLP blocks, train-mode
BatchNorm) on two 20 000-point clouds.  How far do the output features move when the PLAIN torch path's weights are
perturbed by one fp32 ulp, and where does the fused path sit relative to that?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import fused_mlp
from graspbalance_amd.drp import DRP
from graspbalance_amd.scene import make_batch
DEV = "cuda:0"
clouds = torch.from_numpy(make_batch([0, 1], 20000)).to(DEV)

def run(flag, perturb=0.0):
    torch.manual_seed(11)
    drp = DRP().to(DEV).train()
    if perturb:
        torch.manual_seed(123)
        with torch.no_grad():
            for p in drp.parameters():
                p.mul_(1.0 + perturb * torch.randn_like(p))
    fused_mlp.set_enabled(flag)
    with torch.no_grad():
        feats, _, ep = drp(clouds)
    fused_mlp.set_enabled(True)
    return feats.clone(), ep['sa1_features'].clone()

def gap(a, b):
    return tuple(float((x - y).norm() / y.norm()) for x, y in zip(a, b))

plain = run(False)
print("                               fp2 features   sa1(+stage1) features")
print("plain vs plain (again):        %.2e       %.2e" % gap(run(False), plain))
print("plain vs plain (1e-7 perturb): %.2e       %.2e" % gap(run(False, 1e-7), plain))
print("plain vs plain (1e-6 perturb): %.2e       %.2e" % gap(run(False, 1e-6), plain))
print("fused vs plain:                %.2e       %.2e" % gap(run(True), plain))
