"""fused MLPStack gradients vs an fp64 torch autograd of the same rows (X0 -> [linear, BN(batch), ReLU]xL -> max over ns)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from graspbalance_amd import fused_mlp, pointnet2_utils as pu
from tests.seeded import fill_by_key
from tests.golden import make_golden_r2 as mk
DEV = "cuda:0"
xyz = mk.g16_cloud(DEV)
inds = pu.furthest_point_sample(xyz, 2048)
new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
idx = pu.ball_query(0.04, 64, xyz, new_xyz)
def case(widths, scale, eps=1e-5, epsl=None, pool=64):
    epsl = epsl or [eps] * len(widths)
    X0 = (fused_mlp.group_concat_cl(xyz, new_xyz, idx, None, mode=0) * scale).contiguous()
    convs = [nn.Conv2d(a, b, 1, bias=False) for a, b in zip([3] + widths[:-1], widths)]
    bns = [nn.BatchNorm2d(b, eps=e) for b, e in zip(widths, epsl)]
    mods = fill_by_key(nn.ModuleList(convs + bns), seed=16).to(DEV).train()
    L = len(widths)
    out = fused_mlp.conv_bn_act_chain(X0, [(mods[i], mods[L + i]) for i in range(L)], pool_ns=pool)
    torch.manual_seed(1); w = torch.randn_like(out)
    (out * w).sum().backward()
    got = {k: p.grad.double().clone() for k, p in mods.named_parameters()}
    m64 = fill_by_key(nn.ModuleList([nn.Conv2d(a, b, 1, bias=False) for a, b in zip([3] + widths[:-1], widths)] + [nn.BatchNorm2d(b, eps=e) for b, e in zip(widths, epsl)]), seed=16).to(DEV).double().train()
    x = X0.double()
    for i in range(L):
        W = m64[i].weight.view(widths[i], -1)
        y = x @ W.t()
        mean, var = y.mean(0), y.var(0, unbiased=False)
        x = torch.relu((y - mean) / torch.sqrt(var + epsl[i]) * m64[L + i].weight + m64[L + i].bias)
        if i == 0: v0 = var.clone()
    o = x.view(-1, 64, widths[-1]).max(1)[0] if pool else x
    (o * w.double()).sum().backward()
    print(widths, "scale", scale, "eps", epsl, "pool", pool, "fwd %.2e" % float((out.double() - o).norm() / o.norm()), "min var0 %.2e" % float(v0.min()))
    for k, p in m64.named_parameters():
        print("   %-12s %.2e" % (k, float((got[k] - p.grad).norm() / p.grad.norm())))
case([64, 64, 128], 1.0)
case([64, 64, 128], 1.0)
case([64, 64, 128], 25.0)
case([64, 64, 128], 1.0)
case([64, 64, 128], 1.0, epsl=[1e-5, 1e-5, 1.0001e-5])
