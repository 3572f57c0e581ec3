import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import fused_mlp
DEV = "cuda:0"

def part_a():
    """which aten op launches naive_conv kernels in a fused train step"""
    from graspbalance_amd.synthetic import make_training_batch
    from graspbalance_amd.train import Trainer
    from torch.profiler import profile, ProfilerActivity
    tr = Trainer(DEV)
    batch = make_training_batch(range(2), 20000, device=DEV)
    for _ in range(2):
        tr.train_step(batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        tr.train_step(batch)
        torch.cuda.synchronize()
    seen = {}
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ("conv" in ev.name or "miopen" in ev.name):
            kn = [k.name[:40] for k in ev.kernels]
            if any("naive" in k for k in kn):
                key = (ev.name, str(ev.input_shapes)[:150])
                seen[key] = seen.get(key, 0) + 1
    for k, v in seen.items():
        print(v, k)

def part_b():
    from graspbalance_amd.drp import InvResMLP
    from graspbalance_amd.scene import make_batch
    torch.manual_seed(4)
    blk = InvResMLP(in_channels=32, aggr_args={'feature_type': 'dp_fj', "reduction": 'max'}, norm_args={'norm': 'bn'},
                    act_args={'act': 'relu'}, group_args={'NAME': 'ballquery', 'radius': 0.1, 'nsample': 16},
                    conv_args={'order': 'conv-norm-act'}, expansion=4, use_res=True).to(DEV).train()
    p = torch.from_numpy(make_batch([0, 1], 1024)).to(DEV)
    f0 = torch.randn(2, 32, 1024, device=DEV)
    res = {}
    for flag in (True, False):
        fused_mlp.set_enabled(flag)
        b = copy.deepcopy(blk)
        f = f0.clone().requires_grad_(True)
        # stage by stage
        if flag:
            f_cl = f.transpose(1, 2).contiguous()
            from graspbalance_amd.modified_net_tools.group import ball_query
            g = b.convs.grouper
            idx = ball_query(g.radius, g.nsample, p, p)
            x0 = fused_mlp.group_concat_cl(p, p, idx, f_cl, mode=0)
            agg = fused_mlp.conv_bn_act(x0, b.convs.convs[0][0], b.convs.convs[0][1], relu=True, pool_ns=16)
            agg_std = agg.view(2, 1024, 32).transpose(1, 2)
        else:
            agg_std = b.convs([p, f])
        torch.manual_seed(1)
        w = torch.randn(2, 32, 1024, device=DEV)
        (agg_std * w).sum().backward()
        res[flag] = (agg_std.detach().clone(), f.grad.clone(), {k: v.grad.clone() for k, v in b.named_parameters() if v.grad is not None})
    fused_mlp.set_enabled(True)
    def rel(a, b): return float((a - b).abs().max() / (b.abs().max() + 1e-12))
    print("LocalAggregation only: fwd", rel(res[True][0], res[False][0]), "dinput", rel(res[True][1], res[False][1]))
    for k in res[False][2]:
        print("   ", k, rel(res[True][2][k], res[False][2][k]))



def part_c():
    from graspbalance_amd.drp import InvResMLP
    from graspbalance_amd.scene import make_batch
    torch.manual_seed(4)
    blk = InvResMLP(in_channels=32, aggr_args={'feature_type': 'dp_fj', "reduction": 'max'}, norm_args={'norm': 'bn'},
                    act_args={'act': 'relu'}, group_args={'NAME': 'ballquery', 'radius': 0.1, 'nsample': 16},
                    conv_args={'order': 'conv-norm-act'}, expansion=4, use_res=True).to(DEV).train()
    p = torch.from_numpy(make_batch([0, 1], 1024)).to(DEV)
    f0 = torch.randn(2, 32, 1024, device=DEV)
    agg0 = torch.randn(2, 32, 1024, device=DEV)
    def rel(a, b): return float((a - b).abs().max() / (b.abs().max() + 1e-12))
    # pointwise + residual only, same agg input
    res = {}
    for flag in (True, False):
        b = copy.deepcopy(blk)
        agg = agg0.clone().requires_grad_(True)
        f = f0.clone().requires_grad_(True)
        if flag:
            a_cl = agg.transpose(1, 2).reshape(2048, 32)
            f_cl = f.transpose(1, 2).reshape(2048, 32)
            h = fused_mlp.conv_bn_act(a_cl, b.pwconv[0][0], b.pwconv[0][1], relu=True)
            out = fused_mlp.conv_bn_act(h, b.pwconv[1][0], b.pwconv[1][1], relu=True, residual=f_cl)
            out = out.view(2, 1024, 32).transpose(1, 2)
        else:
            y = b.pwconv(agg)
            y = y + f
            out = torch.relu(y)
        torch.manual_seed(1)
        w = torch.randn(2, 32, 1024, device=DEV)
        (out * w).sum().backward()
        res[flag] = (out.detach().clone(), agg.grad.clone(), f.grad.clone(), {k: v.grad.clone() for k, v in b.named_parameters() if v.grad is not None})
    print("pointwise+residual: fwd", rel(res[True][0], res[False][0]), "dagg", rel(res[True][1], res[False][1]), "dres", rel(res[True][2], res[False][2]))
    for k in res[False][3]:
        print("   ", k, rel(res[True][3][k], res[False][3][k]))
    # whole single block
    res = {}
    for flag in (True, False):
        fused_mlp.set_enabled(flag)
        b = copy.deepcopy(blk)
        f = f0.clone().requires_grad_(True)
        _, out = b([p, f])
        torch.manual_seed(1)
        w = torch.randn(2, 32, 1024, device=DEV)
        (out * w).sum().backward()
        res[flag] = (out.detach().clone(), f.grad.clone())
    fused_mlp.set_enabled(True)
    print("single block: fwd", rel(res[True][0], res[False][0]), "dinput", rel(res[True][1], res[False][1]))


if __name__ == "__main__":
    part_c()
