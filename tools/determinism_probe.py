"""Which primitives of the eval forward vary from run to run on identical inputs?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import _lib as L, fused_mlp, pointnet2_modules as pm, pointnet2_utils as pu
from graspbalance_amd.scene import make_batch
from tests.seeded import fill_by_key
DEV = "cuda:0"; lib = L.lib()
def same(name, fn, n=5):
    ref = fn(); bad = []
    for i in range(n):
        o = fn()
        for a, b in zip(ref, o):
            if not torch.equal(a, b):
                d = (a.double() - b.double())
                bad.append("%d differ, rel %.1e" % (int((d != 0).sum()), float(d.norm() / a.double().norm())))
                break
    print("%-40s %s" % (name, "identical" if not bad else "; ".join(bad[:2])))
torch.manual_seed(0)
for (P, K, N) in [(262144, 64, 64), (262144, 64, 128), (131072, 131, 128), (4096, 1024, 256), (4096, 256, 1024), (262144, 3, 64)]:
    X = torch.randn(P, K, device=DEV); W = torch.randn(N, K, device=DEV); Y = torch.empty(P, N, device=DEV)
    aff = torch.rand(2 * K, device=DEV)
    def f(a=None):
        lib.gb_gemm_fwd(L.ptr(X), L.ptr(W), a, L.ptr(Y), None, 1, P, K, N, None, None); torch.cuda.synchronize(); return (Y.clone(),)
    same("gemm_fwd %dx%dx%d" % (P, K, N), f)
    same("gemm_fwd aff %dx%dx%d" % (P, K, N), lambda: f(L.ptr(aff)))
clouds = torch.from_numpy(make_batch([0, 1], 20000)).to(DEV)
sa = fill_by_key(pm.PointnetSAModuleVotes(npoint=2048, radius=0.04, nsample=64, mlp=[0, 64, 64, 128], use_xyz=True, normalize_xyz=True), seed=1).to(DEV).eval()
with torch.no_grad():
    same("sa1 module eval", lambda: tuple(t.clone() for t in sa(clouds)))
    new_xyz, feats, inds = sa(clouds)
    idx = pu.ball_query(0.04, 64, clouds, new_xyz)
    same("group_concat_cl", lambda: (fused_mlp.group_concat_cl(clouds, new_xyz, idx, None, mode=1, scale=0.04).clone(),))
    x0 = fused_mlp.group_concat_cl(clouds, new_xyz, idx, None, mode=1, scale=0.04)
    same("shared_mlp_cl pool", lambda: (fused_mlp.shared_mlp_cl(x0, sa.mlp_module, pool_ns=64).clone(),))
    same("shared_mlp_cl nopool", lambda: (fused_mlp.shared_mlp_cl(x0, sa.mlp_module, pool_ns=0).clone(),))
    sa.train()
    same("shared_mlp_cl pool train", lambda: (fused_mlp.shared_mlp_cl(x0, sa.mlp_module, pool_ns=64).clone(),))

# ---- the DRP backbone, stage by stage -------------------------------------------------------------------------------
from graspbalance_amd.drp import DRP, InvResMLP
net = fill_by_key(DRP(), seed=3).to(DEV).eval()
with torch.no_grad():
    def run():
        feats, xyz, ep = net(clouds)
        return tuple(ep[k].clone() for k in ('sa1_features', 'sa2_features', 'fp2_features'))
    same("DRP eval (sa1, sa2, fp2 features)", run)
    blk = [m for m in net.modules() if isinstance(m, InvResMLP)][0]
    C = blk.convs.convs[0][0].weight.shape[1] - 3 if hasattr(blk.convs, 'convs') else 128
    p = clouds[:, :2048].contiguous()
    f_cl = torch.randn(2, 2048, 128, device=DEV)
    same("InvResMLP.forward_cl", lambda: (blk.forward_cl(p, f_cl).clone(),))
    g = blk.convs.grouper
    idx = pu.ball_query(g.radius, g.nsample, p, p)
    agg_conv, agg_bn = blk.convs.convs[0][0], blk.convs.convs[0][1]
    geo = fused_mlp.LocalGeometry(p, p, idx, mode=0)
    same("local_agg_pool", lambda: (fused_mlp.local_agg_pool(f_cl.reshape(-1, 128), agg_conv, agg_bn, geo).clone(),))
    agg = fused_mlp.local_agg_pool(f_cl.reshape(-1, 128), agg_conv, agg_bn, geo)
    same("pwconv chain", lambda: (fused_mlp.conv_bn_act_chain(agg, [(blk.pwconv[0][0], blk.pwconv[0][1]), (blk.pwconv[1][0], blk.pwconv[1][1])], residual=f_cl.reshape(-1, 128), relu_last=True).clone(),))
