import sys, torch
sys.path.insert(0, '.')
from tests.test_fused_mlp_gpu import _crop_case
from graspbalance_amd import _lib, fused_mlp
wg, xyz, centres, rot, rows = _crop_case(B=4)
wg.train()
names = ["gb_gemm_fwd_pool", "gb_pool_pairs", "gb_gemm_fwd_w"]
fused_mlp.set_crop_pool(True, True)
for it in range(3):
    with _lib.KernelTimer(names, reserve=128) as kt, torch.no_grad():
        out = wg(centres, xyz, rot, rows=rows, channel_last=True)
    torch.cuda.synchronize()
print({k: "%dx%.0f" % (v["launches"], v["mean_ms"] * 1e3) for k, v in kt.summary().items()})
