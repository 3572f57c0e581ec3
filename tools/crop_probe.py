"""Probe: the crop stack's kernels in isolation, per path (GB_CROP_POOL / GB_CROP_LOWRANK)."""
import sys, torch
sys.path.insert(0, '.')
from tests.test_fused_mlp_gpu import _crop_case
from graspbalance_amd import _lib, fused_mlp
wg, xyz, centres, rot, rows = _crop_case(B=4)
wg.train()
names = ["gb_crop_bwd_sparse", "gb_crop_bwd_coef", "gb_crop_bwd_dense", "gb_gemm_gram", "gb_crop_bwd_dw", "gb_gemm_fwd_pool",
         "gb_pool_pairs", "gb_gemm_fwd_w", "gb_affine_relu_maxpool_members", "gb_bn_bwd_stats_pool", "gb_bn_bwd_apply_members",
         "gb_gemm_dgrad", "gb_gemm_wgrad", "gb_bn_bwd_apply_w", "gb_gemm_dgrad_first", "gb_bn_bwd_apply_members_v", "gb_bn_bwd_stats"]
for pool, lowrank in ((False, False), (True, False), (True, True)):
    fused_mlp.set_crop_pool(pool, lowrank)
    for it in range(3):
        with _lib.KernelTimer(names, reserve=128) as kt:
            out = wg(centres, xyz, rot, rows=rows, channel_last=True)
            out.sum().backward()
        torch.cuda.synchronize()
    s = kt.summary()
    print("pool", pool, "lowrank", lowrank, "rows", rows[0].shape[0],
          {k: "%dx%.0f" % (v["launches"], v["mean_ms"] * 1e3) for k, v in s.items()},
          "total %.0f us" % sum(v["launches"] * v["mean_ms"] * 1e3 for v in s.values()))
