"""ObjectBalanceSampling at full size (B=4 clouds x 20000 points, 8 objects): per-object loop vs one segmented launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from graspbalance_amd import modules
from graspbalance_amd.scene import make_batch
B, N, K = 4, 20000, 8
clouds = torch.from_numpy(make_batch(range(B), N)).cuda()
seg = torch.from_numpy(np.random.default_rng(0).integers(0, K + 1, size=(B, N))).cuda()
feats = torch.randn(B, 256, N, device="cuda")
def ep():
    return {'point_clouds': clouds, 'seed_cluster': seg, 'up_sample_features': feats,
            'fp2_inds': torch.zeros(B, 1024, dtype=torch.int32, device="cuda")}
for name, fn in (("per-object loop", modules._object_balance_sampling_loop), ("segmented launch", modules.ObjectBalanceSampling)):
    for _ in range(2): out = fn(ep())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = fn(ep())
    torch.cuda.synchronize()
    print("%-18s %.2f ms" % (name, (time.perf_counter() - t0) / 5 * 1e3))
