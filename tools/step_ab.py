"""A/B: train-step time with own MFMA GEMMs vs torch.mm (rocBLAS) inside the fused channel-last path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import fused_mlp
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
batch = make_training_batch(range(4), 20000, device="cuda:0")
for own in (True, False, True, False):
    fused_mlp.set_own_gemm(own)
    tr = Trainer("cuda:0")
    for _ in range(3): tr.train_step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): tr.train_step(batch)
    torch.cuda.synchronize()
    print("own_gemm=%s: %.2f ms/step" % (own, (time.perf_counter() - t0) / 6 * 1e3))
