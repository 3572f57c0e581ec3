"""Label matching kernels in isolation: gather of labels / offsets (+ width column) / tolerance and the score transform."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import label_generation as lg
from graspbalance_amd.synthetic import make_training_batch
DEV = "cuda:0"
batch = make_training_batch(range(4), 20000, device=DEV)
ep = dict(batch); ep['input_xyz'] = batch['point_clouds']; ep['fp2_xyz'] = batch['point_clouds'][:, :1024].contiguous()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
print("process_grasp_labels (fused): %.0f us" % timeit(lambda: lg.process_grasp_labels(dict(ep))))
