"""Host-side profile of the train step (cProfile over 6 steps): which Python functions the ~16 ms of enqueue work per step
are spent in.  Run on the GPU box."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer

batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0")
for _ in range(4):
    tr.train_step(batch, next_batch=batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(6):
    tr.train_step(batch, next_batch=batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(40)
