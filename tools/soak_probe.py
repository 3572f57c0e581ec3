import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer, _leaves
dev = torch.device("cuda", 0)
pool = [make_training_batch(range(4 * i, 4 * i + 4), 20000, device=dev) for i in range(4)]
n = 0; byt = 0
for p, t in _leaves(pool[0]):
    n += 1; byt += t.numel() * t.element_size()
print("tensors per batch:", n, "bytes: %.2f GB" % (byt / 1e9))
big = sorted(((t.numel() * t.element_size(), p, tuple(t.shape), t.dtype) for p, t in _leaves(pool[0])), reverse=True)[:6]
for b in big: print("   %.1f MB" % (b[0] / 1e6), b[1], b[2], b[3])
tr = Trainer(dev)
for i in range(6):
    tr.train_step(pool[i % 4], next_batch=pool[(i + 1) % 4])
torch.cuda.synchronize()
st = tr._static
import types
orig_load = st.load
T = {"load": 0.0}
def load(batch):
    torch.cuda.synchronize(); t0 = time.perf_counter(); orig_load(batch); torch.cuda.synchronize(); T["load"] += time.perf_counter() - t0
st.load = load
t0 = time.perf_counter()
for i in range(20):
    tr.train_step(pool[i % 4], next_batch=pool[(i + 1) % 4])
torch.cuda.synchronize()
print("per step %.2f ms, of which staging (synchronised) %.2f ms; graphs %d, replays %d" % ((time.perf_counter() - t0) / 20 * 1e3, T["load"] / 20 * 1e3, len(tr._graphs), tr.graph_replays))
