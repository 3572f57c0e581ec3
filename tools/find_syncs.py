"""List the host<->device synchronisation points of one train step (torch sync debug mode)."""
import os, sys, warnings, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0")
for _ in range(2):
    tr.train_step(batch)
torch.cuda.synchronize()
seen = []
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "/root/repo/" in f.filename or "graspbalance_amd" in f.filename]
    st = [f for f in st if "find_syncs" not in f.filename]
    if not st:
        st = [f for f in traceback.extract_stack() if "find_syncs" not in f.filename and "warnings" not in f.filename][-6:]
    seen.append(" <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[-4:][::-1]))
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
tr.train_step(batch)
torch.cuda.set_sync_debug_mode("default")
for s in seen:
    print(s)
print(len(seen), "syncs")
