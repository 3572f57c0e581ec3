"""Which parameter gradients of the captured step differ from the launch-by-launch step's (one step from identical state)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
DEV = "cuda:0"
B = int(os.environ.get("B", 4))
batch = make_training_batch(list(range(B)), num_point=20000, device=DEV)


def first(graph, announce=True, eager_on_graph_trainer=False):
    tr = Trainer(DEV, steps_per_epoch=10, max_epoch=2, graph=graph)
    step = tr.train_step_eager if eager_on_graph_trainer else tr.train_step
    loss = float(step(batch, next_batch=batch if announce else None).detach())
    torch.cuda.synchronize()
    names = [n for n, p in tr.net.named_parameters() if p.requires_grad]
    sizes = [p.numel() for p in tr.optimizer._params]
    return tr.optimizer._flat_g.double().clone(), names, sizes, loss


rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
e1, names, sizes, l1 = first(False)
e2, _, _, l2 = first(False)
g, _, _, lg = first(True)
g2, _, _, lg2 = first(True, announce=False)
h1, _, _, lh1 = first(True, eager_on_graph_trainer=True)
h2, _, _, lh2 = first(True, eager_on_graph_trainer=True)
print("graph trainer, launch by launch (same kernels and arguments as the replay): losses %.8f %.8f" % (lh1, lh2))
print("   eagerG/eagerG %.2e  graph/eagerG %.2e  eagerG/eager %.2e" % (rel(h2, h1), rel(g, h1), rel(h1, e1)))
print("losses eager %.8f %.8f graph %.8f graph(no announce) %.8f" % (l1, l2, lg, lg2))
print("total: eager/eager %.2e graph/eager %.2e graph(no announce)/eager %.2e graph/graph %.2e" % (rel(e2, e1), rel(g, e1), rel(g2, e1), rel(g2, g)))
rows = []
for n, a, b, c in zip(names, e1.split(sizes), e2.split(sizes), g.split(sizes)):
    rows.append((rel(c, a), rel(b, a), float(a.norm()), n, a.numel()))
rows.sort(reverse=True)
for r in rows[:25]:
    print("graph/eager %.2e  eager/eager %.2e  |g| %.3e  %s (%d)" % r)
big = [r for r in rows if r[0] > 10 * r[1] + 1e-5]
print(len(big), "of", len(rows), "tensors beyond 10x their eager spread")
