"""How full are the query groups of the bench workload?  (padded slots repeat the group's first index)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graspbalance_amd import fused_ops, fused_mlp
from graspbalance_amd.synthetic import make_training_batch
from graspbalance_amd.train import Trainer
import graspbalance_amd.pointnet2_utils as pu

batch = make_training_batch(range(4), 20000, device="cuda:0")
tr = Trainer("cuda:0")

def fill(idx, name):
    first = idx[..., :1]
    pad = (idx[..., 1:] == first).sum().item()
    tot = idx.numel()
    print("%-40s groups %8d  ns %3d  unique rows %5.1f%%" % (name, idx[..., 0].numel(), idx.shape[-1], 100.0 * (tot - pad) / tot))

orig_multi = fused_ops.cylinder_query_multi
def multi(*a, **k):
    out = orig_multi(*a, **k)
    for i, t in enumerate(out):
        fill(t, "cylinder_query_multi radius #%d" % i)
    return out
fused_ops.cylinder_query_multi = multi
orig_bq = pu.ball_query
def bq(radius, nsample, xyz, new_xyz):
    out = orig_bq(radius, nsample, xyz, new_xyz)
    fill(out, "ball_query r=%.2f n=%d m=%d" % (radius, xyz.shape[1], new_xyz.shape[1]))
    return out
pu.ball_query = bq
import graspbalance_amd.drp as drp, graspbalance_amd.pointnet2_modules as pm
drp.ball_query = bq
tr.train_step(batch)
