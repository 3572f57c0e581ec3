"""Training harness with the loop semantics of the reference's train.py (Adam :94, OneCycleLR
:107-108, BN-momentum schedule :110-113, per-batch body :138-155) on the GraspBalance model, plus the
RCCL data-parallel gradient reduction.  (The reference script itself cannot run as shipped: it imports
the absent ``graspnet.GraspNet_MSCQ`` and dataset class names that do not exist — SURVEY.md §0.)
"""
import contextlib

import torch
from torch.optim.lr_scheduler import OneCycleLR

from . import fused_mlp
from .data_parallel import FlatGradAllReduce, broadcast_module
from .flat_adam import FlatAdam
from .graspbalance import GraspBalance
from .label_generation import LEAN
from .loss import get_loss
from .pytorch_utils import BNMomentumScheduler

import os
_PREFETCH_AT = os.environ.get("GB_PREFETCH_AT", "sa1")  # A/B switch: "start" | "sa1" | "off"
_NO_CONTEXT = contextlib.nullcontext()

BN_MOMENTUM_INIT = 0.5
BN_MOMENTUM_MAX = 0.001


class Trainer:
    def __init__(self, device, learning_rate=0.001, weight_decay=0.0, bn_decay_step=2, bn_decay_rate=0.5,
                 steps_per_epoch=100, max_epoch=18, num_view=300, seed=1234, distributed=False,
                 bucket_mb=16.0, model=None, time_collectives=False, prefetch_sampling=True, mlp_precision=None,
                 lean_labels=True):
        torch.manual_seed(seed)
        self.device = torch.device(device)
        # 'bf16': BASELINE configs[4].  A property of THIS trainer: every step runs inside fused_mlp.precision(...), the
        # GEMM calls carry it (GbGemmOpts), so two trainers of one process may differ
        self.mlp_precision = mlp_precision if self.device.type == "cuda" else None
        self.lean_labels = bool(lean_labels) and os.environ.get("GB_LEAN_LABELS", "1") != "0"  # A/B switch
        self.net = model if model is not None else GraspBalance(
            input_feature_dim=0, num_view=num_view, num_angle=12, num_depth=4, cylinder_radius=0.08,
            hmin=-0.02, hmax_list=[0.01, 0.02, 0.03, 0.04])
        self.net.to(self.device)
        if distributed:
            broadcast_module(self.net)
        # the reference's optimizer (train.py:94: Adam, default betas / eps) on one flat parameter buffer: one fused
        # launch over 9 M elements instead of torch's multi-tensor lists over 253 tensors (flat_adam.py)
        self.optimizer = FlatAdam(self.net.parameters(), lr=learning_rate, weight_decay=weight_decay)
        self.scheduler = OneCycleLR(self.optimizer, max_lr=learning_rate, steps_per_epoch=steps_per_epoch,
                                    epochs=max_epoch)
        bn_lbmd = lambda it: max(BN_MOMENTUM_INIT * bn_decay_rate ** (int(it / bn_decay_step)), BN_MOMENTUM_MAX)
        self.bnm_scheduler = BNMomentumScheduler(self.net, bn_lambda=bn_lbmd, last_epoch=-1)
        # the buckets ARE slices of the optimizer's flat gradient buffer: the averaged gradient lands where the
        # update reads it (no second 36 MB copy per step)
        self.grads = FlatGradAllReduce(self.net, bucket_mb=bucket_mb, timing=time_collectives,
                                       flat=(self.optimizer._flat_g, self.optimizer._grad_views, self.optimizer._params))
        self.bnm_scheduler.step()
        self.net.train()
        # first-level FPS of the next batch on a side stream (prefetch.py); needs the caller to pass `next_batch`
        self.prefetch = None
        sa1 = getattr(getattr(getattr(self.net, "view_estimator", None), "FeatureExtraction", None), "sa1", None)
        if prefetch_sampling and _PREFETCH_AT != "off" and self.device.type == "cuda" and sa1 is not None and sa1.npoint:
            from .prefetch import SamplingPrefetch
            self.prefetch = SamplingPrefetch(self.device, sa1.npoint)

    def train_step(self, batch, next_batch=None):
        """forward -> loss -> backward -> gradient all-reduce -> Adam step -> LR step.  Returns the
        loss tensor (no host sync here; the reference's per-key .item() logging is the caller's).
        next_batch: the batch of the FOLLOWING call, if the loop already holds it - its first-level furthest-point
        sampling then runs on a side stream under this step (prefetch.py)."""
        with (fused_mlp.precision(self.mlp_precision) if self.mlp_precision is not None else _NO_CONTEXT):
            return self._train_step(batch, next_batch)

    def _train_step(self, batch, next_batch=None):
        if self.device.type == "cuda":
            fused_mlp.begin_step(self.device)  # one re-zeroed arena for the step's small fp64 reduction buffers
        inputs = dict(batch)  # the network adds its outputs to the dict it is given
        if self.lean_labels:
            inputs[LEAN] = True  # label matching builds only what this step reads (label_generation._lean_labels)
        if self.prefetch is not None:
            from .prefetch import AFTER_SA1, KEY
            inds = self.prefetch.take(batch['point_clouds'])
            if inds is not None:
                inputs[KEY] = inds
            if next_batch is not None:
                clouds = next_batch['point_clouds']
                if _PREFETCH_AT == "start":
                    self.prefetch.launch(clouds)
                else:
                    inputs[AFTER_SA1] = lambda: self.prefetch.launch(clouds)
        end_points = self.net(inputs)
        if self.prefetch is not None and next_batch is not None and self.prefetch.pending is None:
            self.prefetch.launch(next_batch['point_clouds'])  # a backbone without the hook: start it now
        loss, end_points = get_loss(end_points)
        loss.backward()
        self.grads.reduce()
        self.optimizer.step()
        self.grads.zero_grad()  # .grad = None: the next backward assigns instead of accumulating
        self.scheduler.step()
        return loss
