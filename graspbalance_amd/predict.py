"""Inference loop step: eval forward + pred_decode, with the first-level furthest-point sampling of the NEXT batch on
a side stream when the loop already holds it (the same mechanism as Trainer.train_step(batch, next_batch=...)).

The reference's inference scripts iterate a DataLoader and call ``net(batch)`` then ``pred_decode`` per batch
(graspbalance.py:122-192); a loop over a dataset or a request queue knows its next batch, and the 20 000 -> 2048
sampling (2 ms of one workgroup per cloud, a fifth of a forward) depends on nothing but the input cloud.
"""
import torch

from . import fused_mlp
from .graspbalance import pred_decode
from .prefetch import AFTER_SA1, KEY, SamplingPrefetch


class Predictor:
    def __init__(self, net, device, prefetch_sampling=True):
        self.net = net.to(device).eval()
        self.device = torch.device(device)
        self.prefetch = None
        # the forward's small zero-initialised buffers (moment / statistic scratch of the fused kernels) come from a private
        # arena: the shared one belongs to training loops (weight gradients are views of it until optimizer.step())
        self._arena = fused_mlp.scoped_arena(self.device) if self.device.type == "cuda" else None
        sa1 = getattr(getattr(getattr(net, "view_estimator", None), "FeatureExtraction", None), "sa1", None)
        if prefetch_sampling and self.device.type == "cuda" and sa1 is not None and sa1.npoint:
            self.prefetch = SamplingPrefetch(self.device, sa1.npoint)

    @torch.no_grad()
    def __call__(self, batch, next_batch=None, decode=True):
        """batch: {'point_clouds': (B,N,3+) ...}.  Returns pred_decode's list of (Ng,17) grasps per cloud (or the
        network's end_points with decode=False).  next_batch: the batch of the following call, if already known."""
        if self._arena is None:
            return self._call(batch, next_batch, decode)
        with self._arena:
            fused_mlp.begin_step(self.device)
            return self._call(batch, next_batch, decode)

    def _call(self, batch, next_batch, decode):
        inputs = dict(batch)
        if self.prefetch is not None:
            inds = self.prefetch.take(batch['point_clouds'])
            if inds is not None:
                inputs[KEY] = inds
            if next_batch is not None:
                clouds = next_batch['point_clouds']
                inputs[AFTER_SA1] = lambda: self.prefetch.launch(clouds)
        end_points = self.net(inputs)
        if self.prefetch is not None and next_batch is not None and self.prefetch.pending is None:
            self.prefetch.launch(next_batch['point_clouds'])  # a backbone without the hook: start it now
        return pred_decode(end_points) if decode else end_points
