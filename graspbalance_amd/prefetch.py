"""Furthest-point sampling of the NEXT batch on a side stream.

The first-level FPS (20 000 -> 2048) is one workgroup per cloud iterating 2047 dependent steps: ~2 ms during which
B of the 256 CUs are busy and nothing else of the step can start, because the whole network hangs off its result.
It depends on nothing but the input cloud, so a training loop that already holds its next batch (any DataLoader
does) can run it one step ahead: ``launch(next_clouds)`` enqueues it on a side stream at the start of step t, it
runs under step t's GEMMs, and step t+1's first set-abstraction level takes the indices through ``take``.  Exactly
one FPS per batch is executed - nothing is cached across steps; a batch that was not announced is sampled inline.

Where it runs matters.  Its workgroups (1024 threads, 80 KB of LDS, 2 ms) share their CUs with whatever the step
runs: next to the tall row-streaming GEMMs (statically partitioned: the launch waits for its slowest workgroup) a
launch takes +11 % and the sampling itself 2.7 instead of 2.0 ms, so the trainer starts it right AFTER the first
set-abstraction level has been enqueued (drp.DRP runs ``AFTER_SA1``), under the few-tile GEMMs of the InvResMLP
stages that leave most of the chip idle anyway; while a sampling is in flight ``fused_mlp.set_reserved_cus`` makes
the persistent GEMMs size their grids for the CUs the sampling does not occupy (GbGemmOpts.reserved_cus, per call),
and ``take`` - the next step's first action - restores the full device.
Measured and rejected: giving the sampling stream CUs of its own with CU masks (hipExtStreamCreateWithCUMask: one CU
of every XCD for the sampling, 248 for the step).  Workgroups are dealt round-robin over the shader engines, so ONE
masked CU makes its engine (7 instead of 8 CUs) the bottleneck of every kernel: a plain grid of 4096 workgroups ran
18 % longer on the 248-CU mask (tools/cumask_probe.hip) and the train step 30.7 instead of 22.6 ms.
"""
import torch

from . import fused_mlp
from . import pointnet2_utils

KEY = '_sa1_inds_prefetched'
AFTER_SA1 = '_after_sa1'  # callable the backbone runs once its first set-abstraction level is enqueued


def _ident(clouds):
    return (clouds.data_ptr(), tuple(clouds.shape), clouds._version)


class SamplingPrefetch:
    def __init__(self, device, npoint):
        self.device = torch.device(device)
        self.npoint = int(npoint)
        self.side = torch.cuda.Stream(device=self.device)
        self.pending = None

    def launch(self, clouds):
        """Start sampling `clouds` ((B,N,3+) fp32 on this device) on the side stream."""
        cur = torch.cuda.current_stream(self.device)
        self.side.wait_stream(cur)  # `clouds` may have just been produced
        with torch.cuda.stream(self.side):
            # a cloud with extra channels makes this a fresh copy: allocated and consumed on the side stream, and kept
            # in `pending` until take() - the caching allocator must not hand its block to the main stream while
            # the 2 ms sampling is still reading it
            xyz = clouds[..., 0:3].contiguous()
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            done = torch.cuda.Event()
            done.record(self.side)
        fused_mlp.set_reserved_cus(min(xyz.shape[0], 128))
        self.pending = (_ident(clouds), clouds, xyz, inds, done)

    def take(self, clouds):
        """The indices launched for exactly this tensor (same storage, shape and version), ordered after the side
        stream's work on the current stream; None when this batch was not announced."""
        rec, self.pending = self.pending, None
        fused_mlp.set_reserved_cus(0)  # nothing of ours is resident on the side stream past this point
        if rec is None:
            return None
        _, _, xyz, inds, done = rec
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)  # also orders the release of `xyz` / a rejected `inds` after the sampling
        if rec[0] != _ident(clouds):
            return None
        inds.record_stream(cur)
        return inds
