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
stages that leave most of the chip idle anyway; ``gb_set_reserved_cus`` makes the persistent GEMM size its grid for
the CUs the sampling does not occupy.
Measured and rejected: giving the sampling stream CUs of its own with CU masks (hipExtStreamCreateWithCUMask,
GB_CU_MASK=1 - one CU of every XCD for the sampling, 248 for the step).  Workgroups are dealt round-robin over the
shader engines, so ONE masked CU makes its engine (7 instead of 8 CUs) the bottleneck of every kernel: a plain grid
of 4096 workgroups ran 18 % longer on the 248-CU mask (tools/cumask_probe.hip) and the train step 30.7 instead of
22.6 ms.
"""
import ctypes
import os

import torch

from . import _lib
from . import pointnet2_utils

KEY = '_sa1_inds_prefetched'
AFTER_SA1 = '_after_sa1'  # callable the backbone runs once its first set-abstraction level is enqueued


def _ident(clouds):
    return (clouds.data_ptr(), tuple(clouds.shape), clouds._version)


RESERVED_CUS = 8  # one per XCD


def _masked_streams(device):
    """(training stream, sampling stream) as torch ExternalStreams over CU-masked HIP streams, or (None, plain side
    stream) when masking is switched off."""
    if os.environ.get("GB_CU_MASK", "0") != "1":  # measured slower (module docstring): off unless asked for
        return None, torch.cuda.Stream(device=device)
    with torch.cuda.device(device):
        n = ctypes.c_int(0)
        _lib.check(_lib.lib().gb_device_cu_count(ctypes.byref(n)), "gb_device_cu_count")
        words = (n.value + 31) // 32
        side = [0] * words
        main = [0] * words
        for cu in range(n.value):
            (side if cu < RESERVED_CUS else main)[cu // 32] |= 1 << (cu % 32)
        out = []
        for mask in (main, side):
            arr = (ctypes.c_uint32 * words)(*mask)
            handle = ctypes.c_void_p()
            _lib.check(_lib.lib().gb_stream_create_cu_mask(ctypes.cast(arr, ctypes.c_void_p), words, ctypes.byref(handle)),
                       "gb_stream_create_cu_mask")
            out.append(torch.cuda.ExternalStream(handle.value, device=device))
    return out[0], out[1]


class SamplingPrefetch:
    def __init__(self, device, npoint):
        self.device = torch.device(device)
        self.npoint = int(npoint)
        self.main, self.side = _masked_streams(self.device)  # main: the stream the training step should run on
        self.pending = None
        self._reserved = 0

    def launch(self, clouds):
        """Start sampling `clouds` ((B,N,3+) fp32 on this device) on the side stream."""
        xyz = clouds[..., 0:3].contiguous()
        B = xyz.shape[0]
        want = RESERVED_CUS if self.main is not None else min(B, 128)
        if want != self._reserved:
            _lib.check(_lib.lib().gb_set_reserved_cus(want), "gb_set_reserved_cus")
            self._reserved = want
        self.side.wait_stream(torch.cuda.current_stream(self.device))  # `clouds` may have just been produced
        with torch.cuda.stream(self.side):
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            done = torch.cuda.Event()
            done.record(self.side)
        self.pending = (_ident(clouds), clouds, inds, done)

    def take(self, clouds):
        """The indices launched for exactly this tensor (same storage, shape and version), ordered after the side
        stream's work on the current stream; None when this batch was not announced."""
        rec, self.pending = self.pending, None
        if rec is None or rec[0] != _ident(clouds):
            return None
        _, _, inds, done = rec
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)
        inds.record_stream(cur)
        return inds
