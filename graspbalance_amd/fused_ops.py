"""Ops that fuse several reference calls into one HIP launch (no reference counterpart)."""
import ctypes

import torch

from . import _lib
from . import pointnet2_utils
from .pointnet2 import _ext as _hip_ext


def cylinder_query_multi(xyz, new_xyz, rot, radii, hmin, hmaxs, nsample):
    """All len(radii) x len(hmaxs) cylinder queries of GraspPoseStage2 (graspbalance.py:84-87,
    modules.py:99-101) in ONE pass over the cloud.  Returns int32 (nr, nh, B, npoint, nsample);
    entry [ir, ih] is bit-identical to ``cylinder_query(radii[ir], hmin, hmaxs[ih], nsample, ...)``.
    CUDA tensors only; anything else goes through the per-query op (which raises on CPU in the
    product, like the reference extension)."""
    B, npoint, _ = new_xyz.size()
    rot9 = rot.view(B, npoint, 9)
    nr, nh = len(radii), len(hmaxs)
    if not xyz.is_cuda or nr > 4 or nh > 4:
        rows = [torch.stack([pointnet2_utils.cylinder_query(r, hmin, h, nsample, xyz, new_xyz, rot9)
                             for h in hmaxs], 0) for r in radii]
        return torch.stack(rows, 0)
    for t, name in ((xyz, "xyz"), (new_xyz, "new_xyz"), (rot9, "rot")):
        if not t.is_contiguous():
            raise RuntimeError("%s must be a contiguous tensor" % name)
        if t.dtype != torch.float32:
            raise RuntimeError("%s must be a float tensor" % name)
    out = torch.empty((nr, nh, B, npoint, nsample), dtype=torch.int32, device=xyz.device)
    ra = (ctypes.c_float * nr)(*[float(r) for r in radii])
    ha = (ctypes.c_float * nh)(*[float(h) for h in hmaxs])
    with _lib.device_ctx(xyz.device):
        stream = _lib.current_stream(xyz.device)
        meta = None
        if _lib.KernelTimer.active is not None:  # bench.py re-runs the timed queries afterwards to count scanned pairs
            meta = {"b": B, "n": xyz.size(1), "m": npoint, "ns": int(nsample), "radii": [float(r) for r in radii],
                    "hmin": float(hmin), "hmaxs": [float(h) for h in hmaxs], "args": (new_xyz, xyz, rot9)}
        _lib.check(_lib.timed("gb_cylinder_query_multi", xyz.device, meta, lambda: _lib.lib().gb_cylinder_query_multi(
            _lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(rot9), _lib.ptr(out), B, xyz.size(1), npoint,
            ctypes.cast(ra, ctypes.c_void_p), nr, float(hmin), ctypes.cast(ha, ctypes.c_void_p), nh,
            int(nsample), stream)), "cylinder_query_multi")
    return out


def fps_segments(xyz_packed, seg_sizes, sample_counts):
    """furthest_point_sample on every segment of a packed (T,3) point list in ONE launch (the per-object loop of
    ObjectBalanceSampling, modules.py:178-221).  seg_sizes / sample_counts: python int lists; returns int64 (sum of
    sample_counts,) indices INTO xyz_packed (segment offset already added), segment after segment."""
    if not xyz_packed.is_cuda:
        raise RuntimeError("fps_segments: CPU not supported")
    if not xyz_packed.is_contiguous() or xyz_packed.dtype != torch.float32:
        raise RuntimeError("xyz_packed must be a contiguous float tensor")
    if any(m > 0 and n <= 0 for n, m in zip(seg_sizes, sample_counts)):
        raise RuntimeError("fps_segments: cannot sample from an empty segment")
    dev = xyz_packed.device
    S = len(seg_sizes)
    offs = [0] * (S + 1)
    outs = [0] * (S + 1)
    for i in range(S):
        offs[i + 1] = offs[i] + int(seg_sizes[i])
        outs[i + 1] = outs[i] + int(sample_counts[i])
    assert offs[S] == xyz_packed.size(0)
    table = torch.tensor([offs, outs], dtype=torch.int32).to(dev, non_blocking=True)
    idx = torch.zeros(outs[S], dtype=torch.int32, device=dev)
    temp = torch.empty(max(offs[S], 1), dtype=torch.float32, device=dev)
    with _lib.device_ctx(dev):
        _lib.check(_lib.lib().gb_fps_segments(_lib.ptr(xyz_packed), _lib.ptr(table[0]), _lib.ptr(table[1]), _lib.ptr(temp),
                                              _lib.ptr(idx), S, max([int(n) for n in seg_sizes], default=0),
                                              _hip_ext.FPS_FLAGS, _lib.current_stream(dev)), "fps_segments")
    base = torch.repeat_interleave(table[0, :S].long(), torch.tensor(sample_counts, device=dev), output_size=outs[S])
    return idx.long() + base
