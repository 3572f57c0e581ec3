"""Ops that fuse several reference calls into one HIP launch (no reference counterpart)."""
import ctypes

import torch

from . import _lib
from . import pointnet2_utils


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
    with torch.cuda.device(xyz.device):
        _lib.check(_lib.lib().gb_cylinder_query_multi(
            _lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(rot9), _lib.ptr(out), B, xyz.size(1), npoint,
            ctypes.cast(ra, ctypes.c_void_p), nr, float(hmin), ctypes.cast(ha, ctypes.c_void_p), nh,
            int(nsample), _lib.current_stream(xyz.device)), "cylinder_query_multi")
    return out
