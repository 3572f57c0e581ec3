"""``pointnet2._ext`` — the 10-function surface of the reference's PN-ext
(PointNet/_ext_src/src/bindings.cpp:12-26), backed by libgraspbal_hip.so.

Same contract as the reference wrappers (sampling.cpp, ball_query.cpp, group_points.cpp,
interpolate.cpp, cylinder_query.cpp): inputs must be contiguous float32 / int32 tensors
(``RuntimeError`` otherwise, the TORCH_CHECK texts of _ext_src/include/utils.h:10-30), outputs are
allocated here (the reference zero-fills them with ``torch::zeros``; the kernels here write every element -
empty ball rows as zeros - so only the scatter-add targets of the *_grad ops still need the fill), CPU tensors
raise ``RuntimeError("CPU not supported")``, launches go to the current stream and never synchronise.
"""
import torch

from .. import _lib

# FPS semantics of PN-ext: near-origin skip + 512-thread tree tie-break (sampling_gpu.cu:64-178)
FPS_FLAGS = _lib.FPS_SKIP_NEAR_ORIGIN | _lib.FPS_TIE_TREE512


def _check_contiguous(t, name):
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)


def _check_float(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be a float tensor" % name)


def _check_int(t, name):
    if t.dtype != torch.int32:
        raise RuntimeError("%s must be an int tensor" % name)


def _check_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)


def _require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("CPU not supported")


def furthest_point_sampling(points, nsamples):
    """(B,N,3) f32 -> (B,nsamples) i32.  sampling.cpp:70-91"""
    _check_contiguous(points, "points")
    _check_float(points, "points")
    _require_gpu(points)
    B, N = points.size(0), points.size(1)
    output = torch.empty((B, nsamples), dtype=torch.int32, device=points.device)  # every element is written by the kernel
    # (the wrapper's running min-distance buffer `tmp = full(1e10)` - sampling.cpp:81 - never leaves it: the library
    #  starts from 1e10 itself when given no buffer, one fill launch less per call)
    tmp = None if N <= 20480 else torch.full((B, N), 1e10, dtype=torch.float32, device=points.device)   # (the streamed forms keep it)
    with _lib.device_ctx(points.device):
        stream = _lib.current_stream(points.device)
        _lib.check(_lib.timed("gb_fps", points.device, {"b": B, "n": N, "m": nsamples},
                              lambda: _lib.fps(points, tmp, output, B, N, nsamples, FPS_FLAGS, stream)),
                   "furthest_point_sampling")
    return output


def gather_points(points, idx):
    """(B,C,N) f32, (B,M) i32 -> (B,C,M).  sampling.cpp:20-43"""
    _check_contiguous(points, "points")
    _check_contiguous(idx, "idx")
    _check_float(points, "points")
    _check_int(idx, "idx")
    if points.is_cuda:
        _check_cuda(idx, "idx")
    _require_gpu(points)
    B, C, N = points.shape
    M = idx.size(1)
    output = torch.empty((B, C, M), dtype=torch.float32, device=points.device)  # every element is written by the kernel
    with _lib.device_ctx(points.device):
        _lib.check(_lib.lib().gb_gather(_lib.ptr(points), _lib.ptr(idx), _lib.ptr(output), B, C, N, M,
                                        _lib.current_stream(points.device)), "gather_points")
    return output


def gather_points_grad(grad_out, idx, n):
    """(B,C,M) f32, (B,M) i32 -> (B,C,n).  sampling.cpp:45-69"""
    _check_contiguous(grad_out, "grad_out")
    _check_contiguous(idx, "idx")
    _check_float(grad_out, "grad_out")
    _check_int(idx, "idx")
    if grad_out.is_cuda:
        _check_cuda(idx, "idx")
    _require_gpu(grad_out)
    B, C, M = grad_out.shape
    output = torch.zeros((B, C, n), dtype=torch.float32, device=grad_out.device)
    with _lib.device_ctx(grad_out.device):
        _lib.check(_lib.lib().gb_gather_grad(_lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(output), B, C, n, M,
                                             _lib.current_stream(grad_out.device)), "gather_points_grad")
    return output


def ball_query(new_xyz, xyz, radius, nsample):
    """(B,M,3), (B,N,3) -> (B,M,nsample) i32.  ball_query.cpp:13-37"""
    _check_contiguous(new_xyz, "new_xyz")
    _check_contiguous(xyz, "xyz")
    _check_float(new_xyz, "new_xyz")
    _check_float(xyz, "xyz")
    if new_xyz.is_cuda:
        _check_cuda(xyz, "xyz")
    _require_gpu(new_xyz)
    B, M = new_xyz.size(0), new_xyz.size(1)
    N = xyz.size(1)
    idx = torch.empty((B, M, nsample), dtype=torch.int32, device=new_xyz.device)  # every element is written by the kernel
    with _lib.device_ctx(new_xyz.device):
        stream = _lib.current_stream(new_xyz.device)
        meta = None
        if _lib.KernelTimer.active is not None:  # bench.py re-runs the timed query afterwards to count scanned pairs
            meta = {"b": B, "n": N, "m": M, "ns": int(nsample), "radius": float(radius), "args": (new_xyz, xyz)}
        _lib.check(_lib.timed("gb_ball_query", new_xyz.device, meta,
                              lambda: _lib.lib().gb_ball_query(_lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(idx), None,
                                                               B, N, M, float(radius), int(nsample), stream)),
                   "ball_query")
    return idx


def cylinder_query(new_xyz, xyz, rot, radius, hmin, hmax, nsample):
    """(B,M,3), (B,N,3), (B,M,9) -> (B,M,nsample) i32.  cylinder_query.cpp:10-47"""
    _check_contiguous(new_xyz, "new_xyz")
    _check_contiguous(xyz, "xyz")
    _check_contiguous(rot, "rot")
    _check_float(new_xyz, "new_xyz")
    _check_float(xyz, "xyz")
    _check_float(rot, "rot")
    if new_xyz.is_cuda:
        _check_cuda(xyz, "xyz")
        _check_cuda(rot, "rot")
    _require_gpu(new_xyz)
    B, M = new_xyz.size(0), new_xyz.size(1)
    N = xyz.size(1)
    idx = torch.empty((B, M, nsample), dtype=torch.int32, device=new_xyz.device)  # every element is written by the kernel
    with _lib.device_ctx(new_xyz.device):
        _lib.check(_lib.lib().gb_cylinder_query(_lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(rot), _lib.ptr(idx),
                                                None, B, N, M, float(radius), float(hmin), float(hmax),
                                                int(nsample), _lib.current_stream(new_xyz.device)),
                   "cylinder_query")
    return idx


def group_points(points, idx):
    """(B,C,N) f32, (B,M,S) i32 -> (B,C,M,S).  group_points.cpp:21-49"""
    _check_contiguous(points, "points")
    _check_contiguous(idx, "idx")
    _check_float(points, "points")
    _check_int(idx, "idx")
    if points.is_cuda:
        _check_cuda(idx, "idx")
    _require_gpu(points)
    B, C, N = points.shape
    M, S = idx.size(1), idx.size(2)
    output = torch.empty((B, C, M, S), dtype=torch.float32, device=points.device)  # every element is written by the kernel
    with _lib.device_ctx(points.device):
        _lib.check(_lib.lib().gb_group(_lib.ptr(points), _lib.ptr(idx), _lib.ptr(output), B, C, N, M, S,
                                       _lib.current_stream(points.device)), "group_points")
    return output


def group_points_grad(grad_out, idx, n):
    """(B,C,M,S) f32, (B,M,S) i32 -> (B,C,n).  group_points.cpp:51-75"""
    _check_contiguous(grad_out, "grad_out")
    _check_contiguous(idx, "idx")
    _check_float(grad_out, "grad_out")
    _check_int(idx, "idx")
    if grad_out.is_cuda:
        _check_cuda(idx, "idx")
    _require_gpu(grad_out)
    B, C = grad_out.size(0), grad_out.size(1)
    M, S = idx.size(1), idx.size(2)
    output = torch.zeros((B, C, n), dtype=torch.float32, device=grad_out.device)
    with _lib.device_ctx(grad_out.device):
        _lib.check(_lib.lib().gb_group_grad(_lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(output), B, C, n, M, S,
                                            _lib.current_stream(grad_out.device)), "group_points_grad")
    return output


def three_nn(unknowns, knows):
    """(B,n,3), (B,m,3) -> [dist2 (B,n,3) f32, idx (B,n,3) i32].  interpolate.cpp:19-45"""
    _check_contiguous(unknowns, "unknowns")
    _check_contiguous(knows, "knows")
    _check_float(unknowns, "unknowns")
    _check_float(knows, "knows")
    if unknowns.is_cuda:
        _check_cuda(knows, "knows")
    _require_gpu(unknowns)
    B, n = unknowns.size(0), unknowns.size(1)
    m = knows.size(1)
    # (m >= 3: the kernel writes all three slots of every row; fewer known points leave slots at the wrapper's zeros)
    alloc = torch.empty if m >= 3 else torch.zeros
    idx = alloc((B, n, 3), dtype=torch.int32, device=unknowns.device)
    dist2 = alloc((B, n, 3), dtype=torch.float32, device=unknowns.device)
    with _lib.device_ctx(unknowns.device):
        _lib.check(_lib.lib().gb_three_nn(_lib.ptr(unknowns), _lib.ptr(knows), _lib.ptr(dist2), _lib.ptr(idx),
                                          B, n, m, _lib.current_stream(unknowns.device)), "three_nn")
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """(B,C,m) f32, (B,n,3) i32, (B,n,3) f32 -> (B,C,n).  interpolate.cpp:47-75"""
    _check_contiguous(points, "points")
    _check_contiguous(idx, "idx")
    _check_contiguous(weight, "weight")
    _check_float(points, "points")
    _check_int(idx, "idx")
    _check_float(weight, "weight")
    if points.is_cuda:
        _check_cuda(idx, "idx")
        _check_cuda(weight, "weight")
    _require_gpu(points)
    B, C, m = points.shape
    n = idx.size(1)
    output = torch.empty((B, C, n), dtype=torch.float32, device=points.device)  # every element is written by the kernel
    with _lib.device_ctx(points.device):
        _lib.check(_lib.lib().gb_three_interpolate(_lib.ptr(points), _lib.ptr(idx), _lib.ptr(weight),
                                                   _lib.ptr(output), B, C, m, n,
                                                   _lib.current_stream(points.device)), "three_interpolate")
    return output


def three_interpolate_grad(grad_out, idx, weight, m):
    """(B,C,n) f32, (B,n,3) i32, (B,n,3) f32 -> (B,C,m).  interpolate.cpp:76-104"""
    _check_contiguous(grad_out, "grad_out")
    _check_contiguous(idx, "idx")
    _check_contiguous(weight, "weight")
    _check_float(grad_out, "grad_out")
    _check_int(idx, "idx")
    _check_float(weight, "weight")
    if grad_out.is_cuda:
        _check_cuda(idx, "idx")
        _check_cuda(weight, "weight")
    _require_gpu(grad_out)
    B, C, n = grad_out.shape
    output = torch.zeros((B, C, m), dtype=torch.float32, device=grad_out.device)
    with _lib.device_ctx(grad_out.device):
        _lib.check(_lib.lib().gb_three_interpolate_grad(_lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(weight),
                                                        _lib.ptr(output), B, C, n, m,
                                                        _lib.current_stream(grad_out.device)),
                   "three_interpolate_grad")
    return output
