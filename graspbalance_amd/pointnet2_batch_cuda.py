"""``pointnet2_batch_cuda`` — the 9-function surface of the reference's PB-ext
(pointnet2_batch/src/pointnet2_api.cpp:10-24), backed by libgraspbal_hip.so.

Contract of the reference wrappers (ball_query.cpp, group_points.cpp, sampling.cpp,
interpolate.cpp): dimensions are passed explicitly, outputs are CALLER-allocated tensors that the
kernels write into (``*_grad`` accumulate into a caller-zeroed tensor), integer wrappers return 1,
the interpolate wrappers return None.  The reference launches on the legacy default stream; here
the current torch stream is used so results are ordered with the caller's other work.
"""
import torch

from . import _lib

# FPS semantics of PB-ext: no near-origin skip, tree tie-break with up to 1024 threads
FPS_FLAGS = _lib.FPS_TIE_TREE1024


def _gpu(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("pointnet2_batch_cuda: tensors must be CUDA tensors (CPU not supported)")
        if not t.is_contiguous():
            raise RuntimeError("pointnet2_batch_cuda: tensors must be contiguous")


def _stream(t):
    return _lib.current_stream(t.device)


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    _gpu(new_xyz, xyz, idx)
    with _lib.device_ctx(xyz.device):
        _lib.check(_lib.lib().gb_ball_query(_lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(idx), None, b, n, m,
                                            float(radius), int(nsample), _stream(xyz)), "ball_query_wrapper")
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    _gpu(points, idx, out)
    with _lib.device_ctx(points.device):
        _lib.check(_lib.lib().gb_group(_lib.ptr(points), _lib.ptr(idx), _lib.ptr(out), b, c, n, npoints, nsample,
                                       _stream(points)), "group_points_wrapper")
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    _gpu(grad_out, idx, grad_points)
    with _lib.device_ctx(grad_out.device):
        _lib.check(_lib.lib().gb_group_grad(_lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(grad_points), b, c, n,
                                            npoints, nsample, _stream(grad_out)), "group_points_grad_wrapper")
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    _gpu(points, idx, out)
    with _lib.device_ctx(points.device):
        _lib.check(_lib.lib().gb_gather(_lib.ptr(points), _lib.ptr(idx), _lib.ptr(out), b, c, n, npoints,
                                        _stream(points)), "gather_points_wrapper")
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    _gpu(grad_out, idx, grad_points)
    with _lib.device_ctx(grad_out.device):
        _lib.check(_lib.lib().gb_gather_grad(_lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(grad_points), b, c, n,
                                             npoints, _stream(grad_out)), "gather_points_grad_wrapper")
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    _gpu(points, temp, idx)
    with _lib.device_ctx(points.device):
        _lib.check(_lib.fps(points, temp, idx, b, n, m, FPS_FLAGS, _stream(points)), "furthest_point_sampling_wrapper")
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    _gpu(unknown, known, dist2, idx)
    with _lib.device_ctx(unknown.device):
        _lib.check(_lib.lib().gb_three_nn(_lib.ptr(unknown), _lib.ptr(known), _lib.ptr(dist2), _lib.ptr(idx), b, n, m,
                                          _stream(unknown)), "three_nn_wrapper")


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _gpu(points, idx, weight, out)
    with _lib.device_ctx(points.device):
        _lib.check(_lib.lib().gb_three_interpolate(_lib.ptr(points), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(out),
                                                   b, c, m, n, _stream(points)), "three_interpolate_wrapper")


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _gpu(grad_out, idx, weight, grad_points)
    with _lib.device_ctx(grad_out.device):
        _lib.check(_lib.lib().gb_three_interpolate_grad(_lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(weight),
                                                        _lib.ptr(grad_points), b, c, n, m, _stream(grad_out)),
                   "three_interpolate_grad_wrapper")
