"""ctypes loader for the CPU oracle (oracle/graspbal_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``graspbalance_amd/`` imports this module.

All functions take / return **CPU** torch tensors (fp32 data, int32 indices) and mirror the
allocate-and-return surface of the reference's ``pointnet2._ext`` (PointNet/_ext_src/src/
bindings.cpp:12-26) so parity tests read like calls into the reference extension.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgraspbal_oracle.so")

FPS_SKIP_NEAR_ORIGIN = 0x1
FPS_TIE_LOWEST = 0x00
FPS_TIE_TREE512 = 0x10
FPS_TIE_TREE1024 = 0x20


def build(force=False):
    """Compile the oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "graspbal_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "CC=gcc"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _f32(t):
    assert t.device.type == "cpu" and t.dtype == torch.float32, "oracle takes CPU float32 tensors"
    return t.contiguous()


def _i32(t):
    assert t.device.type == "cpu" and t.dtype == torch.int32, "oracle takes CPU int32 tensors"
    return t.contiguous()


def furthest_point_sampling(xyz, npoint, flags=FPS_SKIP_NEAR_ORIGIN | FPS_TIE_TREE512, temp=None):
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    idx = torch.zeros(B, npoint, dtype=torch.int32)
    rc = lib().gbo_fps(_p(xyz), _p(temp), _p(idx), B, N, npoint, ctypes.c_uint(flags))
    assert rc == 0
    return idx


def gather_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    B, C, N = points.shape
    M = idx.shape[1]
    out = torch.zeros(B, C, M)
    lib().gbo_gather(_p(points), _p(idx), _p(out), B, C, N, M)
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, M = grad_out.shape
    out = torch.zeros(B, C, n)
    lib().gbo_gather_grad(_p(grad_out), _p(idx), _p(out), B, C, n, M)
    return out


def ball_query(new_xyz, xyz, radius, nsample, return_scanned=False):
    new_xyz, xyz = _f32(new_xyz), _f32(xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.zeros(B, M, nsample, dtype=torch.int32)
    scanned = torch.zeros(B, M, dtype=torch.int32) if return_scanned else None
    lib().gbo_ball_query(_p(new_xyz), _p(xyz), _p(idx), _p(scanned), B, N, M,
                         ctypes.c_float(radius), nsample)
    return (idx, scanned) if return_scanned else idx


def cylinder_query(new_xyz, xyz, rot, radius, hmin, hmax, nsample, return_scanned=False):
    new_xyz, xyz, rot = _f32(new_xyz), _f32(xyz), _f32(rot)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.zeros(B, M, nsample, dtype=torch.int32)
    scanned = torch.zeros(B, M, dtype=torch.int32) if return_scanned else None
    lib().gbo_cylinder_query(_p(new_xyz), _p(xyz), _p(rot), _p(idx), _p(scanned), B, N, M,
                             ctypes.c_float(radius), ctypes.c_float(hmin), ctypes.c_float(hmax),
                             nsample)
    return (idx, scanned) if return_scanned else idx


def group_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    B, C, N = points.shape
    _, M, S = idx.shape
    out = torch.zeros(B, C, M, S)
    lib().gbo_group(_p(points), _p(idx), _p(out), B, C, N, M, S)
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, M, S = grad_out.shape
    out = torch.zeros(B, C, n)
    lib().gbo_group_grad(_p(grad_out), _p(idx), _p(out), B, C, n, M, S)
    return out


def three_nn(unknown, known):
    unknown, known = _f32(unknown), _f32(known)
    B, N, _ = unknown.shape
    M = known.shape[1]
    dist2 = torch.zeros(B, N, 3)
    idx = torch.zeros(B, N, 3, dtype=torch.int32)
    lib().gbo_three_nn(_p(unknown), _p(known), _p(dist2), _p(idx), B, N, M)
    return dist2, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    B, C, M = points.shape
    N = idx.shape[1]
    out = torch.zeros(B, C, N)
    lib().gbo_three_interpolate(_p(points), _p(idx), _p(weight), _p(out), B, C, M, N)
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    B, C, N = grad_out.shape
    out = torch.zeros(B, C, m)
    lib().gbo_three_interpolate_grad(_p(grad_out), _p(idx), _p(weight), _p(out), B, C, N, m)
    return out


def knn(ref, query, k):
    """ref (B,dim,nref), query (B,dim,nq) -> (B,k,nq) int64, 1-based, nearest first, equal distances by index."""
    ref, query = _f32(ref), _f32(query)
    B, D, NR = ref.shape
    NQ = query.shape[2]
    assert 1 <= k <= min(NR, 64)
    idx = torch.zeros(B, k, NQ, dtype=torch.int64)
    assert lib().gbo_knn(_p(ref), _p(query), _p(idx), B, D, NR, NQ, k) == 0
    return idx


def knn1(ref, query):
    """ref (B,dim,nref), query (B,dim,nq) -> (B,1,nq) int64, 1-based (KNN/knn_modules.py:11-18)."""
    ref, query = _f32(ref), _f32(query)
    B, D, NR = ref.shape
    NQ = query.shape[2]
    idx = torch.zeros(B, 1, NQ, dtype=torch.int64)
    lib().gbo_knn1(_p(ref), _p(query), _p(idx), B, D, NR, NQ)
    return idx


class ExtBackend:
    """Object with the 10-function surface of ``pointnet2._ext`` backed by the oracle.

    Tests assign it in place of the HIP-backed ``_ext`` to exercise the python composition
    (QueryAndGroup, SA / FP modules, whole models) on CPU.  FPS uses the PN-ext flags.
    """

    fps_flags = FPS_SKIP_NEAR_ORIGIN | FPS_TIE_TREE512

    def furthest_point_sampling(self, xyz, npoint):
        return furthest_point_sampling(xyz, npoint, self.fps_flags)

    gather_points = staticmethod(gather_points)
    gather_points_grad = staticmethod(gather_points_grad)
    ball_query = staticmethod(lambda new_xyz, xyz, radius, nsample: ball_query(new_xyz, xyz, radius, nsample))
    cylinder_query = staticmethod(
        lambda new_xyz, xyz, rot, radius, hmin, hmax, nsample:
        cylinder_query(new_xyz, xyz, rot, radius, hmin, hmax, nsample))
    group_points = staticmethod(group_points)
    group_points_grad = staticmethod(group_points_grad)
    three_nn = staticmethod(lambda u, k: list(three_nn(u, k)))
    three_interpolate = staticmethod(three_interpolate)
    three_interpolate_grad = staticmethod(three_interpolate_grad)


class PBBackend:
    """The 9-function surface of ``pointnet2_batch_cuda`` (caller-allocated outputs) on the oracle."""

    fps_flags = FPS_TIE_TREE1024

    @staticmethod
    def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
        idx.copy_(ball_query(new_xyz, xyz, radius, nsample))
        return 1

    @staticmethod
    def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
        out.copy_(group_points(points, idx))
        return 1

    @staticmethod
    def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
        grad_points.add_(group_points_grad(grad_out, idx, n))
        return 1

    @staticmethod
    def gather_points_wrapper(b, c, n, npoints, points, idx, out):
        out.copy_(gather_points(points, idx))
        return 1

    @staticmethod
    def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
        grad_points.add_(gather_points_grad(grad_out, idx, n))
        return 1

    def furthest_point_sampling_wrapper(self, b, n, m, points, temp, idx):
        idx.copy_(furthest_point_sampling(points, m, self.fps_flags, temp=temp))
        return 1

    @staticmethod
    def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
        d, i = three_nn(unknown, known)
        dist2.copy_(d)
        idx.copy_(i)

    @staticmethod
    def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
        out.copy_(three_interpolate(points, idx, weight))

    @staticmethod
    def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
        grad_points.add_(three_interpolate_grad(grad_out, idx, weight, m))


def knn_into(ref, query, idx):
    """KNN._C.knn signature (writes 1-based int64 indices into idx)."""
    idx.copy_(knn1(ref, query) if idx.shape[1] == 1 else knn(ref, query, idx.shape[1]))
    return 1
