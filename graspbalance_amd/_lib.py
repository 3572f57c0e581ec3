"""ctypes binding of libgraspbal_hip.so (C-ABI: include/graspbal.h).

The library is hand-written HIP for gfx950 built in-tree by ``graspbalance_amd/csrc/Makefile``
(``__graft_entry__.build()``).  There is NO fallback: if the shared object is missing or a symbol
is absent, importing an op raises — the product never routes through a CPU path.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libgraspbal_hip.so")
ABI_VERSION = 7

GB_OK = 0
_ERRNAMES = {-1: "GB_EINVAL", -2: "GB_ELAUNCH", -3: "GB_ERANGE"}

FPS_SKIP_NEAR_ORIGIN = 0x1
FPS_TIE_LOWEST = 0x00
FPS_TIE_TREE512 = 0x10
FPS_TIE_TREE1024 = 0x20
# gb_fps_pruned: how a register-resident cloud is spread over the waves of its CU (graspbal.h GB_FPS_LAYOUT_*)
FPS_LAYOUT = {"auto": 0x000, "w4": 0x100, "w8": 0x200, "w12": 0x300, "w16": 0x400, "r4": 0x500}

_c = ctypes
_P, _I, _F, _U, _L = _c.c_void_p, _c.c_int, _c.c_float, _c.c_uint, _c.c_longlong

# name -> argtypes; every function returns int.  Kept in one table so tests can check that the
# shared object exports exactly what include/graspbal.h declares.
SIGNATURES = {
    "gb_fps": [_P, _P, _P, _I, _I, _I, _U, _P],
    "gb_fps_pruned": [_P, _P, _P, _P, _I, _I, _I, _U, _P, _P],
    "gb_fps_segments": [_P, _P, _P, _P, _P, _I, _I, _U, _P],
    "gb_fps_cell_order": [_P, _P, _I, _I, _P],
    "gb_fps_row_order": [_P, _P, _I, _I, _P],
    "gb_fps_row_order_ws": [_P, _P, _P, _I, _I, _P],
    "gb_fps_morton_keys": [_P, _P, _I, _I, _P],
    "gb_fps_guarded": [_P, _P, _P, _I, _I, _I, _U, _P, _P, _P, _P],
    "gb_gather": [_P, _P, _P, _I, _I, _I, _I, _P],
    "gb_gather_grad": [_P, _P, _P, _I, _I, _I, _I, _P],
    "gb_ball_query": [_P, _P, _P, _P, _I, _I, _I, _F, _I, _P],
    "gb_cylinder_query": [_P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _I, _P],
    "gb_cylinder_query_multi": [_P, _P, _P, _P, _I, _I, _I, _P, _I, _F, _P, _I, _I, _P],
    "gb_group": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gb_group_grad": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gb_three_nn": [_P, _P, _P, _P, _I, _I, _I, _P],
    "gb_interp_weights": [_P, _P, _L, _P],
    "gb_three_interpolate": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gb_three_interpolate_grad": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gb_knn1": [_P, _P, _P, _I, _I, _I, _I, _P],
    "gb_knn": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gb_group_concat_cl": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "gb_group_concat_cl_grad": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gb_copy_segments": [_P, _I, _P, _P],
    "gb_interp_concat_cl": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gb_interp_concat_cl_grad": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gb_col_stats": [_P, _L, _I, _P, _P, _P],
    "gb_split_col_stats": [_P, _I, _P, _L, _I, _P, _P, _P],
    "gb_split_bn_bwd_stats": [_P, _I, _P, _P, _P, _L, _I, _P, _P],
    "gb_bn_finalize": [_P, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _I, _P],
    "gb_affine_act": [_P, _P, _P, _P, _L, _I, _I, _P],
    "gb_affine_relu_maxpool": [_P, _P, _P, _P, _L, _I, _I, _P],
    "gb_bn_bwd_stats": [_P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P],
    "gb_bn_bwd_apply": [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P],
    "gb_bn_bwd_apply_g": [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P],
    "gb_bn_bwd_stats_pool": [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P],
    "gb_bn_bwd_apply_pool": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P],
    "gb_bn_bwd_reduce": [_P, _I, _I, _P, _P, _P, _P],
    "gb_la_point_stats": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "gb_la_split_w": [_P, _P, _P, _I, _I, _P],
    "gb_la_join_w": [_P, _P, _P, _I, _I, _P],
    "gb_la_col_stats": [_P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P],
    "gb_la_pool": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "gb_la_pool_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "gb_la_pool_bwd_perm": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "gb_la_point_grad": [_P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _P, _P],
    "gb_la_wx_grad": [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P],
    "gb_la_wx_grad_g": [_P, _I, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P],
    "gb_la_wx_grad_gs": [_P, _I, _P, _P, _P, _P, _L, _I, _I, _P, _I, _P, _P, _P],
    "gb_frame_cloud": [_P, _I, _P, _P, _P, _I, _I, _P, _P, _P],
    "gb_frame_mask": [_P, _I, _P, _P, _I, _I, _P, _c.c_double, _P, _P, _P],
    "gb_frame_compact": [_P, _I, _P, _P, _I, _I, _P, _c.c_double, _P, _P, _P],
    "gb_voxel_mean": [_P, _P, _P, _L, _P],
    "gb_collision_counts": [_P, _P, _P, _P, _P, _I, _L, _P],
    "gb_label_gather": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gb_label_scores": [_P, _P, _I, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _P],
    "gb_label_gather_view": [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "gb_label_gather_dt": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gb_label_scores_dt": [_P, _P, _I, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _P],
    "gb_label_gather_view_dt": [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "gb_grasp_loss_fwd": [_P] * 17 + [_I] * 6 + [_F] * 4 + [_P] * 6,
    "gb_grasp_loss_bwd": [_P] * 17 + [_I] * 6 + [_F] * 4 + [_P] * 11,
    "gb_label_finish": [_P, _P, _P, _P, _F, _P, _P, _P, _L, _I, _P],
    "gb_gemm_fwd": [_P, _P, _P, _P, _P, _I, _L, _I, _I, _P, _P, _P],
    "gb_gemm_dgrad": [_P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _P, _P, _P, _P, _P],
    "gb_gemm_wgrad": [_P, _P, _P, _P, _L, _I, _I, _P, _P],
    "gb_gemm_dgrad_wgrad": [_P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "gb_gemm_wgrad_group": [_P, _I, _P, _P],
    "gb_gemm_wgrad_groups": [_L, _I, _I, _I, _I, _U],
    "gb_gemm_uses_rs": [_L, _I, _I, _I, _I, _I],
    "gb_gemm_kernel_for": [_I, _L, _I, _I, _I, _I],
    "gb_gemm_kernel_for2": [_I, _L, _I, _I, _I, _I, _I, _I, _U],
    "gb_gemm_dgrad_first": [_P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _P, _P],
    "gb_moments3": [_P, _P, _L, _P, _P, _P],
    "gb_cyl_unique": [_P, _I, _I, _L, _I, _P, _P, _P, _P],
    "gb_cyl_scan": [_P, _I, _L, _P, _P, _P],
    "gb_cyl_rows": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _L, _P, _P, _P, _P, _P, _P],
    "gb_gemm_fwd_pool": [_P, _P, _P, _P, _P, _P, _L, _L, _P, _P, _I, _L, _I, _I, _I, _P, _P, _P],
    "gb_bn_bwd_apply_members_v": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _L, _I, _P, _P],
    "gb_pool_pairs": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "gb_bn_finalize_lin3": [_P, _P, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P],
    "gb_gemm_fwd_gen3": [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _P, _P, _P],
    "gb_gemm_wgrad_gen3": [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P],
    "gb_gemm_dgrad_first_gen3": [_P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _P, _P],
    "gb_gemm_fwd_w": [_P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _P, _P, _P],
    "gb_affine_relu_maxpool_members": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "gb_bn_bwd_apply_members": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _L, _I, _P, _P],
    "gb_bn_bwd_apply_w": [_P, _P, _P, _P, _P, _L, _L, _I, _I, _P, _P, _P],
}


def build(verbose=False):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return SO_PATH


_lib = None


def lib():
    """The loaded shared object; raises (never falls back) when it is missing or stale."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                "graspbalance_amd: %s not found - build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
        handle = ctypes.CDLL(SO_PATH)
        handle.gb_abi_version.restype = _I
        if handle.gb_abi_version() != ABI_VERSION:
            raise ImportError("graspbalance_amd: %s has ABI %d, expected %d - rebuild"
                              % (SO_PATH, handle.gb_abi_version(), ABI_VERSION))
        handle.gb_last_error.restype = _c.c_char_p
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = _I
        _lib = handle
    return _lib


class BnFinalize(_c.Structure):
    """GbBnFinalize of include/graspbal.h: lets a statistics-producing entry point finish the BatchNorm layer."""
    _fields_ = [("gamma", _c.c_void_p), ("beta", _c.c_void_p), ("running_mean", _c.c_void_p),
                ("running_var", _c.c_void_p), ("ab", _c.c_void_p), ("P", _c.c_longlong), ("eps", _c.c_float),
                ("momentum", _c.c_float), ("training", _c.c_int)]


class GemmOpts(_c.Structure):
    """GbGemmOpts of include/graspbal.h: the per-call options of the gb_gemm_* entry points (precision, CUs left to a
    side-stream kernel, caller-owned split-reduction workspace)."""
    _fields_ = [("precision", _c.c_int), ("reserved_cus", _c.c_int), ("scratch", _c.c_void_p),
                ("scratch_bytes", _c.c_ulonglong), ("rows_dev", _c.c_void_p), ("flags", _c.c_int)]


class WgradItem(_c.Structure):
    """GbWgradItem: one weight gradient of a gb_gemm_wgrad_group call."""
    _fields_ = [("dy", _c.c_void_p), ("x", _c.c_void_p), ("x_aff", _c.c_void_p), ("dw", _c.c_void_p),
                ("P", _c.c_longlong), ("K", _c.c_int), ("N", _c.c_int), ("ldw", _c.c_int)]


PREC_F32, PREC_BF16, PREC_F32_SPLIT3 = 0, 1, 2
GEMM_NO_RING = 1   # GbGemmOpts.flags
GEMM_NO_PAIR = 2
GEMM_NO_DIRECT = 4
GEMM_SCRATCH_BYTES = 320 * 64 * 128 * 4


class KernelTimer:
    """Optional HIP-event timing of C-ABI launches (bench.py's roofline leg).  While active, shims
    that call ``timed(name, fn)`` bracket the launch with events on the launch stream."""

    active = None

    def __init__(self, names, reserve=0):
        """reserve: events created up front (creating them inside the timed region is host time the step pays)."""
        self.names = set(names)
        self.all_names = frozenset(names)
        self.events = {n: [] for n in names}
        self.kept = set()
        self._pool = []
        if reserve:
            import torch
            self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(reserve)]

    def _event(self):
        if self._pool:
            return self._pool.pop()
        import torch
        return torch.cuda.Event(enable_timing=True)

    def sample(self, on):
        """Switch the event bracketing on / off between steps of a timed region: an event pair costs the stream a couple
        of microseconds per launch (~1 ms per train step for 270 launches), so bench.py brackets the launches of every
        4th timed step only and the other steps run as they would without a timer."""
        self.names = set(self.all_names) if on else set()

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def summary(self):
        """name -> (launches, mean ms) ; call after torch.cuda.synchronize()."""
        out = {}
        for n, evs in self.events.items():
            if evs:
                ms = [a.elapsed_time(b) for a, b, _ in evs]
                out[n] = {"launches": len(ms), "mean_ms": sum(ms) / len(ms), "meta": [m for _, _, m in evs][0]}
        return out


def timed(name, device, meta, launch):
    """Run ``launch()`` (a C-ABI call returning rc); time it when a KernelTimer wants `name`."""
    t = KernelTimer.active
    if t is None or name not in t.names:
        return launch()
    import torch
    s = torch.cuda.current_stream(device)
    a, b = t._event(), t._event()
    a.record(s)
    rc = launch()
    b.record(s)
    if meta is not None and "args" in meta:
        # tensor references (for bench.py's re-run of the query) are kept for the FIRST timed launch of a shape only:
        # holding every launch's inputs for the whole timed region pins activations and changes what the caching
        # allocator does under the measured steps
        key = (name, meta.get("n"), meta.get("m"), meta.get("ns"))
        if key in t.kept:
            meta = {k: v for k, v in meta.items() if k != "args"}
        else:
            t.kept.add(key)
    t.events[name].append((a, b, meta))
    return rc


def event_pair_overhead_ms(device, pairs=64):
    """Median elapsed time of an EMPTY start/stop event pair on `device`'s current stream: the bias a HIP-event
    bracket adds to a kernel's duration (a few microseconds - visible on 50 us kernels).  bench.py subtracts it."""
    import torch
    s = torch.cuda.current_stream(device)
    evs = []
    for _ in range(pairs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        b.record(s)
        evs.append((a, b))
    torch.cuda.synchronize(device)
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2]


FPS_PRUNE_MIN_N, FPS_PRUNE_MAX_N, FPS_PRUNE_MIN_M = 8192, 65536, 128  # gb_fps_pruned: n <= 1024 rows of 64
_fps_prune = os.environ.get("GB_FPS_PRUNE", "1") != "0"  # A/B switch
# A/B switch, the visiting order: "rows" (round 5: two levels of equal-count splits, rows are compact), "cell" (round 4:
# counting sort by 32^3 grid cell), "morton" (30-bit Morton keys + a device sort)
_fps_order = os.environ.get("GB_FPS_ORDER", "rows")
_fps_layout = FPS_LAYOUT[os.environ.get("GB_FPS_LAYOUT", "auto")]   # A/B switch: see FPS_LAYOUT
FPS_PREFIX_MAX_N = 4096
_fps_prefix = os.environ.get("GB_FPS_PREFIX", "1") != "0"  # A/B switch


def fps(points, temp, output, b, n, m, flags, stream):
    """gb_fps, or for large clouds its pruned form on a spatially coherent visiting order (identical outputs)."""
    import torch
    if _fps_prune and FPS_PRUNE_MIN_N <= n <= FPS_PRUNE_MAX_N and m >= FPS_PRUNE_MIN_M:
        perm = torch.empty((b, n), dtype=torch.int32, device=points.device)
        scratch = torch.empty((b, n, 4), dtype=torch.float32, device=points.device) if n > 20480 else None
        if _fps_order == "rows":   # one launch (the level-1 order of a large cloud goes through the sampling's own scratch)
            rc = lib().gb_fps_row_order_ws(ptr(points), ptr(perm), ptr(scratch), b, n, stream)
            if rc != GB_OK:
                return rc
        elif _fps_order == "cell":
            rc = lib().gb_fps_cell_order(ptr(points), ptr(perm), b, n, stream)
            if rc != GB_OK:
                return rc
        else:                # full 30-bit Morton sort (keys kernel + torch sort)
            keys = torch.empty((b, n), dtype=torch.int32, device=points.device)
            rc = lib().gb_fps_morton_keys(ptr(points), ptr(keys), b, n, stream)
            if rc != GB_OK:
                return rc
            perm = torch.argsort(keys, dim=1).to(torch.int32)
        return lib().gb_fps_pruned(ptr(points), ptr(perm), ptr(temp), ptr(output), b, n, m, flags | _fps_layout,
                                   ptr(scratch), stream)
    if _fps_prefix and n <= FPS_PREFIX_MAX_N and 64 <= m <= n:
        # small clouds are usually the centres of the previous level, i.e. already in farthest-point order: verify
        # "samples = 0..m-1" in parallel and skip the sequential loop where it holds (identical outputs either way)
        ws = torch.empty(b * (m + n), dtype=torch.float32, device=points.device)
        ok = torch.empty(b, dtype=torch.int32, device=points.device)
        return lib().gb_fps_guarded(ptr(points), ptr(temp), ptr(output), b, n, m, flags, ptr(ws),
                                    _c.c_void_p(ws.data_ptr() + 4 * b * m), ptr(ok), stream)
    return lib().gb_fps(ptr(points), ptr(temp), ptr(output), b, n, m, flags, stream)


def check(rc, what):
    if rc != GB_OK:
        msg = lib().gb_last_error().decode() if rc == -2 else ""
        raise RuntimeError("%s failed: %s %s" % (what, _ERRNAMES.get(rc, rc), msg))


def ptr(t):
    """Address of a tensor for a c_void_p parameter (every entry point has argtypes: a plain int / None converts
    without building a ctypes object - this runs ~5000 times per train step)."""
    return t.data_ptr() if t is not None else None


class _NoContext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CONTEXT = _NoContext()


def device_ctx(device):
    """``torch.cuda.device(device)`` only when `device` is not already current (the guard costs ~4 us per launch)."""
    import torch
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_CONTEXT
    return torch.cuda.device(device)


def current_stream(device):
    import torch
    return _c.c_void_p(torch.cuda.current_stream(device).cuda_stream)
