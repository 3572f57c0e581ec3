"""Channel-last fused execution of the SharedMLP stacks (1x1 conv + BatchNorm + ReLU [+ max over
nsample]) that the reference runs as separate torch passes over (B,C,m,ns) tensors
(pytorch_utils.py:5-182, pointnet2_modules.py:148-188, drp.py:62-117, modules.py:104-124).

Activations are rows ``act[p, c]`` with ``p = (b*m + j)*nsample + k``.  One layer =
GEMM (P,Cin)x(Cin,Cout) -> column statistics -> (affine + ReLU [+ max over ns]) in one pass;
backward = two column reductions + one apply pass + two GEMMs.  The element-wise / reduction
passes are the HIP kernels of csrc/mlp_cl.hip; the GEMMs are ``torch.mm`` here (rocBLAS fp32 MFMA).
The modules keep the reference's parameters (conv weight (Cout,Cin,1,1), BatchNorm affine + running
statistics), so state_dicts are unchanged; results equal the unfused path to fp32 rounding.

``enabled()`` is False on CPU tensors: the python layers then run their plain torch composition
(which only works with an extension bound for CPU, i.e. in the tests).
"""
import ctypes
import weakref
import os
import threading
import time

import torch
from torch.autograd import Function

from . import _lib

_ENABLED = True
STAT_SLOTS = 32   # rows of the BN-statistics buffer the GEMM epilogue spreads its fp64 atomics over
_DGRAD_BN_MAX = int(os.environ.get("GB_DGRAD_BN_MAX", 1 << 40))  # rows*cols above which dgrad's BN-backward sums run as a separate pass (A/B switch)
_OWN_GEMM = True  # hand-written MFMA GEMMs (csrc/gemm_cl.hip); False = torch.mm (rocBLAS) for A/B timing


_FIRST_FUSE = os.environ.get("GB_FIRST_FUSE", "1") != "0"  # A/B switch: closed-form backward of xyz-only first layers
_FIRST_FOLD = os.environ.get("GB_FIRST_FOLD", "1") != "0"  # A/B switch: xyz-only first layers never stored (gen3 kernels)


def set_first_fold(flag):
    global _FIRST_FOLD
    prev, _FIRST_FOLD = _FIRST_FOLD, bool(flag)
    return prev
_LOCAL_AGG = os.environ.get("GB_LOCAL_AGG", "1") != "0"  # A/B switch: 0 = grouped tensor + GEMM for LocalAggregation


# Debug / test facility: a callable (kind, **tensors) that every fused node hands its discrete routing to right after its
# forward - the pre-BatchNorm outputs and (a, b) tables that decide the ReLU masks, the arg-max rows of the pooling - so a
# test can replay exactly those decisions in another implementation (tests/routing_tape.py).  None: no cost.
routing_observer = None


def local_agg_enabled():
    return _LOCAL_AGG


def set_local_agg(flag):
    global _LOCAL_AGG
    _LOCAL_AGG = bool(flag)


def set_enabled(flag):
    global _ENABLED
    _ENABLED = bool(flag)


# ---- per-call GEMM options (include/graspbal.h GbGemmOpts) ------------------------------------------------------------
# The library keeps no state: precision, the CUs left to a side-stream kernel and the split-reduction workspace travel
# with every gb_gemm_* call.  The POLICY lives here, on the caller's side:
#   * precision: a per-thread setting (`precision("bf16")` context / `set_precision`), read when a fused node runs its
#     FORWARD and stored in the autograd context - backward runs on autograd's worker thread, which must use the
#     forward's precision, not its own thread's default.  Two trainers / two threads can hold different settings.
#   * reserved CUs: set by prefetch.SamplingPrefetch while a side-stream sampling is in flight, 0 otherwise.
#   * workspace: one GEMM_SCRATCH_BYTES tensor per (device, stream) from torch's caching allocator, allocated on first
#     use and never resized; only kernels on that stream touch it, so the calls of a stream share it.
# "f32" is the fp32 mode this package runs by default: GB_PREC_F32_SPLIT3 (the tall row-streaming products as three-way
# exact bf16 splits on the matrix cores - fp32 MFMA's error against fp64, twice its inner-loop rate; every other product
# fp32 MFMA) unless GB_SPLIT3=0 (A/B switch), when it is GB_PREC_F32.  "f32_mfma" / "f32_split3" name the two explicitly.
_SPLIT3 = os.environ.get("GB_SPLIT3", "1") != "0"
_F32 = _lib.PREC_F32_SPLIT3 if _SPLIT3 else _lib.PREC_F32
_PREC_CODE = {"f32": _F32, "fp32": _F32, "float32": _F32, "bf16": _lib.PREC_BF16, "bfloat16": _lib.PREC_BF16,
              "f32_mfma": _lib.PREC_F32, "f32_split3": _lib.PREC_F32_SPLIT3}
_PREC_NAME = {_lib.PREC_F32: "f32_mfma", _lib.PREC_BF16: "bf16", _lib.PREC_F32_SPLIT3: "f32_split3"}
_PREC_NAME[_F32] = "f32"
_tls = threading.local()
_RESERVED_CUS = 0
_GEMM_FLAGS = _lib.GEMM_NO_RING if os.environ.get("GB_RING", "1") == "0" else 0   # A/B switch: few-row GEMM kernel
if os.environ.get("GB_DIRECT", "1") == "0":   # A/B switch: tall wgrads on the LDS tiles instead of csrc/gemm_wg.hip
    _GEMM_FLAGS |= _lib.GEMM_NO_DIRECT
_LA_AGG_MIN_M = int(os.environ.get("GB_LA_AGG_MIN_M", "512"))   # smaller stages: the order costs what it saves
_LA_AGG = os.environ.get("GB_LA_AGG", "1") != "0"   # A/B switch: la_pool_bwd's scatter pre-aggregated in LDS over spatially adjacent rows
_PAIR = os.environ.get("GB_PAIR", "1") != "0"   # A/B switch: dgrad + wgrad of a layer through gb_gemm_dgrad_wgrad
_WORKSPACES = {}
_OPTS = {}


def _prec_code(precision):
    return _PREC_CODE[str(precision).replace("torch.", "")]


def set_precision(precision):
    """'f32' (default, see _F32 above) or 'bf16' (BASELINE configs[4], "mixed bf16 MLP / fp32 geometry"): every GEMM
    of the fused SharedMLP path rounds its operands to bf16 on the way into the matrix cores and accumulates in fp32;
    BatchNorm statistics, element-wise passes, geometry and all tensors in HBM stay fp32.  'f32_mfma' / 'f32_split3' name
    the two fp32 modes explicitly (include/graspbal.h GB_PREC_F32 / GB_PREC_F32_SPLIT3: the tall row-streaming products on the
    bf16 matrix cores as a three-way exact split of both operands, six products, fp32 accumulation - fp32 MFMA's error
    against fp64, not its bits).  Applies to fused nodes whose
    forward runs on THIS thread from now on (their backward follows the forward); returns the previous setting."""
    prev = get_precision()
    _tls.prec = _prec_code(precision)
    return prev


def get_precision():
    return _PREC_NAME[getattr(_tls, "prec", _F32)]


class precision:
    """``with fused_mlp.precision("bf16"): ...`` - set_precision for the duration of a block."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = set_precision(self.name)
        return self

    def __exit__(self, *exc):
        set_precision(self.prev)
        return False


def set_reserved_cus(count):
    """Size the persistent GEMM grids for (CUs - count) compute units while a side-stream kernel holds `count` of them
    (0 restores the full device).  Returns the previous value."""
    global _RESERVED_CUS
    prev, _RESERVED_CUS = _RESERVED_CUS, max(0, min(int(count), 128))
    return prev


def _prec():
    return getattr(_tls, "prec", _F32)


def _opts(dev, st, prec, rows_dev=None):
    """ctypes pointer to the GbGemmOpts of a launch on stream `st` (c_void_p) of `dev` at precision code `prec`;
    rows_dev: one-element int64 device tensor holding the call's actual row count (GbGemmOpts.rows_dev) or None."""
    key = (dev.index, st.value, prec, _RESERVED_CUS, _GEMM_FLAGS)
    o = _OPTS.get(key)
    if o is None:
        wkey = (dev.index, st.value)
        ws = _WORKSPACES.get(wkey)
        if ws is None:
            ws = _WORKSPACES[wkey] = torch.empty(_lib.GEMM_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
        o = _OPTS[key] = (ctypes.pointer(_lib.GemmOpts(prec, _RESERVED_CUS, ws.data_ptr(), ws.numel(), None, _GEMM_FLAGS)), ws)
    if rows_dev is None:
        return o[0]
    # (not cached: the count's address changes from step to step; the struct only has to outlive the synchronous call)
    return ctypes.pointer(_lib.GemmOpts(prec, _RESERVED_CUS, o[1].data_ptr(), o[1].numel(), rows_dev.data_ptr(),
                                        _GEMM_FLAGS))


def set_ring_gemm(flag):
    """Few-row fp32 products on the LDS-DMA ring kernel (csrc/gemm_ring.hip; default) or, False, on the register-staged
    tiles of csrc/gemm_cl.hip (GbGemmOpts.flags = GB_GEMM_NO_RING).  -> previous setting."""
    global _GEMM_FLAGS
    prev = not (_GEMM_FLAGS & _lib.GEMM_NO_RING)
    _GEMM_FLAGS = (_GEMM_FLAGS & ~_lib.GEMM_NO_RING) | (0 if flag else _lib.GEMM_NO_RING)
    return prev


def set_own_gemm(flag):
    global _OWN_GEMM
    _OWN_GEMM = bool(flag)


def enabled(t):
    return _ENABLED and t.is_cuda


def _s(t):
    return _lib.current_stream(t.device)


_FN = {}

# ---- per-step zero arena for the small fp64 reduction buffers --------------------------------------------------
# Every fused stack needs a few zero-initialised fp64 buffers (BatchNorm sums, their backward counterparts, moments):
# ~85 of them per train step, each a separate fill launch.  A training loop that knows its own step boundaries
# (train.Trainer) calls begin_step(): the buffers then come out of one arena that is re-zeroed once per step.  Nothing
# taken from the arena may be kept across begin_step() - true for a forward -> backward -> optimizer step; code that
# keeps several graphs alive simply does not call begin_step() and gets plain torch.zeros.
_ARENA = {}
_ARENA32_FLOATS = int(os.environ.get("GB_ZERO_ARENA_MB", "192")) << 18   # fp32 arena capacity (floats); 0 = plain torch.zeros


def _arena_of(device):
    """The zero arena buffers are served from: this thread's scoped one (scoped_arena) or the device's shared one."""
    scoped = getattr(_tls, "arena", None)
    if scoped is not None and scoped.get("dev") == str(device):
        return scoped
    return _ARENA.get(str(device))


class scoped_arena:
    """``with fused_mlp.scoped_arena(device, floats64, floats32):`` - a PRIVATE pair of zero arenas for the calls inside
    (created on first use, kept on the object).  predict.Predictor runs its forwards in one: an evaluation pass between a
    training loop's backward() and optimizer.step() must not re-zero the shared arena - weight gradients are views of it."""

    def __init__(self, device, floats64=1 << 17, floats32=1 << 23):
        self.device, self.sizes, self.arena = torch.device(device), (int(floats64), int(floats32)), None

    def __enter__(self):
        if self.arena is None:
            self.arena = {"buf": torch.zeros(self.sizes[0], dtype=torch.float64, device=self.device), "off": 0,
                          "buf32": torch.zeros(self.sizes[1], dtype=torch.float32, device=self.device)
                          if self.sizes[1] else None, "off32": 0, "live": False, "dev": str(self.device)}
        self.prev = getattr(_tls, "arena", None)
        _tls.arena = self.arena
        return self

    def __exit__(self, *exc):
        self.arena["live"] = False
        _tls.arena = self.prev
        return False


def begin_step(device):
    """Start a train step on `device`: re-zero what the previous step took from the zero arenas and rewind them.
    Two arenas: fp64 (BatchNorm sums, moments: ~85 small buffers per step) and fp32 (the atomic-add targets of a step:
    weight-gradient arenas, scatter targets, per-point sums - ~45 buffers, ~100 MB: one fill instead of 45).  Inside a
    scoped_arena the scoped buffers are the ones rewound and served."""
    ar = _arena_of(device)
    if ar is None:
        ar = _ARENA[str(device)] = {"buf": torch.zeros(1 << 20, dtype=torch.float64, device=device), "off": 0,
                                     "buf32": torch.zeros(_ARENA32_FLOATS, dtype=torch.float32, device=device)
                                     if _ARENA32_FLOATS else None, "off32": 0, "live": True}
    else:
        # up to the HIGH-WATER mark of all steps so far, not just the previous one's: a step replayed from a HIP graph
        # dirties what it took when it was captured, whatever ran (and moved the offsets) on the host since
        ar["hw"], ar["hw32"] = max(ar.get("hw", 0), ar["off"]), max(ar.get("hw32", 0), ar["off32"])
        if ar["hw"]:
            ar["buf"][:ar["hw"]].zero_()
        if ar["hw32"]:
            ar["buf32"][:ar["hw32"]].zero_()
    ar["off"], ar["off32"], ar["live"] = 0, 0, True


def end_arena(device):
    """Stop serving buffers from the arena (e.g. before code that keeps results across steps)."""
    ar = _arena_of(device)
    if ar is not None:
        ar["live"] = False


def _zeros64(n, dev):
    ar = _arena_of(dev)
    if ar is not None and ar["live"]:
        off = ar["off"]
        end = off + (n + 31) // 32 * 32
        if end <= ar["buf"].numel():
            ar["off"] = end
            return ar["buf"][off:off + n]
    return torch.zeros(n, dtype=torch.float64, device=dev)


def _zeros32(n, dev):
    """n zero floats (64-byte aligned) for this step: from the fp32 arena inside a Trainer step, torch.zeros otherwise.
    Like everything taken from the arenas it must not outlive the step (gradients are consumed by the optimizer - and
    dropped by zero_grad - before the next begin_step)."""
    ar = _arena_of(dev)
    if ar is not None and ar["live"] and ar["buf32"] is not None:
        off = ar["off32"]
        end = off + (n + 15) // 16 * 16
        if end <= ar["buf32"].numel():
            ar["off32"] = end
            return ar["buf32"][off:off + n]
    return torch.zeros(n, dtype=torch.float32, device=dev)


def _size_args(args):
    """The integer SIZE arguments of a launch (tensor addresses are plain ints too: they are far above 2^40)."""
    return tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool) and -(1 << 40) < a < (1 << 40))


def _call(name, dev, *args, meta=None):
    """One C-ABI launch.  Fast path (no timer, tensor on the current device): a cached ctypes function and nothing
    else - this runs ~600 times per train step, and the host has to stay ahead of the GPU."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_lib.lib(), name)
    timer = _lib.KernelTimer.active
    wanted = timer is not None and name in timer.names  # a timer only costs the launches it asked for
    if dev.index is None or dev.index == torch.cuda.current_device():
        if not wanted:
            rc = fn(*args)
            if rc:
                _lib.check(rc, name)
            return
        if meta is None:
            meta = {"ints": _size_args(args)}  # sizes, for tools/kernel_breakdown.py
        _lib.check(_lib.timed(name, dev, meta, lambda: fn(*args)), name)
        return
    with torch.cuda.device(dev):
        if meta is None and wanted:
            meta = {"ints": _size_args(args)}
        _lib.check(_lib.timed(name, dev, meta, lambda: fn(*args)), name)


SYNC_WAIT = [0.0]  # seconds the host spent WAITING for the device inside steps (the device -> host read of cylinder_rows
                   # without static rows; the graph loop's three-steps-ahead throttle): bench.py takes it off host work
_TRAIN_TICK = [0]   # bumped by every training-mode BatchNorm finalise: those write the running statistics through raw
                    # pointers, which torch's version counters do not see
_EVAL_AB = {}   # id(running_mean buffer) -> (weak reference to it, key, [a, b, mean, rstd] table)


def invalidate_eval_tables():
    """Forget the cached eval-mode BatchNorm tables.  The cache follows version counters, addresses and this module's
    own training passes; a write through ``.data`` (broadcasts, EMA / weight averaging, a foreign optimizer) moves none
    of them - such code calls this afterwards.  (data_parallel.broadcast_module, FlatAdam.step / load_state_dict do.)"""
    _TRAIN_TICK[0] += 1


def _eval_ab(gamma, beta, running_mean, running_var, eps, N, dev, st):
    """Eval mode: the layer's [a, b, mean, rstd] table depends on parameters and running statistics only, so it is
    computed once (gb_bn_finalize without batch sums) and reused until one of them changes - an eval forward of the
    network used to spend 73 launches per call on it.  Changes are seen through the tensors' version counters and
    addresses (optimizer steps, load_state_dict, .to()) and through _TRAIN_TICK (this module's own training passes)."""
    key = (gamma.data_ptr(), gamma._version, beta.data_ptr(), beta._version, running_mean.data_ptr(),
           running_mean._version, running_var.data_ptr(), running_var._version, float(eps), _TRAIN_TICK[0])
    slot = id(running_mean)   # (tensors compare element-wise: a dictionary keyed by the tensor itself would not do)
    hit = _EVAL_AB.get(slot)
    if hit is not None and hit[0]() is running_mean and hit[1] == key:
        if hit[3] != st.value:   # built on another stream: order this stream behind the launch that wrote the table
            torch.cuda.current_stream(dev).wait_event(hit[4])
        return hit[2]
    ab = torch.empty(4 * N, dtype=torch.float32, device=dev)
    _call("gb_bn_finalize", dev, None, 1, 1, N, _lib.ptr(gamma), _lib.ptr(beta), float(eps), 0.0, _lib.ptr(running_mean),
          _lib.ptr(running_var), _lib.ptr(ab), 0, st)
    done = torch.cuda.Event()
    done.record(torch.cuda.current_stream(dev))
    _EVAL_AB[slot] = (weakref.ref(running_mean, lambda _, slot=slot: _EVAL_AB.pop(slot, None)), key, ab, st.value, done)
    return ab


def _bn_fin(cfg, gamma, beta, ab, P_stat):
    """ctypes GbBnFinalize of a layer, by reference."""
    _TRAIN_TICK[0] += 1
    f = _lib.BnFinalize(gamma.data_ptr(), beta.data_ptr(),
                        cfg.running_mean.data_ptr() if cfg.running_mean is not None else None,
                        cfg.running_var.data_ptr() if cfg.running_var is not None else None, ab.data_ptr(), P_stat,
                        cfg.eps, cfg.momentum, 1)
    return ctypes.byref(f)


def _gemm_meta(kind, P, K, N, fused=False, aff=False, rows_dev=None, prec=None):
    """Timing metadata of a GEMM launch (only built while a KernelTimer is active): FLOP, shape and which
    kernel the C entry dispatches to (gemm_rs_kernel / gemm_cl_kernel).  rows_dev: the launch's device-side row count -
    read back HERE (a synchronisation: timing runs only) so that the FLOP count is that of the rows actually multiplied,
    not of the capacity P."""
    if _lib.KernelTimer.active is None:
        return None
    cap = P
    if rows_dev is not None:
        P = int(rows_dev)
    # asked with the call's own options: the direct wgrad cuts a product with the skeleton its precision selects, the
    # A/B flags veto kernels (with the default options the answer could name another kernel)
    which = _lib.lib().gb_gemm_kernel_for2({"fwd": 0, "dgrad": 1, "wgrad": 2}[kind], cap, K, N, int(fused), int(aff),
                                           _prec() if prec is None else prec, _RESERVED_CUS, _GEMM_FLAGS)
    if rows_dev is not None and kind != "wgrad":
        which = 1      # a device-side row count: the row-streaming kernel whatever the capacity
    if which == 2 and rows_dev is not None:
        which = 0
    kernel = ("gemm_cl_kernel", "gemm_rs_kernel", "gemm_ring_kernel", "wgrad_smallk_kernel", "wgrad_direct_kernel")[which]
    return {"flop": 2.0 * P * K * N, "pkn": (P, K, N), "kernel": kernel}


def _pair_meta(P, K, N):
    """_gemm_meta of a gb_gemm_dgrad_wgrad call that leaves as ONE launch of the ring kernel (both products)."""
    if _lib.KernelTimer.active is None:
        return None
    return {"flop": 4.0 * P * K * N, "pkn": (P, K, N), "kernel": "gemm_ring_kernel", "products": 2}


def _first_meta(P, K, N, rows_dev):
    """_gemm_meta of the closed-form first-layer dgrad (always the row-streaming kernel)."""
    if _lib.KernelTimer.active is None:
        return None
    if rows_dev is not None:
        P = int(rows_dev)
    return {"flop": 2.0 * P * K * N, "pkn": (P, K, N), "kernel": "gemm_rs_kernel"}


# ---- deferred, grouped weight gradients (round 6; VERDICT round 5 #1 i) ----------------------------------------------
# A weight gradient reads its layer's stored input and the dY the layer's backward has formed; nothing reads IT before the
# optimizer.  The few-row ones (the InvResMLP blocks, the feature-propagation stacks, the heads: ~45 per step) used to be
# launched one at a time in the middle of the backward's dependency chain - 10 - 30 us launches a third of which is fixed
# cost.  Inside a WgradQueue the backward functions below only RECORD them; flush() hands the stream's whole list to
# gb_gemm_wgrad_group, which runs up to 63 of them per grid (csrc/gemm_ring.hip gemm_ring_group_kernel).  The queue keeps the
# operands alive until then.  Keyed by (device, stream): autograd runs the backward functions on its own worker thread,
# so a thread-local would not be seen there.
_WGQ = {}
_WGRAD_GROUP = os.environ.get("GB_WGRAD_GROUP", "1") != "0"   # A/B switch: 0 = every wgrad where its layer's backward runs


class WgradQueue:
    """``with WgradQueue(device): loss.backward()`` - the few-row weight gradients of the backward passes that run on
    this device's CURRENT stream inside the block are recorded and leave together on exit (or at flush()).  The
    gradients autograd hands to the parameters inside the block are views of buffers the grouped launch has not
    written yet: whoever reads them (an optimizer, a gradient all-reduce, a hook) must come after flush() on the same
    stream - train.Trainer flushes before it packs / reduces / steps, and does not defer at all while post-accumulate
    hooks issue collectives from inside the backward.  For the same reason the backward must ASSIGN the gradients, not
    accumulate them: a parameter that already holds a .grad gets `grad += incoming` from autograd at once, i.e. the zeros of a
    buffer not written yet (likewise a weight used twice, whose two gradients the engine adds).  With `params` the
    queue CHECKS this when it flushes: every recorded gradient buffer must BE some parameter's .grad (same address) by
    then - else autograd copied or added zeros - and it raises instead of training on them.  (Gradient accumulation over
    micro-steps therefore runs its backward passes outside a queue.)"""

    def __init__(self, device, params=None):
        self.dev = torch.device(device)
        self.items, self.keep, self.prec, self.key, self.launches = [], [], None, None, 0
        self.params = params

    def __enter__(self):
        idx = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        self.st = _lib.current_stream(self.dev)
        self.key = (idx, self.st.value)
        self.outer = _WGQ.get(self.key)
        _WGQ[self.key] = self
        return self

    def __exit__(self, exc_type, *exc):
        if self.outer is not None:
            _WGQ[self.key] = self.outer
        else:
            _WGQ.pop(self.key, None)
        if exc_type is None:
            self.flush()
        else:
            self.items, self.keep = [], []
        return False

    def push(self, dY, X, aff, dW, P, K, N, ldw, prec):
        if self.prec is not None and prec != self.prec:
            self.flush()
        self.prec = prec
        self.items.append((dY.data_ptr(), X.data_ptr(), aff.data_ptr() if aff is not None else None, dW.data_ptr(), P, K, N, ldw))
        self.keep.append((dY, X, aff, dW))

    def _check_assigned(self):
        """Every recorded dW is (columns of) some parameter's .grad: the address autograd was handed is the address it
        kept.  A dW inside a wider matrix (ldw > K: an aggregation conv's columns 3..) starts ldw - K floats in."""
        held = {p.grad.data_ptr() for p in self.params if p.grad is not None}
        for (_, _, _, dw, P, K, N, ldw) in self.items:
            if dw not in held and dw - 4 * (ldw - K) not in held:
                raise RuntimeError("fused_mlp.WgradQueue: a deferred weight gradient (%d x %d, %d rows) is not the .grad of "
                                   "any parameter - it was accumulated into an existing gradient or summed with another "
                                   "use of its weight BEFORE it was computed; run such a backward outside the queue" % (N, K, P))

    def flush(self):
        if not self.items:
            return
        if self.params is not None:
            self._check_assigned()
        arr = (_lib.WgradItem * len(self.items))(*self.items)
        meta = None
        if _lib.KernelTimer.active is not None:
            meta = {"flop": sum(2.0 * i[4] * i[5] * i[6] for i in self.items), "kernel": "gemm_ring_group_kernel",
                    "products": len(self.items), "pkn_list": [(i[4], i[5], i[6]) for i in self.items],
                    "pkn": (sum(i[4] for i in self.items), 0, 0)}
        _call("gb_gemm_wgrad_group", self.dev, ctypes.cast(arr, ctypes.c_void_p), len(self.items),
              _opts(self.dev, self.st, self.prec), self.st, meta=meta)
        self.launches += 1
        self.items, self.keep = [], []


def _wgrad_call(dev, st, dY, X, aff, dW, P, K, N, prec, opts, ldw=None):
    """dW += dY^T f(X): recorded when a WgradQueue collects this stream's few-row weight gradients, launched here
    otherwise.  ldw (a dW that sits in a wider matrix) is only legal when the caller has checked wgrad_groups()."""
    q = _WGQ.get((dev.index, st.value)) if _WGQ else None
    if q is not None and _lib.lib().gb_gemm_wgrad_groups(P, K, N, prec, _RESERVED_CUS, _GEMM_FLAGS):
        q.push(dY, X, aff, dW, P, K, N, K if ldw is None else ldw, prec)
        return
    assert ldw is None or ldw == K
    _call("gb_gemm_wgrad", dev, _lib.ptr(dY), _lib.ptr(X), _lib.ptr(aff), _lib.ptr(dW), P, K, N, opts, st,
          meta=_gemm_meta("wgrad", P, K, N, aff=aff is not None, prec=prec))


def wgrad_deferred(dev, st, P, K, N, prec):
    """Will _wgrad_call record (rather than launch) this product right now?"""
    return bool(_WGQ and (dev.index, st.value) in _WGQ
                and _lib.lib().gb_gemm_wgrad_groups(P, K, N, prec, _RESERVED_CUS, _GEMM_FLAGS))


def _wgrad(dY, X):
    """dW (Cout,Cin) = dY^T X with the reduction over P rows split into S batched slices: a plain
    (Cout x P) x (P x Cin) GEMM has only Cout*Cin/tile^2 output tiles (4 for 64x64), i.e. 4 busy CUs."""
    P = dY.shape[0]
    S = 1
    while S < 64 and P % (2 * S) == 0 and P // (2 * S) >= 2048:
        S *= 2
    if S == 1:
        return torch.mm(dY.t(), X)
    return torch.bmm(dY.view(S, P // S, -1).transpose(1, 2), X.view(S, P // S, -1)).sum(0)


class GroupConcatCL(Function):
    """(xyz (B,N,3), new_xyz (B,m,3), idx (B,m,ns), feat_cl (B,N,C)|None) -> X0 (B*m*ns, 3+C)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, idx, feat_cl, mode, scale, rot):
        B, N, _ = xyz.shape
        m, ns = idx.shape[1], idx.shape[2]
        C = 0 if feat_cl is None else feat_cl.shape[2]
        out = torch.empty((B * m * ns, 3 + C), dtype=torch.float32, device=xyz.device)
        rot9 = rot.reshape(B, m, 9).contiguous() if rot is not None else None
        _call("gb_group_concat_cl", xyz.device, _lib.ptr(xyz), _lib.ptr(new_xyz), _lib.ptr(idx),
              _lib.ptr(feat_cl), _lib.ptr(rot9), _lib.ptr(out), B, N, m, ns, C, int(mode), float(scale), _s(xyz))
        ctx.dims = (B, N, m, ns, C)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, dx0):
        B, N, m, ns, C = ctx.dims
        dfeat = None
        if C > 0 and ctx.needs_input_grad[3]:
            (idx,) = ctx.saved_tensors
            dx0 = dx0.contiguous()
            dfeat = _zeros32(B * N * C, dx0.device).view(B, N, C)
            _call("gb_group_concat_cl_grad", dx0.device, _lib.ptr(dx0), _lib.ptr(idx), _lib.ptr(dfeat), B, N, m, ns,
                  C, _s(dx0))
        return None, None, None, dfeat, None, None, None


class InterpConcatCL(Function):
    """(known_cl (B,m,C2), idx (B,n,3) int32, weight (B,n,3), skip_cl (B,n,C1)|None) -> X0 (B*n, C2+C1): the rows a feature
    propagation stack starts from (csrc/mlp_cl.hip interp_concat_cl_kernel)."""

    @staticmethod
    def forward(ctx, known_cl, idx, weight, skip_cl):
        B, m, C2 = known_cl.shape
        n = idx.shape[1]
        C1 = 0 if skip_cl is None else skip_cl.shape[2]
        out = torch.empty((B * n, C2 + C1), dtype=torch.float32, device=known_cl.device)
        _call("gb_interp_concat_cl", known_cl.device, _lib.ptr(known_cl), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(skip_cl),
              _lib.ptr(out), B, n, m, C2, C1, _s(known_cl))
        ctx.dims = (B, n, m, C2, C1)
        ctx.save_for_backward(idx, weight)
        return out

    @staticmethod
    def backward(ctx, dx0):
        B, n, m, C2, C1 = ctx.dims
        idx, weight = ctx.saved_tensors
        dx0 = dx0.contiguous()
        dev = dx0.device
        dknown = _zeros32(B * m * C2, dev).view(B, m, C2) if ctx.needs_input_grad[0] else None
        dskip = torch.empty((B, n, C1), dtype=torch.float32, device=dev) if (C1 and ctx.needs_input_grad[3]) else None
        _call("gb_interp_concat_cl_grad", dev, _lib.ptr(dx0), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(dknown),
              _lib.ptr(dskip), B, n, m, C2, C1, _s(dx0))
        return dknown, None, None, dskip


def interp_concat_cl(known_cl, idx, weight, skip_cl=None):
    return InterpConcatCL.apply(known_cl.contiguous(), idx.contiguous(), weight.contiguous(),
                                None if skip_cl is None else skip_cl.contiguous())


def group_concat_cl(xyz, new_xyz, idx, feat_cl=None, mode=0, scale=1.0, rot=None):
    return GroupConcatCL.apply(xyz.contiguous(), new_xyz.contiguous(), idx.contiguous(),
                               None if feat_cl is None else feat_cl.contiguous(), mode, scale, rot)


class LinearBNAct(Function):
    """X (P,Cin) -> act(BN(X W^T)) (P,Cout), or its max over groups of `pool_ns` rows (P/ns,Cout).

    forward args: X, W (Cout,Cin), gamma, beta, residual|None, then non-tensor state:
    running_mean, running_var (updated in place in training), momentum, eps, training, relu, pool_ns.
    """

    @staticmethod
    def forward(ctx, X, W, gamma, beta, residual, running_mean, running_var, momentum, eps, training, relu,
                pool_ns):
        dev = X.device
        P, Cout = X.shape[0], W.shape[0]
        ab = torch.empty(4 * Cout, dtype=torch.float32, device=dev)
        slots = STAT_SLOTS if (_OWN_GEMM and P >= 16384) else 1
        stats = torch.zeros(slots * 2 * Cout, dtype=torch.float64, device=dev) if training else None
        if _OWN_GEMM:
            # hand-written fp32 MFMA GEMM; the BatchNorm column statistics come out of its epilogue
            X = X.contiguous()
            Wc = W.contiguous()
            Y = torch.empty((P, Cout), dtype=torch.float32, device=dev)
            st = _s(X)
            _call("gb_gemm_fwd", dev, _lib.ptr(X), _lib.ptr(Wc), None, _lib.ptr(Y), _lib.ptr(stats), slots, P, X.shape[1],
                  Cout, None, _opts(dev, st, _prec()), st, meta=_gemm_meta("fwd", P, X.shape[1], Cout, stats is not None))
        else:
            Y = torch.mm(X, W.t())
            if training:
                _call("gb_col_stats", dev, _lib.ptr(Y), P, Cout, _lib.ptr(stats), None, _s(Y))
        if training:
            _TRAIN_TICK[0] += 1
            _call("gb_bn_finalize", dev, _lib.ptr(stats), slots, P, Cout, _lib.ptr(gamma), _lib.ptr(beta), float(eps),
                  float(momentum), _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(ab), 1, _s(Y))
        else:
            ab = _eval_ab(gamma, beta, running_mean, running_var, eps, Cout, dev, _s(Y))
        ctx.cfg = (P, Cout, bool(training), bool(relu), int(pool_ns))
        ctx.prec = _prec()
        if pool_ns:
            R = P // pool_ns
            out = torch.empty((R, Cout), dtype=torch.float32, device=dev)
            arg = torch.empty((R, Cout), dtype=torch.int32, device=dev)
            _call("gb_affine_relu_maxpool", dev, _lib.ptr(Y), _lib.ptr(ab), _lib.ptr(out), _lib.ptr(arg), R,
                  pool_ns, Cout, _s(Y))
            ctx.save_for_backward(X, W, Y, ab, out, arg)
            return out
        Z = torch.empty_like(Y)
        _call("gb_affine_act", dev, _lib.ptr(Y), _lib.ptr(ab), _lib.ptr(residual), _lib.ptr(Z), P, Cout, int(relu),
              _s(Y))
        ctx.save_for_backward(X, W, Y, ab, residual)
        return Z

    @staticmethod
    def backward(ctx, dout):
        P, Cout, training, relu, pool_ns = ctx.cfg
        dout = dout.contiguous()
        dev = dout.device
        dstats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
        dres = None
        if pool_ns:
            X, W, Y, ab, out, arg = ctx.saved_tensors
            R = P // pool_ns
            dY = torch.empty_like(Y)
            _call("gb_bn_bwd_stats_pool", dev, _lib.ptr(dout), _lib.ptr(out), _lib.ptr(arg), _lib.ptr(Y),
                  _lib.ptr(ab), R, pool_ns, Cout, _lib.ptr(dstats), None, None, _s(Y))
            _call("gb_bn_bwd_apply_pool", dev, _lib.ptr(dout), _lib.ptr(out), _lib.ptr(arg), _lib.ptr(Y),
                  _lib.ptr(ab), _lib.ptr(dstats), R, pool_ns, Cout, int(training), _lib.ptr(dY), _s(Y))
        else:
            X, W, Y, ab, residual = ctx.saved_tensors
            dY = torch.empty_like(Y)
            if residual is not None and ctx.needs_input_grad[4]:
                dres = torch.empty_like(Y)
            _call("gb_bn_bwd_stats", dev, _lib.ptr(dout), _lib.ptr(Y), _lib.ptr(ab), _lib.ptr(residual), P, Cout,
                  int(relu), _lib.ptr(dstats), None, None, _s(Y))
            _call("gb_bn_bwd_apply", dev, _lib.ptr(dout), _lib.ptr(Y), _lib.ptr(ab), _lib.ptr(residual),
                  _lib.ptr(dstats), P, Cout, int(relu), int(training), _lib.ptr(dY), _lib.ptr(dres), _s(Y))
        dgamma = dbeta = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            # dgamma = sum dA*xhat, dbeta = sum dA — exactly the two column sums of pass 1
            dbeta = dstats[:Cout].float()
            dgamma = dstats[Cout:].float()
        dW = dX = None
        if _OWN_GEMM:
            Cin = X.shape[1]
            if ctx.needs_input_grad[1]:
                dW = torch.zeros((Cout, Cin), dtype=torch.float32, device=dev)
                _call("gb_gemm_wgrad", dev, _lib.ptr(dY), _lib.ptr(X), None, _lib.ptr(dW), P, Cin, Cout,
                      _opts(dev, _s(dY), ctx.prec), _s(dY), meta=_gemm_meta("wgrad", P, Cin, Cout, prec=ctx.prec))
            if ctx.needs_input_grad[0]:
                dX = torch.empty((P, Cin), dtype=torch.float32, device=dev)
                W = W.contiguous()
                _call("gb_gemm_dgrad", dev, _lib.ptr(dY), _lib.ptr(W), _lib.ptr(dX), None, None, None, 0, P, Cin, Cout, None,
                      None, None, _opts(dev, _s(dY), ctx.prec), _s(dY),
                      meta=_gemm_meta("dgrad", P, Cin, Cout))
        else:
            dW = _wgrad(dY, X) if ctx.needs_input_grad[1] else None
            dX = torch.mm(dY, W) if ctx.needs_input_grad[0] else None
        return dX, dW, dgamma, dbeta, dres, None, None, None, None, None, None, None


class _LayerCfg:
    __slots__ = ("running_mean", "running_var", "momentum", "eps", "training")

    def __init__(self, bn):
        self.running_mean, self.running_var = bn.running_mean, bn.running_var
        if bn.momentum is None and bn.track_running_stats:
            # torch's cumulative moving average (factor 1 / num_batches_tracked) is not built into gb_bn_finalize;
            # supports() / InvResMLP.fusable() route such layers to the plain composition instead
            raise NotImplementedError("fused BatchNorm path needs a numeric momentum (momentum=None = cumulative average)")
        self.momentum = 0.0 if bn.momentum is None else float(bn.momentum)
        self.eps = float(bn.eps)
        self.training = bool(bn.training or not bn.track_running_stats)


class MLPStack(Function):
    """A whole stack  X0 -> [conv -> BN -> ReLU] x (L-1) -> conv -> BN -> (ReLU | +residual, ReLU | max over
    pool_ns rows)  as ONE autograd node on the hand-written GEMMs:

    * forward: layer l's GEMM reads the PRE-BatchNorm output of layer l-1 and applies relu(a*y+b) while
      loading (gb_gemm_fwd prologue), and emits its own BN column statistics from its epilogue — the
      normalised activations are never written;
    * backward: wgrad re-forms relu(a*y+b) in its operand loader, dgrad's epilogue accumulates the
      BatchNorm-backward sums (dbeta, dgamma) of layer l-1 while it writes that layer's dZ, so the only
      element-wise pass per layer is dy = a*(dA - dbeta/P - xhat*dgamma/P).

    forward(ctx, X0, residual|None, layers: list[_LayerCfg], pool_ns, relu_last, rows: RowSet|None, W0, g0, b0, ...)
    `rows`: X0's rows are the distinct rows of a batch with duplicates (RowSet: multiplicities, cylinder membership);
    statistics / gradients are those of the full batch and the final max runs per cylinder over its members.
    """

    @staticmethod
    def forward(ctx, X0, residual, layers, pool_ns, relu_last, rows, *params):
        dev = X0.device
        st = _s(X0)  # one stream lookup per call, not one per launch
        prec = _prec()
        # rows given by a device-side count (RowSet.rows_dev: X0 and every activation are sized for the row CAPACITY, the
        # kernels read the count): every launch that depends on the row count takes it
        rdev = rows.rows_dev if rows is not None else None
        opts = _opts(dev, st, prec, rdev)
        L = len(layers)
        P_stat = rows.P_total if rows is not None else X0.shape[0]  # rows of the batch the BatchNorm sums stand for
        X0 = X0.contiguous()
        P = X0.shape[0]
        slots = STAT_SLOTS if P >= 16384 else 1
        widths = [params[3 * l].shape[0] for l in range(L)]
        # one zero-filled fp64 arena for every layer's statistics slots, one fp32 arena for the (a,b,mean,rstd) tables
        stat_off = [0]
        for l, cfg in enumerate(layers):
            stat_off.append(stat_off[-1] + (slots * 2 * widths[l] if cfg.training else 0))
        stat_arena = _zeros64(stat_off[-1], dev) if stat_off[-1] else None
        ab_arena = torch.empty(4 * sum(widths), dtype=torch.float32, device=dev)
        Ws, Ys, abs_ = [], [], []
        src, aff, ab_off, pooled = X0, None, 0, None
        # xyz-only first layer folded into its consumers (gb_gemm_fwd_gen3 & co.): its output is never stored - 256 bytes
        # per row written once and read three times (second layer, its wgrad, the first layer's backward) become 12
        fold = (_FIRST_FOLD and _FIRST_FUSE and X0.shape[1] == 3 and L >= 2 and not ctx.needs_input_grad[0]
                and (ctx.needs_input_grad[6] or not any(ctx.needs_input_grad))   # its backward is the closed form
                and widths[0] % 4 == 0 and widths[1] <= 128
                and (L > 2 or not (rows is not None and rows.key is not None and _CROP_POOL))
                and bool(_lib.lib().gb_gemm_uses_rs(P, widths[0], widths[1], 0, 1, 1))
                and bool(_lib.lib().gb_gemm_uses_rs(P, widths[0], widths[1], 1, 2, 0)))
        if rdev is not None and not (fold and L == 3 and rows.key is not None and _CROP_POOL):
            raise RuntimeError("fused_mlp: a device-side row count needs the folded 3-layer crop stack (crop_static_ok)")
        mom0 = None
        for l, cfg in enumerate(layers):
            W = params[3 * l].contiguous()
            gamma, beta = params[3 * l + 1], params[3 * l + 2]
            K, N = (src.shape[1] if src is not None else widths[l - 1]), W.shape[0]
            stats = stat_arena[stat_off[l]:stat_off[l + 1]] if cfg.training else None
            ab = ab_arena[ab_off:ab_off + 4 * N]
            ab_off += 4 * N
            if fold and l == 0:
                # BatchNorm of the folded layer from the 12 moments of its input rows (fp64): sum y = W s, sum y^2 = W M W^T
                if cfg.training or any(ctx.needs_input_grad):   # (the backward's closed-form weight gradient reads them too)
                    mom0 = _zeros64(12, dev)
                    _call("gb_moments3", dev, _lib.ptr(X0), _lib.ptr(rows.w if rows is not None else None), P,
                          _lib.ptr(mom0), _lib.ptr(rdev), st)
                if cfg.training:
                    _TRAIN_TICK[0] += 1
                    _call("gb_bn_finalize_lin3", dev, _lib.ptr(mom0), _lib.ptr(W), P_stat, N, _lib.ptr(gamma), _lib.ptr(beta),
                          cfg.eps, cfg.momentum, _lib.ptr(cfg.running_mean), _lib.ptr(cfg.running_var), _lib.ptr(ab), st)
                else:
                    ab = _eval_ab(gamma, beta, cfg.running_mean, cfg.running_var, cfg.eps, N, dev, st)
                Ws.append(W); Ys.append(None); abs_.append(ab)
                src, aff = None, ab
                continue
            if fold and l == 1:
                Y = _empty_rows(P, N, dev, rows is not None and rdev is None)
                fin = _bn_fin(cfg, gamma, beta, ab, P_stat) if cfg.training else None
                st_buf, st_slots = (stats, slots) if cfg.training else (_zeros64(2 * N, dev), 1)
                _call("gb_gemm_fwd_gen3", dev, _lib.ptr(X0), _lib.ptr(Ws[0]), _lib.ptr(aff), _lib.ptr(W),
                      _lib.ptr(rows.w16 if rows is not None else None), _lib.ptr(Y), _lib.ptr(st_buf), st_slots, P, K, N,
                      fin, opts, st, meta=_gemm_meta("fwd", P, K, N, True, True, rows_dev=rdev))
                if fin is None:
                    ab = _eval_ab(gamma, beta, cfg.running_mean, cfg.running_var, cfg.eps, N, dev, st)
                Ws.append(W); Ys.append(Y); abs_.append(ab)
                src, aff = Y, ab
                continue
            if (l == L - 1 and rows is not None and rows.key is not None and _CROP_POOL
                    and _lib.lib().gb_gemm_uses_rs(P, K, N, 0, 3, int(aff is not None))):
                # the crop stack's last layer: BatchNorm sums and per-(tile, seed, crop) extrema leave the GEMM,
                # gb_pool_pairs finishes the max over each crop's members (no pass over the layer's output)
                fwd_only = not any(ctx.needs_input_grad)   # inference: the layer's output is not stored at all
                Y = None if fwd_only else _empty_rows(P, N, dev, rdev is None)
                if not cfg.training:
                    ab = _eval_ab(gamma, beta, cfg.running_mean, cfg.running_var, cfg.eps, N, dev, st)
                pooled = _pooled_last_layer(dev, st, opts, src, W, aff, gamma, beta, cfg, ab, stats, slots, rows, P, K, N,
                                            P_stat, Y)
                Ws.append(W); Ys.append(Y); abs_.append(ab)
                break
            Y = _empty_rows(P, N, dev, rows is not None)
            # training: the GEMM call finishes the layer's BatchNorm itself (a second launch from the same C call):
            # one Python -> C transition per layer instead of two
            fin = _bn_fin(cfg, gamma, beta, ab, P_stat) if cfg.training else None
            if rows is not None and stats is not None:
                _call("gb_gemm_fwd_w", dev, _lib.ptr(src), _lib.ptr(W), _lib.ptr(aff), _lib.ptr(rows.w16), _lib.ptr(Y),
                      _lib.ptr(stats), slots, P, K, N, fin, opts, st,
                      meta=_gemm_meta("fwd", P, K, N, True, aff is not None, rows_dev=rdev))
            else:
                _call("gb_gemm_fwd", dev, _lib.ptr(src), _lib.ptr(W), _lib.ptr(aff), _lib.ptr(Y), _lib.ptr(stats), slots,
                      P, K, N, fin, opts, st, meta=_gemm_meta("fwd", P, K, N, stats is not None, aff is not None, rows_dev=rdev))
            if fin is None:
                ab = _eval_ab(gamma, beta, cfg.running_mean, cfg.running_var, cfg.eps, N, dev, st)
            Ws.append(W); Ys.append(Y); abs_.append(ab)
            src, aff = Y, ab  # ab[:2N] = [a, b] is exactly the next GEMM's prologue table
        N = widths[-1]
        ctx.cfg = (L, P, int(pool_ns), bool(relu_last), [c.training for c in layers], residual is not None)
        ctx.rows = rows
        ctx.prec = prec
        ctx.by_value = False
        ctx.fold = fold
        ctx.mom0 = mom0
        if not all(c.training for c in layers):
            # eval-mode layers use cached tables (_eval_ab), not slices of the arena: a backward through this node (rare:
            # gradients in eval mode) reads the arena, so gather them once - and skip it when nothing needs a gradient
            ab_arena = torch.cat(abs_) if any(ctx.needs_input_grad) else ab_arena
        if pooled is not None:
            out, ystar = pooled
            if Ys[-1] is not None:   # the backward finds the arg-max rows by value in the stored output
                ctx.save_for_backward(X0, out, ystar, ab_arena, *Ws, *Ys)
                ctx.by_value = True
                if routing_observer is not None:
                    routing_observer("stack", Ys=_observed_ys(Ys, X0, Ws), abs=abs_, out=out, pool_ns=0, relu_last=True,
                                     rows=rows, arg=_arg_rows_by_value(Ys[-1], ystar, rows))
            return out
        if rows is not None:
            RD = rows.R * rows.D
            out = torch.empty((RD, N), dtype=torch.float32, device=dev)
            arg = torch.empty((RD, N), dtype=torch.int32, device=dev)
            _call("gb_affine_relu_maxpool_members", dev, _lib.ptr(Ys[-1]), _lib.ptr(abs_[-1]), _lib.ptr(rows.mem),
                  _lib.ptr(rows.off), _lib.ptr(rows.cnt), _lib.ptr(out), _lib.ptr(arg), rows.R, rows.D, N, st)
            ctx.save_for_backward(X0, out, arg, ab_arena, *Ws, *Ys)
            if routing_observer is not None:
                routing_observer("stack", Ys=_observed_ys(Ys, X0, Ws), abs=abs_, out=out, arg=arg, pool_ns=0, relu_last=True, rows=rows)
            return out
        if pool_ns:
            R = P // pool_ns
            out = torch.empty((R, N), dtype=torch.float32, device=dev)
            arg = torch.empty((R, N), dtype=torch.int32, device=dev)
            _call("gb_affine_relu_maxpool", dev, _lib.ptr(Ys[-1]), _lib.ptr(abs_[-1]), _lib.ptr(out), _lib.ptr(arg), R,
                  pool_ns, N, st)
            ctx.save_for_backward(X0, out, arg, ab_arena, *Ws, *Ys)
            if routing_observer is not None:
                routing_observer("stack", Ys=_observed_ys(Ys, X0, Ws), abs=abs_, out=out, arg=arg, pool_ns=int(pool_ns), relu_last=True, rows=None)
            return out
        if residual is not None:
            residual = residual.contiguous()
        out = torch.empty((P, N), dtype=torch.float32, device=dev)
        _call("gb_affine_act", dev, _lib.ptr(Ys[-1]), _lib.ptr(abs_[-1]), _lib.ptr(residual), _lib.ptr(out), P, N,
              int(relu_last), st)
        ctx.save_for_backward(X0, residual if residual is not None else X0.new_empty(0), X0.new_empty(0), ab_arena,
                              *Ws, *Ys)
        if routing_observer is not None:
            routing_observer("stack", Ys=_observed_ys(Ys, X0, Ws), abs=abs_, out=out, arg=None, pool_ns=0, relu_last=bool(relu_last), rows=None)
        return out

    @staticmethod
    def backward(ctx, dout):
        L, P, pool_ns, relu_last, training, has_res = ctx.cfg
        rows = ctx.rows
        P_stat = rows.P_total if rows is not None else P
        saved = ctx.saved_tensors
        X0, s1, s2, ab_arena = saved[0], saved[1], saved[2], saved[3]
        Ws, Ys = saved[4:4 + L], saved[4 + L:4 + 2 * L]
        widths = [W.shape[0] for W in Ws]
        abs_, off = [], 0
        for n in widths:
            abs_.append(ab_arena[off:off + 4 * n])
            off += 4 * n
        dev = dout.device
        st = _s(dout)  # one stream lookup per call, not one per launch
        rdev = rows.rows_dev if rows is not None else None   # device-side row count: P is the capacity (see forward)
        opts = _opts(dev, st, ctx.prec, rdev)  # the forward's precision (this is autograd's thread)
        quant = rows is not None and rdev is None
        dout = dout.contiguous()
        slots = STAT_SLOTS if P >= 16384 else 1
        # zero-filled arenas: fp64 BatchNorm-backward sums ([2N] last layer, then per layer l<L-1 the slot rows
        # the dgrad epilogue adds into + their total), fp32 weight gradients (wgrad accumulates with atomics)
        fused = [widths[l] * P <= _DGRAD_BN_MAX for l in range(L - 1)]
        d_off = [2 * widths[-1]]
        for l in range(L - 1):
            d_off.append(d_off[-1] + ((slots + 1) * 2 * widths[l] if fused[l] and slots > 1 else 2 * widths[l]))
        d_arena = _zeros64(d_off[-1], dev)
        need_w = [ctx.needs_input_grad[6 + 3 * l] for l in range(L)]
        kin = [X0.shape[1]] + widths[:-1]
        w_off = [0]
        for l in range(L):
            w_off.append(w_off[-1] + (widths[l] * kin[l] if need_w[l] else 0))
        w_arena = _zeros32(w_off[-1], dev) if w_off[-1] else None
        gb_arena = torch.empty(2 * sum(widths), dtype=torch.float32, device=dev)  # [dbeta, dgamma] per layer
        gb_off = [0]
        for n in widths:
            gb_off.append(gb_off[-1] + 2 * n)

        def bn_grads(l):  # (dbeta, dgamma) of layer l: filled by gb_bn_bwd_reduce, launched from the producer's C call
            n = widths[l]
            return gb_arena[gb_off[l]:gb_off[l] + n], gb_arena[gb_off[l] + n:gb_off[l + 1]]

        def param_grads(l, partial, nslots, total):
            dbeta, dgamma = bn_grads(l)
            _call("gb_bn_bwd_reduce", dev, _lib.ptr(partial), nslots, widths[l], _lib.ptr(total), _lib.ptr(dbeta),
                  _lib.ptr(dgamma), st)
            return dgamma, dbeta

        N = widths[-1]
        dstats = d_arena[:2 * N]
        dres = None
        grads = [None] * (3 * L)
        first = L - 1  # the layer the generic loop below starts at (its dY formed here)
        dbeta, dgamma = bn_grads(L - 1)
        pb, pg = _lib.ptr(dbeta), _lib.ptr(dgamma)
        dY = _empty_rows(P, N, dev, quant)
        if rows is not None and ctx.by_value:
            out, ystar = s1, s2
            RD = rows.R * rows.D
            # the crops' extreme y* are saved: the BatchNorm-backward sums need no gather from the layer's output
            _call("gb_bn_bwd_stats", dev, _lib.ptr(dout), _lib.ptr(ystar), _lib.ptr(abs_[-1]), None, RD, N, 1,
                  _lib.ptr(dstats), pb, pg, st)
            _call("gb_bn_bwd_apply_members_v", dev, _lib.ptr(dout), _lib.ptr(out), _lib.ptr(ystar), _lib.ptr(Ys[-1]),
                  _lib.ptr(abs_[-1]), _lib.ptr(dstats), _lib.ptr(rows.w), _lib.ptr(rows.mem),
                  _lib.ptr(rows.off), _lib.ptr(rows.cnt), rows.R, rows.D, N, P_stat, int(training[-1]), _lib.ptr(dY), st)
        elif rows is not None:
            out, arg = s1, s2
            _call("gb_bn_bwd_stats_pool", dev, _lib.ptr(dout), _lib.ptr(out), _lib.ptr(arg), _lib.ptr(Ys[-1]),
                  _lib.ptr(abs_[-1]), rows.R * rows.D, 0, N, _lib.ptr(dstats), pb, pg, st)  # ns = 0: absolute arg rows
            _call("gb_bn_bwd_apply_members", dev, _lib.ptr(dout), _lib.ptr(out), _lib.ptr(arg), _lib.ptr(Ys[-1]),
                  _lib.ptr(abs_[-1]), _lib.ptr(dstats), _lib.ptr(rows.w), _lib.ptr(rows.off), _lib.ptr(rows.cnt), rows.R,
                  rows.D, N, P_stat, int(training[-1]), _lib.ptr(dY), st)
        elif pool_ns:
            out, arg = s1, s2
            R = P // pool_ns
            _call("gb_bn_bwd_stats_pool", dev, _lib.ptr(dout), _lib.ptr(out), _lib.ptr(arg), _lib.ptr(Ys[-1]),
                  _lib.ptr(abs_[-1]), R, pool_ns, N, _lib.ptr(dstats), pb, pg, st)
            _call("gb_bn_bwd_apply_pool", dev, _lib.ptr(dout), _lib.ptr(out), _lib.ptr(arg), _lib.ptr(Ys[-1]),
                  _lib.ptr(abs_[-1]), _lib.ptr(dstats), R, pool_ns, N, int(training[-1]), _lib.ptr(dY), st)
        else:
            residual = s1 if has_res else None
            if has_res and ctx.needs_input_grad[1]:
                dres = torch.empty((P, N), dtype=torch.float32, device=dev)
            # (the apply pass reads the two sums anyway and emits dbeta / dgamma itself: no gb_bn_bwd_reduce launch)
            _call("gb_bn_bwd_stats", dev, _lib.ptr(dout), _lib.ptr(Ys[-1]), _lib.ptr(abs_[-1]), _lib.ptr(residual), P, N,
                  int(relu_last), _lib.ptr(dstats), None, None, st)
            _call("gb_bn_bwd_apply_g", dev, _lib.ptr(dout), _lib.ptr(Ys[-1]), _lib.ptr(abs_[-1]), _lib.ptr(residual),
                  _lib.ptr(dstats), P, N, int(relu_last), int(training[-1]), _lib.ptr(dY), _lib.ptr(dres), pb, pg, st)
        grads[3 * L - 2], grads[3 * L - 1] = dgamma, dbeta
        dX0 = None
        for l in range(first, -1, -1):
            W = Ws[l]
            N, K = W.shape
            src, aff = (X0, None) if l == 0 else (Ys[l - 1], abs_[l - 1])
            # both gradient products of the layer from ONE C call (few-row shapes: one launch, csrc/gemm_ring.hip pair kernel)
            # (a weight gradient that a WgradQueue records is not paired: it leaves later, with the others)
            defer = need_w[l] and rdev is None and not (ctx.fold and l == 1) and wgrad_deferred(dev, st, P, K, N, ctx.prec)
            pair = (_PAIR and need_w[l] and not defer and l >= 1 and fused[l - 1] and rdev is None and not (ctx.fold and l == 1)
                    and not (l == 1 and _FIRST_FUSE and X0.shape[1] == 3 and not ctx.needs_input_grad[0] and need_w[0]
                             and _lib.lib().gb_gemm_uses_rs(P, K, N, 1, 2, 0))
                    and _lib.lib().gb_gemm_kernel_for2(3, P, K, N, 1, 1, ctx.prec, _RESERVED_CUS, _GEMM_FLAGS) == 2)   # (else two calls: per-kernel timing)
            if need_w[l] and not pair:
                dW = w_arena[w_off[l]:w_off[l + 1]]
                if ctx.fold and l == 1:   # the x operand relu(a*y1 + b) is re-formed from the xyz rows
                    _call("gb_gemm_wgrad_gen3", dev, _lib.ptr(dY), _lib.ptr(X0), _lib.ptr(Ws[0]), _lib.ptr(aff), _lib.ptr(dW), P,
                          K, N, opts, st, meta=_gemm_meta("wgrad", P, K, N, aff=True, rows_dev=rdev, prec=ctx.prec))
                elif rdev is not None:
                    _call("gb_gemm_wgrad", dev, _lib.ptr(dY), _lib.ptr(src), _lib.ptr(aff), _lib.ptr(dW), P, K, N, opts, st,
                          meta=_gemm_meta("wgrad", P, K, N, aff=aff is not None, rows_dev=rdev, prec=ctx.prec))
                else:
                    _wgrad_call(dev, st, dY, src, aff, dW, P, K, N, ctx.prec, opts)
                grads[3 * l] = dW.view(N, K)
            if l == 0:
                if ctx.needs_input_grad[0]:
                    dX0 = torch.empty((P, K), dtype=torch.float32, device=dev)
                    _call("gb_gemm_dgrad", dev, _lib.ptr(dY), _lib.ptr(W), _lib.ptr(dX0), None, None, None, 0, P, K, N,
                          None, None, None, opts, st, meta=_gemm_meta("dgrad", P, K, N, rows_dev=rdev))
                break
            if (l == 1 and _FIRST_FUSE and X0.shape[1] == 3 and not ctx.needs_input_grad[0] and need_w[0]
                    and _lib.lib().gb_gemm_uses_rs(P, K, N, 1, 2, 0)):
                # xyz-only first layer: its dZ is never written - five column sums out of the dgrad epilogue give its
                # BatchNorm gradients and, with the 12 moments of the input rows, its weight gradient in closed form
                z = _zeros64(slots * 5 * K + 3 * K + 12, dev)
                sums, u0, mom = z[:slots * 5 * K], z[slots * 5 * K:slots * 5 * K + 3 * K], z[slots * 5 * K + 3 * K:]
                if ctx.fold:
                    _call("gb_gemm_dgrad_first_gen3", dev, _lib.ptr(dY), _lib.ptr(W), _lib.ptr(abs_[0]), _lib.ptr(X0),
                          _lib.ptr(Ws[0]), _lib.ptr(sums), slots, P, K, N, opts, st,
                          meta=_first_meta(P, K, N, rdev))
                    mom = ctx.mom0   # the forward's moments of the same rows
                else:
                    _call("gb_gemm_dgrad_first", dev, _lib.ptr(dY), _lib.ptr(W), _lib.ptr(Ys[0]), _lib.ptr(abs_[0]),
                          _lib.ptr(X0), _lib.ptr(sums), slots, P, K, N, opts, st,
                          meta=_first_meta(P, K, N, rdev))
                    _call("gb_moments3", dev, _lib.ptr(X0), _lib.ptr(rows.w if rows is not None else None), P,
                          _lib.ptr(mom), _lib.ptr(rdev), st)
                dbeta0, dgamma0 = bn_grads(0)
                grads[1], grads[2] = dgamma0, dbeta0
                dW0 = torch.empty((K, 3), dtype=torch.float32, device=dev)
                _call("gb_la_wx_grad_g", dev, _lib.ptr(sums), slots, _lib.ptr(u0), _lib.ptr(mom), _lib.ptr(Ws[0]),
                      _lib.ptr(abs_[0]), P_stat, K, int(training[0]), _lib.ptr(dW0), _lib.ptr(dbeta0), _lib.ptr(dgamma0), st)
                grads[0] = dW0
                break
            # gradient of layer l-1's post-ReLU output + that layer's BatchNorm-backward sums in one launch
            dZ = _empty_rows(P, K, dev, quant)
            region = d_arena[d_off[l - 1]:d_off[l]]
            dbeta, dgamma = bn_grads(l - 1)
            grads[3 * l - 2], grads[3 * l - 1] = dgamma, dbeta
            emit = False   # the apply pass below writes dbeta / dgamma itself (gb_bn_bwd_apply_g)
            if fused[l - 1]:
                dstats = region[slots * 2 * K:] if slots > 1 else region  # the slot rows' total
                # one slot row: the sums ARE the totals, and all gb_bn_bwd_reduce would do is convert them to fp32 - the
                # apply pass reads them anyway and does that (one launch less per layer of the few-row stacks)
                emit = slots == 1 and rows is None
                if pair:
                    dW = w_arena[w_off[l]:w_off[l + 1]]
                    _call("gb_gemm_dgrad_wgrad", dev, _lib.ptr(dY), _lib.ptr(W), _lib.ptr(dZ), _lib.ptr(Ys[l - 1]),
                          _lib.ptr(abs_[l - 1]), _lib.ptr(region), slots, P, K, N, None if emit else _lib.ptr(dstats),
                          None if emit else _lib.ptr(dbeta), None if emit else _lib.ptr(dgamma), _lib.ptr(src),
                          _lib.ptr(aff), _lib.ptr(dW), opts, st, meta=_pair_meta(P, K, N))
                    grads[3 * l] = dW.view(N, K)
                else:
                    _call("gb_gemm_dgrad", dev, _lib.ptr(dY), _lib.ptr(W), _lib.ptr(dZ), _lib.ptr(Ys[l - 1]),
                          _lib.ptr(abs_[l - 1]), _lib.ptr(region), slots, P, K, N, None if emit else _lib.ptr(dstats),
                          None if emit else _lib.ptr(dbeta), None if emit else _lib.ptr(dgamma), opts, st,
                          meta=_gemm_meta("dgrad", P, K, N, fused=True, rows_dev=rdev))
            else:
                # wide + long outputs: the fused epilogue measured slower than a separate column pass
                _call("gb_gemm_dgrad", dev, _lib.ptr(dY), _lib.ptr(W), _lib.ptr(dZ), None, None, None, 0, P, K, N,
                      None, None, None, opts, st, meta=_gemm_meta("dgrad", P, K, N, rows_dev=rdev))
                dstats = region
                _call("gb_bn_bwd_stats", dev, _lib.ptr(dZ), _lib.ptr(Ys[l - 1]), _lib.ptr(abs_[l - 1]), None, P, K, 1,
                      _lib.ptr(dstats), _lib.ptr(dbeta), _lib.ptr(dgamma), st)
            dY = _empty_rows(P, K, dev, quant)
            if rows is not None:
                _call("gb_bn_bwd_apply_w", dev, _lib.ptr(dZ), _lib.ptr(Ys[l - 1]), _lib.ptr(abs_[l - 1]),
                      _lib.ptr(dstats), _lib.ptr(rows.w), P, P_stat, K, int(training[l - 1]), _lib.ptr(dY), _lib.ptr(rdev), st)
            elif emit:
                _call("gb_bn_bwd_apply_g", dev, _lib.ptr(dZ), _lib.ptr(Ys[l - 1]), _lib.ptr(abs_[l - 1]), None,
                      _lib.ptr(dstats), P, K, 1, int(training[l - 1]), _lib.ptr(dY), None, _lib.ptr(dbeta),
                      _lib.ptr(dgamma), st)
            else:
                _call("gb_bn_bwd_apply", dev, _lib.ptr(dZ), _lib.ptr(Ys[l - 1]), _lib.ptr(abs_[l - 1]), None,
                      _lib.ptr(dstats), P, K, 1, int(training[l - 1]), _lib.ptr(dY), None, st)
        return (dX0, dres, None, None, None, None, *grads)


# The crop stacks' last layer (csrc/gemm_rs.hip RS_STATS_POOL_V):
#   GB_CROP_POOL=0     the GEMM stores Y3, gb_affine_relu_maxpool_members pools it in a second pass (the generic form: any
#                      width, any number of crops <= 4)
#   default            the pooling leaves the GEMM epilogue as per-(tile, seed, crop) extreme VALUES + gb_pool_pairs; Y3 is
#                      still stored and the dense backward finds the arg-max rows by value (gb_bn_bwd_apply_members_v),
#                      its BatchNorm sums come from the saved extremes without a gather: one pass over Y3 less
# (Round 3's low-rank + sparse backward of this layer, the bf16 storage of its activations and the multi-pick FPS were
# measured slower than these defaults and removed in round 4: DESIGN.md section 6.)
_CROP_POOL = os.environ.get("GB_CROP_POOL", "1") != "0"


def set_crop_pool(flag):
    """-> previous setting."""
    global _CROP_POOL
    prev, _CROP_POOL = _CROP_POOL, bool(flag)
    return prev


# Row counts that only exist on the device (the distinct rows of the cylinder crops): with static rows the activations
# are sized for the CAPACITY (every slot of every crop distinct) and every kernel reads the count itself
# (GbGemmOpts.rows_dev) - no device -> host read anywhere in the step, which is what lets train.Trainer capture the
# whole step in a HIP graph.  GB_STATIC_ROWS=0: read the counts back and size the activations exactly (one
# synchronisation per step, 2.6x less activation memory for the crop stacks).
_STATIC_ROWS = os.environ.get("GB_STATIC_ROWS", "1") != "0"


def set_static_rows(flag):
    """-> previous setting."""
    global _STATIC_ROWS
    prev, _STATIC_ROWS = _STATIC_ROWS, bool(flag)
    return prev


def crop_static_ok(cap, widths, D):
    """Can a crop stack 3 -> widths[0] -> widths[1] -> widths[2] run on a row CAPACITY of `cap` rows with the count on the
    device?  Only the default execution knows how: first layer folded, second on the row-streaming GEMM, last pooled in
    its epilogue (MLPStack.forward makes the same choices from the same conditions)."""
    L = _lib.lib()
    return bool(_STATIC_ROWS and _FIRST_FOLD and _FIRST_FUSE and _CROP_POOL and len(widths) == 3 and D <= 4
                and widths[0] % 4 == 0 and widths[1] <= 128
                and L.gb_gemm_uses_rs(cap, widths[0], widths[1], 0, 1, 1) and L.gb_gemm_uses_rs(cap, widths[0], widths[1], 1, 2, 0)
                and L.gb_gemm_uses_rs(cap, widths[1], widths[2], 0, 3, 1) and L.gb_gemm_uses_rs(cap, widths[1], widths[2], 1, 1, 0))


def _pooled_last_layer(dev, st, opts, src, W, aff, gamma, beta, cfg, ab, stats, slots, rows, P, K, N, P_stat, Y=None):
    """gb_gemm_fwd_pool + gb_pool_pairs -> (out, ystar), each ((R*D), N).  Y: optional (P, N) buffer that also receives
    the layer's output (a forward-only caller passes None: nothing can find the arg-max rows afterwards)."""
    RD = rows.R * rows.D
    # sized for the row count rounded up like the activations (_empty_rows): the distinct-row count changes every step,
    # and a new allocation size every step means a fresh hipMalloc - a device synchronisation - per radius
    cap_tiles = (P + _ROW_QUANTUM - 1) // _ROW_QUANTUM * (_ROW_QUANTUM // 32)
    pairs = torch.empty((cap_tiles + rows.R) * rows.D * N, dtype=torch.float32, device=dev)
    if cfg.training:
        fin = _bn_fin(cfg, gamma, beta, ab, P_stat)
    else:
        fin, stats, slots = None, _zeros64(2 * N, dev), 1   # the kernel always forms the sums; eval ignores them
    _call("gb_gemm_fwd_pool", dev, _lib.ptr(src), _lib.ptr(W), _lib.ptr(aff), _lib.ptr(rows.key), _lib.ptr(gamma),
          _lib.ptr(pairs), pairs.numel(), rows.R, _lib.ptr(Y), _lib.ptr(stats), slots, P, K, N, rows.D, fin, opts, st,
          meta=_gemm_meta("fwd", P, K, N, True, aff is not None, rows_dev=rows.rows_dev))
    # eval: `ab` is the caller's cached table (_eval_ab)
    out = torch.empty((RD, N), dtype=torch.float32, device=dev)
    ystar = torch.empty((RD, N), dtype=torch.float32, device=dev)
    _call("gb_pool_pairs", dev, _lib.ptr(pairs), _lib.ptr(rows.off), _lib.ptr(rows.cnt), _lib.ptr(ab), _lib.ptr(gamma),
          _lib.ptr(out), _lib.ptr(ystar), rows.R, rows.D, N, st)
    return out, ystar


def _arg_rows_by_value(Y, ystar, rows):
    """Debug / test helper (routing_observer): the arg-max rows the values-only pooled path implies - per (seed, crop,
    column) the first member row whose y equals y* - as gb_affine_relu_maxpool_members would name them."""
    R, D, C = rows.R, rows.D, Y.shape[1]
    P = int(rows.cnt.sum())   # (with a device-side row count Y is sized for the capacity)
    Y = Y[:P]
    seed = torch.repeat_interleave(torch.arange(R, device=Y.device), rows.cnt.long(), output_size=P)
    rowidx = torch.arange(P, device=Y.device, dtype=torch.int64).unsqueeze(1).expand(P, C)
    ys = ystar.view(R, D, C)
    arg = torch.empty((R, D, C), dtype=torch.int64, device=Y.device)
    big = torch.iinfo(torch.int64).max
    for d in range(D):
        hit = (Y == ys[seed, d]) & ((rows.mem[:P].long() >> d) & 1).bool().unsqueeze(1)
        cand = torch.where(hit, rowidx, torch.full_like(rowidx, big))
        first = torch.full((R, C), big, dtype=torch.int64, device=Y.device)
        first.scatter_reduce_(0, seed.unsqueeze(1).expand(P, C), cand, reduce="amin")
        arg[:, d] = torch.where(first == big, rows.off.long().unsqueeze(1).expand(R, C), first)
    return arg.view(R * D, C).to(torch.int32)


def _observed_ys(Ys, X0, Ws):
    """routing_observer helper: the folded first layer's output, re-formed exactly as the kernels do."""
    Ys = [y.float() if (y is not None and y.dtype != torch.float32) else y for y in Ys]
    if Ys[0] is not None:
        return Ys
    W1 = Ws[0]
    y1 = ((X0[:, 0:1] * W1[:, 0]) + (X0[:, 1:2] * W1[:, 1])) + (X0[:, 2:3] * W1[:, 2])
    return [y1] + list(Ys[1:])


class LocalGeometry:
    """The per-point summary of one grouping (xyz, centres, idx) that LocalAggPool needs instead of the grouped
    tensor: cnt (b*n) references per point, dsum (b*n,3) sum of their relative positions, mom fp64 [12] = [sum dp,
    sum dp dp^T].  A stage of InvResMLP blocks over the same points and radius shares one."""

    def __init__(self, xyz, centres, idx, mode=0, scale=1.0):
        self.xyz, self.centres, self.idx = xyz.contiguous(), centres.contiguous(), idx.contiguous()
        self.b, self.n = self.xyz.shape[0], self.xyz.shape[1]
        self.m, self.ns = self.idx.shape[1], self.idx.shape[2]
        self.mode, self.scale = int(mode), float(scale)
        self._perm = None
        dev = xyz.device
        pts = _zeros32(self.b * self.n * 4, dev)
        self.cnt, self.dsum = pts[:self.b * self.n], pts[self.b * self.n:]
        self.mom = _zeros64(12, dev)
        _call("gb_la_point_stats", dev, _lib.ptr(self.xyz), _lib.ptr(self.centres), _lib.ptr(self.idx), self.b, self.n,
              self.m, self.ns, self.mode, self.scale, _lib.ptr(self.cnt), _lib.ptr(self.dsum), _lib.ptr(self.mom),
              _s(xyz))

    @property
    def rows(self):
        return self.b * self.m * self.ns

    def row_perm(self):
        """(b, m) int32: a spatially coherent order of each cloud's centres (csrc/fps.hip gb_fps_row_order: consecutive
        entries are neighbours in space) for gb_la_pool_bwd_perm - one launch per stage, shared by its blocks' backwards."""
        if self._perm is None and _LA_AGG and self.m % 32 == 0 and _LA_AGG_MIN_M <= self.m <= 24576:
            self._perm = torch.empty((self.b, self.m), dtype=torch.int32, device=self.centres.device)
            _call("gb_fps_row_order", self.centres.device, _lib.ptr(self.centres), _lib.ptr(self._perm), self.b, self.m,
                  _s(self.centres))
        return self._perm


def local_agg_supported(C_out, ns):
    return C_out % 4 == 0 and 16 <= C_out <= 1024 and ns <= 64


class LocalAggPool(Function):
    """LocalAggregation's  group -> [dp, fj] -> 1x1 conv -> BatchNorm -> ReLU -> max over neighbours  (reference
    drp.py:32-67) on the point features f (b*n, C) directly: G = f Wf^T, y = G[idx] + dp.Wx is formed on the fly
    and never stored (csrc/local_agg.hip).  forward(ctx, f, W (N,3+C), gamma, beta, geo, cfg) -> (b*m, N)."""

    @staticmethod
    def forward(ctx, f, W, gamma, beta, geo, cfg):
        dev = f.device
        st = _s(f)  # one stream lookup per call, not one per launch
        f = f.contiguous()
        N, C = W.shape[0], W.shape[1] - 3
        W = W.contiguous()
        wbuf = torch.empty(N * (3 + C), dtype=torch.float32, device=dev)
        Wx, Wf = wbuf[:N * 3].view(N, 3), wbuf[N * 3:].view(N, C)  # N*3*4 bytes is a multiple of 16 for N % 4 == 0
        _call("gb_la_split_w", dev, _lib.ptr(W), _lib.ptr(Wx), _lib.ptr(Wf), N, C, st)
        rows, P = geo.b * geo.n, geo.rows
        G = torch.empty((rows, N), dtype=torch.float32, device=dev)
        ctx.prec = _prec()
        _call("gb_gemm_fwd", dev, _lib.ptr(f), _lib.ptr(Wf), None, _lib.ptr(G), None, 1, rows, C, N, None,
              _opts(dev, st, ctx.prec), st, meta=_gemm_meta("fwd", rows, C, N))
        sums = _zeros64(5 * N, dev)  # [sum y, sum y^2, U0, U1, U2]
        stats, u = sums[:2 * N], sums[2 * N:]
        ab = torch.empty(4 * N, dtype=torch.float32, device=dev)
        if cfg.training:  # the same C call also finishes the BatchNorm (ab table, running statistics)
            _call("gb_la_col_stats", dev, _lib.ptr(G), _lib.ptr(geo.cnt), _lib.ptr(geo.dsum), _lib.ptr(Wx),
                  _lib.ptr(geo.mom), rows, N, _lib.ptr(stats), _lib.ptr(u),
                  _bn_fin(cfg, gamma, beta, ab, P), st)
        else:
            ab = _eval_ab(gamma, beta, cfg.running_mean, cfg.running_var, cfg.eps, N, dev, st)
        R = geo.b * geo.m
        out = torch.empty((R, N), dtype=torch.float32, device=dev)
        arg = torch.empty((R, N), dtype=torch.int32, device=dev)
        _call("gb_la_pool", dev, _lib.ptr(G), _lib.ptr(geo.xyz), _lib.ptr(geo.centres), _lib.ptr(geo.idx), _lib.ptr(Wx),
              _lib.ptr(ab), _lib.ptr(out), _lib.ptr(arg), geo.b, geo.n, geo.m, geo.ns, N, geo.mode, geo.scale, st)
        ctx.geo, ctx.training = geo, cfg.training
        ctx.save_for_backward(f, Wx, Wf, G, ab, out, arg, u)
        if routing_observer is not None:
            routing_observer("local_agg", out=out, arg=arg, ns=geo.ns)
        return out

    @staticmethod
    def backward(ctx, dout):
        f, Wx, Wf, G, ab, out, arg, u = ctx.saved_tensors
        geo, training = ctx.geo, int(ctx.training)
        dev = dout.device
        st = _s(dout)  # one stream lookup per call, not one per launch
        dout = dout.contiguous()
        N, C = Wf.shape
        rows, P = geo.b * geo.n, geo.rows
        # one zero fill for the two atomic-add targets: sg (rows, N) and, when the weight gradient is wanted, dWf (N, C)
        zbuf = _zeros32(rows * N + (N * (3 + C) if ctx.needs_input_grad[1] else 0), dev)
        sg = zbuf[:rows * N].view(rows, N)
        red = _zeros64(5 * N, dev)  # [dbeta, dgamma, T0, T1, T2]
        _call("gb_la_pool_bwd_perm", dev, _lib.ptr(dout), _lib.ptr(out), _lib.ptr(arg), _lib.ptr(G), _lib.ptr(geo.xyz),
              _lib.ptr(geo.centres), _lib.ptr(geo.idx), _lib.ptr(Wx), _lib.ptr(ab), _lib.ptr(geo.row_perm()), _lib.ptr(sg),
              _lib.ptr(red), geo.b, geo.n, geo.m, geo.ns, N, geo.mode, geo.scale, st)
        small = torch.empty(5 * N, dtype=torch.float32, device=dev)  # dbeta, dgamma, dWx (N,3)
        dbeta, dgamma, dWx = small[:N], small[N:2 * N], small[2 * N:5 * N].view(N, 3)
        if not ctx.needs_input_grad[1]:   # (otherwise gb_la_wx_grad_g below converts the two sums: one launch less)
            _call("gb_bn_bwd_reduce", dev, _lib.ptr(red), 1, N, None, _lib.ptr(dbeta), _lib.ptr(dgamma), st)
        dG = torch.empty((rows, N), dtype=torch.float32, device=dev)
        _call("gb_la_point_grad", dev, _lib.ptr(sg), _lib.ptr(G), _lib.ptr(geo.cnt), _lib.ptr(geo.dsum), _lib.ptr(Wx),
              _lib.ptr(ab), _lib.ptr(red), P, rows, N, training, _lib.ptr(dG), st)
        dW = None
        if ctx.needs_input_grad[1]:
            if wgrad_deferred(dev, st, rows, C, N, ctx.prec):
                # recorded (WgradQueue): the grouped launch adds dG^T f straight into columns 3.. of the joined (N, 3 + C)
                # gradient (its rows lie 3 + C floats apart), the xyz columns are written by the closed-form kernel at
                # that pitch - no join launch, no strided copy
                dW = zbuf[rows * N:rows * N + N * (3 + C)].view(N, 3 + C)
                _call("gb_la_wx_grad_gs", dev, _lib.ptr(red), 1, _lib.ptr(u), _lib.ptr(geo.mom), _lib.ptr(Wx), _lib.ptr(ab), P,
                      N, training, _lib.ptr(dW), 3 + C, _lib.ptr(dbeta), _lib.ptr(dgamma), st)
                _wgrad_call(dev, st, dG, f, None, dW[:, 3:], rows, C, N, ctx.prec, None, ldw=3 + C)
            else:
                _call("gb_la_wx_grad_g", dev, _lib.ptr(red), 1, _lib.ptr(u), _lib.ptr(geo.mom), _lib.ptr(Wx), _lib.ptr(ab), P,
                      N, training, _lib.ptr(dWx), _lib.ptr(dbeta), _lib.ptr(dgamma), st)
                dWf = zbuf[rows * N:rows * N + N * C].view(N, C)
                _call("gb_gemm_wgrad", dev, _lib.ptr(dG), _lib.ptr(f), None, _lib.ptr(dWf), rows, C, N,
                      _opts(dev, st, ctx.prec), st, meta=_gemm_meta("wgrad", rows, C, N, prec=ctx.prec))
                dW = torch.empty((N, 3 + C), dtype=torch.float32, device=dev)
                _call("gb_la_join_w", dev, _lib.ptr(dWx), _lib.ptr(dWf), _lib.ptr(dW), N, C, st)
        df = None
        if ctx.needs_input_grad[0]:
            df = torch.empty((rows, C), dtype=torch.float32, device=dev)
            _call("gb_gemm_dgrad", dev, _lib.ptr(dG), _lib.ptr(Wf), _lib.ptr(df), None, None, None, 0, rows, C, N,
                  None, None, None, _opts(dev, st, ctx.prec), st, meta=_gemm_meta("dgrad", rows, C, N))
        return df, dW, dgamma, dbeta, None, None


def local_agg_pool(f_cl, conv, bn, geo):
    """conv (1x1, 3+C -> N, no bias) + bn + ReLU + max over the neighbours of `geo`, from point features (b*n, C)."""
    if conv.bias is not None:
        raise NotImplementedError("fused path expects bias-free convs followed by BatchNorm")
    _count_batch(bn)
    W = conv.weight.view(conv.weight.shape[0], -1)
    return LocalAggPool.apply(f_cl, W, bn.weight, bn.bias, geo, _LayerCfg(bn))


_pending_counters = None  # list of num_batches_tracked buffers while inside deferred_counters()


class deferred_counters:
    """Within this context the fused layers' ``num_batches_tracked += 1`` are collected and applied as one
    multi-tensor add on exit (the model's forward wraps itself in it: ~80 one-element launches -> 1)."""

    def __enter__(self):
        global _pending_counters
        self.outer = _pending_counters
        if self.outer is None:
            _pending_counters = []
        return self

    def __exit__(self, *exc):
        global _pending_counters
        if self.outer is None:
            pending, _pending_counters = _pending_counters, None
            if pending:
                torch._foreach_add_(pending, 1)
        return False


def _count_batch(bn):
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if _pending_counters is not None:
            _pending_counters.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)


_ROW_QUANTUM = 32768


def _empty_rows(P, C, dev, quantised, dtype=torch.float32):
    """(P, C) activation buffer (fp32; bf16 for the stored outputs of the bf16 storage mode).  With `quantised` (row counts that change from step to step: the distinct rows
    of the cylinder crops) the allocation is rounded up to a multiple of 32768 rows, so the caching allocator sees
    the same few sizes every step instead of a new one (a new size means a fresh hipMalloc - a device sync)."""
    if not quantised:
        return torch.empty((P, C), dtype=dtype, device=dev)
    cap = (P + _ROW_QUANTUM - 1) // _ROW_QUANTUM * _ROW_QUANTUM
    return torch.empty((cap, C), dtype=dtype, device=dev)[:P]


class RowSet:
    """Distinct rows of a batch with duplicates (the D nested cylinder crops of a seed, csrc/cyl_rows.hip): w / w16
    multiplicities (float / uint16 padded to a multiple of 32 rows), mem member bits, off / cnt the rows of each of
    the R seeds, D crops per seed, P_total rows of the full batch."""
    __slots__ = ("w", "w16", "mem", "off", "cnt", "R", "D", "P_total", "key", "rows_dev")

    def __init__(self, w, w16, mem, off, cnt, R, D, P_total, key=None, rows_dev=None):
        self.w, self.w16, self.mem, self.off, self.cnt = w, w16, mem, off, cnt
        self.R, self.D, self.P_total = int(R), int(D), int(P_total)
        self.key = key  # (seed << 13) | (multiplicity << 4) | member bits per row, zero-padded to 32 rows (D <= 4)
        # one-element int64 device tensor = the number of rows, when the host never learns it: every per-row array (and
        # the x0 that goes with this set) is then sized for the capacity R * D * ns and kernels take the count from here
        self.rows_dev = rows_dev


class _ZeroGradFor(Function):
    """Identity on x that also hands zero gradients to `params`: a convolution bias directly in front of a
    batch-statistics BatchNorm has an exactly zero gradient (the normalisation removes it), but the parameter must
    still RECEIVE one - optimizers and the gradient all-reduce count arrivals."""

    @staticmethod
    def forward(ctx, x, *params):
        ctx.shapes = [(p.shape, p.dtype, p.device) for p in params]
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        # (views of the step's zeroed fp32 arena where there is one: no fill launches)
        def z(sh, dt, dev):
            n = 1
            for d in sh:
                n *= d
            if dt == torch.float32 and dev.type == "cuda":
                return _zeros32(n, dev).view(sh)
            return torch.zeros(sh, dtype=dt, device=dev)
        return (g,) + tuple(z(sh, dt, dev) for sh, dt, dev in ctx.shapes)


def _stack(X, convs_bns, pool_ns=0, residual=None, relu_last=True, rows=None):
    """conv (1x1) + BatchNorm stacks on rows.  A conv WITH bias (the grasp heads, modules.py:49-175) is run without it:
    in front of BatchNorm the bias only shifts the batch mean, i.e. training outputs are unchanged and the running mean
    moves by momentum * bias (applied afterwards); in eval mode it is folded into the running mean handed to the
    finalisation."""
    params, layers, biased, shifts = [], [], [], []
    for conv, bn in convs_bns:
        _count_batch(bn)
        params += [conv.weight.view(conv.weight.shape[0], -1), bn.weight, bn.bias]
        cfg = _LayerCfg(bn)
        if conv.bias is not None:
            biased.append(conv.bias)
            if cfg.training:
                if cfg.running_mean is not None:
                    shifts.append((bn.running_mean, conv.bias.detach(), cfg.momentum))
            else:
                cfg.running_mean = bn.running_mean - conv.bias.detach()
        layers.append(cfg)
    if not torch.is_grad_enabled():
        # needs_input_grad reports the parameters' requires_grad whatever the grad mode: without this a no_grad call
        # (Predictor) would still store the crop stack's last output and gather the eval tables for a backward
        params = [p.detach() for p in params]
    out = MLPStack.apply(X, residual, layers, pool_ns, relu_last, rows, *params)
    if shifts:   # (one launch for the stack's layers)
        torch._foreach_add_([s[0] for s in shifts], [s[1] for s in shifts], alpha=shifts[0][2])
    if biased and any(b.requires_grad for b in biased) and out.requires_grad:
        out = _ZeroGradFor.apply(out, *biased)
    return out


class LinearBias(Function):
    """Y = X W^T + b on rows (a 1x1 convolution with bias and no normalisation: the last layer of the grasp heads,
    the scale-fusion and gate convolutions) on the hand-written GEMMs.  forward(ctx, X (P,K), W (N,K), b (N)|None)."""

    @staticmethod
    def forward(ctx, X, W, b):
        dev = X.device
        st = _s(X)
        X, W = X.contiguous(), W.contiguous()
        P, K = X.shape
        N = W.shape[0]
        Y = torch.empty((P, N), dtype=torch.float32, device=dev)
        ctx.prec = _prec()
        _call("gb_gemm_fwd", dev, _lib.ptr(X), _lib.ptr(W), None, _lib.ptr(Y), None, 1, P, K, N, None,
              _opts(dev, st, ctx.prec), st, meta=_gemm_meta("fwd", P, K, N))
        if b is not None:
            Y.add_(b.detach())     # (one launch; 1 * y + b of the affine pass it replaces is the same value)
        ctx.has_bias = b is not None
        ctx.save_for_backward(X, W)
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, W = ctx.saved_tensors
        dev = dY.device
        st = _s(dY)
        dY = dY.contiguous()
        P, K = X.shape
        N = W.shape[0]
        dX = dW = db = None
        if ctx.needs_input_grad[0]:
            dX = torch.empty((P, K), dtype=torch.float32, device=dev)
            _call("gb_gemm_dgrad", dev, _lib.ptr(dY), _lib.ptr(W), _lib.ptr(dX), None, None, None, 0, P, K, N, None, None,
                  None, _opts(dev, st, ctx.prec), st, meta=_gemm_meta("dgrad", P, K, N))
        if ctx.needs_input_grad[1]:
            dW = _zeros32(N * K, dev).view(N, K)
            _wgrad_call(dev, st, dY, X, None, dW, P, K, N, ctx.prec, _opts(dev, st, ctx.prec))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            sums = _zeros64(2 * N, dev)
            _call("gb_col_stats", dev, _lib.ptr(dY), P, N, _lib.ptr(sums), None, st)
            db = sums[:N].float()
        return dX, dW, db


def linear_bias(X, conv):
    """A 1x1 Conv1d / Conv2d (with or without bias) applied to rows X (P, Cin) -> (P, Cout)."""
    W = conv.weight.view(conv.weight.shape[0], -1)
    return LinearBias.apply(X, W, conv.bias)


def conv_bn_act(X, conv, bn, relu=True, pool_ns=0, residual=None):
    """Apply a 1x1 ``conv`` (Conv1d/Conv2d without bias) + ``bn`` (BatchNorm1d/2d) + optional ReLU to
    channel-last rows X (P,Cin), with the modules' own parameters and running statistics."""
    if _OWN_GEMM:
        return _stack(X, [(conv, bn)], pool_ns=pool_ns, residual=residual, relu_last=relu)
    W = conv.weight.view(conv.weight.shape[0], -1)
    if conv.bias is not None:
        raise NotImplementedError("the torch.mm comparison path expects bias-free convs followed by BatchNorm")
    _count_batch(bn)
    momentum = 0.0 if bn.momentum is None else bn.momentum
    training = bn.training or not bn.track_running_stats
    return LinearBNAct.apply(X, W, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, momentum, bn.eps,
                             training, relu, pool_ns)


def conv_bn_act_chain(X, convs_bns, residual=None, relu_last=True, pool_ns=0, rows=None):
    """Consecutive conv+BN(+ReLU) layers (ReLU after every layer but optionally the last) as one fused stack."""
    if _OWN_GEMM or rows is not None:
        return _stack(X, convs_bns, pool_ns=pool_ns, residual=residual, relu_last=relu_last, rows=rows)
    n = len(convs_bns)
    for i, (conv, bn) in enumerate(convs_bns):
        last = i == n - 1
        X = conv_bn_act(X, conv, bn, relu=relu_last if last else True, pool_ns=pool_ns if last else 0,
                        residual=residual if last else None)
    return X


def shared_mlp_cl(X, shared_mlp, pool_ns=0, rows=None):
    """Run a ``pytorch_utils.SharedMLP`` (layer0..layerK of conv+bn+ReLU) on channel-last rows; the
    last layer is fused with the max over `pool_ns` consecutive rows when pool_ns > 0, or - with `rows`
    (RowSet: X holds the distinct rows of a batch with duplicates) - with the per-crop max over the members."""
    layers = [(layer.conv, layer.bn.bn) for layer in shared_mlp.children()]
    return conv_bn_act_chain(X, layers, pool_ns=pool_ns, rows=rows)


def shared_mlp_widths(shared_mlp):
    """Output widths of a pytorch_utils.SharedMLP's layers."""
    return [layer.conv.weight.shape[0] for layer in shared_mlp.children()]


_CYL_DEDUP = os.environ.get("GB_CYL_DEDUP", "1") != "0"  # A/B switch: distinct rows for the nested cylinder crops


def cyl_dedup_enabled():
    return _CYL_DEDUP


def set_cyl_dedup(flag):
    global _CYL_DEDUP
    _CYL_DEDUP = bool(flag)


def cylinder_rows(idx, xyz, centres, rot, static=False):
    """idx (nr, D, B, m, ns) int32 from fused_ops.cylinder_query_multi -> per radius (x0 (P_u,3), RowSet): the DISTINCT
    (seed, point) rows of the D nested crops of every seed, rotated into the seed's frame (csrc/cyl_rows.hip; three
    launches for all radii).  static=False: one device -> host read of the nr row counts sizes the outputs exactly (the
    step's only synchronisation).  static=True (callers check crop_static_ok): no read - x0 and the per-row arrays have
    the capacity R * D * ns rows and RowSet.rows_dev holds the count on the device."""
    nr, D, B, m, ns = idx.shape
    R, W, dev = B * m, D * ns, idx.device
    idx = idx.contiguous()
    xyz, centres, rot9 = xyz.contiguous(), centres.contiguous(), rot.reshape(B, m, 9).contiguous()
    scratch = torch.empty((2, nr, R, W), dtype=torch.int32, device=dev)
    count = torch.empty((nr, R), dtype=torch.int32, device=dev)
    off = torch.empty((nr, R), dtype=torch.int64, device=dev)
    total = torch.empty(nr, dtype=torch.int64, device=dev)
    st = _s(idx)
    _call("gb_cyl_unique", dev, _lib.ptr(idx), nr, D, R, ns, _lib.ptr(scratch[0]), _lib.ptr(scratch[1]), _lib.ptr(count), st)
    _call("gb_cyl_scan", dev, _lib.ptr(count), nr, R, _lib.ptr(off), _lib.ptr(total), st)
    if static:
        totals = None
        cap = (R * W + 31) // 32 * 32
    else:
        t_sync = time.perf_counter()
        totals = total.tolist()  # the one host synchronisation: row counts size the activations
        SYNC_WAIT[0] += time.perf_counter() - t_sync   # (bench.py: host-bound or GPU-bound? a host that arrives late waits ~0)
        cap = (max(totals) + _ROW_QUANTUM - 1) // _ROW_QUANTUM * _ROW_QUANTUM  # allocation sizes that repeat from step to step
    x0 = torch.empty((nr, cap, 3), dtype=torch.float32, device=dev)
    w = torch.empty((nr, cap), dtype=torch.float32, device=dev)
    w16 = torch.empty((nr, cap), dtype=torch.int16, device=dev)   # uint16 bits; the kernel zeroes the tail of the last tile
    mem = torch.empty((nr, cap), dtype=torch.int32, device=dev)
    key = torch.empty((nr, cap), dtype=torch.int32, device=dev) if (D <= 4 and R < (1 << 18)) else None
    _call("gb_cyl_rows", dev, _lib.ptr(xyz), _lib.ptr(centres), _lib.ptr(rot9), _lib.ptr(scratch[0]), _lib.ptr(scratch[1]),
          _lib.ptr(count), _lib.ptr(off), nr, B, xyz.shape[1], m, W, cap, _lib.ptr(x0), _lib.ptr(w), _lib.ptr(w16),
          _lib.ptr(mem), _lib.ptr(key), st)
    out = []
    for i in range(nr):
        if static:
            rs = RowSet(w[i], w16[i], mem[i], off[i], count[i], R, D, R * W, key[i] if key is not None else None,
                        rows_dev=total[i:i + 1])
            out.append((x0[i], rs))
        else:
            Pu = int(totals[i])
            pad = (Pu + 31) // 32 * 32
            rs = RowSet(w[i, :Pu], w16[i, :pad], mem[i, :Pu], off[i], count[i], R, D, R * W,
                        key[i, :pad] if key is not None else None)
            out.append((x0[i, :Pu], rs))
        if routing_observer is not None:
            routing_observer("cyl_rows", rowset=out[-1][1], sorted=scratch[0, i], idx=idx[i])
    return out


def supports(shared_mlp):
    """True when every layer is conv(no bias) -> bn -> ReLU (the layout SharedMLP(bn=True) builds)."""
    for layer in shared_mlp.children():
        names = [n for n, _ in layer.named_children()]
        if names != ["conv", "bn", "activation"] or layer.conv.bias is not None:
            return False
        if not isinstance(layer.activation, torch.nn.ReLU):
            return False
        bn = getattr(layer.bn, "bn", layer.bn)
        if bn.momentum is None and bn.track_running_stats:
            return False  # cumulative-average running statistics: plain composition
    return True
