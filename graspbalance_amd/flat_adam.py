"""Adam (the reference's optimizer, train.py:94: torch.optim.Adam defaults) on ONE flat parameter buffer.

GraspBalance has 253 parameter tensors, most of them tiny (BatchNorm scales, biases): torch's multi-tensor Adam spends
~1 ms of host time per step building its tensor lists and 0.33 ms of GPU time in 12 chunked launches that move 250 MB
at 0.8 TB/s.  Here every ``p.data`` is a view into one flat fp32 buffer, the moments are flat too, the gradients are
packed with one multi-tensor copy and the update is ONE fused launch over 9 M elements - the same per-element
arithmetic (torch's own fused Adam kernel on CUDA), so parameters follow torch.optim.Adam's trajectory.

``state_dict()`` has torch.optim.Adam's layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``; the tensors
are views of the flat buffers) and ``load_state_dict`` copies into them.  A parameter whose ``.grad`` is None takes
part with a zero gradient (torch skips it); every GraspBalance parameter receives a gradient in every step.
"""
import math
import os

import torch
from torch.optim import Optimizer

_SEGMENT_PACK = os.environ.get("GB_SEGMENT_PACK", "1") != "0"   # A/B switch: the gradient gather as one own launch
_SEG = 8192   # elements per workgroup of gb_copy_segments


class FlatAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam takes one parameter group")
        ps = [p for p in self.param_groups[0]['params'] if p.requires_grad]
        if not ps or any(p.dtype != torch.float32 or p.device != ps[0].device for p in ps):
            raise ValueError("FlatAdam needs float32 parameters on one device")
        self._params = ps
        dev = ps[0].device
        total = sum(p.numel() for p in ps)
        self._flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        self._flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self._exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self._exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._step_t = torch.zeros((), dtype=torch.float32, device=dev)
        # the learning rate as a device scalar: a step captured in a HIP graph (train.Trainer) must not bake the schedule's
        # current value into the update launch - set_lr_tensor(lr) before every replay instead
        self._lr_t = torch.zeros((), dtype=torch.float32, device=dev)
        self.tensor_lr = False
        self._steps = 0
        self._grad_views = []
        off = 0
        for p in ps:
            n = p.numel()
            view = self._flat_p[off:off + n].view_as(p)
            view.copy_(p.data)
            p.data = view  # the parameter now lives in the flat buffer
            self._grad_views.append(self._flat_g[off:off + n].view_as(p))
            self.state[p] = {'step': self._step_t, 'exp_avg': self._exp_avg[off:off + n].view_as(p),
                             'exp_avg_sq': self._exp_avg_sq[off:off + n].view_as(p)}
            off += n
        self._fused = dev.type == "cuda" and hasattr(torch, "_fused_adam_")
        # the gather's tables (pack): pinned buffers allocated HERE - a capture must not allocate host memory
        self._seg_tables, self._seg_captured = {}, []
        self._seg_rows = total // _SEG + len(ps) + 1
        self._seg_pool = ([torch.empty((self._seg_rows, 3), dtype=torch.int64).pin_memory() for _ in range(24)]
                          if (dev.type == "cuda" and _SEGMENT_PACK) else [])

    @torch.no_grad()
    def pack(self, lo=0, hi=None):
        """Gather the parameters' gradients into the flat gradient buffer (one multi-tensor copy; gradients that already
        are views of it - after the flat-bucket all-reduce - cost nothing).  step() does this itself unless told the
        buffer is packed already (a step split around a collective: pack, all-reduce the buffer, step(packed=True)).
        lo, hi: only parameters [lo, hi) of the registration order (a backward cut in two: each part packs its slice)."""
        have_v, have_g = [], []
        for v, p in zip(self._grad_views[lo:hi], self._params[lo:hi]):
            if p.grad is None:
                v.zero_()
            elif p.grad is not v:
                have_v.append(v)
                have_g.append(p.grad)
        if have_v:
            if _SEGMENT_PACK and have_v[0].is_cuda:
                self._pack_segments(have_v, have_g)
            else:
                torch._foreach_copy_(have_v, have_g)

    def _pack_segments(self, views, grads):
        """The gather as ONE launch (csrc/mlp_cl.hip copy_segments_kernel) instead of torch's nine multi-tensor launches
        for 253 tensors: a device table of (source address, destination element, count) per 8192-element piece, sent
        through a pinned buffer.  Launch by launch the table is rebuilt only when an address changed (the allocator hands
        the same blocks out step after step); a capture gets a pinned buffer of its own out of a pool allocated up front
        (its copy node reads that buffer at every replay) - with the pool used up the gather falls back to torch's."""
        import numpy as np
        from . import _lib
        dev = views[0].device
        base = self._flat_g.data_ptr()
        ok = [g.is_contiguous() and g.dtype == torch.float32 and g.device == dev for g in grads]
        if not all(ok):   # (exotic gradients - strided, another dtype - keep torch's copy)
            torch._foreach_copy_([v for v, o in zip(views, ok) if not o], [g for g, o in zip(grads, ok) if not o])
            views = [v for v, o in zip(views, ok) if o]
            grads = [g for g, o in zip(grads, ok) if o]
            if not views:
                return
        capturing = torch.cuda.is_current_stream_capturing()
        key = (tuple(g.data_ptr() for g in grads), tuple(v.data_ptr() for v in views))
        slot = None if capturing else self._seg_tables.get((len(grads), views[0].data_ptr()))
        if slot is None or slot["key"] != key:
            if slot is None:
                if not self._seg_pool:
                    torch._foreach_copy_(views, grads)
                    return
                slot = {"host": self._seg_pool.pop(), "table": None}
                if not capturing:
                    self._seg_tables[(len(grads), views[0].data_ptr())] = slot
                else:
                    self._seg_captured.append(slot)   # (kept alive with the optimizer: a graph replays its copy node)
            elif slot.get("event") is not None:
                # the slot's previous host -> device copy may still read the pinned buffer, and it may have been enqueued
                # on ANOTHER stream than the current one (the capture warm-up stream, a second trainer's): wait for the
                # copy's own event, not for whatever stream is current now
                slot["event"].synchronize()
            src = np.array([g.data_ptr() for g in grads], dtype=np.int64)
            off = np.array([(v.data_ptr() - base) // 4 for v in views], dtype=np.int64)
            num = np.array([g.numel() for g in grads], dtype=np.int64)
            pieces = (num + _SEG - 1) // _SEG
            seg = np.repeat(np.arange(len(grads)), pieces)
            first = np.cumsum(pieces) - pieces
            k = np.arange(int(pieces.sum())) - first[seg]          # piece number within its tensor
            tab = np.stack([src[seg] + 4 * _SEG * k, off[seg] + _SEG * k, np.minimum(num[seg] - _SEG * k, _SEG)], 1)
            rows = int(tab.shape[0])
            slot["host"][:rows].copy_(torch.from_numpy(np.ascontiguousarray(tab)))
            if slot["table"] is None:
                slot["table"] = torch.empty((self._seg_rows, 3), dtype=torch.int64, device=dev)
            slot["table"][:rows].copy_(slot["host"][:rows], non_blocking=True)   # (a copy node when capturing)
            if not capturing:
                if slot.get("event") is None:
                    slot["event"] = torch.cuda.Event()
                slot["event"].record(torch.cuda.current_stream(dev))
            slot["key"], slot["rows"] = key, rows
        with _lib.device_ctx(dev):
            _lib.check(_lib.lib().gb_copy_segments(_lib.ptr(slot["table"]), slot["rows"], _lib.ptr(self._flat_g),
                                                   _lib.current_stream(dev)), "gb_copy_segments")

    def set_lr_tensor(self, lr=None):
        """Write the group's (or the given) learning rate into the device scalar the tensor_lr update reads."""
        self._lr_t.fill_(float(self.param_groups[0]['lr'] if lr is None else lr))

    def count_step(self):
        """Host-side bookkeeping of one update that ran without this object's step() (a replayed HIP graph of it)."""
        self._steps += 1
        from . import fused_mlp
        fused_mlp._TRAIN_TICK[0] += 1

    @torch.no_grad()
    def step(self, closure=None, packed=False):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        lr, (beta1, beta2), eps, wd = group['lr'], group['betas'], group['eps'], group['weight_decay']
        if not packed:
            self.pack()
        if self.tensor_lr and not (self._lr_t.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.set_lr_tensor(lr)   # (a captured step reads whatever the caller put there before each replay)
        self._steps += 1
        self._step_t += 1
        # the parameters are views of ONE flat buffer and the update below writes that buffer: the views' own version
        # counters do not move, so tell the eval-mode BatchNorm table cache (fused_mlp._eval_ab) that parameters changed
        from . import fused_mlp
        fused_mlp._TRAIN_TICK[0] += 1
        if self._fused:
            # tensor_lr: the kernel reads the rate from self._lr_t (fp32: the schedule's value rounded once)
            torch._fused_adam_([self._flat_p], [self._flat_g], [self._exp_avg], [self._exp_avg_sq], [], [self._step_t],
                               lr=self._lr_t if self.tensor_lr else float(lr), beta1=beta1, beta2=beta2, weight_decay=wd,
                               eps=eps, amsgrad=False, maximize=False, grad_scale=None, found_inf=None)
        else:  # torch.optim.Adam's single-tensor arithmetic
            g = self._flat_g
            if wd != 0:
                g = g.add(self._flat_p, alpha=wd)
            self._exp_avg.lerp_(g, 1 - beta1)
            self._exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
            bc1 = 1 - beta1 ** self._steps
            bc2 = 1 - beta2 ** self._steps
            denom = (self._exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
            self._flat_p.addcdiv_(self._exp_avg, denom, value=-lr / bc1)
        return loss

    def load_state_dict(self, state_dict):
        """torch.optim.Adam's layout in; the moments are copied INTO the flat buffers (the views stay views)."""
        ids = state_dict['param_groups'][0]['params']
        for pid, p in zip(ids, self.param_groups[0]['params']):
            st = state_dict['state'].get(pid)
            if st is None or p not in self.state:
                continue
            self.state[p]['exp_avg'].copy_(st['exp_avg'])
            self.state[p]['exp_avg_sq'].copy_(st['exp_avg_sq'])
            self._steps = int(st['step'])
        self._step_t.fill_(float(self._steps))
        for k, v in state_dict['param_groups'][0].items():
            if k != 'params':
                self.param_groups[0][k] = v
        from . import fused_mlp
        fused_mlp.invalidate_eval_tables()
