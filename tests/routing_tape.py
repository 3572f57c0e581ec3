"""Test-only: record the DISCRETE routing of one forward pass - every ReLU mask, every max-pool arg-max - and replay it
in another implementation of the same network, so that both evaluate the same piecewise-linear branch of the function
and their gradients can be compared at rounding level (1e-4) instead of through "no worse than another fp32 run".

Why: behind 19 batch-statistic BatchNorm + max-pool blocks one last-bit difference re-routes a max-pool / ReLU and
moves a gradient by O(1/sqrt(rows)); free-running fp32 paths therefore sit 0.1-0.4 apart in deep gradients and only
relative bounds were possible - bounds that would not catch a wrong kernel whose error stays under that chaos floor
(round 1's gb_gemm_dgrad_first was 56 % off and passed).  With the routing frozen the function is smooth and the
gradient of the fused HIP path must equal the fp64 gradient of the plain composition, tensor by tensor.

Direction: the routing is taken from the path under test (the fused fp32 run: `graspbalance_amd.fused_mlp.
routing_observer` hands over pre-BatchNorm outputs + (a, b) tables and arg-max rows; ReLUs that run through torch are
caught at torch.nn.functional.relu) and forced on the fp64 truth (plain composition: every ReLU goes through
torch.nn.functional.relu, every max-pool through pytorch_utils.max_over_samples - both replaced while replaying).  Both
runs then differentiate the SAME smooth function at the same point; which of two near-tied routes is "right" does not
matter for that comparison.  Entries are consumed strictly in program order and `done()` asserts that the replay used
every one of them, so a fused node that skipped or reordered a layer fails loudly.
"""
import contextlib

import torch
import torch.nn.functional as F


def _rows_to(x, t):
    """t (P, C) rows ordered (b, *spatial) -> the layout of x (B, C, *spatial)."""
    B, C = x.shape[0], x.shape[1]
    sp = tuple(x.shape[2:])
    assert t.shape == (x.numel() // C, C), (tuple(t.shape), tuple(x.shape))
    return t.view(B, *sp, C).movedim(-1, 1)


class RoutingTape:
    def __init__(self):
        self.items, self.pos, self._cyl = [], 0, {}

    # ---- recording --------------------------------------------------------------------------------------------------
    def observe(self, kind, **kw):
        """graspbalance_amd.fused_mlp.routing_observer."""
        if kind == "cyl_rows":
            self._cyl[id(kw["rowset"])] = (kw["rowset"], kw["sorted"].clone(), kw["idx"].clone())
        elif kind == "local_agg":
            # conv -> BN -> ReLU -> max over the neighbours, y never stored: the ReLU matters at the arg-max row only
            self.items.append(("relu", None))
            self.items.append(("pool", kw["arg"].long().clone(), kw["out"] > 0))
        elif kind == "stack":
            Ys, abs_, out, arg, rows = kw["Ys"], kw["abs"], kw["out"], kw["arg"], kw["rows"]
            full = self._full_rows(rows) if rows is not None else None
            for l in range(len(Ys) - 1):
                n = Ys[l].shape[1]
                mask = (Ys[l] * abs_[l][:n] + abs_[l][n:2 * n]) > 0   # relu(a*y + b), one rounding per operation
                self.items.append(("relu", mask if full is None else mask[full]))
            if arg is None:
                if kw["relu_last"]:
                    self.items.append(("relu", out > 0))
                return
            self.items.append(("relu", None))
            if rows is None:
                self.items.append(("pool", arg.long().clone(), out > 0))
            else:
                # arg = absolute distinct-row index; in the full batch the crop (seed r, depth d) holds that row at its
                # first sample k whose distinct row is arg
                rd = rows.R * rows.D
                per_crop = full.view(rd, -1)                                        # (R*D, ns) distinct row of sample k
                hit = per_crop.unsqueeze(-1) == arg.long().unsqueeze(1)             # (R*D, ns, C)
                assert bool(hit.any(dim=1).all()), "an arg-max row is not a member of its crop"
                self.items.append(("pool", hit.float().argmax(dim=1), out > 0))
        else:
            raise AssertionError(kind)

    def _full_rows(self, rows):
        """Distinct-row index of every row (b, seed, depth, sample) of the full batch of a crop stack."""
        rowset, srt, idx = self._cyl[id(rows)]
        D, B, m, ns = idx.shape
        R, W = B * m, D * ns
        ids = idx.permute(1, 2, 0, 3).reshape(R, W).long()
        srt = srt.view(R, W).long().clone()
        srt[torch.arange(W, device=srt.device).unsqueeze(0) >= rowset.cnt.long().unsqueeze(1)] = 1 << 40
        pos = torch.searchsorted(srt, ids)
        assert torch.equal(torch.gather(srt, 1, pos.clamp(max=W - 1)), ids)
        return (rowset.off.long().unsqueeze(1) + pos).reshape(-1)

    # ---- the two modes ----------------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def recording(self):
        from graspbalance_amd import fused_mlp, pytorch_utils
        relu, pool = F.relu, pytorch_utils.max_over_samples

        def rec_relu(x, inplace=False):
            out = relu(x, inplace=inplace)
            self.items.append(("relu_native", (out > 0).clone()))
            return out

        def rec_pool(x, keepdim=False):
            val, arg = torch.max(x, dim=-1, keepdim=keepdim)
            self.items.append(("pool_native", arg.clone(), None))
            return val
        F.relu, pytorch_utils.max_over_samples, fused_mlp.routing_observer = rec_relu, rec_pool, self.observe
        try:
            yield self
        finally:
            F.relu, pytorch_utils.max_over_samples, fused_mlp.routing_observer = relu, pool, None

    def _next(self, *kinds):
        assert self.pos < len(self.items), "the replayed network asks for more routing decisions than were recorded"
        item = self.items[self.pos]
        assert item[0] in kinds, ("routing entry %d" % self.pos, item[0], kinds)
        self.pos += 1
        return item

    @contextlib.contextmanager
    def replaying(self):
        from graspbalance_amd import pytorch_utils
        relu, pool = F.relu, pytorch_utils.max_over_samples
        self.pos = 0

        def play_relu(x, inplace=False):
            kind, mask = self._next("relu", "relu_native")
            if mask is None:
                return x
            if kind == "relu":
                mask = _rows_to(x, mask)
            assert mask.shape == x.shape, (tuple(mask.shape), tuple(x.shape))
            return x * mask.to(device=x.device, dtype=x.dtype)

        def play_pool(x, keepdim=False):
            kind, arg, pos = self._next("pool", "pool_native")
            if kind == "pool":
                grouped = x.reshape(x.shape[0], x.shape[1], -1, x.shape[-1])      # (B, C, groups, ns)
                val = torch.gather(grouped, 3, _rows_to(grouped[..., 0], arg).unsqueeze(-1).to(x.device)).squeeze(-1)
                val = val * _rows_to(val, pos).to(device=x.device, dtype=x.dtype)
                val = val.reshape(x.shape[:-1])
            else:
                val = torch.gather(x, x.dim() - 1, (arg if keepdim else arg.unsqueeze(-1)).to(x.device)).squeeze(-1)
            return val.unsqueeze(-1) if keepdim else val
        F.relu, pytorch_utils.max_over_samples = play_relu, play_pool
        try:
            yield self
        finally:
            F.relu, pytorch_utils.max_over_samples = relu, pool

    def done(self):
        assert self.pos == len(self.items), "replay consumed %d of %d routing entries" % (self.pos, len(self.items))
        return len(self.items)


def grad_errors(named_params, named_params64):
    """{name: ||g - g64|| / max(||g64||, floor)} per parameter tensor; floor = 1e-6 x the largest gradient norm of the
    model, so that the exactly-zero gradients (a convolution bias in front of a batch-statistics BatchNorm) are judged
    absolutely instead of dividing by zero."""
    truth = {k: p.grad for k, p in named_params64}
    top = max(float(g.norm()) for g in truth.values() if g is not None)
    errs = {}
    for k, p in named_params:
        g64 = truth[k]
        if g64 is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        errs[k] = float((p.grad.double() - g64.to(p.grad.device)).norm() / max(float(g64.norm()), 1e-6 * top))
    return errs
