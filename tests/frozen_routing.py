"""Test-only: segment-wise, frozen-routing gradient parity of a WHOLE train step against an fp64 truth.

One train step of the network under test (the fused HIP path; on CPU the plain fp32 composition, to validate the
harness) runs once, forward + loss + backward, exactly as shipped.  While it runs,

  * `tests/routing_tape.RoutingTape` records every discrete decision (ReLU masks, max-pool arg-max rows), and
  * every SEGMENT of the network - the four set-abstraction levels, the four InvResMLP stages, the two
    feature-propagation levels, the graspable / view head, stage 2 (four cylinder-crop stacks + fusion + both depth
    heads) - has its inputs, outputs and the gradient arriving at its outputs captured (its float inputs are handed
    over as aliases, so the gradient the segment itself sends back is captured too).

Then each segment is re-run ALONE as the plain composition in fp64 (`tests/f64_truth`) on the captured inputs, with the
recorded routing replayed and the captured output gradient applied.  By the chain rule the step's gradient is right iff
every segment's (outputs, input gradients, parameter gradients) are right given its inputs and its incoming gradient -
and per segment the comparison is rounding-level: asserted 1e-5 on outputs, 1e-4 on every gradient tensor.

Why segments: a batch-statistics BatchNorm removes the mean of its input, which carries ~1/3 of the energy behind a
ReLU, but not of a perturbation - every conv+BN+ReLU layer of a randomly initialised stack multiplies the RELATIVE
rounding error by ~1.2 whatever the routing (measured on CPU, routing frozen, plain fp32 vs its own fp64 run: sa1
1e-6, sa2 2e-5, sa3 3e-4, sa4 6e-3 over 57 such layers).  A whole-network 1e-4 bound is therefore out of reach of ANY
fp32 implementation; a segment is at most 18 layers deep.
"""
import contextlib
import copy

import torch

from tests import f64_truth
from tests.routing_tape import RoutingTape


class _Rec:
    def __init__(self, name):
        self.name, self.args, self.inputs, self.outputs = name, None, {}, {}
        self.in_grads, self.out_grads, self.tape = {}, {}, (0, 0)


def _alias(rec, key, t):
    """A fresh autograd alias of input `t`: its gradient is what THIS segment sends back, not the sum over consumers."""
    rec.inputs[key] = t.detach()
    if not (t.is_floating_point() and t.requires_grad):
        return t
    a = t.view_as(t)
    a.register_hook(lambda g, rec=rec, key=key: rec.in_grads.__setitem__(key, g.detach().clone()))
    return a


def _watch(rec, key, t):
    rec.outputs[key] = t.detach()
    if t.requires_grad:
        t.register_hook(lambda g, rec=rec, key=key: rec.out_grads.__setitem__(key, g.detach().clone()))


OUT_KEYS_GD = ('objectness_score', 'view_score')
OUT_KEYS_S2 = ('grasp_score_pred', 'grasp_angle_cls_pred', 'grasp_width_pred', 'grasp_tolerance_pred')


@contextlib.contextmanager
def probed(net, tape):
    """Wrap the segments of `net` (a GraspBalance) for one forward pass; yields {name: _Rec}."""
    from graspbalance_amd import drp as drp_mod
    recs, undo = {}, []
    fe = net.view_estimator.FeatureExtraction

    def wrap_positional(owner, name, out_index):
        inner = owner.forward

        def forward(*args):
            rec = recs[name] = _Rec(name)
            start = len(tape.items)
            args = tuple(_alias(rec, i, a) if torch.is_tensor(a) else a for i, a in enumerate(args))
            rec.args = args
            out = inner(*args)
            _watch(rec, "out", out if out_index is None else out[out_index])
            rec.tape = (start, len(tape.items))
            return out
        owner.forward = forward
        undo.append(lambda: owner.__dict__.pop("forward"))

    for level in (1, 2, 3, 4):
        wrap_positional(getattr(fe, "sa%d" % level), "sa%d" % level, 1)
    wrap_positional(fe.fp1, "fp1", None)
    wrap_positional(fe.fp2, "fp2", None)

    inner_stage = drp_mod.run_stage
    stage_of = {id(getattr(fe, "InvResMLP_blocks%d" % l)): "stage%d" % l for l in (1, 2, 3, 4)}

    def run_stage(blocks, p, f):
        name = stage_of.get(id(blocks))
        if name is None:
            return inner_stage(blocks, p, f)
        rec = recs[name] = _Rec(name)
        start = len(tape.items)
        f = _alias(rec, "f", f)
        rec.inputs["p"] = p.detach()
        p_out, f_out = inner_stage(blocks, p, f)
        _watch(rec, "out", f_out)
        rec.tape = (start, len(tape.items))
        return p_out, f_out
    drp_mod.run_stage = run_stage
    undo.append(lambda: setattr(drp_mod, "run_stage", inner_stage))

    gd = net.view_estimator.GraspableClasification
    inner_gd = gd.forward

    def gd_forward(seed_xyz, seed_features, end_points, record=True):
        rec = recs["graspable"] = _Rec("graspable")
        start = len(tape.items)
        rec.inputs["seed_xyz"] = seed_xyz.detach()
        out = inner_gd(seed_xyz, _alias(rec, "seed_features", seed_features), end_points, record)
        for k in OUT_KEYS_GD:
            _watch(rec, k, out[k])
        rec.tape = (start, len(tape.items))
        return out
    gd.forward = gd_forward
    undo.append(lambda: gd.__dict__.pop("forward"))

    gen = net.grasp_generator
    inner_gen = gen.forward

    def gen_forward(end_points):
        rec = recs["grasp_stage2"] = _Rec("grasp_stage2")
        start = len(tape.items)
        end_points['fp2_features'] = _alias(rec, "fp2_features", end_points['fp2_features'])
        rec.args = dict(end_points)   # shallow: the entries as they are on entry (label matching re-binds keys later)
        out = inner_gen(end_points)
        for k in OUT_KEYS_S2:
            _watch(rec, k, out[k])
        rec.tape = (start, len(tape.items))
        return out
    gen.forward = gen_forward
    undo.append(lambda: gen.__dict__.pop("forward"))
    try:
        yield recs
    finally:
        for u in reversed(undo):
            u()


def _d(t):
    return t.detach().double() if torch.is_tensor(t) and t.is_floating_point() else t


def _leaf(t):
    return t.detach().double().requires_grad_(True)


def _truth_segment(name, rec, net64, views):
    """Run segment `name` of the fp64 model on the captured inputs -> ({out key: tensor}, {in key: leaf})."""
    from graspbalance_amd import drp as drp_mod
    fe = net64.view_estimator.FeatureExtraction
    if name.startswith("sa") or name.startswith("fp"):
        leaves, args = {}, []
        for i, a in enumerate(rec.args):
            if torch.is_tensor(a) and a.is_floating_point() and i in rec.in_grads:
                leaves[i] = _leaf(a)
                args.append(leaves[i])
            else:
                args.append(_d(a))
        out = getattr(fe, name)(*args)
        return {"out": out[1] if name.startswith("sa") else out}, leaves
    if name.startswith("stage"):
        f = _leaf(rec.inputs["f"])
        _, out = drp_mod.run_stage(getattr(fe, "InvResMLP_blocks" + name[-1]), _d(rec.inputs["p"]), f)
        return {"out": out}, {"f": f}
    if name == "graspable":
        gd = net64.view_estimator.GraspableClasification
        gd._top_view = lambda vs: (torch.gather(vs, 2, views.to(vs.device).unsqueeze(-1)).squeeze(-1), views.to(vs.device))
        f = _leaf(rec.inputs["seed_features"])
        out = gd(_d(rec.inputs["seed_xyz"]), f, {})
        return {k: out[k] for k in OUT_KEYS_GD}, {"seed_features": f}
    if name == "grasp_stage2":
        ep = dict(rec.args)
        f = _leaf(rec.inputs["fp2_features"])
        ep['fp2_features'] = f
        ep['input_xyz'] = _d(ep['input_xyz'])
        with f64_truth.double_stage2_inputs():
            out = net64.grasp_generator(ep)
        return {k: out[k] for k in OUT_KEYS_S2}, {"fp2_features": f}
    raise AssertionError(name)


def _segment_module(net, name):
    fe = net.view_estimator.FeatureExtraction
    if name.startswith("stage"):
        return getattr(fe, "InvResMLP_blocks" + name[-1])
    if name == "graspable":
        return net.view_estimator.GraspableClasification
    if name == "grasp_stage2":
        return net.grasp_generator
    return getattr(fe, name)


def frozen_routing_train_step(base, batch, views, prior=None, tamper=None):
    """One train step of a copy of `base` (whatever path is active: fused HIP on the GPU, plain on CPU) with routing
    recorded and segments probed, then every segment's fp64 truth.  -> {segment: {"out/<key>" | "din/<key>" |
    "dparam/<name>": relative error}}, number of routing entries, loss."""
    from graspbalance_amd import fused_mlp
    from graspbalance_amd.loss import get_loss
    net = copy.deepcopy(base)
    if tamper is not None:
        tamper(net)    # negative control: break the implementation under test on purpose

    def force(n):
        n.view_estimator.GraspableClasification._top_view = \
            lambda vs: (torch.gather(vs, 2, views.to(vs.device).unsqueeze(-1)).squeeze(-1), views.to(vs.device))
    force(net)
    tape = RoutingTape()
    with tape.recording(), probed(net, tape) as recs:
        ep = net(dict(batch))
    loss, ep = get_loss(ep) if prior is None else get_loss(ep, prior)
    # as train.Trainer runs it: on the GPU the few-row weight gradients are recorded and launched together at the end
    dev = batch['point_clouds'].device
    with (fused_mlp.WgradQueue(dev) if (dev.type == "cuda" and fused_mlp._WGRAD_GROUP) else contextlib.nullcontext()):
        loss.backward()
    net64 = f64_truth.double_model(base)
    report = {}
    was = fused_mlp._ENABLED
    fused_mlp.set_enabled(False)
    try:
        with f64_truth.torch_geometry():
            for name, rec in recs.items():
                assert rec.out_grads, "segment %s received no gradient" % name
                with tape.replaying():
                    tape.pos = rec.tape[0]
                    outs, leaves = _truth_segment(name, rec, net64, views)
                    assert tape.pos == rec.tape[1], (name, "replay used entries", rec.tape[0], tape.pos, rec.tape[1])
                keys = [k for k in outs if k in rec.out_grads]
                torch.autograd.backward([outs[k] for k in keys], [rec.out_grads[k].double() for k in keys])
                errs = {}
                for k, t in outs.items():
                    errs["out/%s" % k] = f64_truth.rel(rec.outputs[k], t)
                for k, leaf in leaves.items():
                    errs["din/%s" % k] = f64_truth.rel(rec.in_grads[k], leaf.grad)
                mod, mod64 = _segment_module(net, name), _segment_module(net64, name)
                truth = dict(mod64.named_parameters())
                top = max(float(g.grad.norm()) for g in truth.values() if g.grad is not None)
                for k, p in mod.named_parameters():
                    g64 = truth[k].grad
                    if g64 is None:
                        assert p.grad is None or float(p.grad.abs().max()) == 0.0, (name, k)
                        continue
                    errs["dparam/%s" % k] = float((p.grad.double() - g64).norm() / max(float(g64.norm()), 1e-3 * top))
                report[name] = errs
                net64.zero_grad(set_to_none=True)
    finally:
        fused_mlp.set_enabled(was)
    return report, len(tape.items), float(loss.detach())


def summarise(report):
    """{segment: (worst output error, worst input-gradient error, worst parameter-gradient error (name))}"""
    out = {}
    for seg, errs in report.items():
        def worst(prefix):
            sel = {k: v for k, v in errs.items() if k.startswith(prefix)}
            if not sel:
                return (0.0, "-")
            k = max(sel, key=sel.get)
            return (sel[k], k[len(prefix):])
        out[seg] = (worst("out/"), worst("din/"), worst("dparam/"))
    return out
