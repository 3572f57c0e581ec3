"""Test-only: an fp64 "truth" run of the PLAIN composition (the reference's op sequence: group -> cat -> conv -> BN ->
ReLU -> max ...), used to judge fp32 implementations by their distance from it instead of from each other.

Inside ``torch_geometry()`` the wrapper functions of pointnet2_utils / modified_net_tools are replaced:
  * index-producing ops (FPS, ball / cylinder query, three_nn) run the bound fp32 implementation on fp32 casts of
    their inputs - the inputs are fp32-representable point coordinates, so the indices are THE indices of the fp32
    paths (frozen routing); three_nn's distances are then recomputed in the input dtype from those indices;
  * value ops (gather, group, three_interpolate) are differentiable torch indexing in the input dtype.
``double_model()`` deep-copies a module to fp64 and keeps the pieces of the pipeline that are fp32 *data* in the
product (view templates -> rotations, label matching) in fp32, cast to fp64 at the seam.
"""
import contextlib
import copy

import torch


def _gather(features, idx):
    B, C, _ = features.shape
    return torch.gather(features, 2, idx.long().unsqueeze(1).expand(-1, C, -1))


def _group(features, idx):
    B, C, _ = features.shape
    m, ns = idx.shape[1], idx.shape[2]
    return torch.gather(features, 2, idx.long().reshape(B, 1, m * ns).expand(-1, C, -1)).view(B, C, m, ns)


def _three_interpolate(features, idx, weight):
    B, C, _ = features.shape
    n = idx.shape[1]
    g = torch.gather(features, 2, idx.long().reshape(B, 1, n * 3).expand(-1, C, -1)).view(B, C, n, 3)
    return (g * weight.unsqueeze(1)).sum(-1)


@contextlib.contextmanager
def torch_geometry():
    from graspbalance_amd import graspbalance, modules, pointnet2_utils as pu
    from graspbalance_amd.modified_net_tools import group, subsample, upsampling
    orig = {"fps": pu.furthest_point_sample, "ball": pu.ball_query, "cyl": pu.cylinder_query, "nn": pu.three_nn,
            "pb_ball": group.ball_query, "pb_fps": subsample.furthest_point_sample, "pb_nn": upsampling.three_nn}

    def f32(t):
        return t.float().contiguous()

    def three_nn_via(fn):
        def three_nn(unknown, known):
            _, idx = fn(f32(unknown), f32(known))
            B, n, _ = unknown.shape
            nb = torch.gather(known, 1, idx.long().reshape(B, n * 3, 1).expand(-1, -1, 3)).view(B, n, 3, 3)
            return (nb - unknown.unsqueeze(2)).pow(2).sum(-1).sqrt(), idx
        return three_nn

    patches = [
        (pu, "furthest_point_sample", lambda xyz, m: orig["fps"](f32(xyz), m)),
        (pu, "ball_query", lambda r, ns, xyz, new_xyz: orig["ball"](r, ns, f32(xyz), f32(new_xyz))),
        (pu, "cylinder_query", lambda r, hmin, hmax, ns, xyz, new_xyz, rot:
            orig["cyl"](r, hmin, hmax, ns, f32(xyz), f32(new_xyz), f32(rot))),
        (pu, "three_nn", three_nn_via(orig["nn"])),
        (pu, "gather_operation", _gather), (pu, "grouping_operation", _group),
        (pu, "three_interpolate", _three_interpolate),
        (group, "ball_query", lambda r, ns, xyz, new_xyz: orig["pb_ball"](r, ns, f32(xyz), f32(new_xyz))),
        (group, "grouping_operation", _group), (group, "gather_operation", _gather),
        (subsample, "furthest_point_sample", lambda xyz, m: orig["pb_fps"](f32(xyz), m)),
        (subsample, "gather_operation", _gather),
        (upsampling, "three_nn", three_nn_via(orig["pb_nn"])), (upsampling, "three_interpolate", _three_interpolate),
        (modules, "furthest_point_sample", lambda xyz, m: orig["fps"](f32(xyz), m)),
        (graspbalance, "three_nn", three_nn_via(orig["nn"])), (graspbalance, "three_interpolate", _three_interpolate),
    ]
    saved = [(mod, name, getattr(mod, name)) for mod, name, _ in patches]
    for mod, name, fn in patches:
        setattr(mod, name, fn)
    try:
        yield
    finally:
        for mod, name, fn in saved:
            setattr(mod, name, fn)


def double_model(net):
    """fp64 copy of `net`; a GraspBalance keeps the fp32 data seams of the product (rotations from the fp32 view
    templates; label matching on the fp32 labels) and casts their results to fp64."""
    net64 = copy.deepcopy(net).double()
    if hasattr(net64, "grasp_generator"):
        net64.grasp_generator.fused_cylinder = False  # the fused multi-query op is fp32-only: 16 plain queries
    gd = getattr(getattr(net64, "view_estimator", None), "GraspableClasification", None)
    if gd is not None:
        inner = gd.forward

        def forward(seed_xyz, seed_features, end_points, record=True):
            out = inner(seed_xyz, seed_features, end_points, record)
            if record:
                out['grasp_top_view_rot'] = out['grasp_top_view_rot'].double()
            return out
        gd.forward = forward
    return net64


@contextlib.contextmanager
def double_stage2_inputs():
    """Training mode: stage 2 takes its seeds and rotations from the (fp32) label matching - cast at the seam."""
    from graspbalance_amd import graspbalance
    inner = graspbalance._stage2_inputs

    def stage2_inputs(end_points, is_training):
        seed_xyz, rot, end_points = inner(end_points, is_training)
        for k in ('batch_grasp_view_label', 'batch_grasp_label', 'batch_grasp_offset', 'batch_grasp_tolerance',
                  'batch_grasp_label_all', 'batch_grasp_offset_all'):
            if k in end_points:  # the loss then runs in fp64 on the same label values
                end_points[k] = end_points[k].double()
        return seed_xyz.double(), rot.double(), end_points
    graspbalance._stage2_inputs = stage2_inputs
    try:
        yield
    finally:
        graspbalance._stage2_inputs = inner


def rel(a, b):
    """relative L2 distance of a from b (both moved to fp64 on b's device)."""
    b = b.detach().double()
    return float((a.detach().double().to(b.device) - b).norm() / (b.norm() + 1e-300))


def elem(a, b):
    """Largest ELEMENT-wise deviation of a from b, relative to max(1, max|b|) (VERDICT round 4 #5a: a norm over half a
    million elements does not see a handful of wrong ones; this does)."""
    b = b.detach().double()
    d = (a.detach().double().to(b.device) - b).abs()
    return float(d.max() / max(1.0, float(b.abs().max()))) if d.numel() else 0.0
