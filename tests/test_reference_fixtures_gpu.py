"""GPU: the HIP path against fixtures produced by RUNNING THE REFERENCE's own python on CPU over the oracle
(tests/golden/make_golden_r2.py; the CPU twin of this file, tests/test_reference_fixtures_cpu.py, shows this repo's
python layers reproduce those fixtures bit for bit on the same oracle path).  What differs here is everything the
product replaces: HIP geometry kernels (indices must still be identical), the channel-last fused SharedMLP stacks on
the hand-written MFMA GEMMs, LocalAggregation without the grouped tensor, distinct-row cylinder crops, fused label
matching and loss kernels — fp32 everywhere, different summation orders.

Tolerances are relative L2 over the stored sample of each tensor.  north_star's bar is 1e-5 on grouped features and
grasp scores: it is asserted wherever the tensor is that well conditioned (eval mode throughout; train mode through
the first set-abstraction levels); deeper train-mode tensors sit behind up to 19 batch-statistic BatchNorm + max-pool
blocks and carry the measured bound next to them."""
import pytest
import torch

from tests.golden import make_golden_r2 as mk
from tests.seeded import assert_errors, fill_by_key
from tests import test_reference_fixtures_cpu as cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_label_matching_matches_reference_gpu(golden):
    """f1 on the fused HIP path (gb_knn1, gb_label_gather, gb_label_finish)."""
    from graspbalance_amd import label_generation as lg
    from tests.seeded import check_summary
    g12 = golden.load("g12_labels")
    ep_in = mk.g12_inputs(DEV)
    assert lg._fusable(ep_in)
    ep = lg.process_grasp_labels(ep_in)
    for k in ('batch_grasp_point', 'batch_grasp_view', 'batch_grasp_view_rot', 'batch_grasp_offset',
              'batch_grasp_tolerance'):
        check_summary(g12, k, ep[k], 1e-6)
    for k in ('batch_grasp_label', 'batch_grasp_view_label'):  # log(u_max / label): device logf vs libm
        check_summary(g12, k, ep[k], 1e-6)
    rot, labels, offsets, tol, ep = lg.match_grasp_view_and_label(ep)
    for k, t in (('top_view_rot', rot), ('top_label', labels), ('top_offset', offsets), ('top_tolerance', tol),
                 ('top_view', ep['batch_grasp_view'])):
        check_summary(g12, k, t, 1e-6)


def test_loss_matches_reference_gpu(golden):
    """f2 on the fused loss kernels (csrc/loss.hip) with the reference's scale prior."""
    from graspbalance_amd import loss
    g13, ep, preds = cases._loss_case(DEV, golden)
    assert loss._fused_loss_ok(ep)
    cases.check_loss_against_reference(g13, ep, preds, 2e-6)


def test_pred_decode_matches_reference_gpu(golden):
    from graspbalance_amd.graspbalance import pred_decode
    from tests.seeded import check_summary
    g14 = golden.load("g14_pred_decode")
    preds = pred_decode(mk.g14_inputs(DEV))
    for i, p in enumerate(preds):
        check_summary(g14, "cloud%d" % i, p, 1e-6)
    # one read-back of the objectness mask + index gathers == the reference's 7 masked selections per cloud, bit for bit
    from graspbalance_amd.graspbalance import _pred_decode_loop
    loop = _pred_decode_loop(mk.g14_inputs(DEV))
    assert len(loop) == len(preds) and all(torch.equal(a, b) for a, b in zip(loop, preds))


def test_object_balance_sampling_matches_reference_gpu(golden):
    """f3 on the HIP path (gb_three_nn 4096 x 1024, gb_three_interpolate, ONE gb_fps_segments launch for all objects
    of all clouds) vs the reference's ObjectBalanceSampling + obs branch run (g20): seed indices bit-exact (asserted
    inside the case), re-sampled features 1e-6, the whole obs=True inference network 1e-5."""
    errs = cases.run_obs_case(DEV, golden.load("g20_obs"))
    _report("obs", errs)
    assert_errors(errs, {"net/top_view_flips": 0, "net/top_view_gap": 0.0, "branch/fp2_xyz": 0.0, "net/fp2_xyz": 0.0,
                         "branch/": 1e-6, "net/up_sample_features": 1e-5, "net/fp2_features": 1e-5,
                         "net/pred_decode": 3e-5}, 1e-5)


def _report(name, errs):
    print(name, {k: "%.2e" % v for k, v in errs.items()})


def _fused_and_plain(run):
    """run() on the fused HIP path and on the plain torch composition over the same HIP geometry kernels."""
    from graspbalance_amd import fused_mlp
    fused = run()
    fused_mlp.set_enabled(False)
    try:
        plain = run()
    finally:
        fused_mlp.set_enabled(True)
    return fused, plain


def _no_worse_than_plain(fused, plain, prefixes, factor, floor):
    """Train-mode tensors behind many batch-statistic BatchNorm + max-pool blocks amplify last-bit differences (both
    fp32 paths sit equally far from the CPU run): there the fused path must be no farther from the reference run than
    `factor` x the plain torch composition is."""
    bad = {k: (fused[k], plain[k]) for k in fused
           if k.startswith(prefixes) and not fused[k] <= factor * plain[k] + floor}
    assert not bad, bad


def test_backbone_matches_reference_gpu(golden):
    """a13 on the HIP path: FPS / ball-query indices identical; eval features and train features through the SA
    levels 1e-5 (2e-5 after the two FP levels: 18 batch-statistic BatchNorms behind it); gradients: no farther from
    the reference run than 2x the plain torch composition on the same kernels (measured: both 5e-3 .. 0.3)."""
    from graspbalance_amd.backbone import Pointnet2Backbone

    def run():
        net = fill_by_key(Pointnet2Backbone(), seed=16).to(DEV)
        return cases.run_backbone_case(net, mk.g16_cloud(DEV), golden.load("g16_backbone"))
    fused, plain = _fused_and_plain(run)
    _report("backbone fused", fused)
    _report("backbone plain", plain)
    assert_errors(fused, {"grad/": 1.0, "train/fp2_features": 2e-5}, 1e-5)
    _no_worse_than_plain(fused, plain, ("grad/", "train/"), 2.0, 1e-5)


def test_heads_match_reference_gpu(golden):
    """a14 on the HIP path (cylinder queries, distinct-row stacks, Conv1d heads): 1e-5."""
    errs = cases.run_heads_case(DEV, golden.load("g17_heads"))
    _report("heads", errs)
    assert_errors(errs, {}, 1e-5)


def test_heads_on_the_fused_stack_match_reference_gpu(golden, monkeypatch):
    """The optional own-kernel execution of the heads (GB_HEADS_FUSED=1: conv+bias in front of BatchNorm folded,
    LinearBias for the last layers) against the same reference-run fixture."""
    from graspbalance_amd import modules
    monkeypatch.setattr(modules, "_HEADS_FUSED", True)
    monkeypatch.setattr(modules, "_GD_FUSED", True)
    errs = cases.run_heads_case(DEV, golden.load("g17_heads"))
    _report("heads (fused stack)", errs)
    assert_errors(errs, {}, 1e-5)


# Bounds = about 3x what MI355X measures (printed by the test; DESIGN.md section 3 keeps the table): relative L2 of
# the HIP path vs the reference's own CPU run.  Eval mode meets north_star's 1e-5 on every tensor.  Train mode
# (batch statistics at B = 2) amplifies last-bit differences level by level - sa1 2.8e-6, sa2 5.6e-5, sa3 6e-4,
# sa4 1.2e-2 - identically for torch's plain composition on the GPU (asserted: fused <= 2 x plain).
NETWORK_TOL = {
    "eval/": 1e-5, "eval/pred_decode": 3e-5, "eval/top_view_flips": 0, "eval/top_view_gap": 0.0,
    "train/top_view_flips": 80, "train/top_view_gap": 2e-2,
    "train/sa1_features": 1e-5, "train/sa2_features": 2e-4, "train/sa3_features": 2e-3,
    "train/sa4_features": 4e-2, "train/fp2_features": 3e-2, "train/objectness_score": 5e-2, "train/view_score": 4e-2,
    "train/grasp_top_view_rot": 1e-6, "train/grasp_": 2e-2,
    "train/loss/": 2e-3, "train/loss/overall_loss": 2e-4,
    "train/batch_grasp_point": 1e-6, "train/batch_grasp_view_label": 1e-6,
    "grad/": 2.0,                       # deep gradients are chaotic for ANY fp32 run; bounded relative to plain below
    "grad/view_estimator.GraspableClasification.conv3.weight": 1e-2,
    "grad/grasp_generator.GraspParameters.conv3.weight": 2e-2, "grad/total_norm": 0.15,
}


def test_whole_network_matches_reference_gpu(golden):
    """a15: GraspBalance on the HIP path vs the reference's own run (N = 4096, B = 2, by-key weights).
    What this test is evidence of: the EVAL tensors and the decode (bounded tightly below).  Its train-mode bounds
    (NETWORK_TOL: `sa4_features` 4e-2, `grad/` 2.0, then "at most twice as far as the plain composition") are GUARDS against
    gross defects only - a free-running fp32 train step amplifies rounding by ~1.2 per conv + BN + ReLU layer whatever
    the implementation.  The evidence for the train step's gradients is the frozen-routing harness
    (tests/test_frozen_routing_gpu.py: every segment's outputs 1e-5, all 253 parameter gradients 1e-4 against fp64)."""
    def run():
        return cases.run_network_case(DEV, golden.load("g15_graspbalance"), cases._prior(golden.load("g13_loss")))
    fused, plain = _fused_and_plain(run)
    _report("network fused", fused)
    _report("network plain", plain)
    assert_errors(fused, NETWORK_TOL, 1e-5)
    per_tensor = {k: v for k, v in fused.items() if k != "grad/total_norm"}  # a scalar: bounded by NETWORK_TOL only
    _no_worse_than_plain(per_tensor, plain, ("train/sa", "train/fp2", "train/grasp_", "train/view", "train/obj", "grad/"),
                         2.0, 1e-5)
