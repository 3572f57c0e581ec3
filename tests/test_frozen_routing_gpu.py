"""GPU: frozen-routing gradient parity of the fused HIP path (tests/frozen_routing.py).  One whole train step
(forward, label matching, fused loss, backward) runs as shipped - fused channel-last stacks on the hand-written MFMA
GEMMs, LocalAggregation without the grouped tensor, distinct rows of the nested cylinder crops, closed-form first-layer
backward - while its ReLU masks and max-pool arg-max rows are recorded and every segment's inputs / outputs / incoming
gradient captured; then each segment's plain composition runs alone in fp64 on the same inputs with the same routing
and the same incoming gradient.  Asserted per segment: outputs 1e-5, the gradient it sends back 1e-4, EVERY parameter
gradient tensor 1e-4 (relative L2; tensors whose true gradient is below 1e-3 of the segment's largest one - a conv
bias in front of a batch-statistics BatchNorm - against that floor).  No "no worse than the plain path" escape: a
wrong kernel (round 1's gb_gemm_dgrad_first: SA1's first-layer gradients 56 % off) fails its segment outright.
The CPU twin (tests/test_routing_tape_cpu.py) shows the plain fp32 composition meets the same bounds."""
import copy

import pytest
import torch

from tests.frozen_routing import frozen_routing_train_step, summarise
from tests.seeded import fill_by_key

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SEGMENTS = {"sa1", "sa2", "sa3", "sa4", "stage1", "stage2", "stage3", "stage4", "fp1", "fp2", "graspable",
            "grasp_stage2"}


def _check(net, batch, what):
    with torch.no_grad():
        views = copy.deepcopy(net)(dict(batch))['grasp_top_view_inds'].clone()
    report, entries, loss = frozen_routing_train_step(net, batch, views)
    assert set(report) == SEGMENTS
    table = summarise(report)
    print(what, "loss %.4f, %d routing entries;" % (loss, entries),
          {k: "out %.1e din %.1e dparam %.1e (%s)" % (v[0][0], v[1][0], v[2][0], v[2][1]) for k, v in table.items()})
    n_grads = sum(1 for errs in report.values() for k in errs if k.startswith("dparam/"))
    assert n_grads == len(list(net.parameters())) == 253   # every parameter tensor of GraspBalance, exactly once
    bad = {seg: v for seg, v in table.items() if not (v[0][0] <= 1e-5 and v[1][0] <= 1e-4 and v[2][0] <= 1e-4)}
    assert not bad, bad
    return report


def test_train_step_gradients_toy_size_frozen_routing():
    """The real GraspBalance class with shrunken set-abstraction levels, B = 2 x 3000 points."""
    from tests.test_model_cpu import _tiny_net
    from graspbalance_amd.synthetic import make_training_batch
    batch = make_training_batch(range(2), num_point=3000, num_objects=2, grasp_points_per_object=20, num_view=30,
                                device=DEV)
    _check(fill_by_key(_tiny_net(), seed=9).to(DEV).train(), batch, "toy")


def test_train_step_gradients_full_size_frozen_routing():
    """The real network (SA_SPECS of backbone.py, 300 views) on B = 2 clouds of 20 000 points: the shapes of BASELINE
    configs[3] at half the batch (the fp64 plain composition of stage 2 holds (2,256,4096,64) doubles several times)."""
    from graspbalance_amd.graspbalance import GraspBalance
    from graspbalance_amd.synthetic import make_training_batch
    batch = make_training_batch([0, 1], num_point=20000, device=DEV)
    _check(fill_by_key(GraspBalance(), seed=11).to(DEV).train(), batch, "full size")
