"""CPU: the frozen-routing harness itself (tests/routing_tape.py, tests/frozen_routing.py) on the oracle path, with the
plain fp32 composition as the implementation under test: its recorded ReLU masks / max-pool arg-max rows are replayed
by the fp64 truth segment by segment, and every output and EVERY gradient tensor of a whole train step then agrees to
rounding level - the property tests/test_frozen_routing_gpu.py asserts for the fused HIP path.  Free-running, the same
two runs sit 0.1-0.4 apart in the deep gradients (tests/test_f64_truth_cpu.py)."""
import copy

import pytest
import torch

from tests.frozen_routing import frozen_routing_train_step, summarise
from tests.seeded import fill_by_key
from tests.test_model_cpu import _tiny_batch, _tiny_net

SEGMENTS = {"sa1", "sa2", "sa3", "sa4", "stage1", "stage2", "stage3", "stage4", "fp1", "fp2", "graspable",
            "grasp_stage2"}


@pytest.fixture()
def cpu(monkeypatch):
    from tests import cpu_backend
    cpu_backend.install(monkeypatch)


def test_frozen_routing_makes_fp32_and_fp64_agree_segment_by_segment(cpu):
    net = fill_by_key(_tiny_net(), seed=9).train()
    batch = _tiny_batch()
    with torch.no_grad():
        views = copy.deepcopy(net)(dict(batch))['grasp_top_view_inds'].clone()
    report, entries, loss = frozen_routing_train_step(net, batch, views)
    assert entries > 100                       # every ReLU and every max-pool of 19 blocks + heads was recorded
    assert set(report) == SEGMENTS
    table = summarise(report)
    print({k: "out %.1e din %.1e dparam %.1e (%s)" % (v[0][0], v[1][0], v[2][0], v[2][1]) for k, v in table.items()})
    n_grads = sum(1 for errs in report.values() for k in errs if k.startswith("dparam/"))
    assert n_grads == len(list(net.parameters()))          # each parameter tensor belongs to exactly one segment
    for seg, (out, din, dparam) in table.items():
        assert out[0] <= 1e-5 and din[0] <= 1e-4 and dparam[0] <= 1e-4, (seg, out, din, dparam)


def test_harness_flags_a_one_percent_gradient_error(cpu):
    """Negative control: one parameter gradient of the implementation under test scaled by 1.01 (SA1's first conv) and
    one activation gradient by 1.01 (what stage 3 sends back) - exactly those two entries leave the bound."""
    net = fill_by_key(_tiny_net(), seed=9).train()
    batch = _tiny_batch()
    with torch.no_grad():
        views = copy.deepcopy(net)(dict(batch))['grasp_top_view_inds'].clone()

    def tamper(n):
        fe = n.view_estimator.FeatureExtraction
        fe.sa1.mlp_module.layer0.conv.weight.register_hook(lambda g: g * 1.01)
        first = fe.InvResMLP_blocks3[0].convs.convs[0][0]
        first.register_full_backward_hook(lambda m, gin, gout: tuple(None if g is None else g * 1.01 for g in gin))
    report, _, _ = frozen_routing_train_step(net, batch, views, tamper=tamper)
    over = {(seg, k) for seg, errs in report.items() for k, v in errs.items() if v > 1e-4}
    assert ("sa1", "dparam/mlp_module.layer0.conv.weight") in over, over
    assert any(seg == "stage3" and k.startswith("din/") for seg, k in over), over
    assert all(seg in ("sa1", "stage3") for seg, _ in over), over
