"""GPU: the HIP path judged by its distance from an fp64 run of the plain composition (tests/f64_truth.py), next to the
distance of an independent fp32 implementation of the same composition from that truth:

  * full size (BASELINE configs[2]: the real GraspBalance, 20 000-point clouds), eval mode: HIP fused path vs the CPU
    oracle path vs the truth - indices identical, every stage within 1e-5 of the truth and no farther from it than
    twice the CPU path, the same bits on every run;
  * the exact configs[1] module (one SA layer, npoint 1024, r 0.04, ns 32, MLP [3,64,128]) on 20 000-point clouds;
  * train mode (batch statistics, backward): fused HIP path vs plain torch composition on the GPU vs the truth, with
    the top-view arg-max frozen - the fused path may be at most twice as far from the truth as the plain path is;
  * a configs[4]-shaped step (N = 50 000 points): FPS / ball-query indices against the oracle, finite loss.
"""
import copy

import pytest
import torch

from tests import f64_truth
from tests.f64_truth import rel
from tests.seeded import fill_by_key

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VALUE_KEYS = ('sa1_features', 'sa2_features', 'sa3_features', 'sa4_features', 'fp2_features', 'objectness_score',
              'view_score', 'grasp_score_pred', 'grasp_angle_cls_pred', 'grasp_width_pred', 'grasp_tolerance_pred')


def _force(net, views):
    def top_view(vs):
        idx = views.to(vs.device)
        return torch.gather(vs, 2, idx.unsqueeze(-1)).squeeze(-1), idx
    net.view_estimator.GraspableClasification._top_view = top_view


def _truth_forward(net_gpu, batch64, views, train=False):
    from graspbalance_amd import fused_mlp
    net64 = f64_truth.double_model(net_gpu)
    _force(net64, views)
    fused_mlp.set_enabled(False)
    try:
        with f64_truth.torch_geometry(), f64_truth.double_stage2_inputs():
            return net64, net64(batch64)
    finally:
        fused_mlp.set_enabled(True)


def test_full_size_eval_forward_hip_vs_oracle_path_vs_f64_truth(monkeypatch):
    """BASELINE configs[2] at its stated shape (B = 4 clouds of 20 000 points, real SA_SPECS, by-key weights and running
    statistics)."""
    from graspbalance_amd.graspbalance import GraspBalance
    from graspbalance_amd.scene import make_batch
    from tests import cpu_backend
    net = fill_by_key(GraspBalance(is_training=False), seed=21).eval()
    clouds = torch.from_numpy(make_batch([0, 1, 2, 3], 20000))
    gpu = copy.deepcopy(net).to(DEV)
    with torch.no_grad():
        free = gpu({'point_clouds': clouds.to(DEV)})
    with monkeypatch.context() as mp:
        cpu_backend.install(mp)
        with torch.no_grad():
            cpu = net({'point_clouds': clouds})
    views = cpu['grasp_top_view_inds']
    flips = int((free['grasp_top_view_inds'].cpu() != views).sum())
    _force(gpu, views)
    with torch.no_grad():
        got = gpu({'point_clouds': clouds.to(DEV)})
        again = gpu({'point_clouds': clouds.to(DEV)})
        _, truth = _truth_forward(gpu, {'point_clouds': clouds.double().to(DEV)}, views)
    # the eval forward is bit-reproducible (split reductions add their partial products in a fixed order; before
    # that, fp32 atomics made the grasp heads wander between 4e-6 and 2e-5 of the truth from run to run)
    for k in VALUE_KEYS:
        assert torch.equal(got[k], again[k]), k
    for k in ('sa1_inds', 'sa2_inds', 'fp2_inds'):
        assert torch.equal(got[k].cpu(), cpu[k]) and torch.equal(got[k], truth[k]), k
    for k in ('sa1_xyz', 'sa2_xyz', 'sa3_xyz', 'sa4_xyz', 'fp2_xyz'):
        assert torch.equal(got[k].cpu(), cpu[k]), k
    # judged per cloud (eval mode: the clouds are independent).  north_star's bar - features and grasp scores within
    # 1e-5 - is asserted wherever fp32 itself can meet it: where the CPU oracle path (the reference's own arithmetic,
    # torch fp32 over the oracle) is itself farther than 2.5e-6 from the truth, the HIP path may be up to 4 x as far
    # as the CPU path.  That clause is needed for ONE spot of this by-key random network: the sigmoid gate of
    # stage 2 (graspbalance.py:113-116) multiplies seed features of rms 1.7e3 (eval-mode BatchNorm with random running
    # statistics does not normalise them), and on clouds 1 and 2 that turns the 1e-6 of fp2_features into 1e-5 / 1e-4
    # in the four grasp tensors for BOTH fp32 paths (tools/eval_b4_probe.py: the crop stacks feeding the same sum are
    # at 1.8e-7); clouds 0 and 3 and every backbone / stage-1 tensor of all four clouds meet 1e-5 outright.
    report, outright = {}, 0
    B = clouds.shape[0]
    for k in VALUE_KEYS:
        for i in range(B):
            e_hip, e_cpu = rel(got[k][i], truth[k][i]), rel(cpu[k][i], truth[k][i])
            report[(k, i)] = (e_hip, e_cpu)
            outright += e_hip <= 1e-5
    print("full-size eval: top-view flips (free HIP vs CPU arg-max) %d of %d;" % (flips, views.numel()),
          {"%s[%d]" % k: "hip %.1e cpu %.1e" % v for k, v in report.items()})
    for (k, i), (e_hip, e_cpu) in report.items():
        if k.startswith("grasp_"):
            assert e_hip <= max(1e-5, 4.0 * e_cpu), (k, i, e_hip, e_cpu)   # per cloud the ratio is noisy (measured 0.7 .. 2.9)
        else:
            assert e_hip <= 1e-5, (k, i, e_hip)           # backbone and stage 1: 1e-5 outright on every cloud
    for k in VALUE_KEYS:                                  # the whole batch: never more than twice the CPU path's distance
        e_hip, e_cpu = rel(got[k], truth[k]), rel(cpu[k], truth[k])
        assert e_hip <= 2.0 * e_cpu + 2e-7, (k, e_hip, e_cpu)
    assert outright >= len(report) - 8, (outright, len(report))   # at most the 4 grasp tensors of 2 clouds use the clause
    assert flips <= 8, flips


def test_config3_stated_initialisation_every_tensor_within_1e_5():
    """SURVEY section 8d C3 exactly as stated: B = 4 clouds of 20 000 points (make_scene seeds 0..3), eval mode, weights
    from ``torch.manual_seed(1234)`` default initialisation (running statistics 0 / 1).  north_star's bar - features AND
    grasp scores within 1e-5 of the fp64 truth - asserted outright on every tensor of every cloud, no clause relative to
    another fp32 path (the by-key random network of the test above needs one for its un-normalised seed features; a
    network as initialised does not).  The truth replays the HIP run's top-view arg-max (a discrete choice)."""
    from graspbalance_amd.graspbalance import GraspBalance
    from graspbalance_amd.scene import make_batch
    torch.manual_seed(1234)
    gpu = GraspBalance(is_training=False).to(DEV).eval()
    clouds = torch.from_numpy(make_batch([0, 1, 2, 3], 20000)).to(DEV)
    with torch.no_grad():
        got = gpu({'point_clouds': clouds})
        views = got['grasp_top_view_inds'].clone()
        _, truth = _truth_forward(gpu, {'point_clouds': clouds.double()}, views.cpu())
    for k in ('sa1_inds', 'sa2_inds', 'fp2_inds'):
        assert torch.equal(got[k], truth[k]), k
    report = {(k, i): rel(got[k][i], truth[k][i]) for k in VALUE_KEYS for i in range(clouds.shape[0])}
    print("configs[2]/C3 at the stated initialisation:", {"%s[%d]" % k: "%.1e" % v for k, v in report.items()})
    worst = max(report.items(), key=lambda kv: kv[1])
    assert worst[1] <= 1e-5, worst
    # ... and element by element (north_star: "grouped features and grasp scores within 1e-5 fp32"): no single value of any
    # tensor of any cloud is further than 1e-5 * max(1, max|truth|) from the truth
    report_e = {(k, i): f64_truth.elem(got[k][i], truth[k][i]) for k in VALUE_KEYS for i in range(clouds.shape[0])}
    print("  element-wise:", {"%s[%d]" % k: "%.1e" % v for k, v in report_e.items()})
    worst_e = max(report_e.items(), key=lambda kv: kv[1])
    assert worst_e[1] <= 1e-5, worst_e
    # the free arg-max of the truth picks the same views (near-ties aside)
    free64 = torch.max(truth['view_score'], dim=2)[1]
    assert int((free64 != views).sum()) <= 4


@pytest.mark.parametrize("B", [1, 4])
def test_config1_sa_layer_on_20k_cloud(orc, B):
    """BASELINE configs[1]: PointnetSAModuleVotes(npoint=1024, radius=0.04, nsample=32, mlp=[0,64,128]) forward on
    20 000-point clouds, train mode: FPS / ball-query indices == oracle; features vs the fp64 truth."""
    from graspbalance_amd import fused_mlp, pointnet2_modules as pm, pointnet2_utils as pu
    from graspbalance_amd.scene import make_batch
    sa = fill_by_key(pm.PointnetSAModuleVotes(npoint=1024, radius=0.04, nsample=32, mlp=[0, 64, 128], use_xyz=True,
                                              normalize_xyz=True), seed=1).to(DEV).train()
    plain, sa64 = copy.deepcopy(sa), copy.deepcopy(sa).double()
    clouds = torch.from_numpy(make_batch(list(range(10, 10 + B)), 20000))
    xyz = clouds.to(DEV)
    new_xyz, feats, inds = sa(xyz)
    want_inds = orc.furthest_point_sampling(clouds, 1024, orc.FPS_SKIP_NEAR_ORIGIN | orc.FPS_TIE_TREE512)
    assert torch.equal(inds.cpu(), want_inds)
    want_ball = orc.ball_query(new_xyz.cpu(), clouds, 0.04, 32)
    assert torch.equal(pu.ball_query(0.04, 32, xyz, new_xyz).cpu(), want_ball)
    assert feats.shape == (B, 128, 1024)
    fused_mlp.set_enabled(False)
    try:
        _, feats_plain, _ = plain(xyz)
        with f64_truth.torch_geometry():
            _, feats64, _ = sa64(xyz.double())
    finally:
        fused_mlp.set_enabled(True)
    e_fused, e_plain = rel(feats, feats64), rel(feats_plain, feats64)
    m_fused, m_plain = f64_truth.elem(feats, feats64), f64_truth.elem(feats_plain, feats64)
    print("configs[1] B=%d: fused %.2e plain %.2e; element-wise fused %.2e plain %.2e" % (B, e_fused, e_plain, m_fused, m_plain))
    assert e_fused <= 1e-5 and e_fused <= 2.0 * e_plain + 2e-7, (e_fused, e_plain)
    assert m_fused <= 1e-5, m_fused   # every grouped feature, not only their norm (north_star's 1e-5)


def _grad_gap(params, params64):
    num = sum(float((p.grad.double() - q.grad).norm()) ** 2 for p, q in zip(params, params64)) ** 0.5
    den = sum(float(q.grad.norm()) ** 2 for q in params64) ** 0.5
    return num / den


def test_invresmlp_stage_fused_and_plain_vs_f64_truth():
    """Two stacked InvResMLP blocks, train mode, forward and every gradient: the fused path (LocalAggregation without
    the grouped tensor + the fused pointwise pair) may be at most twice as far from the fp64 truth as torch's plain
    composition is (replaces a 2e-2 fused-vs-plain bound that could not fail)."""
    import torch.nn as nn
    from graspbalance_amd import fused_mlp
    from graspbalance_amd.drp import InvResMLP, run_stage
    from graspbalance_amd.scene import make_batch
    blocks = nn.Sequential(*[InvResMLP(in_channels=32, aggr_args={'feature_type': 'dp_fj', "reduction": 'max'},
                                       norm_args={'norm': 'bn'}, act_args={'act': 'relu'},
                                       group_args={'NAME': 'ballquery', 'radius': 0.1, 'nsample': 16},
                                       conv_args={'order': 'conv-norm-act'}, expansion=4, use_res=True)
                             for _ in range(2)])
    blocks = fill_by_key(blocks, seed=4).to(DEV).train()
    p = torch.from_numpy(make_batch([0, 1], 1024)).to(DEV)
    torch.manual_seed(4)
    f0 = torch.randn(2, 32, 1024, device=DEV)
    w = torch.randn(2, 32, 1024, device=DEV)

    def run(mod, flag, dtype):
        mod = copy.deepcopy(mod).to(dtype)
        f = f0.detach().to(dtype).clone().requires_grad_(True)
        fused_mlp.set_enabled(flag)
        try:
            if dtype == torch.float64:
                with f64_truth.torch_geometry():
                    _, out = run_stage(mod, p.to(dtype), f)
            else:
                _, out = run_stage(mod, p, f)
            (out * w.to(dtype)).sum().backward()
        finally:
            fused_mlp.set_enabled(True)
        return out.detach(), f.grad, list(mod.parameters())

    fused, plain, truth = run(blocks, True, torch.float32), run(blocks, False, torch.float32), run(blocks, False, torch.float64)
    e = {"fwd": (rel(fused[0], truth[0]), rel(plain[0], truth[0])),
         "dinput": (rel(fused[1], truth[1]), rel(plain[1], truth[1])),
         "dparams": (_grad_gap(fused[2], truth[2]), _grad_gap(plain[2], truth[2]))}
    print("InvResMLP stage (fused, plain) vs fp64:", {k: "%.2e %.2e" % v for k, v in e.items()})
    assert e["fwd"][0] <= 1e-5
    for k, (ef, ep) in e.items():
        assert ef <= 2.0 * ep + 1e-6, (k, ef, ep)


def _train_step(net, batch, views, fused, dtype=torch.float32):
    from graspbalance_amd import fused_mlp
    from graspbalance_amd.loss import get_loss
    _force(net, views)
    fused_mlp.set_enabled(fused)
    try:
        if dtype == torch.float64:
            b = dict(batch)
            b['point_clouds'] = batch['point_clouds'].double()
            with f64_truth.torch_geometry(), f64_truth.double_stage2_inputs():
                loss, ep = get_loss(net(b))
                loss.backward()
        else:
            loss, ep = get_loss(net(dict(batch)))
            loss.backward()
    finally:
        fused_mlp.set_enabled(True)
    return loss.detach(), ep, list(net.parameters())


def test_whole_network_train_step_fused_and_plain_vs_f64_truth():
    """One train step (forward, label matching, loss, backward) of the toy-size GraspBalance with the top-view arg-max
    frozen: loss, backbone features, grasp scores and the whole gradient of the fused HIP path vs the plain torch
    composition vs the fp64 truth (replaces 0.25 / 0.6 fused-vs-plain bounds that could not fail)."""
    from tests.test_model_cpu import _tiny_net
    from graspbalance_amd.synthetic import make_training_batch
    batch = make_training_batch(range(2), num_point=3000, num_objects=2, grasp_points_per_object=20, num_view=30,
                                device=DEV)
    base = fill_by_key(_tiny_net(), seed=9).to(DEV).train()
    with torch.no_grad():
        views = copy.deepcopy(base)(dict(batch))['grasp_top_view_inds'].clone()
    fused = _train_step(copy.deepcopy(base), batch, views, True)
    plain = _train_step(copy.deepcopy(base), batch, views, False)
    truth = _train_step(f64_truth.double_model(base), batch, views, False, torch.float64)
    e = {"loss": (rel(fused[0], truth[0]), rel(plain[0], truth[0])),
         "dparams": (_grad_gap(fused[2], truth[2]), _grad_gap(plain[2], truth[2]))}
    for k in ('sa1_features', 'fp2_features', 'view_score', 'grasp_score_pred', 'grasp_width_pred'):
        e[k] = (rel(fused[1][k], truth[1][k]), rel(plain[1][k], truth[1][k]))
    print("train step (fused, plain) vs fp64:", {k: "%.2e %.2e" % v for k, v in e.items()})
    assert e["sa1_features"][0] <= 1e-5
    assert e["loss"][0] <= 2e-3  # one scalar behind 4e-3 feature noise: not comparable path against path
    for k, (ef, ep) in e.items():
        if k != "loss":
            assert ef <= 2.0 * ep + 1e-6, (k, ef, ep)


def test_drp_backbone_fused_and_plain_vs_f64_truth():
    """The DRP backbone alone (4 SA levels, 15 InvResMLP blocks at toy sizes, 2 FP levels), train mode, random
    projection as loss: features and gradients of the fused path vs the plain path vs the fp64 truth."""
    from tests.test_model_cpu import _tiny_net
    from graspbalance_amd import fused_mlp
    from graspbalance_amd.scene import make_batch
    clouds = torch.from_numpy(make_batch([0, 1], 3000)).to(DEV)
    base = fill_by_key(_tiny_net().view_estimator.FeatureExtraction, seed=5).to(DEV).train()
    torch.manual_seed(7)
    w = torch.randn(2, 256, 128, device=DEV)

    def run(flag, dtype):
        drp = copy.deepcopy(base).to(dtype)
        fused_mlp.set_enabled(flag)
        try:
            if dtype == torch.float64:
                with f64_truth.torch_geometry():
                    feats, _, ep = drp(clouds.double())
            else:
                feats, _, ep = drp(clouds)
            (feats * w.to(dtype)).sum().backward()
        finally:
            fused_mlp.set_enabled(True)
        return feats.detach(), ep['sa1_features'].detach(), list(drp.parameters())

    fused, plain, truth = run(True, torch.float32), run(False, torch.float32), run(False, torch.float64)
    e = {"sa1": (rel(fused[1], truth[1]), rel(plain[1], truth[1])),
         "fp2": (rel(fused[0], truth[0]), rel(plain[0], truth[0])),
         "dparams": (_grad_gap(fused[2], truth[2]), _grad_gap(plain[2], truth[2]))}
    print("DRP backbone (fused, plain) vs fp64:", {k: "%.2e %.2e" % v for k, v in e.items()})
    assert e["sa1"][0] <= 1e-5
    for k, (ef, ep) in e.items():
        assert ef <= 2.0 * ep + 1e-6, (k, ef, ep)


def test_config4_shaped_step_at_50000_points(orc):
    """BASELINE configs[4] shape at B = 2: one train step on 50 000-point clouds (first-level FPS on the
    streamed-rows kernel); FPS and first-level ball-query indices against the oracle, finite loss and gradients."""
    from graspbalance_amd import pointnet2_utils as pu
    from graspbalance_amd.synthetic import make_training_batch
    from graspbalance_amd.train import Trainer
    batch = make_training_batch([0, 1], num_point=50000, device=DEV)
    trainer = Trainer(DEV, steps_per_epoch=10, max_epoch=2)
    clouds = batch['point_clouds']
    inds = pu.furthest_point_sample(clouds, 2048)
    want = orc.furthest_point_sampling(clouds.cpu(), 2048, orc.FPS_SKIP_NEAR_ORIGIN | orc.FPS_TIE_TREE512)
    assert torch.equal(inds.cpu(), want)
    new_xyz = torch.gather(clouds, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    assert torch.equal(pu.ball_query(0.04, 64, clouds, new_xyz).cpu(), orc.ball_query(new_xyz.cpu(), clouds.cpu(), 0.04, 64))
    loss = trainer.train_step(batch)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss))
    loss = trainer.train_step(batch)
    assert bool(torch.isfinite(loss)) and all(bool(torch.isfinite(p).all()) for p in trainer.net.parameters())


def test_config4_stated_shape_b8_50000_points_bf16(orc):
    """BASELINE configs[4] at its stated shape: B = 8 clouds of 50 000 points per GPU, nsample 64, bf16 MLP mode
    (fused_mlp.set_precision("bf16"): operands rounded to bf16 on the way into the matrix cores) / fp32 geometry.
    Geometry is untouched by the mode: first-level FPS (the streamed-rows kernel, 20 480 < N <= 64 512) and ball-query
    indices of all 8 clouds == oracle.  Eval forward bf16 vs fp32 on the same weights: within the per-level bound
    measured in tests/test_bf16_gpu.py (2e-2) after SA1 and a stage of three InvResMLP blocks, and after the whole
    backbone + heads.  Two train steps in bf16: finite loss, finite parameters."""
    from graspbalance_amd import fused_mlp, pointnet2_utils as pu
    from graspbalance_amd.graspbalance import GraspBalance
    from graspbalance_amd.synthetic import make_training_batch
    from graspbalance_amd.train import Trainer
    batch = make_training_batch(range(8), num_point=50000, device=DEV)
    clouds = batch['point_clouds']
    assert clouds.shape == (8, 50000, 3)
    inds = pu.furthest_point_sample(clouds, 2048)
    want = orc.furthest_point_sampling(clouds.cpu(), 2048, orc.FPS_SKIP_NEAR_ORIGIN | orc.FPS_TIE_TREE512)
    assert torch.equal(inds.cpu(), want)
    new_xyz = torch.gather(clouds, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    assert torch.equal(pu.ball_query(0.04, 64, clouds, new_xyz).cpu(), orc.ball_query(new_xyz.cpu(), clouds.cpu(), 0.04, 64))
    net = fill_by_key(GraspBalance(is_training=False), seed=44).to(DEV).eval()
    out = {}
    try:
        for mode in ("f32", "bf16"):
            fused_mlp.set_precision(mode)
            with torch.no_grad():
                out[mode] = net({'point_clouds': clouds.clone()})
        errs = {k: rel(out["bf16"][k], out["f32"][k]) for k in VALUE_KEYS}
        flips = float((out["bf16"]['grasp_top_view_inds'] != out["f32"]['grasp_top_view_inds']).float().mean())
        print("configs[4] eval forward, bf16 vs fp32:", {k: "%.1e" % v for k, v in errs.items()}, "top-view flips %.3f" % flips)
        assert torch.equal(out["bf16"]['sa1_inds'], out["f32"]['sa1_inds'])
        assert 1e-5 < errs['sa1_features']              # the mode really changed the arithmetic
        # The bound is DERIVED, not read off a run.  Rounding both operands of a contraction to bf16 (8 significant bits,
        # round to nearest: relative error <= 2^-9 each, independent) perturbs its output by ~sqrt(2) 2^-9 relative; the
        # perturbations of the L contractions in front of a tensor add in quadrature (independent roundings through layers
        # that BatchNorm keeps at unit scale): e(L) ~ 2^-9 sqrt(2 L).  A factor 4 on top covers what a conv + BatchNorm +
        # ReLU layer does to an incoming perturbation (x 1.2 per layer measured in fp32, DESIGN.md section 3.2, never x 4
        # in aggregate over these depths).  Depths: contractions on the longest path from the cloud to the tensor.
        depth = {'sa1_features': 3, 'sa2_features': 3 + 9 + 3, 'sa3_features': 15 + 18 + 3, 'sa4_features': 36 + 9 + 3,
                 'fp2_features': 48 + 9 + 4, 'objectness_score': 61 + 3, 'view_score': 61 + 3}
        for k in VALUE_KEYS:
            L = depth.get(k, 61 + 3 + 1 + 3)            # grasp tensors: crop stack (3) + scale fusion (1) + depth head (3)
            bound = 4.0 * 2.0 ** -9 * (2.0 * L) ** 0.5
            assert errs[k] <= bound, (k, errs[k], bound, L)
        tr = Trainer(DEV, steps_per_epoch=10, max_epoch=2, mlp_precision="bf16")
        losses = [float(tr.train_step(batch).detach()) for _ in range(2)]
        torch.cuda.synchronize()
        assert all(l == l and abs(l) < 1e4 for l in losses), losses
        assert all(bool(torch.isfinite(p).all()) for p in tr.net.parameters())
    finally:
        fused_mlp.set_precision("f32")
