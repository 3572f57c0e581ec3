"""CPU: this repo's python layers (label matching, loss, pred_decode, backbone, heads, the whole GraspBalance network)
on the oracle-backed extension hooks against fixtures produced by RUNNING THE REFERENCE's own python
(tests/golden/make_golden_r2.py: label_generation.py:18-157, TrainModel/loss.py:29-179, graspbalance.py:122-192,
backbone.py:14-98, modules.py:49-175, drp.py:150-307).  Same seeded inputs, same by-key weights; both sides are torch
CPU fp32 over the same oracle geometry, so the tolerances are rounding-level."""
import numpy as np
import pytest
import torch

from tests.golden import make_golden_r2 as mk
from tests.seeded import assert_errors, check_summary, fill_by_key


@pytest.fixture()
def cpu(monkeypatch):
    from tests import cpu_backend
    cpu_backend.install(monkeypatch)


def _prior(g13):
    from graspbalance_amd.loss import ScalePrior
    return ScalePrior(g13["prior_num"], g13["prior_interval"])


def test_label_matching_matches_reference(cpu, golden):
    """f1: process_grasp_labels + match_grasp_view_and_label (label_generation.py:18-157)."""
    from graspbalance_amd import label_generation as lg
    g12 = golden.load("g12_labels")
    ep = lg.process_grasp_labels(mk.g12_inputs())
    for k in ('batch_grasp_point', 'batch_grasp_view', 'batch_grasp_view_rot', 'batch_grasp_label', 'batch_grasp_offset',
              'batch_grasp_tolerance', 'batch_grasp_view_label'):
        check_summary(g12, k, ep[k], 1e-6)
    rot, labels, offsets, tol, ep = lg.match_grasp_view_and_label(ep)
    for k, t in (('top_view_rot', rot), ('top_label', labels), ('top_offset', offsets), ('top_tolerance', tol),
                 ('top_view', ep['batch_grasp_view'])):
        check_summary(g12, k, t, 1e-6)


def _loss_case(device, golden):
    from graspbalance_amd import label_generation as lg
    from graspbalance_amd.loss import get_loss
    g13 = golden.load("g13_loss")
    ep = lg.process_grasp_labels(mk.g12_inputs(device))
    ep = lg.match_grasp_view_and_label(ep)[-1]
    preds = {k: v.clone().requires_grad_(True) for k, v in mk.g13_predictions(device).items()}
    ep.update(preds)
    loss, ep = get_loss(ep, _prior(g13))
    loss.backward()
    return g13, ep, preds


def check_loss_against_reference(g13, ep, preds, rtol):
    for k in g13.files:
        if not k.endswith("::shape") or k.startswith("grad__") or k.startswith("prior"):
            continue
        name = k[:-len("::shape")].replace("__", "/")
        want = g13[k[:-len("::shape")] + "::sample"]
        if name == "graspable_mask":
            assert np.array_equal(ep[name].float().cpu().numpy().reshape(-1)[::int(g13[k[:-7] + "::step"])], want)
            continue
        got = float(ep[name].detach() if torch.is_tensor(ep[name]) else ep[name])
        assert abs(got - float(want[0])) <= rtol * max(1.0, abs(float(want[0]))), (name, got, float(want[0]))
    for k, p in preds.items():
        check_summary(g13, "grad/" + k, p.grad, rtol * 10, what="d loss / d " + k)


def test_loss_matches_reference(cpu, golden):
    """f2: get_loss (TrainModel/loss.py:29-179) values, accuracies, graspable mask and gradients w.r.t. the six
    prediction tensors, with the reference's scale prior."""
    g13, ep, preds = _loss_case("cpu", golden)
    check_loss_against_reference(g13, ep, preds, 1e-6)


def test_pred_decode_matches_reference(golden):
    """f3: pred_decode (graspbalance.py:139-192)."""
    from graspbalance_amd.graspbalance import pred_decode
    g14 = golden.load("g14_pred_decode")
    preds = pred_decode(mk.g14_inputs())
    for i, p in enumerate(preds):
        check_summary(g14, "cloud%d" % i, p, 1e-6)
    # the batched formulation vs the reference's cloud-by-cloud composition: the same bits
    from graspbalance_amd.graspbalance import _pred_decode_loop
    loop = _pred_decode_loop(mk.g14_inputs())
    assert len(loop) == len(preds) and all(torch.equal(a, b) for a, b in zip(loop, preds))


def test_backbone_matches_reference(cpu, golden):
    """a13: Pointnet2Backbone (backbone.py:14-98) train-mode forward + backward and eval forward."""
    from graspbalance_amd.backbone import Pointnet2Backbone
    net = fill_by_key(Pointnet2Backbone(), seed=16)
    assert_errors(run_backbone_case(net, mk.g16_cloud(), golden.load("g16_backbone")), {"grad/": 2e-5}, 2e-6)


def run_backbone_case(net, cloud, g16):
    """-> {key: relative error vs the reference run}; FPS-derived indices and coordinates are asserted identical."""
    seen = {}
    for name in ("sa1", "sa2", "sa4"):
        getattr(net, name).register_forward_hook(lambda m, i, o, name=name: seen.__setitem__(name, o[1].detach()))
    net.train()
    feats, xyz, ep = net(cloud.clone())
    torch.manual_seed(160)
    w = torch.randn(feats.shape).to(feats.device)
    (feats * w).sum().backward()
    errs = {}
    assert np.array_equal(ep['fp2_inds'].cpu().numpy().reshape(-1)[::int(g16["train__fp2_inds::step"])],
                          g16["train__fp2_inds::sample"].astype(np.int32))
    check_summary(g16, "train/fp2_xyz", xyz, 0.0)
    for k, t in (("train/sa1_features", seen["sa1"]), ("train/sa4_features", seen["sa4"]), ("train/fp2_features", feats),
                 ("train/sa1_running_mean", net.sa1.mlp_module.layer0.bn.bn.running_mean),
                 ("train/sa1_running_var", net.sa1.mlp_module.layer0.bn.bn.running_var)):
        errs[k] = check_summary(g16, k, t, None)
    for k, p in net.named_parameters():
        if "grad__" + k.replace("/", "__") + "::shape" in g16.files:
            errs["grad/" + k] = check_summary(g16, "grad/" + k, p.grad, None)
    net.eval()
    with torch.no_grad():
        feats, xyz, ep = net(cloud.clone())
    errs["eval/fp2_features"] = check_summary(g16, "eval/fp2_features", feats, None)
    errs["eval/sa2_features"] = check_summary(g16, "eval/sa2_features", seen["sa2"], None)
    return errs


def run_heads_case(device, g17):
    from graspbalance_amd import modules
    x = mk.g17_inputs(device)
    errs = {}
    gd = fill_by_key(modules.GraspableDetection(300, 256), seed=17).to(device).train()
    ep = gd(x['seed_xyz'], x['seed_features'], {})
    errs["gd/objectness_score"] = check_summary(g17, "gd/objectness_score", ep['objectness_score'], None)
    errs["gd/view_score"] = check_summary(g17, "gd/view_score", ep['view_score'], None)
    inds = ep['grasp_top_view_inds'].cpu().numpy().reshape(-1)[::int(g17["gd__top_view_inds::step"])]
    assert np.array_equal(inds, g17["gd__top_view_inds::sample"].astype(np.int64))
    errs["gd/top_view_rot"] = check_summary(g17, "gd/top_view_rot", ep['grasp_top_view_rot'], None)
    rot = ep['grasp_top_view_rot'][:, :128].contiguous()
    wg = fill_by_key(modules.GraspWidthGrouping(64, 3, 0.06, -0.02, [0.01, 0.02, 0.03, 0.04]), seed=18).to(device)
    wg.train()
    errs["wg/vp_features"] = check_summary(g17, "wg/vp_features", wg(x['wg_seeds'], x['cloud'], rot), None)
    wg.eval()
    with torch.no_grad():
        errs["wg/vp_features_eval"] = check_summary(g17, "wg/vp_features_eval", wg(x['wg_seeds'], x['cloud'], rot), None)
    pp = fill_by_key(modules.GraspPoseParametersDetection(12, 4), seed=19).to(device).train()
    ep2 = pp(x['vp_features'], {})
    for k in ('grasp_score_pred', 'grasp_angle_cls_pred', 'grasp_width_pred'):
        errs["pp/" + k] = check_summary(g17, "pp/" + k, ep2[k], None)
    tn = fill_by_key(modules.ToleranceNet(12, 4), seed=20).to(device).train()
    errs["tn"] = check_summary(g17, "tn/grasp_tolerance_pred", tn(x['vp_features'], {})['grasp_tolerance_pred'], None)
    return errs


def test_heads_match_reference(cpu, golden):
    """a14: GraspableDetection, GraspWidthGrouping, GraspPoseParametersDetection, ToleranceNet (modules.py:49-175)."""
    assert_errors(run_heads_case("cpu", golden.load("g17_heads")), {}, 2e-6)


INDEX_KEYS = ('sa1_inds', 'sa2_inds', 'fp2_inds')
EXACT_KEYS = ('sa1_xyz', 'sa2_xyz', 'sa4_xyz')


def _stored(g15, name):
    key = name.replace("/", "__")
    assert int(g15[key + "::step"]) == 1
    return torch.from_numpy(g15[key + "::sample"]).view(*[int(v) for v in g15[key + "::shape"]])


def _force_views(net, inds):
    """Freeze the one discrete routing decision that depends on float scores (the top-view arg-max, modules.py:74):
    `inds` (B,Ns) from the reference run replace the arg-max, so a last-bit score difference cannot swap a seed's
    rotation and labels.  How often the free arg-max agrees is measured separately (_top_view_flips)."""
    def top_view(view_score):
        idx = inds.to(view_score.device)
        return torch.gather(view_score, 2, idx.unsqueeze(-1)).squeeze(-1), idx
    net.view_estimator.GraspableClasification._top_view = top_view


def _top_view_flips(ep, want):
    """Free-running arg-max vs the reference's -> (number of differing seeds, largest gap between OUR best score and
    OUR score of the reference's pick, relative to the score scale): a flip is legitimate only as a near-tie."""
    got = ep['grasp_top_view_inds'].cpu()
    diff = got != want
    if not bool(diff.any()):
        return 0, 0.0
    vs = ep['view_score'].detach().cpu()
    ours = torch.gather(vs, 2, got.unsqueeze(-1)).squeeze(-1)
    theirs = torch.gather(vs, 2, want.unsqueeze(-1)).squeeze(-1)
    return int(diff.sum()), float((ours - theirs)[diff].max() / vs.abs().max())


def run_network_case(device, g15, prior, do_train=True):
    """The whole network against the reference run -> {key: relative error}.  FPS indices / coordinates are asserted
    identical; the top-view arg-max is reported (flip count, gap) and then frozen to the reference's picks so that
    every later tensor is compared seed by seed."""
    from graspbalance_amd.graspbalance import GraspBalance, pred_decode
    from graspbalance_amd.loss import get_loss
    batch = mk.g15_batch(device)
    errs = {}
    net = fill_by_key(GraspBalance(is_training=False), seed=15).to(device).eval()
    want_views = _stored(g15, "eval/grasp_top_view_inds").long()
    with torch.no_grad():
        free = net({'point_clouds': batch['point_clouds'].clone()})
    errs["eval/top_view_flips"], errs["eval/top_view_gap"] = _top_view_flips(free, want_views)
    _force_views(net, want_views)
    with torch.no_grad():
        ep = net({'point_clouds': batch['point_clouds'].clone()})
    _check_keys(g15, "eval/", ep, errs)
    for i, p in enumerate(pred_decode(ep)):
        if tuple(p.shape) == tuple(int(v) for v in g15["eval__pred_decode%d::shape" % i]):
            errs["eval/pred_decode%d" % i] = check_summary(g15, "eval/pred_decode%d" % i, p, None)
        else:  # an objectness arg-max flipped: a different number of grasps survives
            errs["eval/pred_decode%d" % i] = float("inf")
    if not do_train:
        return errs
    net = fill_by_key(GraspBalance(is_training=True), seed=15).to(device).train()
    want_views = _stored(g15, "train/grasp_top_view_inds").long()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    free = net(dict(batch))
    errs["train/top_view_flips"], errs["train/top_view_gap"] = _top_view_flips(free, want_views)
    net.load_state_dict(state)  # the free pass moved the BatchNorm running statistics
    _force_views(net, want_views)
    ep = net(dict(batch))
    loss, ep = get_loss(ep, prior)
    loss.backward()
    _check_keys(g15, "train/", ep, errs)
    for k in g15.files:
        if k.startswith("train__loss") and k.endswith("::sample"):
            name = k[len("train__"):-len("::sample")].replace("__", "/")
            want = float(g15[k][0])
            errs["train/" + name] = abs(float(ep[name].detach()) - want) / max(1.0, abs(want))
    errs["train/batch_grasp_point"] = check_summary(g15, "train/batch_grasp_point", ep['batch_grasp_point'], None)
    errs["train/batch_grasp_view_label"] = check_summary(g15, "train/batch_grasp_view_label",
                                                         ep['batch_grasp_view_label'], None)
    params = dict(net.named_parameters())
    for k in g15.files:
        if k.startswith("grad__") and k.endswith("::shape") and "total_norm" not in k:
            name = k[len("grad__"):-len("::shape")]
            errs["grad/" + name] = check_summary(g15, "grad/" + name, params[name].grad, None)
    total = sum(float(p.grad.double().pow(2).sum()) for p in net.parameters() if p.grad is not None) ** 0.5
    want = float(g15["grad__total_norm::sample"][0])
    errs["grad/total_norm"] = abs(total - want) / want
    return errs


def _check_keys(g15, prefix, ep, errs):
    for k in INDEX_KEYS:
        key = (prefix + k).replace("/", "__")
        got = ep[k].cpu().numpy().reshape(-1)[::int(g15[key + "::step"])]
        assert np.array_equal(got, g15[key + "::sample"].astype(got.dtype)), prefix + k
    for k in EXACT_KEYS:
        check_summary(g15, prefix + k, ep[k], 0.0)
    for k in mk.EVAL_KEYS:
        if k in INDEX_KEYS or k in EXACT_KEYS or k == 'grasp_top_view_inds':
            continue
        errs[prefix + k] = check_summary(g15, prefix + k, ep[k].float(), None)


def run_obs_case(device, g20, net_too=True):
    """f3: the object-balanced re-sampling against the reference's own run (g20).  Seed indices (fp2_inds, and the
    FPS indices they replace) are asserted identical; -> {key: relative error} for the float tensors."""
    from graspbalance_amd import modules
    from graspbalance_amd.graspbalance import GraspBalance, pred_decode
    from graspbalance_amd.pointnet2_utils import three_interpolate, three_nn
    x = mk.g20_inputs(device)
    B = x['point_clouds'].size(0)
    ep = {'point_clouds': x['point_clouds'], 'seed_cluster': x['seed_cluster'],
          'fp2_inds': torch.arange(1024, dtype=torch.int32, device=device).unsqueeze(0).repeat(B, 1)}
    dist, idx = three_nn(x['point_clouds'], x['seed_xyz'])
    dist_recip = 1.0 / (dist + 1e-8)
    weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
    ep['up_sample_features'] = three_interpolate(x['seed_features'], idx, weight)
    ep = modules.ObjectBalanceSampling(ep)
    assert ep['fp2_inds'].dtype == torch.int32
    assert np.array_equal(ep['fp2_inds'].cpu().numpy(), g20["branch_fp2_inds"])
    assert np.array_equal(ep['fp2_inds_fps'].cpu().numpy(), g20["branch_fp2_inds_fps"])
    errs = {"branch/" + k: check_summary(g20, "branch/" + k, ep[k], None)
            for k in ('up_sample_features', 'fp2_xyz', 'fp2_features')}
    if not net_too:
        return errs
    net = fill_by_key(GraspBalance(is_training=False, obs=True), seed=15).to(device).eval()
    want_views = _stored(g20, "net/grasp_top_view_inds").long()
    feed = lambda: {'point_clouds': x['point_clouds'].clone(), 'seed_cluster': x['seed_cluster'].clone()}
    with torch.no_grad():
        free = net(feed())
    errs["net/top_view_flips"], errs["net/top_view_gap"] = _top_view_flips(free, want_views)
    _force_views(net, want_views)
    with torch.no_grad():
        ep = net(feed())
    assert np.array_equal(ep['fp2_inds'].cpu().numpy(), g20["net_fp2_inds"])
    assert np.array_equal(ep['fp2_inds_fps'].cpu().numpy(), g20["net_fp2_inds_fps"])
    for k in ('up_sample_features', 'fp2_xyz', 'fp2_features', 'objectness_score', 'view_score', 'grasp_score_pred',
              'grasp_angle_cls_pred', 'grasp_width_pred', 'grasp_tolerance_pred'):
        errs["net/" + k] = check_summary(g20, "net/" + k, ep[k].float(), None)
    for i, p in enumerate(pred_decode(ep)):
        if tuple(p.shape) == tuple(int(v) for v in g20["net__pred_decode%d::shape" % i]):
            errs["net/pred_decode%d" % i] = check_summary(g20, "net/pred_decode%d" % i, p, None)
        else:
            errs["net/pred_decode%d" % i] = float("inf")
    return errs


def test_object_balance_sampling_matches_reference(cpu, golden):
    """f3: ObjectBalanceSampling (modules.py:178-221) + the obs branch of GraspPoseStage1 (graspbalance.py:35-46), alone
    on seeded backbone outputs and inside the reference's whole inference network (obs=True), vs the reference run."""
    errs = run_obs_case("cpu", golden.load("g20_obs"))
    print({k: "%.2e" % v for k, v in errs.items()})
    assert_errors(errs, {"net/top_view_flips": 0, "net/top_view_gap": 0.0, "branch/fp2_xyz": 0.0, "net/fp2_xyz": 0.0}, 1e-6)


def test_whole_network_matches_reference(cpu, golden):
    """a15: the reference's GraspBalance (graspbalance.py:122-136, DRP backbone, heads, label matching, loss) run
    on CPU over the oracle vs this repo's GraspBalance on the same oracle path: eval forward + pred_decode, train
    forward + loss + gradients.  Both are torch CPU fp32 in the same operation order: identical results."""
    errs = run_network_case("cpu", golden.load("g15_graspbalance"), _prior(golden.load("g13_loss")))
    print({k: "%.2e" % v for k, v in errs.items()})
    assert_errors(errs, {"grad/": 1e-5}, 1e-6)
