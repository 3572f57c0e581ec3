"""GPU: the whole network on the HIP path against the same network on the CPU oracle path (same
weights, same inputs): indices bit-exact, head outputs within 1e-5 (fp32 conv/BN on different BLAS
back ends; the geometry feeding them is bit-identical)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nets():
    from tests.seeded import fill_by_key
    from tests.test_model_cpu import _tiny_net
    cpu_net = fill_by_key(_tiny_net(training=False), seed=2)  # by-key weights incl. non-trivial running statistics
    gpu_net = _tiny_net(training=False)
    gpu_net.load_state_dict(cpu_net.state_dict())
    return cpu_net.eval(), gpu_net.to(DEV).eval()


def test_eval_forward_matches_cpu_oracle_path(monkeypatch):
    """Toy-size eval forward, HIP path vs CPU oracle path: indices identical, every output within 1e-5 with the
    top-view arg-max frozen to the CPU path's picks (the free arg-max may differ only at near-ties).  The full-size
    version with an fp64 truth is tests/test_parity_f64_gpu.py."""
    from tests.test_model_cpu import _tiny_batch
    from tests import cpu_backend
    cpu_net, gpu_net = _nets()
    clouds = _tiny_batch(B=2, N=3000)['point_clouds']
    with monkeypatch.context() as mp:
        cpu_backend.install(mp)
        with torch.no_grad():
            want = cpu_net({'point_clouds': clouds})
    views = want['grasp_top_view_inds']
    with torch.no_grad():
        free = gpu_net({'point_clouds': clouds.to(DEV)})
    assert int((free['grasp_top_view_inds'].cpu() != views).sum()) <= 2
    gpu_net.view_estimator.GraspableClasification._top_view = \
        lambda vs: (torch.gather(vs, 2, views.to(DEV).unsqueeze(-1)).squeeze(-1), views.to(DEV))
    with torch.no_grad():
        got = gpu_net({'point_clouds': clouds.to(DEV)})
        gpu_net.grasp_generator.fused_cylinder = False
        got_sep = gpu_net({'point_clouds': clouds.to(DEV)})
    for k in ('sa1_inds', 'sa2_inds', 'fp2_inds', 'grasp_top_view_inds'):
        assert torch.equal(got[k].cpu(), want[k]), k
    for k in ('sa1_xyz', 'sa4_xyz', 'fp2_xyz'):
        assert torch.equal(got[k].cpu(), want[k]), k
    for k in ('fp2_features', 'objectness_score', 'view_score', 'grasp_score_pred', 'grasp_angle_cls_pred',
              'grasp_width_pred', 'grasp_tolerance_pred'):
        err = float((got[k].cpu() - want[k]).norm() / (want[k].norm() + 1e-12))
        assert err < 1e-5, (k, err)
        # 16 separate cylinder queries (unfused head) vs the fused query + channel-last MLP: same indices
        # (test_cylinder_query_multi_equals_16_single_queries), values equal to fp32 rounding
        err = float((got[k] - got_sep[k]).norm() / (got_sep[k].norm() + 1e-12))
        assert err < 1e-5, (k, err)


def test_train_step_runs_and_updates(monkeypatch):
    from tests.test_model_cpu import _tiny_net
    from graspbalance_amd.synthetic import make_training_batch
    from graspbalance_amd.train import Trainer
    net = _tiny_net()
    trainer = Trainer(DEV, num_view=30, model=net, steps_per_epoch=10, max_epoch=2)
    batch = make_training_batch(range(2), num_point=3000, num_objects=2, grasp_points_per_object=20, num_view=30,
                                device=DEV)
    before = [p.detach().clone() for p in trainer.net.parameters()]
    losses = [float(trainer.train_step(batch)) for _ in range(3)]
    assert all(l == l for l in losses)
    changed = sum(int(not torch.equal(a, b)) for a, b in zip(before, trainer.net.parameters()))
    assert changed > 200


def test_full_size_forward_properties():
    """Config 3 shapes (B=2 here): full-size GraspBalance eval forward; size-independent checks."""
    from graspbalance_amd.graspbalance import GraspBalance, pred_decode
    from graspbalance_amd.scene import make_batch
    torch.manual_seed(1234)
    net = GraspBalance(is_training=False).to(DEV).eval()
    clouds = torch.from_numpy(make_batch([0, 1], 20000)).to(DEV)
    with torch.no_grad():
        out = net({'point_clouds': clouds})
    assert out['fp2_features'].shape == (2, 256, 1024)
    assert out['grasp_score_pred'].shape == (2, 12, 1024, 4) and out['view_score'].shape == (2, 1024, 300)
    assert torch.equal(out['fp2_inds'], out['sa1_inds'][:, :1024])
    # FPS prefix property: seeds are the first 1024 FPS picks of the input cloud
    assert torch.equal(out['fp2_xyz'], torch.gather(clouds, 1, out['fp2_inds'].long()[:, :, None].expand(-1, -1, 3)))
    for k in ('grasp_score_pred', 'grasp_width_pred', 'grasp_tolerance_pred', 'objectness_score'):
        assert bool(torch.isfinite(out[k]).all())
    rot = out['grasp_top_view_rot']
    eye = torch.matmul(rot, rot.transpose(-1, -2))
    assert float((eye - torch.eye(3, device=DEV)).abs().max()) < 1e-5  # rotations stay orthonormal
    preds = pred_decode(out)
    assert len(preds) == 2 and preds[0].shape[1] == 17


@pytest.mark.parametrize("ragged", [False, True])
def test_fused_label_matching_equals_per_object_composition(monkeypatch, ragged):
    """gb_label_gather path of process_grasp_labels (all objects' pose transforms as one batched matmul, composed
    index maps, fused score transform) == the reference-style per-object composition, bit for bit - also with a
    different number of grasp points per object."""
    from graspbalance_amd import label_generation as lg
    from graspbalance_amd.synthetic import make_training_batch
    batch = make_training_batch(range(2), num_point=6000, num_objects=3, grasp_points_per_object=40, num_view=60,
                                device=DEV)
    if ragged:
        keep = {(0, 1): 23, (1, 0): 31, (1, 1): 1}
        for key in ('grasp_points_list', 'grasp_labels_list', 'grasp_offsets_list', 'grasp_tolerance_list'):
            batch[key] = [[t[:keep.get((i, k), t.size(0))] for k, t in enumerate(per)] for i, per in enumerate(batch[key])]
    ep = dict(batch)
    ep['input_xyz'] = batch['point_clouds']
    ep['fp2_xyz'] = batch['point_clouds'][:, :256].contiguous()
    assert lg._fusable(ep)
    fused = lg.process_grasp_labels(dict(ep))
    monkeypatch.setattr(lg, "_fusable", lambda e: False)
    plain = lg.process_grasp_labels(dict(ep))
    for k in ('batch_grasp_point', 'batch_grasp_label', 'batch_grasp_offset', 'batch_grasp_tolerance',
              'batch_grasp_view_label'):
        assert torch.equal(fused[k], plain[k]), k
    for k in ('batch_grasp_view', 'batch_grasp_view_rot'):
        assert torch.equal(fused[k], plain[k]), k


def _segmented_clouds(B, N, objects, seed):
    import numpy as np
    from graspbalance_amd.scene import make_batch
    clouds = torch.from_numpy(make_batch(list(range(seed, seed + B)), N))
    rng = np.random.default_rng(seed)
    seg = torch.from_numpy(rng.integers(0, objects + 1, size=(B, N))).long()
    seg[0, :40] = 0
    seg[0][seg[0] == objects] = 0          # cloud 0 has one object fewer
    seg[-1][seg[-1] == 1] = 0
    seg[-1, 5:25] = 1                      # ... and the last one an object smaller than its share of seeds
    return clouds, seg


def test_object_balance_sampling_batched_equals_per_object_loop():
    """ObjectBalanceSampling (TrainModel/modules.py:178-221) as one segmented FPS launch == the reference's
    per-object composition: same seed indices, coordinates and features."""
    from graspbalance_amd import modules
    clouds, seg = _segmented_clouds(3, 6000, 7, seed=40)
    feats = torch.randn(3, 256, 6000)
    def ep():
        return {'point_clouds': clouds.to(DEV), 'seed_cluster': seg.to(DEV), 'up_sample_features': feats.to(DEV),
                'fp2_inds': torch.zeros(3, 1024, dtype=torch.int32, device=DEV)}
    a = modules.ObjectBalanceSampling(ep())
    b = modules._object_balance_sampling_loop(ep())
    for k in ('fp2_inds', 'fp2_xyz', 'fp2_features'):
        assert torch.equal(a[k], b[k]), k
    assert a['fp2_inds'].dtype == torch.int32 and a['fp2_inds'].shape == (3, 1024)
    # every seed lies on an object, objects get 1024 // K seeds each (remainder to the last)
    picked = torch.gather(seg.to(DEV), 1, a['fp2_inds'].long())
    assert bool((picked != 0).all())
    labels, counts = torch.unique(picked[1], return_counts=True)
    assert counts.tolist() == [1024 // 7] * 6 + [1024 // 7 + 1024 % 7]
    # label 0 absent: the reference runs out of per-object shares (IndexError); so does the build
    e = ep()
    e['seed_cluster'] = torch.ones_like(e['seed_cluster'])
    with pytest.raises((IndexError, ZeroDivisionError)):
        modules.ObjectBalanceSampling(e)


def test_obs_eval_forward():
    """Inference with object-balanced seeds (GraspPoseStage1 obs branch, graspbalance.py:35-42): runs end to end,
    seeds are the sampled object points, features are the 3-NN up-sampled backbone features at those points."""
    from tests.test_model_cpu import _tiny_net
    from graspbalance_amd.graspbalance import pred_decode
    net = _tiny_net(training=False)
    net.view_estimator.obs = True
    net = net.to(DEV).eval()
    clouds, seg = _segmented_clouds(2, 3000, 4, seed=50)
    with torch.no_grad():
        out = net({'point_clouds': clouds.to(DEV), 'seed_cluster': seg.to(DEV)})
    inds = out['fp2_inds'].long()
    assert torch.equal(out['fp2_xyz'], torch.gather(clouds.to(DEV), 1, inds[:, :, None].expand(-1, -1, 3)))
    up = out['up_sample_features']  # (B,256,N)
    assert torch.equal(out['fp2_features'], torch.gather(up, 2, inds[:, None, :].expand(-1, up.size(1), -1)))
    assert bool((torch.gather(seg.to(DEV), 1, inds) != 0).all())
    assert out['grasp_score_pred'].shape[0] == 2 and bool(torch.isfinite(out['grasp_score_pred']).all())
    assert len(pred_decode(out)) == 2


def test_flat_adam_gpu_follows_torch_fused_adam():
    """FlatAdam's CUDA path (torch's fused Adam kernel on one flat buffer) == torch.optim.Adam(fused=True) per tensor,
    under the OneCycleLR schedule (lr and beta1 change every step)."""
    from torch.optim.lr_scheduler import OneCycleLR
    from graspbalance_amd.flat_adam import FlatAdam
    from tests.test_flat_adam_cpu import _net

    def run(make_opt):
        net = _net(0).to(DEV)
        opt = make_opt(net.parameters())
        sched = OneCycleLR(opt, max_lr=1e-2, steps_per_epoch=4, epochs=3)
        torch.manual_seed(1)
        for _ in range(9):
            x = torch.randn(9, 7, device=DEV)
            for p in net.parameters():
                p.grad = None
            net(x).square().mean().backward()
            opt.step()
            sched.step()
        return net, opt

    a, _ = run(lambda ps: torch.optim.Adam(ps, lr=1e-2, fused=True))
    b, opt = run(lambda ps: FlatAdam(ps, lr=1e-2))
    assert opt._fused
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-8)


def test_sampling_prefetch_on_side_stream_equals_inline_sampling():
    """prefetch.SamplingPrefetch: the first-level FPS of an announced batch, run on a side stream, is consumed by the
    next step and equals the inline sampling; an unannounced (or modified) batch is sampled inline."""
    from tests.test_model_cpu import _tiny_net
    from graspbalance_amd import pointnet2_utils as pu
    from graspbalance_amd.prefetch import KEY, SamplingPrefetch
    from graspbalance_amd.synthetic import make_training_batch
    from graspbalance_amd.train import Trainer
    a = make_training_batch(range(2), num_point=3000, num_objects=2, grasp_points_per_object=20, num_view=30, device=DEV)
    b = make_training_batch(range(2, 4), num_point=3000, num_objects=2, grasp_points_per_object=20, num_view=30, device=DEV)
    pf = SamplingPrefetch(torch.device(DEV), 256)
    pf.launch(b['point_clouds'])
    assert pf.take(a['point_clouds']) is None and pf.pending is None      # announced b, asked for a: nothing
    pf.launch(b['point_clouds'])
    inds = pf.take(b['point_clouds'])
    torch.cuda.synchronize()
    assert torch.equal(inds, pu.furthest_point_sample(b['point_clouds'], 256))
    pf.launch(b['point_clouds'])
    b['point_clouds'].add_(0.0)                                            # modified in place after the announcement
    assert pf.take(b['point_clouds']) is None
    # through the trainer (its launch-by-launch execution: the HIP-graph step keeps the samples in static buffers instead,
    # tests/test_graph_step_gpu.py): the announced batch's indices reach the network (and are the ones it would compute)
    tr = Trainer(DEV, num_view=30, model=_tiny_net(), steps_per_epoch=10, max_epoch=2, graph=False)
    seen = {}
    sa1 = tr.net.view_estimator.FeatureExtraction.sa1
    inner = sa1.forward

    def spy(xyz, features=None, inds=None):
        seen['given'] = inds
        out = inner(xyz, features, inds)
        seen['used'] = out[2]
        return out
    sa1.forward = spy
    tr.train_step(a, next_batch=b)
    assert seen['given'] is None
    tr.train_step(b, next_batch=a)
    assert seen['given'] is not None and seen['given'] is seen['used']
    torch.cuda.synchronize()
    assert torch.equal(seen['used'], pu.furthest_point_sample(b['point_clouds'], 256))
    loss = tr.train_step(a)
    assert seen['given'] is not None and bool(torch.isfinite(loss))
    _ = KEY


def test_sampling_prefetch_keeps_its_xyz_copy_of_a_six_channel_cloud_alive():
    """A (B,N,6) cloud (xyz + colour): the side stream samples a fresh contiguous copy of the coordinates.  While the
    sampling is pending, the main stream allocates and overwrites same-sized blocks - had the copy been released when
    launch() returned, the caching allocator would hand its block out here - and the indices must still be the inline
    ones."""
    from graspbalance_amd import pointnet2_utils as pu
    from graspbalance_amd.prefetch import SamplingPrefetch
    from graspbalance_amd.scene import make_scene
    xyz = torch.stack([torch.as_tensor(make_scene(40 + i, 20000)) for i in range(2)]).float().to(DEV)
    cloud6 = torch.cat([xyz, torch.rand_like(xyz)], dim=2).contiguous()
    want = pu.furthest_point_sample(xyz.contiguous(), 2048)
    pf = SamplingPrefetch(torch.device(DEV), 2048)
    for _ in range(3):
        pf.launch(cloud6)
        junk = [torch.full((2, 20000, 3), float(k), device=DEV) for k in range(8)]   # same size as the xyz copy
        inds = pf.take(cloud6)
        torch.cuda.synchronize()
        assert inds is not None and torch.equal(inds, want)
        del junk


def test_lean_label_matching_equals_full():
    """The training step's lean label matching (label_generation._lean_labels: gb_label_gather max pass, gb_label_scores,
    gb_label_gather_view - none of the (B,Ns,V,A,D[,3]) tensors is built) against the full path on the same network
    outputs: every key the step reads - view labels and their positions, the picked view's labels / offsets / tolerance,
    rotations, the loss with all its terms and the gradients w.r.t. the predictions."""
    from graspbalance_amd import label_generation as lg
    from graspbalance_amd.graspbalance import match_grasp_view_and_label
    from graspbalance_amd.loss import get_loss
    from tests.golden import make_golden_r2 as mk
    res = {}
    for lean in (False, True):
        ep = mk.g12_inputs(DEV)
        if lean:
            ep[lg.LEAN] = True
        assert lg._fusable(ep)
        ep = lg.process_grasp_labels(ep)
        assert ('_lean' in ep) == lean and ('batch_grasp_label' in ep) != lean
        rot, labels, offsets, tol, ep = match_grasp_view_and_label(ep)
        preds = {k: v.clone().requires_grad_(True) for k, v in mk.g13_predictions(DEV).items()}
        ep.update(preds)
        loss, ep = get_loss(ep)
        loss.backward()
        res[lean] = (ep, rot, labels, offsets, tol, {k: p.grad for k, p in preds.items()})
    full, lean = res[False], res[True]
    for k in ('batch_grasp_view_label', '_view_label_arg', 'batch_grasp_point', 'batch_grasp_view', 'batch_grasp_view_rot',
              'batch_grasp_offset', 'batch_grasp_tolerance', 'graspable_mask'):
        assert torch.equal(full[0][k], lean[0][k]), k
    assert torch.equal(full[1], lean[1]) and torch.equal(full[3], lean[3]) and torch.equal(full[4], lean[4])
    assert torch.allclose(full[2], lean[2], rtol=2e-7, atol=0) and float(lean[2].abs().max()) > 0   # log on two code paths
    for k in full[0]:
        if k.startswith('loss/') or 'acc' in k:
            assert torch.allclose(full[0][k], lean[0][k], rtol=1e-6, atol=1e-7, equal_nan=True), k
    for k, g in full[5].items():
        assert torch.allclose(g, lean[5][k], rtol=1e-5, atol=1e-9), k
    assert 'batch_grasp_label_all' not in lean[0] and 'batch_grasp_offset_all' in full[0]


def test_capacity_form_label_matching_equals_the_packed_form():
    """Round 5: label matching on a LabelGeometry + device-side pointer tables - the form a captured
    train step runs, whose launches do not depend on how many objects a cloud has or how many grasp points an object
    brings - against the lean matching on the packed lists, for clouds with different numbers of objects and objects with
    different numbers of points: view labels and their positions, matched points, views, rotations, the picked view's
    labels / offsets / tolerances, the seed widths."""
    from graspbalance_amd import label_generation as lg
    from tests.golden import make_golden_r2 as mk

    def inputs():
        ep = mk.g12_inputs(DEV)
        for key in lg.LIST_KEYS:
            ep[key] = [per[:-1] if i == 0 else list(per) for i, per in enumerate(ep[key])]   # cloud 0 has one object less
        ep[lg.LEAN] = True
        return ep
    packed = lg.process_grasp_labels(inputs())
    ep = inputs()
    B = len(ep['grasp_points_list'])
    geo = lg.LabelGeometry(B, 4, 64, DEV)
    geo.vad = tuple(ep['grasp_labels_list'][0][0].shape[1:4])
    geo.load(ep)
    tables = {}
    for key in lg.BY_REFERENCE:
        ts = [t for per in ep[key] for t in per]
        at = [ts[0].data_ptr()] * (B * geo.kc)
        for t, sl in zip(ts, geo.slots(ep)):
            at[sl] = t.data_ptr()
        tables[key] = torch.tensor(at, dtype=torch.int64, device=DEV)
    ep[lg.GEOMETRY], ep[lg.TABLES] = geo, tables
    cap = lg.process_grasp_labels(ep)
    assert '_lean' in cap and '_lean' in packed
    for k in ('batch_grasp_view_label', '_view_label_arg', 'batch_grasp_view', 'batch_grasp_view_rot'):
        assert torch.equal(packed[k], cap[k]), k
    assert torch.allclose(packed['batch_grasp_point'], cap['batch_grasp_point'], rtol=0, atol=1e-6)
    for k in ('label', 'offset', 'tolerance', 'seed_width'):
        assert torch.equal(packed['_lean'][k], cap['_lean'][k]), k
    assert float(cap['_lean']['label'].abs().max()) > 0


def test_predictor_with_announced_next_batch_returns_the_same_grasps():
    """predict.Predictor: eval forward + pred_decode; with the next batch announced its first-level sampling runs on a
    side stream under the current forward and is consumed by the next call - same grasps bit for bit as the inline
    path, for alternating batches, and an unannounced batch is simply sampled inline."""
    from tests.seeded import fill_by_key
    from tests.test_model_cpu import _tiny_net
    from graspbalance_amd.predict import Predictor
    from graspbalance_amd.scene import make_batch
    net = fill_by_key(_tiny_net(training=False), seed=5)
    a = {'point_clouds': torch.from_numpy(make_batch([0, 1], 3000)).to(DEV)}
    b = {'point_clouds': torch.from_numpy(make_batch([2, 3], 3000)).to(DEV)}
    plain = Predictor(net, DEV, prefetch_sampling=False)
    want_a, want_b = plain(a), plain(b)
    piped = Predictor(net, DEV)
    got = [piped(a, next_batch=b), piped(b, next_batch=a), piped(a), piped(b)]
    torch.cuda.synchronize()
    for g, w in zip(got, (want_a, want_b, want_a, want_b)):
        assert len(g) == len(w) and all(torch.equal(x, y) for x, y in zip(g, w))
    assert piped.prefetch.pending is None


def test_eval_tables_follow_trainer_steps():
    """The eval-mode BatchNorm tables are cached (fused_mlp._eval_ab).  A Trainer updates the parameters through
    FlatAdam - views of one flat buffer, whose own version counters do not move - and the running statistics through the
    kernels' raw pointers: after train steps the fused eval forward must still equal the plain composition on the SAME
    (updated) network, i.e. the cache must have been dropped."""
    from tests.test_model_cpu import _tiny_net
    from graspbalance_amd import fused_mlp
    from graspbalance_amd.synthetic import make_training_batch
    from graspbalance_amd.train import Trainer
    tr = Trainer(DEV, num_view=30, model=_tiny_net(), steps_per_epoch=10, max_epoch=2)
    batch = make_training_batch(range(2), num_point=3000, num_objects=2, grasp_points_per_object=20, num_view=30,
                                device=DEV)
    backbone = tr.net.view_estimator.FeatureExtraction   # (the whole network's stage 2 wants labels when built for training)

    def eval_out(fused):
        backbone.eval()
        fused_mlp.set_enabled(fused)
        try:
            with torch.no_grad():
                return backbone(batch['point_clouds'])[0].clone()
        finally:
            fused_mlp.set_enabled(True)
            backbone.train()
    before = eval_out(True)                     # fills the cache
    for _ in range(2):
        tr.train_step(batch)
    after, plain = eval_out(True), eval_out(False)
    torch.cuda.synchronize()
    assert float((after - before).norm() / before.norm()) > 1e-3          # the network did change
    assert float((after - plain).norm() / plain.norm()) < 1e-4, "stale eval-mode BatchNorm tables"


def test_predictor_call_between_backward_and_step_leaves_the_gradients_alone():
    """An evaluation pass in the middle of a training step (validation between backward() and optimizer.step(), or
    between gradient-accumulation micro-steps): the weight gradients of the fused stacks are views of the shared zero
    arena, which a Predictor call used to rewind and re-zero.  It now works in a private arena."""
    from tests.test_model_cpu import _tiny_net
    from graspbalance_amd import fused_mlp
    from graspbalance_amd.loss import get_loss
    from graspbalance_amd.predict import Predictor
    from graspbalance_amd.synthetic import make_training_batch
    net = _tiny_net().to(DEV).train()
    batch = make_training_batch(range(2), num_point=3000, num_objects=2, grasp_points_per_object=20, num_view=30,
                                device=DEV)
    fused_mlp.begin_step(torch.device(DEV))
    loss, _ = get_loss(net(dict(batch)))
    loss.backward()
    torch.cuda.synchronize()
    before = [p.grad.clone() for p in net.parameters()]
    assert sum(float(g.abs().sum()) for g in before) > 0
    import copy
    ev = copy.deepcopy(net)
    ev.is_training = ev.view_estimator.is_training = ev.grasp_generator.is_training = False
    Predictor(ev, DEV)({'point_clouds': batch['point_clouds']}, decode=False)
    torch.cuda.synchronize()
    assert all(torch.equal(a, p.grad) for a, p in zip(before, net.parameters()))
    fused_mlp.end_arena(torch.device(DEV))


def test_whole_network_backward_inside_a_wgrad_queue_gives_the_same_253_gradients():
    """Round 6: train.Trainer runs its backward inside fused_mlp.WgradQueue - the few-row weight gradients (InvResMLP
    blocks, aggregation convs written straight into their joined (N, 3 + C) gradient, feature propagation, heads) are
    recorded and leave as grouped launches at the end.  ONE forward (so the routing is the same), the backward twice -
    outside a queue and inside one: every one of the 253 parameter gradients agrees to the order of the fp32 atomics, the
    gradients autograd handed out inside the queue ALIAS the buffers the grouped launch wrote (no clone in between: a
    clone would have frozen zeros), and the recorded products really were many."""
    from tests.test_model_cpu import _tiny_net
    from graspbalance_amd import fused_mlp
    from graspbalance_amd.loss import get_loss
    from graspbalance_amd.synthetic import make_training_batch
    net = _tiny_net().to(DEV).train()
    batch = make_training_batch(range(2), num_point=3000, num_objects=2, grasp_points_per_object=20, num_view=30,
                                device=DEV)
    loss, _ = get_loss(net(dict(batch)))
    loss.backward(retain_graph=True)
    torch.cuda.synchronize()
    plain = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    with fused_mlp.WgradQueue(DEV) as q:
        loss.backward()
        recorded = [(k[3], i[3], i[4:8]) for k, i in zip(q.keep, q.items)]
        assert len(recorded) >= 30 and q.launches == 0
        ptrs = {p.grad.data_ptr() for p in net.parameters()}
        # a recorded dW is a parameter's .grad itself, or columns 3.. of one (an aggregation conv's joined gradient)
        assert all(dw_ptr in ptrs or dw_ptr - 12 in ptrs for _, dw_ptr, _ in recorded)
    torch.cuda.synchronize()
    assert q.launches >= 1
    worst = 0.0
    for (name, p), g0 in zip(net.named_parameters(), plain):
        err = float((p.grad - g0).norm()) / max(float(g0.norm()), 1e-12)
        worst = max(worst, err)
        assert err < 1e-5, (name, err)
    print("253 gradients, queue vs plain: worst relative difference %.2e over %d recorded products" % (worst, len(recorded)))
