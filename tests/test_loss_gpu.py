"""GPU: the fused loss (csrc/loss.hip behind loss._FusedGraspLoss) against the torch formulation of the same file,
which follows TrainModel/loss.py term by term: every reported value and the gradients of all six prediction tensors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

LOSS_KEYS = ['loss/overall_loss', 'loss/stage1_graspable_loss', 'loss/stage1_view_loss', 'loss/stage2_grasp_score_loss',
             'loss/stage2_grasp_angle_class_loss', 'loss/stage2_grasp_width_loss', 'loss/stage2_grasp_tolerance_loss',
             'stage1_graspable_acc', 'stage1_graspable_prec', 'stage1_graspable_recall', 'stage1_pos_view_pred_count',
             'stage2_grasp_angle_class_acc/0_degree', 'stage2_grasp_angle_class_acc/15_degree',
             'stage2_grasp_angle_class_acc/30_degree']


def _inputs(seed, B=3, Ns=70, V=40, A=12, D=4, N=500, empty=False, with_view_arg=False):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    feats = (torch.randn(B, 2 + V, Ns, generator=g)).to(DEV).requires_grad_(True)
    head = torch.randn(B, 3 * A, Ns, D, generator=g).to(DEV).requires_grad_(True)
    tol = (0.05 * r(B, A, Ns, D)).to(DEV).requires_grad_(True)
    labels_all = r(B, Ns, V, A, D) * (r(B, Ns, V, 1, 1) > 0.3)
    view_label = labels_all.view(B, Ns, V, -1).max(3)[0]
    top = torch.randint(0, V, (B, Ns), generator=g)
    take = lambda t: torch.gather(t, 2, top.view(B, Ns, 1, *([1] * (t.dim() - 3))).expand(B, Ns, 1, *t.shape[3:])).squeeze(2)
    offsets_all = torch.stack([r(B, Ns, V, A, D), r(B, Ns, V, A, D), 0.12 * r(B, Ns, V, A, D)], -1)
    tol_all = 0.05 * r(B, Ns, V, A, D)
    obj = (r(B, N) > (2.0 if empty else 0.35)).long()
    ep = {
        'objectness_score': feats[:, :2, :], 'view_score': feats[:, 2:, :].transpose(1, 2).contiguous(),
        'grasp_score_pred': head[:, :A], 'grasp_angle_cls_pred': head[:, A:2 * A], 'grasp_width_pred': head[:, 2 * A:],
        'grasp_tolerance_pred': tol,
        'batch_grasp_view_label': view_label.to(DEV), 'batch_grasp_label_all': labels_all.to(DEV),
        'batch_grasp_offset_all': offsets_all.to(DEV), 'batch_grasp_label': take(labels_all).to(DEV),
        'batch_grasp_offset': take(offsets_all).to(DEV), 'batch_grasp_tolerance': take(tol_all).to(DEV),
        'objectness_label': obj.to(DEV), 'fp2_inds': torch.randint(0, N, (B, Ns), generator=g).int().to(DEV),
    }
    if with_view_arg:  # what gb_label_finish leaves behind: the loss kernel then forms the seed weights itself
        ep['_view_label_arg'] = labels_all.view(B, Ns, V, -1).argmax(3).int().to(DEV)
        ep['_view_label_source'] = ep['batch_grasp_label_all']
    return ep, (feats, head, tol)


@pytest.mark.parametrize("with_view_arg", [False, True])
@pytest.mark.parametrize("seed,empty", [(0, False), (1, False), (2, True)])
def test_fused_loss_equals_torch_formulation(monkeypatch, seed, empty, with_view_arg):
    from graspbalance_amd import loss as L
    prior = L.ScalePrior(np.arange(1, 33)[::-1].copy(), np.linspace(0.1 / 33, 0.1, 33))
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(L, "_FUSED_LOSS", fused)
        ep, leaves = _inputs(seed, empty=empty, with_view_arg=with_view_arg)
        assert L._fused_loss_ok(ep) == fused
        loss, ep = L.get_loss(ep, prior)
        grads = torch.autograd.grad(loss, leaves)
        res[fused] = (ep, grads)
    a, b = res[True][0], res[False][0]
    for k in LOSS_KEYS:
        x, y = a[k].double(), b[k].double()
        assert x.shape == y.shape and a[k].dtype == b[k].dtype, k
        assert torch.allclose(x, y, rtol=2e-5, atol=1e-7, equal_nan=True), (k, float(x), float(y))
    assert torch.equal(a['graspable_mask'], b['graspable_mask'])
    for name, gx, gy in zip(("objectness/view", "score/angle/width", "tolerance"), res[True][1], res[False][1]):
        scale = float(gy.abs().max()) + 1e-12
        assert float((gx - gy).abs().max()) / scale < 2e-5, (name, float((gx - gy).abs().max()), scale)
    if empty:  # no object seeds: the masked means are 0 / 1e-6 = 0 and the selections are empty (NaN fractions)
        assert float(a['loss/stage1_view_loss'].detach()) == 0.0 and bool(torch.isnan(a['stage2_grasp_angle_class_acc/0_degree']))


def test_fused_loss_partial_backward(monkeypatch):
    """Gradients of single terms (not only of the overall loss) flow through the fused op."""
    from graspbalance_amd import loss as L
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(L, "_FUSED_LOSS", fused)
        ep, leaves = _inputs(5)
        _, ep = L.get_loss(ep)
        target = ep['loss/stage2_grasp_width_loss'] * 3.0 + ep['loss/stage1_view_loss']
        outs[fused] = torch.autograd.grad(target, leaves, allow_unused=True)
    for gx, gy in zip(outs[True], outs[False]):
        if gy is None:
            assert gx is None or float(gx.abs().max()) == 0.0
        else:
            assert float((gx - gy).abs().max()) / (float(gy.abs().max()) + 1e-12) < 2e-5
