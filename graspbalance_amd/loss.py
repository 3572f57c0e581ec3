"""Training loss with the terms of the reference's TrainModel/loss.py (generate_reweight_mask :29,
get_loss :44, compute_robust_graspable_loss :55, compute_weighted_view_loss :80,
compute_weighted_grasp_loss :118): objectness CE + scale-reweighted view MSE + 0.2 x (score Huber +
angle CE + width Huber + tolerance Huber).

The reference loads its object-scale prior from ScaleDistribution/objects_scales.npy at import time
and ``.cuda()``s it (loss.py:18-26); that data file is not part of this repo, so the prior is an
argument: ``ScalePrior.uniform()`` (all weights 1, used with synthetic data) or ``ScalePrior.from_npy``.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from .loss_utils import GRASP_MAX_TOLERANCE, GRASP_MAX_WIDTH, THRESH_BAD, THRESH_GOOD, huber_loss


class ScalePrior:
    """Per-width-bin loss weights ``1 - log(count / max count)`` over 32 bins of grasp width."""

    def __init__(self, num, interval):
        num = torch.as_tensor(np.asarray(num), dtype=torch.float32)
        self.intervals = [float(v) for v in np.asarray(interval).tolist()]
        self.weights = -(num / num.max()).log() + 1
        self._dev = {}

    @classmethod
    def uniform(cls, bins=32, max_width=GRASP_MAX_WIDTH):
        return cls(np.ones(bins), np.linspace(max_width / (bins + 1), max_width, bins + 1))

    @classmethod
    def from_npy(cls, path):
        d = np.load(path, allow_pickle=True).item()
        return cls(d['num'], d['interval'])

    def _on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (self.weights.to(device), torch.tensor(self.intervals, dtype=torch.float32, device=device))
        return self._dev[key]

    def lookup(self, widths):
        """weights[b] for the bin with intervals[b] < width < intervals[b+1] (both strict, as the reference's
        loop of masked assignments loss.py:32-36; a width on a boundary or outside keeps bin 0) — as one
        bucketize instead of 32 x (two compares, and, masked store), and without the boolean-index host sync."""
        weights, edges = self._on(widths.device)
        nb = edges.numel() - 1
        i = torch.bucketize(widths, edges)  # edges[i-1] < w <= edges[i]
        inside = (i >= 1) & (i <= nb) & (widths != edges[i.clamp(max=nb)])
        return weights[torch.where(inside, i - 1, torch.zeros_like(i))]


_DEFAULT_PRIOR = ScalePrior.uniform()


def generate_reweight_mask(end_points, prior=None):
    prior = prior or _DEFAULT_PRIOR
    if '_seed_width' in end_points:   # lean label matching: the width at the seed's best label was gathered directly
        return prior.lookup(end_points['_seed_width'])
    labels = end_points['batch_grasp_label_all']
    B, Ns = labels.shape[:2]
    best = torch.argmax(labels.reshape(B, Ns, -1), dim=2, keepdim=True)
    # == gather(offsets[..., 2].reshape(B,Ns,-1), 2, best) without materialising the strided width slice
    offsets = end_points['batch_grasp_offset_all'].reshape(B, Ns, -1)
    return prior.lookup(torch.gather(offsets, 2, best * 3 + 2).squeeze(2))


def _masked_fraction(flags, mask):
    """== flags[mask].float().mean() (NaN for an empty selection, like the mean of an empty tensor) without the
    boolean-index gather, whose output size costs a device->host synchronisation."""
    return (flags & mask).sum().float() / mask.sum().float()


def _seed_objectness(end_points):
    return torch.gather(end_points['objectness_label'], 1, end_points['fp2_inds'].long())


def _graspable_label(end_points, objectness_label):
    per_view = end_points.get('batch_grasp_view_label')  # process_grasp_labels already took this max of `labels`
    if '_seed_width' not in end_points:
        labels = end_points['batch_grasp_label_all']
        B, Ns, V = labels.shape[:3]
        if per_view is None or end_points.get('_view_label_source') is not labels:
            per_view = labels.view(B, Ns, V, -1).max(3)[0]
    graspable_cnt = torch.sum((per_view > THRESH_BAD).long(), dim=2)
    return (graspable_cnt > 10) * objectness_label


def compute_robust_graspable_loss(end_points):
    objectness_score = end_points['objectness_score']
    graspable_label = _graspable_label(end_points, _seed_objectness(end_points))
    end_points['graspable_mask'] = graspable_label
    loss = nn.functional.cross_entropy(objectness_score, graspable_label, reduction='mean')
    end_points['loss/stage1_graspable_loss'] = loss
    pred = torch.argmax(objectness_score, 1)
    correct = pred == graspable_label.long()
    end_points['stage1_graspable_acc'] = correct.float().mean()
    end_points['stage1_graspable_prec'] = _masked_fraction(correct, pred == 1)
    end_points['stage1_graspable_recall'] = _masked_fraction(correct, graspable_label == 1)
    return loss, end_points


def compute_weighted_view_loss(end_points, weight_mask, width_weight_mask=None):
    view_score = end_points['view_score']
    view_label = end_points['batch_grasp_view_label']
    V = view_label.size(2)
    objectness_label = _seed_objectness(end_points)
    graspable_label = _graspable_label(end_points, objectness_label) * objectness_label
    objectness_mask = (graspable_label > 0).unsqueeze(-1).repeat(1, 1, V)
    if width_weight_mask is not None:
        weight_mask = weight_mask * width_weight_mask
    loss_mask = objectness_mask.float() * weight_mask.unsqueeze(-1).repeat(1, 1, V)
    loss = nn.functional.mse_loss(view_score, view_label, reduction='none')
    loss = torch.sum(loss * loss_mask) / (loss_mask.sum() + 1e-6)
    end_points['loss/stage1_view_loss'] = loss
    end_points['stage1_pos_view_pred_count'] = ((view_score >= THRESH_GOOD) & objectness_mask).long().sum()
    return loss, end_points


def compute_weighted_grasp_loss(end_points, weight_mask):
    objectness_mask = _seed_objectness(end_points).bool()
    labels = end_points['batch_grasp_label']       # (B,Ns,A,D) of the top view
    offsets = end_points['batch_grasp_offset']     # (B,Ns,A,D,3)
    tolerance = end_points['batch_grasp_tolerance']
    A = labels.size(2)
    widths = offsets[:, :, :, :, 2]
    best_angle = torch.argmax(labels, dim=2, keepdim=True)  # (B,Ns,1,D)
    target_labels = torch.gather(labels, 2, best_angle).squeeze(2)
    target_widths = torch.gather(widths, 2, best_angle).squeeze(2)
    target_tolerance = torch.gather(tolerance, 2, best_angle).squeeze(2)
    graspable_mask = target_labels > THRESH_BAD
    loss_mask = (objectness_mask.unsqueeze(-1).expand_as(graspable_mask) & graspable_mask).float() \
        * weight_mask.unsqueeze(-1).expand_as(graspable_mask)

    def masked_mean(x, mask):
        return torch.sum(x * mask) / (mask.sum() + 1e-6)

    # 1. grasp score (every depth of a seed counts as soon as one depth is graspable)
    depth_loss_mask = loss_mask.max(dim=2)[0].unsqueeze(-1).expand_as(loss_mask)
    pick = best_angle.transpose(1, 2)  # (B,1,Ns,D)
    grasp_score = torch.gather(end_points['grasp_score_pred'], 1, pick).squeeze(1)
    grasp_score_loss = masked_mean(huber_loss(grasp_score - target_labels, delta=1.0), depth_loss_mask)
    end_points['loss/stage2_grasp_score_loss'] = grasp_score_loss
    # 2. in-plane rotation class
    target_angles_cls = best_angle.squeeze(2)
    angle_scores = end_points['grasp_angle_cls_pred']
    angle_loss = masked_mean(nn.functional.cross_entropy(angle_scores, target_angles_cls, reduction='none'), loss_mask)
    end_points['loss/stage2_grasp_angle_class_loss'] = angle_loss
    angle_pred = torch.argmax(angle_scores, 1)
    diff = torch.abs(angle_pred - target_angles_cls)
    sel = loss_mask.bool()
    end_points['stage2_grasp_angle_class_acc/0_degree'] = _masked_fraction(angle_pred == target_angles_cls, sel)
    end_points['stage2_grasp_angle_class_acc/15_degree'] = _masked_fraction((diff <= 1) | (diff >= A - 1), sel)
    end_points['stage2_grasp_angle_class_acc/30_degree'] = _masked_fraction((diff <= 2) | (diff >= A - 2), sel)
    # 3. width, 4. tolerance
    width_pred = torch.gather(end_points['grasp_width_pred'], 1, pick).squeeze(1)
    width_loss = masked_mean(huber_loss((width_pred - target_widths) / GRASP_MAX_WIDTH, delta=1), loss_mask)
    end_points['loss/stage2_grasp_width_loss'] = width_loss
    tol_pred = torch.gather(end_points['grasp_tolerance_pred'], 1, pick).squeeze(1)
    tol_loss = masked_mean(huber_loss((tol_pred - target_tolerance) / GRASP_MAX_TOLERANCE, delta=1), loss_mask)
    end_points['loss/stage2_grasp_tolerance_loss'] = tol_loss
    return grasp_score_loss + angle_loss + width_loss + tol_loss, end_points


_FUSED_LOSS = os.environ.get("GB_FUSED_LOSS", "1") != "0"  # A/B switch: the loss terms as three HIP launches


def _dense_inner(t):
    """(B,C,...) tensor whose dimensions after the batch are dense (a channel slice of a contiguous tensor is)."""
    return t.dim() >= 2 and t[0].is_contiguous() and t.dtype == torch.float32 and t.is_cuda


class _FusedGraspLoss(torch.autograd.Function):
    """Every term of get_loss (csrc/loss.hip).  forward -> (out (14,), graspable_mask (B,Ns) int64)."""

    @staticmethod
    def forward(ctx, obj_score, view_score, score_pred, angle_pred, width_pred, tol_pred, view_label, obj_label, weight,
                labels, offsets, tolerance, view_arg=None, offsets_all=None, edges=None, prior_w=None):
        from . import _lib
        dev = view_score.device
        B, Ns, V = view_score.shape
        A, D = labels.shape[2], labels.shape[3]
        S = B * Ns
        strides = (ctypes.c_longlong * 5)(obj_score.stride(0), score_pred.stride(0), angle_pred.stride(0),
                                          width_pred.stride(0), tol_pred.stride(0))
        work = torch.empty(S * 40 + 3, dtype=torch.float32, device=dev)
        partial, aux, den = work[:S * 20], work[S * 20:S * 40], work[S * 40:]
        graspable = torch.empty((B, Ns), dtype=torch.int64, device=dev)
        out = torch.empty(14, dtype=torch.float32, device=dev)
        ins = (obj_score, view_score, view_label, obj_label, weight, labels, offsets, tolerance, score_pred, angle_pred,
               width_pred, tol_pred)
        ctx.dims = (B, Ns, V, A, D)
        ctx.strides = strides
        extra = (view_arg, offsets_all, edges, prior_w)
        ctx.nb = int(prior_w.numel()) if prior_w is not None else 0
        with _lib.device_ctx(dev):
            _lib.check(_lib.lib().gb_grasp_loss_fwd(*[_lib.ptr(t) for t in ins], ctypes.cast(strides, ctypes.c_void_p),
                                                    *[_lib.ptr(t) for t in extra], ctx.nb, B, Ns, V, A, D, THRESH_BAD, THRESH_GOOD, GRASP_MAX_WIDTH,
                                                    GRASP_MAX_TOLERANCE, _lib.ptr(partial), _lib.ptr(aux),
                                                    _lib.ptr(graspable), _lib.ptr(out), _lib.ptr(den),
                                                    _lib.current_stream(dev)), "gb_grasp_loss_fwd")
        ctx.save_for_backward(*ins, aux, graspable, den)  # backward reads the masks from aux, not the weights
        ctx.mark_non_differentiable(graspable)
        return out, graspable

    @staticmethod
    def backward(ctx, g_out, _g_mask):
        from . import _lib
        *ins, aux, graspable, den = ctx.saved_tensors
        B, Ns, V, A, D = ctx.dims
        dev = g_out.device
        g_out = g_out.contiguous().float()
        d_obj = torch.empty((B, 2, Ns), dtype=torch.float32, device=dev)
        d_view = torch.empty((B, Ns, V), dtype=torch.float32, device=dev)
        d_preds = torch.empty((4, B, A, Ns, D), dtype=torch.float32, device=dev)
        with _lib.device_ctx(dev):
            _lib.check(_lib.lib().gb_grasp_loss_bwd(*[_lib.ptr(t) for t in ins], ctypes.cast(ctx.strides, ctypes.c_void_p),
                                                    None, None, None, None, 0, B, Ns, V, A, D, THRESH_BAD, THRESH_GOOD, GRASP_MAX_WIDTH,
                                                    GRASP_MAX_TOLERANCE, _lib.ptr(aux), _lib.ptr(graspable), _lib.ptr(den),
                                                    _lib.ptr(g_out), _lib.ptr(d_obj), _lib.ptr(d_view), _lib.ptr(d_preds[0]),
                                                    _lib.ptr(d_preds[1]), _lib.ptr(d_preds[2]), _lib.ptr(d_preds[3]),
                                                    _lib.current_stream(dev)), "gb_grasp_loss_bwd")
        return (d_obj, d_view, d_preds[0], d_preds[1], d_preds[2], d_preds[3]) + (None,) * 10


def _fused_loss_ok(end_points):
    keys = ('objectness_score', 'view_score', 'grasp_score_pred', 'grasp_angle_cls_pred', 'grasp_width_pred',
            'grasp_tolerance_pred')
    if not _FUSED_LOSS or not all(_dense_inner(end_points[k]) for k in keys):
        return False
    labels = end_points['batch_grasp_label']
    return (labels.dim() == 4 and labels.size(3) <= 8 and end_points['objectness_score'].size(1) == 2
            and end_points['grasp_score_pred'].shape[1:] == (labels.size(2), labels.size(1), labels.size(3)))


def _get_loss_fused(end_points, prior):
    """get_loss through _FusedGraspLoss: same keys, same values (masked means accumulated in fp64)."""
    f = lambda t: t.contiguous().float()
    view_label = f(end_points['batch_grasp_view_label'])
    extra = ()
    weight = None
    lean = '_seed_width' in end_points
    labels_all = offsets_all = view_arg = None
    if not lean:
        labels_all, view_arg = end_points['batch_grasp_label_all'], end_points.get('_view_label_arg')
        offsets_all = end_points['batch_grasp_offset_all']
    if lean:
        weight = f(generate_reweight_mask(end_points, prior))
    elif (view_arg is not None and end_points.get('_view_label_source') is labels_all and offsets_all.is_contiguous()
            and offsets_all.dtype == torch.float32 and view_arg.shape == view_label.shape):
        # per-view maxima and their positions came out of gb_label_finish: the arg-max over all views of a seed, the
        # width gather and the prior lookup of generate_reweight_mask happen inside the loss kernel
        prior_w, edges = (prior or _DEFAULT_PRIOR)._on(view_label.device)
        extra = (view_arg, offsets_all, edges, prior_w)
    else:
        weight = f(generate_reweight_mask(end_points, prior))
    out, graspable = _FusedGraspLoss.apply(
        end_points['objectness_score'], f(end_points['view_score']), end_points['grasp_score_pred'],
        end_points['grasp_angle_cls_pred'], end_points['grasp_width_pred'], end_points['grasp_tolerance_pred'],
        view_label, _seed_objectness(end_points).contiguous(), weight,
        f(end_points['batch_grasp_label']), f(end_points['batch_grasp_offset']), f(end_points['batch_grasp_tolerance']),
        *extra)
    # (the metrics from a detached view: `out.unbind(0)` under autograd made the engine materialise a zero gradient for
    #  each of the 13 unused scalars and stack them - 15 launches for nothing)
    vals = list(out.detach().unbind(0))
    for k in range(7):       # the loss and its six terms stay differentiable (views: nothing is launched for them)
        vals[k] = out[k]
    end_points['graspable_mask'] = graspable
    for k, name in enumerate(('loss/overall_loss', 'loss/stage1_graspable_loss', 'loss/stage1_view_loss',
                              'loss/stage2_grasp_score_loss', 'loss/stage2_grasp_angle_class_loss',
                              'loss/stage2_grasp_width_loss', 'loss/stage2_grasp_tolerance_loss', 'stage1_graspable_acc',
                              'stage1_graspable_prec', 'stage1_graspable_recall', None,
                              'stage2_grasp_angle_class_acc/0_degree', 'stage2_grasp_angle_class_acc/15_degree',
                              'stage2_grasp_angle_class_acc/30_degree')):
        if name:
            end_points[name] = vals[k]
    end_points['stage1_pos_view_pred_count'] = vals[10].long()
    return vals[0], end_points


def get_loss(end_points, prior=None):
    if _fused_loss_ok(end_points):
        return _get_loss_fused(end_points, prior)
    reweight_mask = generate_reweight_mask(end_points, prior)
    objectness_loss, end_points = compute_robust_graspable_loss(end_points)
    view_loss, end_points = compute_weighted_view_loss(end_points, reweight_mask.clone())
    grasp_loss, end_points = compute_weighted_grasp_loss(end_points, reweight_mask.clone())
    loss = objectness_loss + view_loss + 0.2 * grasp_loss
    end_points['loss/overall_loss'] = loss
    return loss, end_points
