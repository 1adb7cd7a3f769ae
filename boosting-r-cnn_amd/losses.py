"""Loss leaf functions of the hot path, registered under the reference's names.

Reference: mmdet/models/losses/{utils.py:28-101, focal_loss.py:12-182, varifocal_loss.py:10-134,
cross_entropy_loss.py:10-251, iou_loss.py:14-50,175-236,456-534, mse_loss.py:9-57, smooth_l1_loss.py:10-146,
accuracy.py:6-79}.  The device-resident train step does not come through here (its loss arithmetic lives in the fused
HIP kernels `brcnn_rpn_loss_*` / `brcnn_boost_loss_*`); these modules serve the reference-signature entry points --
host tensors, the per-image chain the tests compare against, off-path callers -- with the reference's fp32 operation
order.  Every class is an element-wise term plus ONE shared reduction rule (`_LossModule`):

    weight (broadcast per row if needed) -> 'none' | sum / avg_factor | mean | sum        (utils.py:28-55)

with `avg_factor` valid only for 'mean' / 'none', `reduction_override`, and `loss_weight` applied last.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .core import bbox_overlaps
from .registry import LOSSES

_REDUCTIONS = ('none', 'mean', 'sum')


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """utils.py:28-55"""
    if reduction not in _REDUCTIONS:
        raise ValueError(f'{reduction} is not a valid value for reduction')
    if weight is not None:
        loss = loss * weight
    if avg_factor is not None:
        if reduction == 'sum':
            raise ValueError('avg_factor can not be used with reduction="sum"')
        return loss.sum() / avg_factor if reduction == 'mean' else loss
    return loss if reduction == 'none' else (loss.mean() if reduction == 'mean' else loss.sum())


def _row_weight(weight, loss):
    """a per-sample weight against a per-(sample, class) loss: one column, broadcast (focal_loss.py:41-55)"""
    if weight is None or weight.shape == loss.shape:
        return weight
    if weight.size(0) == loss.size(0):
        return weight.view(-1, 1)
    assert weight.numel() == loss.numel()
    return weight.view(loss.size(0), -1)


class _LossModule(nn.Module):
    """`forward(pred, target, weight, avg_factor, reduction_override)` of every loss class: the subclass supplies
    `term(pred, target, **kw)` (unreduced) and may reshape the weight in `prepare_weight`"""

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def _reduction(self, override):
        assert override in (None,) + _REDUCTIONS
        return override if override else self.reduction

    def prepare_weight(self, weight, loss):
        return weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        loss = self.term(pred, target, **kwargs)
        return self.loss_weight * weight_reduce_loss(loss, self.prepare_weight(weight, loss),
                                                     self._reduction(reduction_override), avg_factor)


# ----------------------------------------------------------------------------- focal / varifocal
def py_sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None):
    """focal_loss.py:12-57 (host form; `target` one-hot): BCE-with-logits x alpha_t x (1 - p_t)^gamma"""
    p = pred.sigmoid()
    t = target.type_as(pred)
    miss = (1 - p) * t + p * (1 - t)
    loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * ((alpha * t + (1 - alpha) * (1 - t)) * miss.pow(gamma))
    return weight_reduce_loss(loss, _row_weight(weight, loss), reduction, avg_factor)


def sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None):
    """device form: the HIP kernel with reduction 'none' (class-index targets), weights / avg_factor applied here --
    what the reference does around mmcv's CUDA op (focal_loss.py:60-103)"""
    from . import ops
    loss = ops.sigmoid_focal_loss(pred.contiguous(), target.contiguous(), gamma, alpha, None, 'none')
    return weight_reduce_loss(loss, _row_weight(weight, loss), reduction, avg_factor)


@LOSSES.register_module()
class FocalLoss(_LossModule):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__(reduction, loss_weight)
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = self._reduction(reduction_override)
        if pred.is_cuda:            # focal_loss.py:163-170: the native op on the device, the python form on the host
            fn = sigmoid_focal_loss
        else:
            fn, target = py_sigmoid_focal_loss, F.one_hot(target, num_classes=pred.size(1) + 1)[:, :pred.size(1)]
        return self.loss_weight * fn(pred, target, weight, gamma=self.gamma, alpha=self.alpha, reduction=reduction,
                                     avg_factor=avg_factor)


def varifocal_loss(pred, target, weight=None, alpha=0.75, gamma=2.0, iou_weighted=True, reduction='mean', avg_factor=None):
    """varifocal_loss.py:10-58: BCE-with-logits against the IoU-aware target; positives weighted by the target (or 1),
    negatives by alpha |sigmoid - target|^gamma"""
    assert pred.size() == target.size()
    t = target.type_as(pred)
    pos = (t > 0.0).float()
    neg_w = alpha * (pred.sigmoid() - t).abs().pow(gamma) * (t <= 0.0).float()
    w = (t * pos if iou_weighted else pos) + neg_w
    return weight_reduce_loss(F.binary_cross_entropy_with_logits(pred, t, reduction='none') * w, weight, reduction, avg_factor)


@LOSSES.register_module()
class VarifocalLoss(_LossModule):
    """the RPN classification loss of the VOC recipe"""

    def __init__(self, use_sigmoid=True, alpha=0.75, gamma=2.0, iou_weighted=True, reduction='mean', loss_weight=1.0):
        super().__init__(reduction, loss_weight)
        assert use_sigmoid is True, 'Only sigmoid varifocal loss supported now.'
        assert alpha >= 0.0
        self.use_sigmoid, self.alpha, self.gamma, self.iou_weighted = use_sigmoid, alpha, gamma, iou_weighted

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        return self.loss_weight * varifocal_loss(pred, target, weight, alpha=self.alpha, gamma=self.gamma,
                                                 iou_weighted=self.iou_weighted,
                                                 reduction=self._reduction(reduction_override), avg_factor=avg_factor)


# ----------------------------------------------------------------------------- cross entropy
def cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None, ignore_index=-100):
    """cross_entropy_loss.py:10-58: softmax CE per sample"""
    loss = F.cross_entropy(pred, label, weight=class_weight, reduction='none',
                           ignore_index=-100 if ignore_index is None else ignore_index)
    return weight_reduce_loss(loss, None if weight is None else weight.float(), reduction, avg_factor)


def _one_hot_with_weights(labels, label_weights, channels, ignore_index):
    """class indices -> (n, channels) 0/1 targets; rows with a negative / ignored label weigh 0 (:61-80)"""
    usable = (labels >= 0) & (labels != ignore_index)
    onehot = labels.new_zeros((labels.size(0), channels))
    rows = torch.nonzero(usable & (labels < channels), as_tuple=False).reshape(-1)
    if rows.numel() > 0:
        onehot[rows, labels[rows]] = 1
    w = usable.view(-1, 1).expand(labels.size(0), channels).float()
    if label_weights is not None:
        w = label_weights.view(-1, 1).repeat(1, channels) * w
    return onehot, w


def binary_cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None, ignore_index=-100):
    """cross_entropy_loss.py:83-135: sigmoid CE; class-index labels are expanded to one-hot rows first"""
    if pred.dim() != label.dim():
        label, weight = _one_hot_with_weights(label, weight, pred.size(-1), -100 if ignore_index is None else ignore_index)
    loss = F.binary_cross_entropy_with_logits(pred, label.float(), pos_weight=class_weight, reduction='none')
    return weight_reduce_loss(loss, None if weight is None else weight.float(), reduction, avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(_LossModule):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, ignore_index=None,
                 loss_weight=1.0):
        super().__init__(reduction, loss_weight)
        assert not use_mask, 'mask cross entropy is outside the hot path'
        self.use_sigmoid, self.use_mask = use_sigmoid, use_mask
        self.class_weight, self.ignore_index = class_weight, ignore_index
        self.cls_criterion = binary_cross_entropy if use_sigmoid else cross_entropy

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, ignore_index=None, **kwargs):
        cw = None if self.class_weight is None else cls_score.new_tensor(self.class_weight)
        return self.loss_weight * self.cls_criterion(
            cls_score, label, weight, class_weight=cw, reduction=self._reduction(reduction_override), avg_factor=avg_factor,
            ignore_index=self.ignore_index if ignore_index is None else ignore_index, **kwargs)


# ----------------------------------------------------------------------------- box losses
class _BoxLoss(_LossModule):
    """IoU-type losses over (n, 4) corner boxes: a (n, 4) weight is averaged per box, and an all-zero weight
    short-circuits to a zero that still depends on `pred` (iou_loss.py:505-519 / 291-300)"""

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        reduction = self._reduction(reduction_override)
        if weight is not None and not torch.any(weight > 0) and (reduction != 'none' or not self.keep_none_on_zero_weight):
            return (pred * (weight.unsqueeze(1) if pred.dim() == weight.dim() + 1 else weight)).sum()
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        return self.loss_weight * weight_reduce_loss(self.term(pred, target, **kwargs), weight, reduction, avg_factor)


def iou_loss(pred, target, weight=None, linear=False, mode='log', eps=1e-6, reduction='mean', avg_factor=None):
    """iou_loss.py:14-50: -log(IoU) (or 1 - IoU, 1 - IoU^2) of aligned boxes, IoU floored at eps"""
    assert mode in ('linear', 'square', 'log')
    ious = bbox_overlaps(pred, target, is_aligned=True).clamp(min=eps)
    loss = 1 - ious if (linear or mode == 'linear') else (1 - ious ** 2 if mode == 'square' else -ious.log())
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class IoULoss(_BoxLoss):
    keep_none_on_zero_weight = True      # iou_loss.py:505: the shortcut is skipped for reduction 'none'

    def __init__(self, linear=False, eps=1e-6, reduction='mean', loss_weight=1.0, mode='log'):
        super().__init__(reduction, loss_weight)
        assert mode in ('linear', 'square', 'log')
        self.mode, self.linear, self.eps = ('linear' if linear else mode), linear, eps

    def term(self, pred, target, **kwargs):
        return iou_loss(pred, target, None, mode=self.mode, eps=self.eps, reduction='none', **kwargs)


def ciou_loss(pred, target, weight=None, eps=1e-7, reduction='mean', avg_factor=None):
    """Complete-IoU loss (iou_loss.py:175-236): 1 - (IoU - rho^2 / c^2 - alpha v), clamped to [-1, 1] before the 1 -"""
    inner = (torch.min(pred[:, 2:], target[:, 2:]) - torch.max(pred[:, :2], target[:, :2])).clamp(min=0)
    overlap = inner[:, 0] * inner[:, 1]
    w1, h1 = pred[:, 2] - pred[:, 0], pred[:, 3] - pred[:, 1]
    w2, h2 = target[:, 2] - target[:, 0], target[:, 3] - target[:, 1]
    ious = overlap / (w1 * h1 + w2 * h2 - overlap + eps)
    outer = (torch.max(pred[:, 2:], target[:, 2:]) - torch.min(pred[:, :2], target[:, :2])).clamp(min=0)
    diag2 = outer[:, 0] ** 2 + outer[:, 1] ** 2 + eps
    rho2 = ((target[:, 0] + target[:, 2]) - (pred[:, 0] + pred[:, 2])) ** 2 / 4 + \
           ((target[:, 1] + target[:, 3]) - (pred[:, 1] + pred[:, 3])) ** 2 / 4
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / (h2 + eps)) - torch.atan(w1 / (h1 + eps)), 2)
    with torch.no_grad():
        alpha = (ious > 0.5).float() * v / (1 - ious + v)
    loss = 1 - (ious - (rho2 / diag2 + alpha * v)).clamp(min=-1.0, max=1.0)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class CIoULoss(_BoxLoss):
    keep_none_on_zero_weight = False     # iou_loss.py:291: the shortcut applies whatever the reduction

    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__(reduction, loss_weight)
        self.eps = eps

    def term(self, pred, target, **kwargs):
        return ciou_loss(pred, target, None, eps=self.eps, reduction='none', **kwargs)


@LOSSES.register_module()
class MSELoss(_LossModule):
    def term(self, pred, target):
        return F.mse_loss(pred, target, reduction='none')


def _abs_diff(pred, target):
    if target.numel() == 0:
        return None
    assert pred.size() == target.size()
    return torch.abs(pred - target)


@LOSSES.register_module()
class L1Loss(_LossModule):
    def term(self, pred, target):
        d = _abs_diff(pred, target)
        return pred.sum() * 0 if d is None else d


@LOSSES.register_module()
class SmoothL1Loss(_LossModule):
    """smooth_l1_loss.py:10-32 (the box loss of the DynamicRoIHead variant: its beta is updated while training)"""

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__(reduction, loss_weight)
        self.beta = beta

    def term(self, pred, target):
        assert self.beta > 0
        d = _abs_diff(pred, target)
        if d is None:
            return pred.sum() * 0
        return torch.where(d < self.beta, 0.5 * d * d / self.beta, d - 0.5 * self.beta)


# ----------------------------------------------------------------------------- accuracy
def accuracy(pred, target, topk=1, thresh=None):
    """accuracy.py:6-51: top-k accuracy in percent (an int `topk` gives one tensor, a tuple a list)"""
    ks = (topk,) if isinstance(topk, int) else tuple(topk)
    if pred.size(0) == 0:
        res = [pred.new_tensor(0.) for _ in ks]
    else:
        assert pred.ndim == 2 and target.ndim == 1 and pred.size(0) == target.size(0)
        assert max(ks) <= pred.size(1), f'maxk {max(ks)} exceeds pred dimension {pred.size(1)}'
        value, label = pred.topk(max(ks), dim=1)
        hit = label.t().eq(target.view(1, -1).expand(max(ks), -1))
        if thresh is not None:
            hit = hit & (value > thresh).t()
        res = [hit[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / pred.size(0)) for k in ks]
    return res[0] if isinstance(topk, int) else res
