"""Loss leaf functions of the hot path, registered under the reference's names.

Restates mmdet/models/losses/{utils.py:28-101, focal_loss.py:12-182,
cross_entropy_loss.py:10-251, iou_loss.py:14-50,456-534, mse_loss.py:9-57,
smooth_l1_loss.py:35-146, accuracy.py:6-79}: the `weight_reduce_loss` rule (`avg_factor`
with 'mean' => sum/avg_factor, with 'sum' => error), `reduction_override`, the zero-weight
shortcut of IoULoss.  On device tensors the focal loss runs the HIP kernel (the reference's
CUDA path, focal_loss.py:86); on host tensors it uses the reference's own python form
(focal_loss.py:12-57), exactly as `FocalLoss.forward` dispatches at focal_loss.py:163-170.
"""
import functools

import torch
import torch.nn as nn
import torch.nn.functional as F

from .core import bbox_overlaps
from .registry import LOSSES


def reduce_loss(loss, reduction):
    reduction_enum = F._Reduction.get_enum(reduction)
    if reduction_enum == 0:
        return loss
    if reduction_enum == 1:
        return loss.mean()
    return loss.sum()


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        loss = reduce_loss(loss, reduction)
    elif reduction == 'mean':
        loss = loss.sum() / avg_factor
    elif reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def weighted_loss(loss_func):
    @functools.wraps(loss_func)
    def wrapper(pred, target, weight=None, reduction='mean', avg_factor=None, **kwargs):
        loss = loss_func(pred, target, **kwargs)
        return weight_reduce_loss(loss, weight, reduction, avg_factor)
    return wrapper


# ----------------------------------------------------------------------------- focal
def py_sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean',
                          avg_factor=None):
    pred_sigmoid = pred.sigmoid()
    target = target.type_as(pred)
    pt = (1 - pred_sigmoid) * target + pred_sigmoid * (1 - target)
    focal_weight = (alpha * target + (1 - alpha) * (1 - target)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, target, reduction='none') * focal_weight
    if weight is not None:
        if weight.shape != loss.shape:
            if weight.size(0) == loss.size(0):
                weight = weight.view(-1, 1)
            else:
                assert weight.numel() == loss.numel()
                weight = weight.view(loss.size(0), -1)
        assert weight.ndim == loss.ndim
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


def sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean',
                       avg_factor=None):
    """device path: HIP kernel with reduction 'none', weights/avg_factor applied here."""
    from . import ops
    loss = ops.sigmoid_focal_loss(pred.contiguous(), target.contiguous(), gamma, alpha, None, 'none')
    if weight is not None:
        if weight.shape != loss.shape:
            if weight.size(0) == loss.size(0):
                weight = weight.view(-1, 1)
            else:
                assert weight.numel() == loss.numel()
                weight = weight.view(loss.size(0), -1)
        assert weight.ndim == loss.ndim
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if pred.is_cuda:
            fn = sigmoid_focal_loss
        else:
            num_classes = pred.size(1)
            target = F.one_hot(target, num_classes=num_classes + 1)[:, :num_classes]
            fn = py_sigmoid_focal_loss
        return self.loss_weight * fn(pred, target, weight, gamma=self.gamma, alpha=self.alpha,
                                     reduction=reduction, avg_factor=avg_factor)


# ----------------------------------------------------------------------------- cross entropy
def varifocal_loss(pred, target, weight=None, alpha=0.75, gamma=2.0, iou_weighted=True, reduction='mean',
                   avg_factor=None):
    """losses/varifocal_loss.py:10-58: BCE-with-logits against the IoU-aware target, positives
    weighted by the target, negatives by alpha * |sigmoid - target|^gamma"""
    assert pred.size() == target.size()
    pred_sigmoid = pred.sigmoid()
    target = target.type_as(pred)
    pos = (target > 0.0).float()
    neg = (target <= 0.0).float()
    if iou_weighted:
        focal_weight = target * pos + alpha * (pred_sigmoid - target).abs().pow(gamma) * neg
    else:
        focal_weight = pos + alpha * (pred_sigmoid - target).abs().pow(gamma) * neg
    loss = F.binary_cross_entropy_with_logits(pred, target, reduction='none') * focal_weight
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class VarifocalLoss(nn.Module):
    """losses/varifocal_loss.py:61-134 (the RPN classification loss of the VOC recipe)"""

    def __init__(self, use_sigmoid=True, alpha=0.75, gamma=2.0, iou_weighted=True, reduction='mean',
                 loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid varifocal loss supported now.'
        assert alpha >= 0.0
        self.use_sigmoid, self.alpha, self.gamma = use_sigmoid, alpha, gamma
        self.iou_weighted, self.reduction, self.loss_weight = iou_weighted, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * varifocal_loss(pred, target, weight, alpha=self.alpha, gamma=self.gamma,
                                                 iou_weighted=self.iou_weighted, reduction=reduction,
                                                 avg_factor=avg_factor)


def cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None,
                  ignore_index=-100):
    ignore_index = -100 if ignore_index is None else ignore_index
    loss = F.cross_entropy(pred, label, weight=class_weight, reduction='none',
                           ignore_index=ignore_index)
    if weight is not None:
        weight = weight.float()
    return weight_reduce_loss(loss, weight=weight, reduction=reduction, avg_factor=avg_factor)


def _expand_onehot_labels(labels, label_weights, label_channels, ignore_index):
    bin_labels = labels.new_full((labels.size(0), label_channels), 0)
    valid_mask = (labels >= 0) & (labels != ignore_index)
    inds = torch.nonzero(valid_mask & (labels < label_channels), as_tuple=False)
    if inds.numel() > 0:
        bin_labels[inds, labels[inds]] = 1
    valid_mask = valid_mask.view(-1, 1).expand(labels.size(0), label_channels).float()
    if label_weights is None:
        bin_label_weights = valid_mask
    else:
        bin_label_weights = label_weights.view(-1, 1).repeat(1, label_channels)
        bin_label_weights *= valid_mask
    return bin_labels, bin_label_weights


def binary_cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None,
                         class_weight=None, ignore_index=-100):
    ignore_index = -100 if ignore_index is None else ignore_index
    if pred.dim() != label.dim():
        label, weight = _expand_onehot_labels(label, weight, pred.size(-1), ignore_index)
    if weight is not None:
        weight = weight.float()
    loss = F.binary_cross_entropy_with_logits(pred, label.float(), pos_weight=class_weight,
                                              reduction='none')
    return weight_reduce_loss(loss, weight, reduction=reduction, avg_factor=avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None,
                 ignore_index=None, loss_weight=1.0):
        super().__init__()
        assert (use_sigmoid is False) or (use_mask is False)
        assert not use_mask, 'mask cross entropy is outside the hot path'
        self.use_sigmoid, self.use_mask = use_sigmoid, use_mask
        self.reduction, self.loss_weight = reduction, loss_weight
        self.class_weight, self.ignore_index = class_weight, ignore_index
        self.cls_criterion = binary_cross_entropy if use_sigmoid else cross_entropy

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None,
                ignore_index=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if ignore_index is None:
            ignore_index = self.ignore_index
        class_weight = cls_score.new_tensor(self.class_weight) if self.class_weight is not None \
            else None
        return self.loss_weight * self.cls_criterion(
            cls_score, label, weight, class_weight=class_weight, reduction=reduction,
            avg_factor=avg_factor, ignore_index=ignore_index, **kwargs)


# ----------------------------------------------------------------------------- box losses
@weighted_loss
def iou_loss(pred, target, linear=False, mode='log', eps=1e-6):
    assert mode in ['linear', 'square', 'log']
    if linear:
        mode = 'linear'
    ious = bbox_overlaps(pred, target, is_aligned=True).clamp(min=eps)
    if mode == 'linear':
        return 1 - ious
    if mode == 'square':
        return 1 - ious ** 2
    return -ious.log()


@LOSSES.register_module()
class IoULoss(nn.Module):
    def __init__(self, linear=False, eps=1e-6, reduction='mean', loss_weight=1.0, mode='log'):
        super().__init__()
        assert mode in ['linear', 'square', 'log']
        if linear:
            mode = 'linear'
        self.mode, self.linear, self.eps = mode, linear, eps
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if (weight is not None) and (not torch.any(weight > 0)) and (reduction != 'none'):
            if pred.dim() == weight.dim() + 1:
                weight = weight.unsqueeze(1)
            return (pred * weight).sum()
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        return self.loss_weight * iou_loss(pred, target, weight, mode=self.mode, eps=self.eps,
                                           reduction=reduction, avg_factor=avg_factor, **kwargs)


@weighted_loss
def mse_loss(pred, target):
    return F.mse_loss(pred, target, reduction='none')


@LOSSES.register_module()
class MSELoss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * mse_loss(pred, target, weight, reduction=reduction,
                                           avg_factor=avg_factor)


@weighted_loss
def l1_loss(pred, target):
    if target.numel() == 0:
        return pred.sum() * 0
    assert pred.size() == target.size()
    return torch.abs(pred - target)


@LOSSES.register_module()
class L1Loss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * l1_loss(pred, target, weight, reduction=reduction,
                                          avg_factor=avg_factor)


@weighted_loss
def smooth_l1_loss(pred, target, beta=1.0):
    assert beta > 0
    if target.numel() == 0:
        return pred.sum() * 0
    assert pred.size() == target.size()
    diff = torch.abs(pred - target)
    return torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kw):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * smooth_l1_loss(pred, target, weight, beta=self.beta,
                                                 reduction=reduction, avg_factor=avg_factor, **kw)


# ----------------------------------------------------------------------------- accuracy
def accuracy(pred, target, topk=1, thresh=None):
    assert isinstance(topk, (int, tuple))
    return_single = isinstance(topk, int)
    if return_single:
        topk = (topk,)
    maxk = max(topk)
    if pred.size(0) == 0:
        accu = [pred.new_tensor(0.) for _ in range(len(topk))]
        return accu[0] if return_single else accu
    assert pred.ndim == 2 and target.ndim == 1
    assert pred.size(0) == target.size(0)
    assert maxk <= pred.size(1), f'maxk {maxk} exceeds pred dimension {pred.size(1)}'
    pred_value, pred_label = pred.topk(maxk, dim=1)
    pred_label = pred_label.t()
    correct = pred_label.eq(target.view(1, -1).expand_as(pred_label))
    if thresh is not None:
        correct = correct & (pred_value > thresh).t()
    res = []
    for k in topk:
        correct_k = correct[:k].reshape(-1).float().sum(0, keepdim=True)
        res.append(correct_k.mul_(100.0 / pred.size(0)))
    return res[0] if return_single else res


# ----------------------------------------------------------------------------- CIoU
@weighted_loss
def ciou_loss(pred, target, eps=1e-7):
    """Complete-IoU loss (iou_loss.py:175-236): 1 - (IoU - rho^2/c^2 - alpha*v), clamped"""
    import math
    lt = torch.max(pred[:, :2], target[:, :2])
    rb = torch.min(pred[:, 2:], target[:, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    ap = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    ag = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    union = ap + ag - overlap + eps
    ious = overlap / union
    enclose_wh = (torch.max(pred[:, 2:], target[:, 2:]) - torch.min(pred[:, :2], target[:, :2])).clamp(min=0)
    c2 = enclose_wh[:, 0] ** 2 + enclose_wh[:, 1] ** 2 + eps
    w1, h1 = pred[:, 2] - pred[:, 0], pred[:, 3] - pred[:, 1] + eps
    w2, h2 = target[:, 2] - target[:, 0], target[:, 3] - target[:, 1] + eps
    left = ((target[:, 0] + target[:, 2]) - (pred[:, 0] + pred[:, 2])) ** 2 / 4
    right = ((target[:, 1] + target[:, 3]) - (pred[:, 1] + pred[:, 3])) ** 2 / 4
    rho2 = left + right
    factor = 4 / math.pi ** 2
    v = factor * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = (ious > 0.5).float() * v / (1 - ious + v)
    cious = ious - (rho2 / c2 + alpha * v)
    return 1 - cious.clamp(min=-1.0, max=1.0)


@LOSSES.register_module()
class CIoULoss(nn.Module):
    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        if weight is not None and not torch.any(weight > 0):
            if pred.dim() == weight.dim() + 1:
                weight = weight.unsqueeze(1)
            return (pred * weight).sum()
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        return self.loss_weight * ciou_loss(pred, target, weight, eps=self.eps, reduction=reduction,
                                            avg_factor=avg_factor, **kwargs)
