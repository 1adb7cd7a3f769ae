"""Train-step operators on libbrcnn_hip.so: whole-batch target assignment, RoI sampling and the
fused losses (C ABI: `brcnn_assign_max_iou`, `brcnn_rcnn_sample`, `brcnn_rpn_loss_*`,
`brcnn_boost_loss_*`, include/brcnn_hip.h).

The reference does this work per image and per level with chains of small torch ops and host
synchronisations (mmdet/core/bbox/assigners/max_iou_assigner.py:61-213,
core/bbox/samplers/random_sampler.py:32-82, models/dense_heads/atss_rpn_head.py:299-464,
models/roi_heads/prob_roi_head.py:23-154).  Here the batch is one unit: ground truth travels as
one flat (sum G, 4) tensor plus host-side offsets, every entry is one or two launches, and the
only device->host read of a train step is the (batch, 2) positive / negative counts the seeded
host `randperm` of the RandomSampler needs.

Device tensors only; a CPU tensor raises (there is no fallback).
"""
import ctypes
import math

import torch
import torch.distributed as dist
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import lib as _L
from .ops import _ptr, _require_gpu, _stream

MAX_IMAGES = 64


def _ints(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def _floats(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def flatten_gts(gt_bboxes, gt_labels=None):
    """list of (G_b, 4) [+ (G_b,)] -> flat (sum G, 4) fp32, (sum G,) int64 or None, host offsets"""
    offs = [0]
    for g in gt_bboxes:
        offs.append(offs[-1] + int(g.shape[0]))
    gts = torch.cat([g.reshape(-1, 4) for g in gt_bboxes], 0).float().contiguous()
    labels = None
    if gt_labels is not None:
        labels = torch.cat([l.reshape(-1) for l in gt_labels], 0).long().contiguous()
    return gts, labels, offs


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, eps=1e-6):
    """bbox_overlaps of two (n, 4|5) device box lists on the HIP kernel (values only, no autograd)"""
    _require_gpu(bboxes1, bboxes2)
    b1, b2 = bboxes1.detach().float().contiguous(), bboxes2.detach().float().contiguous()
    n1, n2 = b1.shape[0], b2.shape[0]
    out = torch.empty((n1,) if is_aligned else (n1, n2), dtype=torch.float32, device=b1.device)
    st = _L.load().brcnn_bbox_overlaps(_ptr(b1), b1.shape[1] if n1 else 4, n1, _ptr(b2), b2.shape[1] if n2 else 4, n2,
                                       {'iou': 0, 'iof': 1, 'giou': 2}[mode], int(bool(is_aligned)), float(eps), _ptr(out),
                                       _stream())
    _L.check(st, 'brcnn_bbox_overlaps')
    return out


def _neg_range(neg_iou_thr):
    if isinstance(neg_iou_thr, (tuple, list)):
        assert len(neg_iou_thr) == 2
        return float(neg_iou_thr[0]), float(neg_iou_thr[1])
    return 0.0, float(neg_iou_thr)


def assign_max_iou(boxes, gts, gt_offsets, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, match_low_quality=True,
                   num_boxes=None, batch=None, geom=None, valid_hw=None, img_hw=None, allowed_border=-1,
                   want_overlaps=False, want_counts=False):
    """MaxIoUAssigner for a whole batch.

    boxes: (n, 4|5) shared by every image (RPN anchors; `batch` given) or (B, n, 4|5) per image
    (proposals, `num_boxes` (B,) int32 real rows).  geom = (level_starts[L+1], level_widths[L], A)
    with valid_hw (B, L, 2) int32 / img_hw (B, 2) + allowed_border for the anchor validity flags.
    Returns gt_inds (B, n) int32 [, max_overlaps (B, n)] [, counts (B, 2) int32]."""
    _require_gpu(boxes, gts, num_boxes, valid_hw, img_hw)
    assert boxes.dtype == torch.float32 and boxes.is_contiguous()
    if boxes.dim() == 2:
        assert batch is not None
        B, n, bstride = int(batch), boxes.shape[0], 0
    else:
        B, n = boxes.shape[:2]
        bstride = boxes.stride(0)
    rstride = boxes.shape[-1]
    assert B <= MAX_IMAGES and len(gt_offsets) == B + 1
    dev = boxes.device
    total_gt = gt_offsets[-1]
    gt_inds = torch.empty((B, n), dtype=torch.int32, device=dev)
    mo = torch.empty((B, n), dtype=torch.float32, device=dev) if want_overlaps else None
    counts = torch.empty((B, 2), dtype=torch.int32, device=dev) if want_counts else None
    ws = torch.empty((max(total_gt, 1),), dtype=torch.int32, device=dev) if match_low_quality else None
    if geom is not None:
        starts, widths, A = geom
        L = len(widths)
        ls, lw = _ints(starts), _ints(widths)
    else:
        L, A, ls, lw = 0, 1, None, None
    lo, hi = _neg_range(neg_iou_thr)
    st = _L.load().brcnn_assign_max_iou(
        _ptr(boxes), bstride, rstride, _ptr(num_boxes), n, B, _ptr(gts) if total_gt else None, _ints(gt_offsets), L,
        ls, lw, int(A), _ptr(valid_hw), _ptr(img_hw), float(allowed_border), float(pos_iou_thr), lo, hi,
        float(min_pos_iou), int(bool(match_low_quality)), _ptr(ws), _ptr(gt_inds), _ptr(mo), _ptr(counts), _stream())
    _L.check(st, 'brcnn_assign_max_iou')
    out = (gt_inds,)
    if want_overlaps:
        out += (mo,)
    if want_counts:
        out += (counts,)
    return out if len(out) > 1 else gt_inds


def sample_counts(n_pos, n_neg, num, num_expected_pos, neg_pos_ub=-1):
    """rows the RandomSampler keeps for an image with n_pos / n_neg candidates
    (base_sampler.py:78-100): (sampled_pos, sampled_neg, expected_neg)"""
    spos = min(n_pos, num_expected_pos)
    eneg = num - spos
    if neg_pos_ub >= 0:
        eneg = min(eneg, int(neg_pos_ub * max(1, spos)))
    return spos, min(n_neg, eneg), eneg


def draw_sampler_perms(counts, num, num_expected_pos, neg_pos_ub=-1):
    """The host random stream of RandomSampler.random_choice (random_sampler.py:58,
    `torch.randperm(gallery.numel())[:num]`) for every image in the reference's order (positives
    then negatives, image by image), as one (B, num_expected_pos + num) int32 tensor, plus the
    (B+1) output row offsets.  `counts`: [(n_pos, n_neg)] host ints."""
    B = len(counts)
    perm = torch.zeros((B, num_expected_pos + num), dtype=torch.int32)
    rows = [0]
    for b, (n_pos, n_neg) in enumerate(counts):
        spos, sneg, eneg = sample_counts(n_pos, n_neg, num, num_expected_pos, neg_pos_ub)
        if n_pos > num_expected_pos:
            perm[b, :num_expected_pos] = torch.randperm(n_pos)[:num_expected_pos].to(torch.int32)
        if n_neg > eneg:
            perm[b, num_expected_pos:num_expected_pos + eneg] = torch.randperm(n_neg)[:eneg].to(torch.int32)
        rows.append(rows[-1] + spos + sneg)
    return perm, rows


def rcnn_sample(proposals, num_props, gt_inds, max_overlaps, gts, gt_labels, gt_offsets, perm, row_offsets, num,
                num_expected_pos, neg_pos_ub, num_classes, means, stds, add_gt_as_proposals=True,
                reg_decoded_bbox=False, want_ious=False, want_pos_flags=False):
    """Sampled RoIs + targets of the second stage for the whole batch in one launch.
    Returns dict(rois (N,5), labels (N) int64, bbox_targets (N,4), priors (N) [, ious (N)] [, pos_flags (N) int32])."""
    _require_gpu(proposals, num_props, gt_inds, max_overlaps, gts, gt_labels, perm)
    B, K, five = proposals.shape
    assert five == 5 and proposals.is_contiguous() and proposals.dtype == torch.float32
    assert gt_inds.shape == (B, K) and gt_inds.dtype == torch.int32 and gt_inds.is_contiguous()
    assert perm.dtype == torch.int32 and perm.shape == (B, num_expected_pos + num) and perm.is_contiguous()
    dev = proposals.device
    N = int(row_offsets[-1])
    gmax = max(gt_offsets[b + 1] - gt_offsets[b] for b in range(B))
    stride = K + gmax
    lists = torch.empty((B, 2, stride), dtype=torch.int32, device=dev)
    out = dict(rois=torch.empty((N, 5), dtype=torch.float32, device=dev),
               labels=torch.empty((N,), dtype=torch.int64, device=dev),
               bbox_targets=torch.empty((N, 4), dtype=torch.float32, device=dev),
               priors=torch.empty((N,), dtype=torch.float32, device=dev))
    if want_ious:
        assert max_overlaps is not None
        out['ious'] = torch.empty((N,), dtype=torch.float32, device=dev)
    if want_pos_flags:
        out['pos_flags'] = torch.empty((N,), dtype=torch.int32, device=dev)
    if N == 0:
        return out
    total_gt = gt_offsets[-1]
    st = _L.load().brcnn_rcnn_sample(
        _ptr(proposals), _ptr(num_props), K, B, _ptr(gt_inds), _ptr(max_overlaps), _ptr(gts) if total_gt else None,
        _ptr(gt_labels) if total_gt else None, _ints(gt_offsets), int(bool(add_gt_as_proposals)), int(num),
        int(num_expected_pos), float(neg_pos_ub), _ptr(perm), _ints(row_offsets), int(num_classes),
        int(bool(reg_decoded_bbox)), _floats(means), _floats(stds), _ptr(lists), stride, _ptr(out['rois']),
        _ptr(out['labels']), _ptr(out['bbox_targets']), _ptr(out['priors']), _ptr(out.get('ious')),
        _ptr(out.get('pos_flags')), _stream())
    _L.check(st, 'brcnn_rcnn_sample')
    return out


# ----------------------------------------------------------------------------- RPN loss
class RPNLossMeta:
    """host-side description of one fused RPN loss call (shapes, anchors, loss configuration)"""

    def __init__(self, batch, sizes, strides, base_anchors, num_anchors, gt_offsets, focal_gamma, focal_alpha,
                 pos_weight, iou_gamma, means, stds, wh_ratio_clip, with_aug, lw_cls, lw_bbox, lw_aug, lw_iou,
                 cls_mode=0, reg_mode=0):
        """cls_mode: 0 FocalLoss, 1 / 2 VarifocalLoss (iou_weighted / not; focal_gamma / focal_alpha are then
        the varifocal ones); reg_mode: 0 decoded boxes with IoULoss('log') [+ MSE aug], 1 CIoULoss on the raw
        deltas (reg_decoded_bbox=False)"""
        self.batch, self.sizes, self.A = int(batch), [tuple(s) for s in sizes], int(num_anchors)
        self.L = len(self.sizes)
        self.hs, self.ws = _ints([h for h, _ in self.sizes]), _ints([w for _, w in self.sizes])
        self.sw = _ints([s if isinstance(s, int) else s[0] for s in strides])
        self.sh = _ints([s if isinstance(s, int) else s[1] for s in strides])
        self.base = base_anchors                      # list of (A,4) device tensors (kept alive here)
        self.base_ptrs = (ctypes.c_void_p * self.L)(*[b.data_ptr() for b in base_anchors])
        self.gt_offsets = _ints(gt_offsets)
        self.total_gt = int(gt_offsets[-1])
        self.cfg = _floats([focal_gamma, focal_alpha, pos_weight, iou_gamma] + list(means) + list(stds) +
                           [abs(math.log(wh_ratio_clip)), 1.0 if with_aug else 0.0, lw_cls, lw_bbox, lw_aug, lw_iou,
                            float(cls_mode), float(reg_mode)])
        self.rows = sum(self.batch * h * w for h, w in self.sizes)
        self.anchors_per_image = sum(h * w for h, w in self.sizes) * self.A


def _reduce_mean_(t):
    """mean over ranks in place (mmdet/core/utils/dist_utils.py:67-73); identity when not distributed"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return t


class RPNLossFunction(Function):
    """losses3 = [loss_rpn_cls, loss_rpn_bbox, loss_rpn_iou] (summed over the pyramid levels) from the
    fused head output y (rows, ystride) of all levels and the per-level Scale parameters."""

    @staticmethod
    def forward(ctx, y, scales, gt_inds, gts, meta):
        _require_gpu(y, scales, gt_inds, gts)
        assert y.dim() == 2 and y.dtype == torch.float32 and y.is_contiguous() and y.shape[0] == meta.rows
        assert gt_inds.dtype == torch.int32 and gt_inds.shape == (meta.batch, meta.anchors_per_image)
        lib = _L.load()
        dev = y.device
        sc = scales.detach().float().contiguous()
        nb = lib.brcnn_rpn_loss_workspace_bytes(meta.batch, meta.L, meta.hs, meta.ws, meta.A)
        ws = torch.empty((nb + 3) // 4, dtype=torch.float32, device=dev)
        small = torch.empty((meta.L * 6 + 2,), dtype=torch.float32, device=dev)
        sums, coef = small[:meta.L * 6], small[meta.L * 6:]
        totals = torch.empty((2,), dtype=torch.float32, device=dev)
        losses3 = torch.empty((3,), dtype=torch.float32, device=dev)
        per_level = torch.empty((3 * meta.L,), dtype=torch.float32, device=dev)
        gp = _ptr(gts) if meta.total_gt else None
        st = lib.brcnn_rpn_loss_forward(_ptr(y), y.shape[1], meta.batch, meta.L, meta.hs, meta.ws, meta.sw, meta.sh,
                                        meta.base_ptrs, meta.A, _ptr(sc), _ptr(gt_inds), gp, meta.gt_offsets, meta.cfg,
                                        _ptr(ws), nb, _ptr(sums), _ptr(totals), _stream())
        _L.check(st, 'brcnn_rpn_loss_forward')
        _reduce_mean_(totals)           # ONE collective for both normalisers (num_pos, sum iou_target)
        st = lib.brcnn_rpn_loss_finalize(_ptr(sums), _ptr(totals), meta.L, meta.cfg, _ptr(losses3), _ptr(per_level),
                                         _ptr(coef), _stream())
        _L.check(st, 'brcnn_rpn_loss_finalize')
        ctx.save_for_backward(y, sc, gt_inds, gts, coef)
        ctx.meta = meta
        ctx.scales_dtype = scales.dtype
        per_level = per_level.view(3, meta.L)
        ctx.mark_non_differentiable(per_level, totals)
        return losses3, per_level, totals

    @staticmethod
    @once_differentiable
    def backward(ctx, g3, _gl, _gt):
        y, sc, gt_inds, gts, coef = ctx.saved_tensors
        meta = ctx.meta
        lib = _L.load()
        nb = lib.brcnn_rpn_loss_workspace_bytes(meta.batch, meta.L, meta.hs, meta.ws, meta.A)
        ws = torch.empty((nb + 3) // 4, dtype=torch.float32, device=y.device)
        dy = torch.empty_like(y)
        dsc = torch.empty((meta.L,), dtype=torch.float32, device=y.device)
        g3 = g3.float().contiguous()
        gp = _ptr(gts) if meta.total_gt else None
        st = lib.brcnn_rpn_loss_backward(_ptr(y), y.shape[1], meta.batch, meta.L, meta.hs, meta.ws, meta.sw, meta.sh,
                                         meta.base_ptrs, meta.A, _ptr(sc), _ptr(gt_inds), gp, meta.gt_offsets, meta.cfg,
                                         _ptr(g3), _ptr(coef), _ptr(ws), nb, _ptr(dy), _ptr(dsc), _stream())
        _L.check(st, 'brcnn_rpn_loss_backward')
        return dy, dsc.to(ctx.scales_dtype), None, None, None


def rpn_loss(y, scales, gt_inds, gts, meta):
    """(losses3 (3,), per_level (3, L), totals (2,) = rank means of [num_pos, sum iou_target])"""
    return RPNLossFunction.apply(y, scales, gt_inds, gts, meta)


# ----------------------------------------------------------------------------- boosting loss
class BoostLossFunction(Function):
    """out3 = [loss_cls, loss_bbox, acc] of the boosting-reweighted R-CNN loss
    (prob_roi_head.py:107-154 over convfc_bbox_head.py:332-418)."""

    @staticmethod
    def forward(ctx, cls_score, bbox_pred, labels, priors, ious, bbox_targets, num_classes, agnostic, cfg6):
        _require_gpu(cls_score, bbox_pred, labels, priors, ious, bbox_targets)
        n = cls_score.shape[0]
        assert cls_score.shape == (n, num_classes + 1) and labels.dtype == torch.int64
        assert bbox_pred.shape == (n, 4 if agnostic else 4 * num_classes)
        cls_c, bb_c = cls_score.detach().float().contiguous(), bbox_pred.detach().float().contiguous()
        labels, priors = labels.contiguous(), priors.detach().float().contiguous()
        tg = bbox_targets.detach().float().contiguous()
        io = ious.detach().float().contiguous() if ious is not None else None
        lib = _L.load()
        dev = cls_score.device
        nb = lib.brcnn_boost_loss_workspace_bytes(n)
        ws = torch.empty((nb + 3) // 4, dtype=torch.float32, device=dev)
        out3 = torch.empty((3,), dtype=torch.float32, device=dev)
        coef = torch.empty((2,), dtype=torch.float32, device=dev)
        cfg = _floats(cfg6)
        st = lib.brcnn_boost_loss_forward_ex(_ptr(cls_c), _ptr(bb_c), _ptr(labels), _ptr(priors), _ptr(io), _ptr(tg), n,
                                             int(num_classes), int(bool(agnostic)), cfg, _ptr(ws), nb, _ptr(out3),
                                             _ptr(coef), _stream())
        _L.check(st, 'brcnn_boost_loss_forward_ex')
        ctx.save_for_backward(cls_c, bb_c, labels, priors, tg, coef, *([io] if io is not None else []))
        ctx.cfg = (n, int(num_classes), bool(agnostic), tuple(cfg6), cls_score.dtype, bbox_pred.dtype)
        return out3

    @staticmethod
    @once_differentiable
    def backward(ctx, g3):
        cls_c, bb_c, labels, priors, tg, coef, *rest = ctx.saved_tensors
        io = rest[0] if rest else None
        n, num_classes, agnostic, cfg6, cdt, bdt = ctx.cfg
        dcls, dbb = torch.empty_like(cls_c), torch.empty_like(bb_c)
        g3 = g3.float().contiguous()
        st = _L.load().brcnn_boost_loss_backward_ex(_ptr(cls_c), _ptr(bb_c), _ptr(labels), _ptr(priors), _ptr(io), _ptr(tg),
                                                    n, num_classes, int(agnostic), _floats(cfg6), _ptr(g3), _ptr(coef),
                                                    _ptr(dcls), _ptr(dbb), _stream())
        _L.check(st, 'brcnn_boost_loss_backward_ex')
        return dcls.to(cdt), dbb.to(bdt), None, None, None, None, None, None, None


def boost_loss(cls_score, bbox_pred, labels, priors, bbox_targets, num_classes, gamma, alpha=0.0, ious=None,
               iou_gamma=0.0, loss_cls_weight=1.0, loss_bbox_weight=1.0, reg_norm='bbox_num', reg_class_agnostic=False,
               plain_label_weights=False, smooth_l1_beta=0.0):
    """[loss_cls, loss_bbox, acc] as a (3,) tensor (differentiable w.r.t. cls_score and bbox_pred).
    `plain_label_weights`: the boosted weights enter the head's own loss as label weights (DyProbRoIHead,
    prob_roi_head.py:604-623) instead of ProbRoIHead's norm_loss; `smooth_l1_beta` > 0: SmoothL1Loss box term"""
    cfg8 = (float(gamma), float(alpha), float(iou_gamma), float(loss_cls_weight), float(loss_bbox_weight),
            1.0 if reg_norm == 'mean' else 0.0, 1.0 if plain_label_weights else 0.0, float(smooth_l1_beta))
    return BoostLossFunction.apply(cls_score, bbox_pred, labels, priors, ious, bbox_targets, int(num_classes),
                                   bool(reg_class_agnostic), cfg8)
