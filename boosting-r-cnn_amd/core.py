"""Host-side box machinery of the hot path (`mmdet/core/*` counterpart), torch tensors on
any device.  These are the small index/elementwise pieces around the HIP kernels: anchor
generation, the delta<->box coder, IoU, MaxIoU assignment, sampling, RoI packing, result
formatting.  Each class keeps the reference's registry name and constructor arguments.

Reference files restated: core/anchor/anchor_generator.py:12-458, core/anchor/utils.py:5-47,
core/bbox/coder/delta_xywh_bbox_coder.py:99-272, core/bbox/iou_calculators/
iou2d_calculator.py:30-261, core/bbox/assigners/{max_iou_assigner.py:61-213,
assign_result.py}, core/bbox/samplers/{base_sampler.py:35-102, random_sampler.py:32-82,
pseudo_sampler.py:24-42, sampling_result.py:26-55}, core/bbox/transforms.py:59-117,
core/utils/misc.py:11-43.
"""
from functools import partial

import numpy as np
import torch

from .registry import (BBOX_ASSIGNERS, BBOX_CODERS, BBOX_SAMPLERS, IOU_CALCULATORS,
                       PRIOR_GENERATORS, build_iou_calculator)


# ----------------------------------------------------------------------------- misc
def multi_apply(func, *args, **kwargs):
    pfunc = partial(func, **kwargs) if kwargs else func
    map_results = map(pfunc, *args)
    return tuple(map(list, zip(*map_results)))


def unmap(data, count, inds, fill=0):
    """scatter a subset back to the full set of `count` items (core/utils/misc.py:29-43)"""
    if data.dim() == 1:
        ret = data.new_full((count,), fill)
        ret[inds.type(torch.bool)] = data
    else:
        new_size = (count,) + data.size()[1:]
        ret = data.new_full(new_size, fill)
        ret[inds.type(torch.bool), :] = data
    return ret


def images_to_levels(target, num_levels):
    """[img][all anchors] -> [level](img, anchors of level)  (core/anchor/utils.py:5-18)"""
    target = torch.stack(target, 0)
    level_targets = []
    start = 0
    for n in num_levels:
        end = start + n
        level_targets.append(target[:, start:end])
        start = end
    return level_targets


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    img_h, img_w = img_shape[:2]
    if allowed_border >= 0:
        return valid_flags & \
            (flat_anchors[:, 0] >= -allowed_border) & (flat_anchors[:, 1] >= -allowed_border) & \
            (flat_anchors[:, 2] < img_w + allowed_border) & \
            (flat_anchors[:, 3] < img_h + allowed_border)
    return valid_flags


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


# ----------------------------------------------------------------------------- anchors
@PRIOR_GENERATORS.register_module()
class AnchorGenerator:
    """RetinaNet-style anchor generator (ratio-major when scale_major=True; centre offset 0
    puts the anchor centre on the cell's top-left pixel)."""

    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True,
                 octave_base_scale=None, scales_per_octave=None, centers=None, center_offset=0.):
        if center_offset != 0:
            assert centers is None, f'center cannot be set when center_offset!=0, {centers} is given.'
        if not (0 <= center_offset <= 1):
            raise ValueError(f'center_offset should be in range [0, 1], {center_offset} is given.')
        if centers is not None:
            assert len(centers) == len(strides)
        self.strides = [_pair(s) for s in strides]
        self.base_sizes = [min(s) for s in self.strides] if base_sizes is None else base_sizes
        assert len(self.base_sizes) == len(self.strides)
        assert ((octave_base_scale is not None and scales_per_octave is not None) ^
                (scales is not None)), \
            'scales and octave_base_scale with scales_per_octave cannot be set at the same time'
        if scales is not None:
            self.scales = torch.Tensor(scales)
        else:
            octave_scales = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
            self.scales = torch.Tensor(octave_scales * octave_base_scale)
        self.octave_base_scale, self.scales_per_octave = octave_base_scale, scales_per_octave
        self.ratios = torch.Tensor(ratios)
        self.scale_major = scale_major
        self.centers = centers
        self.center_offset = center_offset
        self.base_anchors = self.gen_base_anchors()

    @property
    def num_base_anchors(self):
        return [b.size(0) for b in self.base_anchors]

    num_base_priors = num_base_anchors

    @property
    def num_levels(self):
        return len(self.strides)

    def gen_base_anchors(self):
        out = []
        for i, base_size in enumerate(self.base_sizes):
            center = self.centers[i] if self.centers is not None else None
            out.append(self.gen_single_level_base_anchors(base_size, self.scales, self.ratios, center))
        return out

    def gen_single_level_base_anchors(self, base_size, scales, ratios, center=None):
        w = h = base_size
        if center is None:
            x_center, y_center = self.center_offset * w, self.center_offset * h
        else:
            x_center, y_center = center
        h_ratios = torch.sqrt(ratios)
        w_ratios = 1 / h_ratios
        if self.scale_major:
            ws = (w * w_ratios[:, None] * scales[None, :]).view(-1)
            hs = (h * h_ratios[:, None] * scales[None, :]).view(-1)
        else:
            ws = (w * scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (h * scales[:, None] * h_ratios[None, :]).view(-1)
        return torch.stack([x_center - 0.5 * ws, y_center - 0.5 * hs, x_center + 0.5 * ws,
                            y_center + 0.5 * hs], dim=-1)

    @staticmethod
    def _meshgrid(x, y, row_major=True):
        xx = x.repeat(y.shape[0])
        yy = y.view(-1, 1).repeat(1, x.shape[0]).view(-1)
        return (xx, yy) if row_major else (yy, xx)

    def single_level_grid_anchors(self, base_anchors, featmap_size, stride=(16, 16), device='cuda'):
        feat_h, feat_w = featmap_size
        shift_x = torch.arange(0, feat_w, device=device) * stride[0]
        shift_y = torch.arange(0, feat_h, device=device) * stride[1]
        shift_xx, shift_yy = self._meshgrid(shift_x, shift_y)
        shifts = torch.stack([shift_xx, shift_yy, shift_xx, shift_yy], dim=-1).type_as(base_anchors)
        return (base_anchors[None, :, :] + shifts[:, None, :]).view(-1, 4)

    def grid_anchors(self, featmap_sizes, device='cuda'):
        """anchors of all levels; cached per (map sizes, device): they are constants of the
        geometry, and rebuilding them every step costs an H2D copy (a stream sync) per level"""
        assert self.num_levels == len(featmap_sizes)
        key = (tuple(tuple(int(v) for v in s) for s in featmap_sizes), str(device))
        cache = self.__dict__.setdefault('_grid_cache', {})
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            cache[key] = [self.single_level_grid_anchors(self.base_anchors[i].to(device), featmap_sizes[i],
                                                         self.strides[i], device=device)
                          for i in range(self.num_levels)]
        return list(cache[key])

    grid_priors = grid_anchors

    def single_level_valid_flags(self, featmap_size, valid_size, num_base_anchors, device='cuda'):
        feat_h, feat_w = featmap_size
        valid_h, valid_w = valid_size
        assert valid_h <= feat_h and valid_w <= feat_w
        valid_x = torch.zeros(feat_w, dtype=torch.bool, device=device)
        valid_y = torch.zeros(feat_h, dtype=torch.bool, device=device)
        valid_x[:valid_w] = 1
        valid_y[:valid_h] = 1
        xx, yy = self._meshgrid(valid_x, valid_y)
        valid = xx & yy
        return valid[:, None].expand(valid.size(0), num_base_anchors).contiguous().view(-1)

    def all_valid(self, featmap_sizes, pad_shape):
        """host-side knowledge: True when `valid_flags` would be all ones (the padded image
        covers every cell of every level) -- lets the target code skip the masked gather /
        scatter and their device->host syncs"""
        h, w = pad_shape[:2]
        for i, (feat_h, feat_w) in enumerate(featmap_sizes):
            stride = self.strides[i]
            if int(np.ceil(h / stride[1])) < feat_h or int(np.ceil(w / stride[0])) < feat_w:
                return False
        return True

    def valid_flags(self, featmap_sizes, pad_shape, device='cuda'):
        assert self.num_levels == len(featmap_sizes)
        key = (tuple(tuple(int(v) for v in s) for s in featmap_sizes), tuple(int(v) for v in pad_shape[:2]), str(device))
        cache = self.__dict__.setdefault('_flag_cache', {})
        if key in cache:
            return list(cache[key])
        if len(cache) > 256:
            cache.clear()
        flags = []
        for i in range(self.num_levels):
            stride = self.strides[i]
            feat_h, feat_w = featmap_sizes[i]
            h, w = pad_shape[:2]
            valid_feat_h = min(int(np.ceil(h / stride[1])), feat_h)
            valid_feat_w = min(int(np.ceil(w / stride[0])), feat_w)
            flags.append(self.single_level_valid_flags((feat_h, feat_w),
                                                       (valid_feat_h, valid_feat_w),
                                                       self.num_base_anchors[i], device=device))
        cache[key] = flags
        return list(flags)


_CONST_CACHE = {}


def const_like(values, like):
    """small constant tensor on `like`'s device / dtype, uploaded once (new_tensor of a python
    list is an H2D copy, i.e. a stream sync, on every call)"""
    key = (tuple(float(v) for v in values), str(like.device), like.dtype)
    t = _CONST_CACHE.get(key)
    if t is None:
        if len(_CONST_CACHE) > 256:
            _CONST_CACHE.clear()
        t = _CONST_CACHE[key] = like.new_tensor(list(values))
    return t


def const_rows(rows, like):
    """(R, C) constant tensor (per-image host metadata: image shapes, scale factors) on `like`'s
    device / dtype, uploaded once per distinct value set -- keeps the device-resident test path
    free of H2D copies so that it can be captured in a HIP graph"""
    key = ('rows', tuple(tuple(float(v) for v in r) for r in rows), str(like.device), like.dtype)
    t = _CONST_CACHE.get(key)
    if t is None:
        if len(_CONST_CACHE) > 256:
            _CONST_CACHE.clear()
        t = _CONST_CACHE[key] = like.new_tensor([list(map(float, r)) for r in rows])
    return t


# ----------------------------------------------------------------------------- coder
# The host forms below serve the reference-signature entry points on CPU tensors (tests, the CPU plumbing run of
# BASELINE configs[0]) and autograd callers; device tensors take the HIP kernels.  Each keeps the reference's fp32
# operation ORDER (the parity bar is bit-exact against its goldens g2 / g3 / g4), written once per function over whole
# (x, y) / (w, h) pairs instead of per coordinate, and only with the options the Boosting R-CNN recipes set.
def _centre_size(boxes):
    """(..., 4) corner boxes -> centre (..., 2), size (..., 2)"""
    lo, hi = boxes[..., :2], boxes[..., 2:4]
    return (lo + hi) * 0.5, hi - lo


def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    """delta_xywh_bbox_coder.py:99-141: ((g_ctr - p_ctr) / p_size, log(g_size / p_size)), normalised"""
    assert proposals.size() == gt.size()
    p_ctr, p_size = _centre_size(proposals.float())
    g_ctr, g_size = _centre_size(gt.float())
    deltas = torch.cat([(g_ctr - p_ctr) / p_size, torch.log(g_size / p_size)], dim=-1)
    return deltas.sub_(const_like(means, deltas)).div_(const_like(stds, deltas))


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip=16 / 1000, clip_border=True):
    """delta_xywh_bbox_coder.py:145-272 (the recipes' form: no centre clamp).  `deltas` (..., 4) or (..., 4C):
    class-wise quadruples decode against the same roi; `max_shape` (H, W[, ...]) or one row per leading batch entry."""
    quads = deltas.reshape(deltas.shape[:-1] + (deltas.shape[-1] // 4, 4))     # (..., C, 4)
    d = quads * const_like(stds, deltas) + const_like(means, deltas)
    ctr, size = _centre_size(rois)
    ctr, size = ctr.unsqueeze(-2), size.unsqueeze(-2)
    limit = float(np.abs(np.log(wh_ratio_clip)))
    new_size = size * d[..., 2:].clamp(min=-limit, max=limit).exp()
    new_ctr = ctr + size * d[..., :2]
    half = new_size * 0.5
    boxes = torch.cat([new_ctr - half, new_ctr + half], dim=-1)
    if clip_border and max_shape is not None:
        if not isinstance(max_shape, torch.Tensor):
            flat = all(not isinstance(v, (list, tuple, np.ndarray)) for v in max_shape)
            max_shape = const_like(max_shape, boxes) if flat else const_rows(max_shape, boxes)
        hw = max_shape[..., :2].type_as(boxes)
        if hw.ndim == 2:        # one border per image of a (B, n, ...) batch
            assert boxes.ndim == 4 and hw.size(0) == boxes.size(0)
            hw = hw[:, None, None, :]
        upper = torch.cat([hw.flip(-1), hw.flip(-1)], dim=-1)                  # (W, H, W, H): clips to W / H, not W-1
        boxes = torch.minimum(boxes.clamp(min=0), upper)
    return boxes.reshape(deltas.shape)


@BBOX_CODERS.register_module()
class DeltaXYWHBBoxCoder:
    def __init__(self, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.),
                 clip_border=True, add_ctr_clamp=False, ctr_clamp=32):
        if add_ctr_clamp:
            raise NotImplementedError('add_ctr_clamp: not used by the Boosting R-CNN recipes')
        self.means, self.stds = target_means, target_stds
        self.clip_border, self.add_ctr_clamp, self.ctr_clamp = clip_border, False, ctr_clamp

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0) and bboxes.size(-1) == gt_bboxes.size(-1) == 4
        return bbox2delta(bboxes, gt_bboxes, self.means, self.stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        if pred_bboxes.ndim == 3:
            assert pred_bboxes.size(1) == bboxes.size(1)
        return delta2bbox(bboxes, pred_bboxes, self.means, self.stds, max_shape, wh_ratio_clip, self.clip_border)


# ----------------------------------------------------------------------------- IoU
def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, eps=1e-6):
    """iou2d_calculator.py:75-261: IoU / IoF / GIoU of corner boxes, pairwise (rows x cols) or aligned; no +1
    offset, union floored at `eps`"""
    assert mode in ('iou', 'iof', 'giou'), f'Unsupported mode {mode}'
    assert (bboxes1.size(-1) == 4 or bboxes1.size(0) == 0) and (bboxes2.size(-1) == 4 or bboxes2.size(0) == 0)
    assert bboxes1.shape[:-2] == bboxes2.shape[:-2]
    rows, cols = bboxes1.size(-2), bboxes2.size(-2)
    assert not is_aligned or rows == cols
    if rows * cols == 0:
        return bboxes1.new(bboxes1.shape[:-2] + ((rows,) if is_aligned else (rows, cols)))
    if bboxes1.is_cuda and bboxes1.dim() == 2 and bboxes1.dtype == torch.float32 and \
            not (torch.is_grad_enabled() and (bboxes1.requires_grad or bboxes2.requires_grad)):
        from .train_ops import bbox_overlaps as _dev       # values only: the HIP table kernel
        return _dev(bboxes1, bboxes2, mode, is_aligned, eps)
    a, b = (bboxes1, bboxes2) if is_aligned else (bboxes1.unsqueeze(-2), bboxes2.unsqueeze(-3))
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    inner = (torch.min(a[..., 2:], b[..., 2:]) - torch.max(a[..., :2], b[..., :2])).clamp(min=0)
    overlap = inner[..., 0] * inner[..., 1]
    floor = const_like([eps], overlap)
    union = torch.max(area_a + area_b - overlap if mode != 'iof' else area_a.expand_as(overlap), floor)
    ious = overlap / union
    if mode != 'giou':
        return ious
    outer = (torch.max(a[..., 2:], b[..., 2:]) - torch.min(a[..., :2], b[..., :2])).clamp(min=0)
    hull = torch.max(outer[..., 0] * outer[..., 1], floor)
    return ious - (hull - union) / hull


@IOU_CALCULATORS.register_module()
class BboxOverlaps2D:
    def __init__(self, scale=1., dtype=None):
        self.scale, self.dtype = scale, dtype

    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        assert bboxes1.size(-1) in (0, 4, 5) and bboxes2.size(-1) in (0, 4, 5)
        return bbox_overlaps(bboxes1[..., :4], bboxes2[..., :4], mode, is_aligned)      # a 5th column is a score


# ----------------------------------------------------------------------------- assign
class AssignResult:
    """gt_inds: -1 ignore, 0 background, k > 0 matched to ground truth k-1 (assign_result.py)"""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    @property
    def num_preds(self):
        return len(self.gt_inds)

    def add_gt_(self, gt_labels):
        """the ground truths join the proposals in front, each matched to itself with IoU 1 (:191-205)"""
        k = len(gt_labels)
        self.gt_inds = torch.cat([torch.arange(1, k + 1, dtype=torch.long, device=gt_labels.device), self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(k), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner:
    """max_iou_assigner.py:61-213 with the settings of the recipes: scalar `neg_iou_thr`, every box tied with a ground
    truth's best IoU joins it (`gt_max_assign_all`)."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True,
                 ignore_iof_thr=-1, ignore_wrt_candidates=True, match_low_quality=True,
                 gpu_assign_thr=-1, iou_calculator=dict(type='BboxOverlaps2D')):
        if not gt_max_assign_all or isinstance(neg_iou_thr, (tuple, list)):
            raise NotImplementedError('MaxIoUAssigner: gt_max_assign_all=False / a neg_iou_thr interval are not used '
                                      'by the Boosting R-CNN recipes')
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, float(neg_iou_thr), min_pos_iou
        self.gt_max_assign_all = True
        self.ignore_iof_thr, self.ignore_wrt_candidates = ignore_iof_thr, ignore_wrt_candidates
        self.gpu_assign_thr = gpu_assign_thr
        self.match_low_quality = match_low_quality
        self.iou_calculator = build_iou_calculator(iou_calculator)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        has_ignore = self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0
        if bboxes.is_cuda and bboxes.dim() == 2 and bboxes.shape[0] > 0 and \
                type(self.iou_calculator) is BboxOverlaps2D and not has_ignore:
            return self._assign_device(bboxes, gt_bboxes, gt_labels)
        overlaps = self.iou_calculator(gt_bboxes, bboxes)
        if has_ignore and bboxes.numel() > 0:
            if self.ignore_wrt_candidates:
                covered = self.iou_calculator(bboxes, gt_bboxes_ignore, mode='iof').max(dim=1)[0]
            else:
                covered = self.iou_calculator(gt_bboxes_ignore, bboxes, mode='iof').max(dim=0)[0]
            overlaps[:, covered > self.ignore_iof_thr] = -1
        return self.assign_wrt_overlaps(overlaps, gt_labels)

    def _assign_device(self, bboxes, gt_bboxes, gt_labels):
        """the same assignment for one image on the device kernel (`brcnn_assign_max_iou`, the IoU matrix is
        never materialised); the whole-batch train step calls the kernel once for all images instead"""
        from . import train_ops
        gts = gt_bboxes.reshape(-1, gt_bboxes.shape[-1])[:, :4].float().contiguous()
        n_gt = gts.shape[0]
        gi, mo = train_ops.assign_max_iou(bboxes.float().contiguous(), gts, [0, n_gt], self.pos_iou_thr,
                                          self.neg_iou_thr, self.min_pos_iou, self.match_low_quality, batch=1,
                                          want_overlaps=True)
        gt_inds = gi[0].long()
        labels = None
        if gt_labels is not None:
            if n_gt == 0:
                labels = gt_inds.new_full(gt_inds.shape, -1)
            else:
                labels = torch.where(gt_inds > 0, gt_labels[(gt_inds - 1).clamp(min=0)], gt_inds.new_full((), -1))
        return AssignResult(n_gt, gt_inds, mo[0], labels=labels)

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        """(num_gts, n) IoU table -> AssignResult.  Every step is a `where` over the n boxes (no boolean index
        assignment: that would synchronise with the host on a device table)."""
        num_gts, n = overlaps.shape
        gt_inds = overlaps.new_full((n,), -1, dtype=torch.long)
        if num_gts == 0 or n == 0:
            if num_gts == 0:
                gt_inds[:] = 0                      # no ground truth: everything is background
            labels = None if gt_labels is None else overlaps.new_full((n,), -1, dtype=torch.long)
            return AssignResult(num_gts, gt_inds, overlaps.new_zeros((n,)), labels=labels)
        best_iou, best_gt = overlaps.max(dim=0)
        gt_inds = torch.where((best_iou >= 0) & (best_iou < self.neg_iou_thr), gt_inds.new_zeros(()), gt_inds)
        gt_inds = torch.where(best_iou >= self.pos_iou_thr, best_gt + 1, gt_inds)
        if self.match_low_quality:
            # every ground truth keeps the boxes that reach its own best IoU (>= min_pos_iou); where two ground
            # truths claim a box the later one wins, as the reference's loop over ground truths leaves it (:194-200)
            top = overlaps.max(dim=1)[0]
            claims = (overlaps == top[:, None]) & (top >= self.min_pos_iou)[:, None]
            order = torch.arange(1, num_gts + 1, device=overlaps.device)[:, None]
            winner = (claims * order).max(dim=0)[0]
            gt_inds = torch.where(winner > 0, winner, gt_inds)
        labels = None
        if gt_labels is not None:
            labels = torch.where(gt_inds > 0, gt_labels[(gt_inds - 1).clamp(min=0)], gt_inds.new_full((), -1))
        return AssignResult(num_gts, gt_inds, best_iou, labels=labels)


class SamplingResult:
    """sampling_result.py:26-55: the sampled rows split into positives / negatives with their targets"""

    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            assert self.pos_assigned_gt_inds.numel() == 0
            self.pos_gt_bboxes = torch.empty_like(gt_bboxes).view(-1, 4)
        else:
            self.pos_gt_bboxes = gt_bboxes.view(-1, 4)[self.pos_assigned_gt_inds.long(), :]
        self.pos_gt_labels = None if assign_result.labels is None else assign_result.labels[pos_inds]

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


def _rows_where(mask):
    return torch.nonzero(mask, as_tuple=False).reshape(-1)


@BBOX_SAMPLERS.register_module()
class PseudoSampler:
    """pseudo_sampler.py:24-42: no sampling -- every assigned box is kept"""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        flags = bboxes.new_zeros(bboxes.shape[0], dtype=torch.uint8)
        return SamplingResult(_rows_where(assign_result.gt_inds > 0).unique(), _rows_where(assign_result.gt_inds == 0).unique(),
                              bboxes, gt_bboxes, assign_result, flags)


@BBOX_SAMPLERS.register_module()
class RandomSampler:
    """base_sampler.py:35-102 + random_sampler.py:32-82: up to int(num * pos_fraction) positives, negatives fill up
    to `num` (capped at neg_pos_ub x positives); ground truths join the proposals first."""

    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        self.num, self.pos_fraction = num, pos_fraction
        self.neg_pos_ub, self.add_gt_as_proposals = neg_pos_ub, add_gt_as_proposals

    @staticmethod
    def random_choice(gallery, num):
        """`num` rows of the index tensor `gallery`: the first `num` entries of a HOST randperm over its length -- the
        reference's draw (random_sampler.py:58), so a seeded run picks the same samples"""
        assert len(gallery) >= num
        return gallery[torch.randperm(gallery.numel())[:num].to(device=gallery.device)]

    def _draw(self, candidates, limit):
        return candidates if candidates.numel() <= limit else self.random_choice(candidates, limit)

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, **kwargs):
        bboxes = bboxes.reshape(-1, bboxes.shape[-1])[:, :4]
        gt_flags = bboxes.new_zeros((bboxes.shape[0],), dtype=torch.uint8)
        if self.add_gt_as_proposals and len(gt_bboxes) > 0:
            if gt_labels is None:
                raise ValueError('gt_labels must be given when add_gt_as_proposals is True')
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            assign_result.add_gt_(gt_labels)
            gt_flags = torch.cat([bboxes.new_ones(gt_bboxes.shape[0], dtype=torch.uint8), gt_flags])
        pos = self._draw(_rows_where(assign_result.gt_inds > 0), int(self.num * self.pos_fraction)).unique()
        room = self.num - pos.numel()
        if self.neg_pos_ub >= 0:
            room = min(room, int(self.neg_pos_ub * max(1, pos.numel())))
        neg = self._draw(_rows_where(assign_result.gt_inds == 0), room).unique()
        return SamplingResult(pos, neg, bboxes, gt_bboxes, assign_result, gt_flags)


# ----------------------------------------------------------------------------- transforms
def bbox2roi(bbox_list):
    """list[(n,4|5)] -> (sum n, 5) [batch_ind, x1, y1, x2, y2] (core/bbox/transforms.py:59-78)"""
    rows = [torch.cat([b.new_full((b.size(0), 1), i), b[:, :4]], dim=-1) if b.size(0) > 0 else b.new_zeros((0, 5))
            for i, b in enumerate(bbox_list)]
    return torch.cat(rows, 0)


def bbox2result(bboxes, labels, num_classes):
    """(n,5),(n,) -> list[num_classes] of ndarray (k,5) (core/bbox/transforms.py:100-117)"""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes, labels = bboxes.detach().cpu().numpy(), labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None, return_inds=False):
    """core/post_processing/bbox_nms.py:8-95 on this package's batched_nms: (n, 4 | 4C) boxes and (n, C+1) scores ->
    the (roi, class) candidates above `score_thr` through class-aware NMS, best `max_num`"""
    from .ops import batched_nms
    n, num_classes = multi_scores.size(0), multi_scores.size(1) - 1
    boxes = multi_bboxes.view(n, -1, 4) if multi_bboxes.shape[1] > 4 else multi_bboxes[:, None].expand(n, num_classes, 4)
    scores = multi_scores[:, :num_classes]
    keep_mask = (scores > score_thr).reshape(-1)                    # thresholded BEFORE any score factor, as the reference
    if score_factors is not None:
        scores = scores * score_factors.view(-1, 1)
    inds = _rows_where(keep_mask)
    labels = torch.arange(num_classes, dtype=torch.long, device=scores.device).repeat(n)[inds]
    boxes, scores = boxes.reshape(-1, 4)[inds], scores.reshape(-1)[inds]
    if boxes.numel() == 0:
        dets = torch.cat([boxes, scores[:, None]], -1)
        return (dets, labels, inds) if return_inds else (dets, labels)
    dets, keep = batched_nms(boxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return (dets, labels[keep], inds[keep]) if return_inds else (dets, labels[keep])
