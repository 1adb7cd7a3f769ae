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
def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    assert proposals.size() == gt.size()
    proposals = proposals.float()
    gt = gt.float()
    px = (proposals[..., 0] + proposals[..., 2]) * 0.5
    py = (proposals[..., 1] + proposals[..., 3]) * 0.5
    pw = proposals[..., 2] - proposals[..., 0]
    ph = proposals[..., 3] - proposals[..., 1]
    gx = (gt[..., 0] + gt[..., 2]) * 0.5
    gy = (gt[..., 1] + gt[..., 3]) * 0.5
    gw = gt[..., 2] - gt[..., 0]
    gh = gt[..., 3] - gt[..., 1]
    dx = (gx - px) / pw
    dy = (gy - py) / ph
    dw = torch.log(gw / pw)
    dh = torch.log(gh / ph)
    deltas = torch.stack([dx, dy, dw, dh], dim=-1)
    means = const_like(means, deltas).unsqueeze(0)
    stds = const_like(stds, deltas).unsqueeze(0)
    return deltas.sub_(means).div_(stds)


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip=16 / 1000, clip_border=True, add_ctr_clamp=False, ctr_clamp=32):
    means = const_like(means, deltas).view(1, -1).repeat(1, deltas.size(-1) // 4)
    stds = const_like(stds, deltas).view(1, -1).repeat(1, deltas.size(-1) // 4)
    denorm = deltas * stds + means
    dx, dy, dw, dh = denorm[..., 0::4], denorm[..., 1::4], denorm[..., 2::4], denorm[..., 3::4]
    x1, y1, x2, y2 = rois[..., 0], rois[..., 1], rois[..., 2], rois[..., 3]
    px = ((x1 + x2) * 0.5).unsqueeze(-1).expand_as(dx)
    py = ((y1 + y2) * 0.5).unsqueeze(-1).expand_as(dy)
    pw = (x2 - x1).unsqueeze(-1).expand_as(dw)
    ph = (y2 - y1).unsqueeze(-1).expand_as(dh)
    dx_width = pw * dx
    dy_height = ph * dy
    max_ratio = np.abs(np.log(wh_ratio_clip))
    if add_ctr_clamp:
        dx_width = torch.clamp(dx_width, max=ctr_clamp, min=-ctr_clamp)
        dy_height = torch.clamp(dy_height, max=ctr_clamp, min=-ctr_clamp)
        dw = torch.clamp(dw, max=max_ratio)
        dh = torch.clamp(dh, max=max_ratio)
    else:
        dw = dw.clamp(min=-max_ratio, max=max_ratio)
        dh = dh.clamp(min=-max_ratio, max=max_ratio)
    gw = pw * dw.exp()
    gh = ph * dh.exp()
    gx = px + dx_width
    gy = py + dy_height
    x1 = gx - gw * 0.5
    y1 = gy - gh * 0.5
    x2 = gx + gw * 0.5
    y2 = gy + gh * 0.5
    bboxes = torch.stack([x1, y1, x2, y2], dim=-1).view(deltas.size())
    if clip_border and max_shape is not None:
        if not isinstance(max_shape, torch.Tensor):
            flat = all(not isinstance(v, (list, tuple, np.ndarray)) for v in max_shape)
            max_shape = const_like(max_shape, x1) if flat else const_rows(max_shape, x1)
        max_shape = max_shape[..., :2].type_as(x1)
        if max_shape.ndim == 2:
            assert bboxes.ndim == 3
            assert max_shape.size(0) == bboxes.size(0)
        min_xy = const_like([0], x1)[0]
        max_xy = torch.cat([max_shape] * (deltas.size(-1) // 2), dim=-1).flip(-1).unsqueeze(-2)
        bboxes = torch.where(bboxes < min_xy, min_xy, bboxes)
        bboxes = torch.where(bboxes > max_xy, max_xy, bboxes)
    return bboxes


@BBOX_CODERS.register_module()
class DeltaXYWHBBoxCoder:
    def __init__(self, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.),
                 clip_border=True, add_ctr_clamp=False, ctr_clamp=32):
        self.means, self.stds = target_means, target_stds
        self.clip_border, self.add_ctr_clamp, self.ctr_clamp = clip_border, add_ctr_clamp, ctr_clamp

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == gt_bboxes.size(-1) == 4
        return bbox2delta(bboxes, gt_bboxes, self.means, self.stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        if pred_bboxes.ndim == 3:
            assert pred_bboxes.size(1) == bboxes.size(1)
        return delta2bbox(bboxes, pred_bboxes, self.means, self.stds, max_shape, wh_ratio_clip,
                          self.clip_border, self.add_ctr_clamp, self.ctr_clamp)


# ----------------------------------------------------------------------------- IoU
def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, eps=1e-6):
    assert mode in ['iou', 'iof', 'giou'], f'Unsupported mode {mode}'
    assert (bboxes1.size(-1) == 4 or bboxes1.size(0) == 0)
    assert (bboxes2.size(-1) == 4 or bboxes2.size(0) == 0)
    assert bboxes1.shape[:-2] == bboxes2.shape[:-2]
    batch_shape = bboxes1.shape[:-2]
    rows, cols = bboxes1.size(-2), bboxes2.size(-2)
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return bboxes1.new(batch_shape + ((rows,) if is_aligned else (rows, cols)))
    if bboxes1.is_cuda and bboxes1.dim() == 2 and bboxes1.dtype == torch.float32 and \
            not (torch.is_grad_enabled() and (bboxes1.requires_grad or bboxes2.requires_grad)):
        from .train_ops import bbox_overlaps as _dev       # values only: the HIP table kernel
        return _dev(bboxes1, bboxes2, mode, is_aligned, eps)
    area1 = (bboxes1[..., 2] - bboxes1[..., 0]) * (bboxes1[..., 3] - bboxes1[..., 1])
    area2 = (bboxes2[..., 2] - bboxes2[..., 0]) * (bboxes2[..., 3] - bboxes2[..., 1])
    if is_aligned:
        lt = torch.max(bboxes1[..., :2], bboxes2[..., :2])
        rb = torch.min(bboxes1[..., 2:], bboxes2[..., 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1 + area2 - overlap if mode in ['iou', 'giou'] else area1
        if mode == 'giou':
            enclosed_lt = torch.min(bboxes1[..., :2], bboxes2[..., :2])
            enclosed_rb = torch.max(bboxes1[..., 2:], bboxes2[..., 2:])
    else:
        lt = torch.max(bboxes1[..., :, None, :2], bboxes2[..., None, :, :2])
        rb = torch.min(bboxes1[..., :, None, 2:], bboxes2[..., None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1[..., None] + area2[..., None, :] - overlap if mode in ['iou', 'giou'] \
            else area1[..., None]
        if mode == 'giou':
            enclosed_lt = torch.min(bboxes1[..., :, None, :2], bboxes2[..., None, :, :2])
            enclosed_rb = torch.max(bboxes1[..., :, None, 2:], bboxes2[..., None, :, 2:])
    eps = const_like([eps], union)
    union = torch.max(union, eps)
    ious = overlap / union
    if mode in ['iou', 'iof']:
        return ious
    enclose_wh = (enclosed_rb - enclosed_lt).clamp(min=0)
    enclose_area = torch.max(enclose_wh[..., 0] * enclose_wh[..., 1], eps)
    return ious - (enclose_area - union) / enclose_area


@IOU_CALCULATORS.register_module()
class BboxOverlaps2D:
    def __init__(self, scale=1., dtype=None):
        self.scale, self.dtype = scale, dtype

    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        assert bboxes1.size(-1) in [0, 4, 5]
        assert bboxes2.size(-1) in [0, 4, 5]
        if bboxes2.size(-1) == 5:
            bboxes2 = bboxes2[..., :4]
        if bboxes1.size(-1) == 5:
            bboxes1 = bboxes1[..., :4]
        return bbox_overlaps(bboxes1, bboxes2, mode, is_aligned)


# ----------------------------------------------------------------------------- assign
class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = \
            num_gts, gt_inds, max_overlaps, labels

    @property
    def num_preds(self):
        return len(self.gt_inds)

    def add_gt_(self, gt_labels):
        self_inds = torch.arange(1, len(gt_labels) + 1, dtype=torch.long, device=gt_labels.device)
        self.gt_inds = torch.cat([self_inds, self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(len(gt_labels)), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner:
    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True,
                 ignore_iof_thr=-1, ignore_wrt_candidates=True, match_low_quality=True,
                 gpu_assign_thr=-1, iou_calculator=dict(type='BboxOverlaps2D')):
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all = gt_max_assign_all
        self.ignore_iof_thr, self.ignore_wrt_candidates = ignore_iof_thr, ignore_wrt_candidates
        self.gpu_assign_thr = gpu_assign_thr
        self.match_low_quality = match_low_quality
        self.iou_calculator = build_iou_calculator(iou_calculator)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        if bboxes.is_cuda and bboxes.dim() == 2 and bboxes.shape[0] > 0 and self.gt_max_assign_all and \
                type(self.iou_calculator) is BboxOverlaps2D and \
                not (self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0):
            return self._assign_device(bboxes, gt_bboxes, gt_labels)
        overlaps = self.iou_calculator(gt_bboxes, bboxes)
        if (self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None
                and gt_bboxes_ignore.numel() > 0 and bboxes.numel() > 0):
            if self.ignore_wrt_candidates:
                ignore_max, _ = self.iou_calculator(bboxes, gt_bboxes_ignore, mode='iof').max(dim=1)
            else:
                ignore_max, _ = self.iou_calculator(gt_bboxes_ignore, bboxes, mode='iof').max(dim=0)
            overlaps[:, ignore_max > self.ignore_iof_thr] = -1
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
        num_gts, num_bboxes = overlaps.size(0), overlaps.size(1)
        assigned = overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            max_overlaps = overlaps.new_zeros((num_bboxes,))
            if num_gts == 0:
                assigned[:] = 0
            labels = None if gt_labels is None else overlaps.new_full((num_bboxes,), -1,
                                                                      dtype=torch.long)
            return AssignResult(num_gts, assigned, max_overlaps, labels=labels)
        max_overlaps, argmax_overlaps = overlaps.max(dim=0)
        gt_max_overlaps, gt_argmax_overlaps = overlaps.max(dim=1)
        # masked assignments as `where` (same values; boolean index_put_ would sync with the host)
        zero = assigned.new_zeros(())
        if isinstance(self.neg_iou_thr, float):
            assigned = torch.where((max_overlaps >= 0) & (max_overlaps < self.neg_iou_thr), zero, assigned)
        elif isinstance(self.neg_iou_thr, tuple):
            assert len(self.neg_iou_thr) == 2
            assigned = torch.where((max_overlaps >= self.neg_iou_thr[0]) & (max_overlaps < self.neg_iou_thr[1]),
                                   zero, assigned)
        pos_inds = max_overlaps >= self.pos_iou_thr
        assigned = torch.where(pos_inds, argmax_overlaps + 1, assigned)
        if self.match_low_quality:
            # the reference loops over gts in order, later gts overriding earlier ones
            # (max_iou_assigner.py:194-200); the same result without a host loop:
            ok = gt_max_overlaps >= self.min_pos_iou
            if self.gt_max_assign_all:
                hit = (overlaps == gt_max_overlaps[:, None]) & ok[:, None]
                rank = torch.arange(1, num_gts + 1, device=overlaps.device)[:, None]
                best = (hit * rank).max(dim=0)[0]
                assigned = torch.where(best > 0, best, assigned)
            else:
                for i in torch.nonzero(ok, as_tuple=False).flatten().tolist():
                    assigned[gt_argmax_overlaps[i]] = i + 1
        if gt_labels is not None:
            labels = torch.where(assigned > 0, gt_labels[(assigned - 1).clamp(min=0)],
                                 assigned.new_full((), -1))
        else:
            labels = None
        return AssignResult(num_gts, assigned, max_overlaps, labels=labels)


class SamplingResult:
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
            if len(gt_bboxes.shape) < 2:
                gt_bboxes = gt_bboxes.view(-1, 4)
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds.long(), :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None \
            else None

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


@BBOX_SAMPLERS.register_module()
class PseudoSampler:
    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        gt_flags = bboxes.new_zeros(bboxes.shape[0], dtype=torch.uint8)
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)


@BBOX_SAMPLERS.register_module()
class RandomSampler:
    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        self.num, self.pos_fraction = num, pos_fraction
        self.neg_pos_ub, self.add_gt_as_proposals = neg_pos_ub, add_gt_as_proposals

    def random_choice(self, gallery, num):
        assert len(gallery) >= num
        is_tensor = isinstance(gallery, torch.Tensor)
        if not is_tensor:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 'cpu'
            gallery = torch.tensor(gallery, dtype=torch.long, device=device)
        # host randperm, as in the reference (random_sampler.py:58): seeded parity
        perm = torch.randperm(gallery.numel())[:num].to(device=gallery.device)
        rand_inds = gallery[perm]
        if not is_tensor:
            rand_inds = rand_inds.cpu().numpy()
        return rand_inds

    def _sample_pos(self, assign_result, num_expected, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False)
        if pos_inds.numel() != 0:
            pos_inds = pos_inds.squeeze(1)
        if pos_inds.numel() <= num_expected:
            return pos_inds
        return self.random_choice(pos_inds, num_expected)

    def _sample_neg(self, assign_result, num_expected, **kwargs):
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False)
        if neg_inds.numel() != 0:
            neg_inds = neg_inds.squeeze(1)
        if len(neg_inds) <= num_expected:
            return neg_inds
        return self.random_choice(neg_inds, num_expected)

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, **kwargs):
        if len(bboxes.shape) < 2:
            bboxes = bboxes[None, :]
        bboxes = bboxes[:, :4]
        gt_flags = bboxes.new_zeros((bboxes.shape[0],), dtype=torch.uint8)
        if self.add_gt_as_proposals and len(gt_bboxes) > 0:
            if gt_labels is None:
                raise ValueError('gt_labels must be given when add_gt_as_proposals is True')
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            assign_result.add_gt_(gt_labels)
            gt_ones = bboxes.new_ones(gt_bboxes.shape[0], dtype=torch.uint8)
            gt_flags = torch.cat([gt_ones, gt_flags])
        num_expected_pos = int(self.num * self.pos_fraction)
        pos_inds = self._sample_pos(assign_result, num_expected_pos, bboxes=bboxes, **kwargs)
        pos_inds = pos_inds.unique()
        num_sampled_pos = pos_inds.numel()
        num_expected_neg = self.num - num_sampled_pos
        if self.neg_pos_ub >= 0:
            _pos = max(1, num_sampled_pos)
            neg_upper_bound = int(self.neg_pos_ub * _pos)
            if num_expected_neg > neg_upper_bound:
                num_expected_neg = neg_upper_bound
        neg_inds = self._sample_neg(assign_result, num_expected_neg, bboxes=bboxes, **kwargs)
        neg_inds = neg_inds.unique()
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)


# ----------------------------------------------------------------------------- transforms
def bbox2roi(bbox_list):
    """list[(n,4|5)] -> (sum n, 5) [batch_ind, x1, y1, x2, y2] (core/bbox/transforms.py:59-78)"""
    rois_list = []
    for img_id, bboxes in enumerate(bbox_list):
        if bboxes.size(0) > 0:
            img_inds = bboxes.new_full((bboxes.size(0), 1), img_id)
            rois = torch.cat([img_inds, bboxes[:, :4]], dim=-1)
        else:
            rois = bboxes.new_zeros((0, 5))
        rois_list.append(rois)
    return torch.cat(rois_list, 0)


def bbox2result(bboxes, labels, num_classes):
    """(n,5),(n,) -> list[num_classes] of ndarray (k,5) (core/bbox/transforms.py:100-117)"""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes = bboxes.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1,
                   score_factors=None, return_inds=False):
    """core/post_processing/bbox_nms.py:8-95 on top of this package's batched_nms."""
    from .ops import batched_nms
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
    scores = multi_scores[:, :-1]
    labels = torch.arange(num_classes, dtype=torch.long, device=scores.device)
    labels = labels.view(1, -1).expand_as(scores)
    bboxes = bboxes.reshape(-1, 4)
    scores = scores.reshape(-1)
    labels = labels.reshape(-1)
    valid_mask = scores > score_thr
    if score_factors is not None:
        score_factors = score_factors.view(-1, 1).expand(multi_scores.size(0), num_classes)
        scores = scores * score_factors.reshape(-1)
    inds = valid_mask.nonzero(as_tuple=False).squeeze(1)
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    if bboxes.numel() == 0:
        dets = torch.cat([bboxes, scores[:, None]], -1)
        return (dets, labels, inds) if return_inds else (dets, labels)
    dets, keep = batched_nms(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets = dets[:max_num]
        keep = keep[:max_num]
    if return_inds:
        return dets, labels[keep], inds[keep]
    return dets, labels[keep]
