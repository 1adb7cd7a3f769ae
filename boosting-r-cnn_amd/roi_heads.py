"""Boosting R-CNN second stage: RoI feature extraction, the 2-FC box head, the boosting
re-weighted loss and the prior-fused test-time scoring.

Mirrors mmdet/models/roi_heads/{prob_roi_head.py:10-283 (ProbRoIHead),
standard_roi_head.py, base_roi_head.py}, roi_extractors/{base_roi_extractor.py:37-60,
single_level_roi_extractor.py:36-115}, bbox_heads/{bbox_head.py:19-253,
convfc_bbox_head.py:25-192,283-451 (ProbConvFCBBoxHead)}: same registry names, constructor
arguments and parameter names (`bbox_head.{shared_fcs.{i},fc_cls,fc_reg}`).  Execution:
  * RoI extraction = one fused launch (level mapping + RoIAlign on NHWC maps);
  * the FCs are MFMA GEMMs on the (K,7,7,C) RoI tensor; the first FC's weight columns are
    permuted once from the reference's (C,7,7) flatten order to (7,7,C);
  * test-time: sqrt(softmax * prior), per-class decode, threshold, class-aware NMS for the
    whole batch stay on the device (postprocess.batched_nms_images).
"""
import os as _os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .blocks import PackedCache, to_nhwc
from .core import bbox2result, bbox2roi, bbox_overlaps, const_rows, multi_apply, multiclass_nms
from .losses import SmoothL1Loss, accuracy
from .postprocess import batched_nms_images
from .profiling import stage_mark
from .registry import (HEADS, ROI_EXTRACTORS, build_assigner, build_bbox_coder, build_head,
                       build_loss, build_roi_extractor, build_sampler)

# BRCNN_TIME_SYNC=1: [seconds the host blocked on the sampler counts, calls] (tools/experiments/host_slack.py)
SYNC_WAIT = [0.0, 0] if _os.environ.get('BRCNN_TIME_SYNC') == '1' else None


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


# ----------------------------------------------------------------------------- extractor
@ROI_EXTRACTORS.register_module()
class SingleRoIExtractor(nn.Module):
    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56, init_cfg=None):
        super().__init__()
        self.roi_layers = self.build_roi_layers(roi_layer, featmap_strides)
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
        self.finest_scale = finest_scale

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def build_roi_layers(self, layer_cfg, featmap_strides):
        """`getattr(ops, type)`: the plugin seam of base_roi_extractor.py:54-60, resolved
        against this package's operator namespace."""
        cfg = dict(layer_cfg)
        layer_type = cfg.pop('type')
        assert hasattr(ops, layer_type)
        layer_cls = getattr(ops, layer_type)
        return nn.ModuleList([layer_cls(spatial_scale=1 / s, **cfg) for s in featmap_strides])

    def map_roi_levels(self, rois, num_levels):
        scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
        target_lvls = torch.floor(torch.log2(scale / self.finest_scale + 1e-6))
        return target_lvls.clamp(min=0, max=num_levels - 1).long()

    def _fusable(self):
        l0 = self.roi_layers[0]
        return all(isinstance(l, ops.RoIAlign) and l.pool_mode == 'avg' and l.aligned and
                   l.output_size == l0.output_size and l.sampling_ratio == l0.sampling_ratio
                   for l in self.roi_layers)

    def forward_nhwc(self, feats_nhwc, rois):
        """(K, ph, pw, C) RoI features from NHWC maps, one launch."""
        assert self._fusable()
        l0 = self.roi_layers[0]
        if torch.is_grad_enabled() and any(f.requires_grad for f in feats_nhwc):
            from .autograd import roi_extract_autograd
            # a 16-bit pyramid is read as it is and its gradient written in the same dtype (the gather backward
            # accumulates in fp32): no widening / narrowing casts around the extractor
            from .autograd import ROI_BACKWARD_GATHER
            ph_pw = l0.output_size if isinstance(l0.output_size, (tuple, list)) else (l0.output_size, l0.output_size)
            if feats_nhwc[0].dtype != torch.float32 and not (ROI_BACKWARD_GATHER and max(ph_pw) <= 7):
                feats_nhwc = [f.float() for f in feats_nhwc]
            return roi_extract_autograd(list(feats_nhwc[:self.num_inputs]), rois, l0.output_size,
                                        self.featmap_strides, self.finest_scale, l0.sampling_ratio)
        out, _ = ops.roi_extract(list(feats_nhwc[:self.num_inputs]), rois, l0.output_size,
                                 self.featmap_strides, self.finest_scale, l0.sampling_ratio)
        return out

    def forward(self, feats, rois, roi_scale_factor=None):
        """reference signature: feats tuple of (N,C,h,w), rois (K,5) -> (K,C,ph,pw)"""
        assert roi_scale_factor is None
        if self._fusable() and feats[0].shape[1] % 4 == 0:
            out = self.forward_nhwc([to_nhwc(f).contiguous() for f in feats], rois)
            return out.permute(0, 3, 1, 2)
        out_size = self.roi_layers[0].output_size
        num_levels = len(feats)
        roi_feats = feats[0].new_zeros(rois.size(0), self.out_channels, *out_size)
        if num_levels == 1:
            return roi_feats if len(rois) == 0 else self.roi_layers[0](feats[0], rois)
        target_lvls = self.map_roi_levels(rois, num_levels)
        for i in range(num_levels):
            inds = (target_lvls == i).nonzero(as_tuple=False).squeeze(1)
            if inds.numel() > 0:
                roi_feats[inds] = self.roi_layers[i](feats[i], rois[inds])
        return roi_feats


# ----------------------------------------------------------------------------- bbox head
@HEADS.register_module()
class ProbConvFCBBoxHead(nn.Module):
    """ConvFCBBoxHead (convfc_bbox_head.py:15-200): shared convs -> shared fcs, then separate
    cls / reg branches of convs and fcs.  The COCO / UTDAC recipes build two shared FCs (fast
    path: the two predictors share one GEMM); the VOC recipe builds 2 cls FCs and 4 GN reg
    convs (general path: ConvModules on the NHWC RoI features, FCs on the MFMA linear kernel)."""

    def __init__(self, num_shared_convs=0, num_shared_fcs=0, num_cls_convs=0, num_cls_fcs=0,
                 num_reg_convs=0, num_reg_fcs=0, conv_out_channels=256, fc_out_channels=1024,
                 conv_cfg=None, norm_cfg=None, init_cfg=None, focal_reg=False, gamma=1,
                 with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7,
                 in_channels=256, num_classes=80,
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True,
                                 target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2]),
                 reg_class_agnostic=False, reg_decoded_bbox=False,
                 reg_predictor_cfg=dict(type='Linear'), cls_predictor_cfg=dict(type='Linear'),
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)):
        super().__init__()
        assert with_cls and with_reg and not with_avg_pool
        assert num_shared_convs + num_shared_fcs + num_cls_convs + num_cls_fcs + num_reg_convs + num_reg_fcs > 0
        if num_cls_convs > 0 or num_reg_convs > 0:
            assert num_shared_fcs == 0
        assert reg_predictor_cfg.get('type') == 'Linear' and cls_predictor_cfg.get('type') == 'Linear'
        self.with_cls, self.with_reg, self.with_avg_pool = with_cls, with_reg, with_avg_pool
        self.roi_feat_size = _pair(roi_feat_size)
        self.roi_feat_area = self.roi_feat_size[0] * self.roi_feat_size[1]
        self.in_channels, self.num_classes = in_channels, num_classes
        self.reg_class_agnostic, self.reg_decoded_bbox = reg_class_agnostic, reg_decoded_bbox
        self.num_shared_fcs, self.fc_out_channels = num_shared_fcs, fc_out_channels
        self.focal_reg, self.gamma = focal_reg, gamma
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.num_shared_convs, self.num_cls_convs, self.num_cls_fcs = num_shared_convs, num_cls_convs, num_cls_fcs
        self.num_reg_convs, self.num_reg_fcs = num_reg_convs, num_reg_fcs
        self.conv_out_channels, self.conv_cfg, self.norm_cfg = conv_out_channels, conv_cfg, norm_cfg
        self.shared_convs, self.shared_fcs, last = self._add_conv_fc_branch(
            num_shared_convs, num_shared_fcs, in_channels, True)
        self.shared_out_channels = last
        self.cls_convs, self.cls_fcs, self.cls_last_dim = self._add_conv_fc_branch(
            num_cls_convs, num_cls_fcs, self.shared_out_channels)
        self.reg_convs, self.reg_fcs, self.reg_last_dim = self._add_conv_fc_branch(
            num_reg_convs, num_reg_fcs, self.shared_out_channels)
        if num_shared_fcs == 0:
            if num_cls_fcs == 0:
                self.cls_last_dim *= self.roi_feat_area
            if num_reg_fcs == 0:
                self.reg_last_dim *= self.roi_feat_area
        self.relu = nn.ReLU(inplace=True)
        self.fc_cls = nn.Linear(self.cls_last_dim, num_classes + 1)
        self.fc_reg = nn.Linear(self.reg_last_dim, 4 if reg_class_agnostic else 4 * num_classes)
        self._simple = (num_shared_convs == num_cls_convs == num_reg_convs == 0 and
                        num_cls_fcs == num_reg_fcs == 0 and num_shared_fcs > 0)
        self._fc_caches = {}
        self._caches = [PackedCache() for _ in range(num_shared_fcs + 1)]
        if self._simple:
            # tags for optim.FusedSGD.register_conv_weights: the shared FCs' 16-bit operands are written by the
            # optimizer step (the first one in the (ph, pw, C) column order the NHWC RoI features multiply)
            ph_pw = self.roi_feat_size if isinstance(self.roi_feat_size, (tuple, list)) else (self.roi_feat_size,) * 2
            for i, fc in enumerate(self.shared_fcs):
                if i == 0:
                    fc._brcnn_fc_perm = (self.in_channels, int(ph_pw[0]), int(ph_pw[1]))
                else:
                    fc._brcnn_pack_linear = True
        self.init_weights()

    custom_cls_channels = False
    custom_activation = False
    custom_accuracy = False

    def _add_conv_fc_branch(self, num_branch_convs, num_branch_fcs, in_channels, is_shared=False):
        """convfc_bbox_head.py:114-152"""
        from .blocks import ConvModule
        last = in_channels
        convs = nn.ModuleList()
        for i in range(num_branch_convs):
            convs.append(ConvModule(last if i == 0 else self.conv_out_channels, self.conv_out_channels, 3,
                                    padding=1, conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
        if num_branch_convs > 0:
            last = self.conv_out_channels
        fcs = nn.ModuleList()
        if num_branch_fcs > 0:
            if is_shared or self.num_shared_fcs == 0:
                last *= self.roi_feat_area
            for i in range(num_branch_fcs):
                fcs.append(nn.Linear(last if i == 0 else self.fc_out_channels, self.fc_out_channels))
            last = self.fc_out_channels
        return convs, fcs, last

    def init_weights(self):
        """Xavier for shared / cls / reg fcs (convfc_bbox_head.py:102-112), Normal .01/.001 for
        fc_cls/fc_reg (bbox_head.py:83-94)"""
        for fc in list(self.shared_fcs) + list(self.cls_fcs) + list(self.reg_fcs):
            nn.init.xavier_normal_(fc.weight, gain=1)
            nn.init.constant_(fc.bias, 0)
        nn.init.normal_(self.fc_cls.weight, 0, 0.01)
        nn.init.constant_(self.fc_cls.bias, 0)
        nn.init.normal_(self.fc_reg.weight, 0, 0.001)
        nn.init.constant_(self.fc_reg.bias, 0)

    # ---- execution -------------------------------------------------------------------
    def forward_nhwc(self, roi_feats):
        """roi_feats (K, ph, pw, C) -> (cls_score (K,C+1), bbox_pred (K,4C))"""
        if not self._simple:
            return self._forward_general(roi_feats)
        k, ph, pw, c = roi_feats.shape
        x = roi_feats.reshape(k, ph * pw * c)
        from .autograd import linear_autograd, permuted_fc_weight, wants_grad
        if wants_grad(x, self.fc_cls.weight, self.shared_fcs[0].weight):
            from .blocks import compute_dtype
            if compute_dtype() != torch.float32:
                x = x.to(compute_dtype())     # 16-bit modes: the FC GEMMs (fwd / dgrad / wgrad) on bf16 / fp16 MFMA
            for i, fc in enumerate(self.shared_fcs):
                w = fc.weight
                if i == 0:   # (out, C*ph*pw) columns -> (ph,pw,C) order, differentiable (its gradient's way back on the second stream)
                    w = permuted_fc_weight(w, c, ph, pw, x.dtype)
                x = linear_autograd(x, w, fc.bias, relu=True)       # (ReLU in the GEMM's epilogue)
            # fc_cls | fc_reg as one GEMM through one autograd node (cat + pad; its backward hands out views of dW, so the
            # weight-gradient launch leaves the main stream: autograd.FusedHeadWeights)
            from .autograd import fused_head_weights
            w, b = fused_head_weights((self.fc_cls, self.fc_reg), 32 if x.dtype == torch.float32 else 64)
            from .autograd import SplitColumns
            y = linear_autograd(x, w, b, out_f32=True)          # fp32 scores / deltas straight from the kernel
            return SplitColumns.apply(y, self.fc_cls.out_features, self.fc_reg.out_features)
        for i, fc in enumerate(self.shared_fcs):
            if i == 0:
                def builder(fc=fc):   # (out, C*ph*pw) columns -> (ph,pw,C) order
                    w = fc.weight.detach().float().view(-1, c, ph, pw).permute(0, 2, 3, 1)
                    return w.reshape(fc.out_features, -1).to(x.dtype).contiguous(), \
                        fc.bias.detach().float().contiguous()
            else:
                def builder(fc=fc):
                    return fc.weight.detach().float().to(x.dtype).contiguous(), \
                        fc.bias.detach().float().contiguous()
            w, b = self._caches[i].get([fc.weight, fc.bias], builder)
            x = ops.linear_nhwc(x, w, b, True)

        def head_builder():   # fc_cls and fc_reg share the input: one GEMM
            w = torch.cat([self.fc_cls.weight, self.fc_reg.weight], 0).detach().float().to(x.dtype).contiguous()
            b = torch.cat([self.fc_cls.bias, self.fc_reg.bias], 0).detach().float().contiguous()
            return w, b
        w, b = self._caches[-1].get([self.fc_cls.weight, self.fc_cls.bias, self.fc_reg.weight,
                                     self.fc_reg.bias], head_builder)
        y = ops.linear_nhwc(x, w, b, False, out_f32=True)   # fp32 scores / deltas in either mode
        nc = self.fc_cls.out_features
        return y[:, :nc], y[:, nc:]

    # ---- general ConvFC structure (VOC recipe) ------------------------------------------
    def _fc(self, x, fc, relu, spatial=None, out_f32=False):
        """x (K, features) through nn.Linear `fc` on the MFMA linear kernel.  `spatial` =
        (ph, pw, C) when x is a flattened NHWC map: the reference flattens (C, ph, pw), so the
        weight columns are permuted once to the NHWC order."""
        from .autograd import linear_autograd, permuted_fc_weight, wants_grad
        if wants_grad(x, fc.weight, fc.bias):
            w = fc.weight
            if spatial is not None:
                ph, pw, c = spatial
                w = w.view(-1, c, ph, pw).permute(0, 2, 3, 1).reshape(fc.out_features, -1)
            y = linear_autograd(x.float(), w, fc.bias)
            return y.relu() if relu else y
        cache = self._fc_caches.setdefault(id(fc), PackedCache())

        def builder():
            w = fc.weight.detach().float()
            if spatial is not None:
                ph, pw, c = spatial
                w = w.view(-1, c, ph, pw).permute(0, 2, 3, 1).reshape(fc.out_features, -1)
            return w.to(x.dtype).contiguous(), fc.bias.detach().float().contiguous()
        w, b = cache.get([fc.weight, fc.bias], builder)
        return ops.linear_nhwc(x.contiguous(), w, b, relu, out_f32=out_f32)

    def _branch(self, x, spatial, convs, fcs):
        for conv in convs:
            x = conv.forward_nhwc(x)
            spatial = (x.shape[1], x.shape[2], x.shape[3])
        if x.dim() > 2:
            x = x.reshape(x.shape[0], -1)
        for i, fc in enumerate(fcs):
            x = self._fc(x, fc, True, spatial if i == 0 else None)
            spatial = None
        return x, spatial

    def _forward_general(self, roi_feats):
        k, ph, pw, c = roi_feats.shape
        x, spatial = roi_feats, (ph, pw, c)
        for conv in self.shared_convs:
            x = conv.forward_nhwc(x)
            spatial = (x.shape[1], x.shape[2], x.shape[3])
        if self.num_shared_fcs > 0:
            x = x.reshape(k, -1)
            for i, fc in enumerate(self.shared_fcs):
                x = self._fc(x, fc, True, spatial if i == 0 else None)
            spatial = None
        x_cls, sp_cls = self._branch(x, spatial, self.cls_convs, self.cls_fcs)
        x_reg, sp_reg = self._branch(x, spatial, self.reg_convs, self.reg_fcs)
        cls_score = self._fc(x_cls, self.fc_cls, False, sp_cls, out_f32=True)
        bbox_pred = self._fc(x_reg, self.fc_reg, False, sp_reg, out_f32=True)
        return cls_score.float(), bbox_pred.float()

    def forward(self, x):
        """reference signature: x (K,C,ph,pw) -> cls_score, bbox_pred"""
        return self.forward_nhwc(to_nhwc(x).contiguous())

    # ---- targets / loss ----------------------------------------------------------------
    def _get_target_single(self, pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, cfg):
        num_pos, num_neg = pos_bboxes.size(0), neg_bboxes.size(0)
        num_samples = num_pos + num_neg
        labels = pos_bboxes.new_full((num_samples,), self.num_classes, dtype=torch.long)
        label_weights = pos_bboxes.new_zeros(num_samples)
        bbox_targets = pos_bboxes.new_zeros(num_samples, 4)
        bbox_weights = pos_bboxes.new_zeros(num_samples, 4)
        if num_pos > 0:
            labels[:num_pos] = pos_gt_labels
            label_weights[:num_pos] = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
            if not self.reg_decoded_bbox:
                pos_bbox_targets = self.bbox_coder.encode(pos_bboxes, pos_gt_bboxes)
            else:
                pos_bbox_targets = pos_gt_bboxes
            bbox_targets[:num_pos, :] = pos_bbox_targets
            bbox_weights[:num_pos, :] = 1
        if num_neg > 0:
            label_weights[-num_neg:] = 1.0
        return labels, label_weights, bbox_targets, bbox_weights

    def get_targets(self, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg, concat=True):
        out = multi_apply(self._get_target_single,
                          [r.pos_bboxes for r in sampling_results],
                          [r.neg_bboxes for r in sampling_results],
                          [r.pos_gt_bboxes for r in sampling_results],
                          [r.pos_gt_labels for r in sampling_results], cfg=rcnn_train_cfg)
        if concat:
            out = tuple(torch.cat(o, 0) for o in out)
        return out

    def loss(self, cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_weights,
             reduction_override=None):
        losses = dict()
        if bbox_pred is not None:
            bg = self.num_classes
            pos = ((labels >= 0) & (labels < bg)).type(torch.bool)
            if pos.any():
                if self.reg_decoded_bbox:
                    bbox_pred = self.bbox_coder.decode(rois[:, 1:], bbox_pred)
                if self.reg_class_agnostic:
                    pos_bbox_pred = bbox_pred.view(bbox_pred.size(0), 4)[pos]
                else:
                    pos_bbox_pred = bbox_pred.view(bbox_pred.size(0), -1, 4)[pos, labels[pos]]
                if self.focal_reg:
                    # (the reference evaluates iou_target unconditionally but only uses it here)
                    if self.reg_decoded_bbox:
                        iou_target = bbox_overlaps(pos_bbox_pred.detach(), bbox_targets[pos],
                                                   is_aligned=True)
                    else:
                        dp = self.bbox_coder.decode(rois[pos, 1:], pos_bbox_pred)
                        dt = self.bbox_coder.decode(rois[pos, 1:], bbox_targets[pos])
                        iou_target = bbox_overlaps(dp.detach(), dt, is_aligned=True)
                    bw = bbox_weights[pos] * iou_target[pos, None] ** self.gamma
                    losses['loss_bbox'] = self.loss_bbox(
                        pos_bbox_pred, bbox_targets[pos], bw.clamp(min=1e-12),
                        avg_factor=iou_target[pos].sum(), reduction_override=reduction_override)
                else:
                    losses['loss_bbox'] = self.loss_bbox(
                        pos_bbox_pred, bbox_targets[pos], bbox_weights[pos],
                        avg_factor=bbox_targets.size(0), reduction_override=reduction_override)
            else:
                losses['loss_bbox'] = bbox_pred[pos].sum()
        if cls_score is not None:
            avg_factor = max(torch.sum(label_weights > 0).float().item(), 1.)
            if cls_score.numel() > 0:
                losses['loss_cls'] = self.loss_cls(cls_score, labels, label_weights,
                                                   avg_factor=avg_factor,
                                                   reduction_override=reduction_override)
                losses['acc'] = accuracy(cls_score, labels)
        return losses

    # ---- test ------------------------------------------------------------------------
    def get_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False,
                   cfg=None):
        """ProbConvFCBBoxHead.get_bboxes (convfc_bbox_head.py:294-330): scores arrive already
        fused (no softmax here)."""
        scores = cls_score
        if bbox_pred is not None:
            bboxes = self.bbox_coder.decode(rois[..., 1:], bbox_pred, max_shape=img_shape)
        else:
            bboxes = rois[:, 1:].clone()
            if img_shape is not None:
                bboxes[:, [0, 2]].clamp_(min=0, max=img_shape[1])
                bboxes[:, [1, 3]].clamp_(min=0, max=img_shape[0])
        if rescale and bboxes.size(0) > 0:
            scale_factor = bboxes.new_tensor(scale_factor)
            bboxes = (bboxes.view(bboxes.size(0), -1, 4) / scale_factor).view(bboxes.size()[0], -1)
        if cfg is None:
            return bboxes, scores
        return multiclass_nms(bboxes, scores, cfg.score_thr, cfg.nms, cfg.max_per_img)


# ----------------------------------------------------------------------------- roi head
@HEADS.register_module()
class ProbRoIHead(nn.Module):
    def __init__(self, alpha=0, gamma=0.1, boost=False, prob=True, ams=False, quality=False,
                 iou_gamma=0, reg_norm='bbox_num', bbox_roi_extractor=None, bbox_head=None,
                 mask_roi_extractor=None, mask_head=None, shared_head=None, train_cfg=None,
                 test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__()
        assert mask_head is None and shared_head is None and not ams, \
            'mask / shared heads are outside the hot path'
        self.alpha, self.gamma, self.boost, self.prob = alpha, gamma, boost, prob
        self.quality, self.iou_gamma, self.reg_norm = quality, iou_gamma, reg_norm
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.bbox_roi_extractor = build_roi_extractor(bbox_roi_extractor)
        self.bbox_head = build_head(bbox_head)
        self.bbox_assigner = self.bbox_sampler = None
        if self.train_cfg:
            self.bbox_assigner = build_assigner(self.train_cfg.assigner)
            self.bbox_sampler = build_sampler(self.train_cfg.sampler, context=self)

    with_bbox = True
    with_mask = False
    with_shared_head = False

    # ---- train -------------------------------------------------------------------------
    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels,
                      gt_bboxes_ignore=None, gt_masks=None):
        num_imgs = len(img_metas)
        if gt_bboxes_ignore is None:
            gt_bboxes_ignore = [None for _ in range(num_imgs)]
        sampling_results, priors, ious = [], [], []
        for i in range(num_imgs):
            assign_result = self.bbox_assigner.assign(proposal_list[i], gt_bboxes[i],
                                                      gt_bboxes_ignore[i], gt_labels[i])
            sampling_result = self.bbox_sampler.sample(assign_result, proposal_list[i],
                                                       gt_bboxes[i], gt_labels[i],
                                                       feats=[lvl[i][None] for lvl in x])
            sampling_results.append(sampling_result)
            # prior extraction (prob_roi_head.py:51-64): assumes the first num_gts sampled
            # positives are the GT boxes themselves
            num_gts = assign_result.num_gts
            pos_inds = sampling_result.pos_inds[num_gts:].clone() - num_gts
            neg_inds = sampling_result.neg_inds.clone() - num_gts
            pos_prior = proposal_list[i][pos_inds, -1].clone()
            neg_prior = 1 - proposal_list[i][neg_inds, -1].clone()
            gt_weights = pos_prior.new_zeros(num_gts)
            if self.quality:
                pos_ious = assign_result.max_overlaps[sampling_result.pos_inds]
                neg_ious = 1 - assign_result.max_overlaps[sampling_result.neg_inds]
                ious.append(torch.cat([pos_ious, neg_ious], dim=0).detach())
            priors.append(torch.cat([gt_weights, pos_prior, neg_prior], dim=0).detach())
        priors = torch.cat(priors, dim=0)
        ious = torch.cat(ious, dim=0) if self.quality else None
        losses = dict()
        if self.boost:
            bbox_results = self._bbox_forward_train_boost(x, sampling_results, gt_bboxes, gt_labels,
                                                          img_metas, priors, ious)
        else:
            bbox_results = self._bbox_forward_train(x, sampling_results, gt_bboxes, gt_labels,
                                                    img_metas)
        losses.update(bbox_results['loss_bbox'])
        return losses

    # ---- device-resident train step ---------------------------------------------------------
    def device_train_ok(self):
        """the configuration the whole-batch assignment / sampling / boosting-loss kernels cover: the
        shipped ProbRoIHead recipes (boost=True, CrossEntropy + L1, MaxIoU assignment, RandomSampler)"""
        from .losses import L1Loss
        return bool(type(self) is ProbRoIHead and self.boost and self._device_common_ok() and
                    type(self.bbox_head.loss_bbox) is L1Loss)

    def _device_common_ok(self):
        """what every whole-batch variant needs: MaxIoU assignment, RandomSampler, softmax cross-entropy, encoded box
        targets, the fused RoI extractor"""
        from .core import MaxIoUAssigner, RandomSampler
        from .losses import CrossEntropyLoss
        h = self.bbox_head
        return bool(
            self.train_cfg is not None and
            type(self.bbox_assigner) is MaxIoUAssigner and self.bbox_assigner.ignore_iof_thr <= 0 and
            (not self.bbox_assigner.match_low_quality or self.bbox_assigner.gt_max_assign_all) and
            type(self.bbox_sampler) is RandomSampler and self.bbox_sampler.num <= 2048 and
            type(h.loss_cls) is CrossEntropyLoss and not h.loss_cls.use_sigmoid and h.loss_cls.class_weight is None and
            h.loss_cls.ignore_index is None and not h.focal_reg and
            not h.reg_decoded_bbox and self.train_cfg.pos_weight <= 0 and self.bbox_roi_extractor._fusable())

    _needs_overlaps = False     # (DyProbRoIHead: the assignment's max_overlaps also without `quality`)

    def sample_device(self, dets, num, gt_flat, overlap_work=None, proposal_stream=None):
        """assignment + RandomSampler + targets + priors of the whole batch (prob_roi_head.py:23-69,
        bbox_head.py:122-253) from the padded proposals `dets` (B,K,5) / `num` (B,).  The one host
        read of the train step happens here: the (B,2) candidate counts the seeded host `randperm`
        needs; `overlap_work()` is called while that copy is in flight (the caller queues the RPN
        loss kernels there, so the device has work while the host waits and draws)."""
        from . import train_ops
        gts, gt_labels, offs = gt_flat
        B, K, _ = dets.shape
        a, sp = self.bbox_assigner, self.bbox_sampler
        host = self.__dict__.setdefault('_count_host', {})
        if host.get('B') != B:
            host.update(B=B, counts=torch.empty((B, 2), dtype=torch.int32).pin_memory(),
                        perm=torch.empty((B, int(sp.num * sp.pos_fraction) + sp.num), dtype=torch.int32).pin_memory())
        # `proposal_stream`: the stream the proposals were produced on (detectors.py, early_rpn_backward): assignment
        # and the count copy stay on it, `overlap_work` is queued on the current stream meanwhile, which joins it below
        import contextlib
        with (torch.cuda.stream(proposal_stream) if proposal_stream is not None else contextlib.nullcontext()):
            res = train_ops.assign_max_iou(dets, gts, offs, a.pos_iou_thr, a.neg_iou_thr, a.min_pos_iou,
                                           a.match_low_quality, num_boxes=num,
                                           want_overlaps=self.quality or self._needs_overlaps, want_counts=True)
            gt_inds, mo, counts = (res[0], res[1], res[2]) if (self.quality or self._needs_overlaps) else (res[0], None, res[1])
            host['counts'].copy_(counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        extra = overlap_work() if overlap_work is not None else None
        if SYNC_WAIT is not None:       # BRCNN_TIME_SYNC=1: how long the host waits here = its slack over the device
            import time
            t0 = time.perf_counter()
            ev.synchronize()
            SYNC_WAIT[0] += time.perf_counter() - t0
            SYNC_WAIT[1] += 1
        else:
            ev.synchronize()
        if proposal_stream is not None:
            cur = torch.cuda.current_stream(dets.device)
            cur.wait_stream(proposal_stream)
            for t in (dets, num, gt_inds, mo):
                if t is not None:
                    t.record_stream(cur)
        num_pos = int(sp.num * sp.pos_fraction)
        cnt = []
        for b, (p_, n_) in enumerate(host['counts'].tolist()):
            g = offs[b + 1] - offs[b]
            cnt.append((p_ + (g if sp.add_gt_as_proposals else 0), n_))
            if min(cnt[-1][0], num_pos) < g:
                raise RuntimeError(f'image {b}: {g} ground-truth boxes exceed the {num_pos} positive slots of the '
                                   f'sampler (the reference prior extraction fails on this too)')
        perm, rows = train_ops.draw_sampler_perms(cnt, sp.num, num_pos, sp.neg_pos_ub)
        host['perm'].copy_(perm)
        perm_dev = host['perm'].to(dets.device, non_blocking=True)
        coder = self.bbox_head.bbox_coder
        out = train_ops.rcnn_sample(dets, num, gt_inds, mo, gts, gt_labels, offs, perm_dev, rows, sp.num, num_pos,
                                    sp.neg_pos_ub, self.bbox_head.num_classes, coder.means, coder.stds,
                                    add_gt_as_proposals=sp.add_gt_as_proposals,
                                    reg_decoded_bbox=self.bbox_head.reg_decoded_bbox, want_ious=self.quality)
        out['rows'] = rows
        # host-side knowledge of the draw: sampled positives of the batch, the proposals' overlaps (DyProbRoIHead)
        out['num_pos_host'] = sum(train_ops.sample_counts(p_, n_, sp.num, num_pos, sp.neg_pos_ub)[0] for p_, n_ in cnt)
        out['max_overlaps'], out['num_gts_host'] = mo, [offs[b + 1] - offs[b] for b in range(B)]
        return out, extra

    def forward_train_device(self, feats_nhwc, img_metas, dets, num, gt_flat, overlap_work=None, proposal_stream=None):
        """ProbRoIHead.forward_train on the padded device proposals; returns (losses, overlap_work's result).
        `feats_nhwc` may be a callable that yields the pyramid once `overlap_work` has run."""
        from . import train_ops
        smp, extra = self.sample_device(dets, num, gt_flat, overlap_work, proposal_stream)
        stage_mark('sampler')                # (incl. the wait for the proposal stream: proposals, assignment, counts)
        if callable(feats_nhwc):
            feats_nhwc = feats_nhwc()
        roi_feats = self.bbox_roi_extractor.forward_nhwc(feats_nhwc, smp['rois'])
        stage_mark('roi_align')
        cls_score, bbox_pred = self.bbox_head.forward_nhwc(roi_feats)
        stage_mark('fc_head')
        from . import autograd as _A
        _A.release_held_weight_gradients()       # (held since the early RPN backward pass, if the detector said so)
        h = self.bbox_head
        out3 = train_ops.boost_loss(cls_score, bbox_pred, smp['labels'], smp['priors'], smp['bbox_targets'],
                                    h.num_classes, self.gamma, self.alpha, smp.get('ious'), self.iou_gamma,
                                    h.loss_cls.loss_weight, h.loss_bbox.loss_weight, self.reg_norm, h.reg_class_agnostic)
        stage_mark('boost_loss')
        self.last_samples = smp
        return dict(loss_cls=out3[0], loss_bbox=out3[1], acc=out3[2].reshape(1)), extra

    def _bbox_forward(self, x, rois):
        bbox_feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)
        cls_score, bbox_pred = self.bbox_head(bbox_feats)
        return dict(cls_score=cls_score, bbox_pred=bbox_pred, bbox_feats=bbox_feats)

    def _bbox_forward_train(self, x, sampling_results, gt_bboxes, gt_labels, img_metas):
        rois = bbox2roi([res.bboxes for res in sampling_results])
        bbox_results = self._bbox_forward(x, rois)
        bbox_targets = self.bbox_head.get_targets(sampling_results, gt_bboxes, gt_labels,
                                                  self.train_cfg)
        loss_bbox = self.bbox_head.loss(bbox_results['cls_score'], bbox_results['bbox_pred'], rois,
                                        *bbox_targets)
        bbox_results.update(loss_bbox=loss_bbox)
        return bbox_results

    def boost_weights(self, cls_score, labels, priors, ious=None):
        """label_weights_new of prob_roi_head.py:116-129"""
        if ious is not None:
            p = torch.gather(cls_score.clone().softmax(1).detach(), 1, labels.reshape(-1, 1))
            w = (ious - p).abs() ** self.iou_gamma * (1 - priors) ** self.gamma
        else:
            w = (1 - priors) ** self.gamma
        if self.alpha != 0:
            w = w * self.alpha
        return w

    def _bbox_forward_train_boost(self, x, sampling_results, gt_bboxes, gt_labels, img_metas,
                                  priors, ious=None):
        rois = bbox2roi([res.bboxes for res in sampling_results])
        bbox_results = self._bbox_forward(x, rois)
        labels, label_weights, bbox, bbox_weights = self.bbox_head.get_targets(
            sampling_results, gt_bboxes, gt_labels, self.train_cfg)
        label_weights_new = self.boost_weights(bbox_results['cls_score'], labels, priors, ious)
        # NB: the un-boosted label_weights go into the head loss (prob_roi_head.py:130)
        loss_bbox = self.bbox_head.loss(bbox_results['cls_score'], bbox_results['bbox_pred'], rois,
                                        labels, label_weights, bbox, bbox_weights,
                                        reduction_override='none')
        loss_bbox['loss_cls'] = self.norm_loss(loss_bbox['loss_cls'], label_weights_new,
                                               label_weights_new.shape[0])
        if self.reg_norm == 'mean':
            loss_bbox['loss_bbox'] = loss_bbox['loss_bbox'].mean()
        else:
            loss_bbox['loss_bbox'] = loss_bbox['loss_bbox'].sum() / bbox.size(0)
        bbox_results.update(loss_bbox=loss_bbox)
        return bbox_results

    @staticmethod
    def norm_loss(loss, weights, avg_factor):
        new_weights = weights * (loss.sum() / (weights * loss).sum())
        return (loss * new_weights.detach()).sum() / avg_factor

    # ---- test --------------------------------------------------------------------------
    def fuse_scores(self, cls_score, prior):
        """sqrt(softmax(cls_score) * prior) over all C+1 columns (prob_roi_head.py:232-240)"""
        if not self.prob:
            return cls_score
        return (cls_score.softmax(1) * prior.reshape(-1, 1)) ** 0.5

    def simple_test_bboxes(self, x, img_metas, proposals, rcnn_test_cfg, rescale=False):
        prior = torch.cat([boxes[:, -1] for boxes in proposals], dim=0)
        return self._simple_test_bboxes_with_prior(x, img_metas, proposals, prior, rcnn_test_cfg, rescale)

    def _simple_test_bboxes_with_prior(self, x, img_metas, proposals, prior, rcnn_test_cfg, rescale):
        rois = bbox2roi(proposals)
        if rois.shape[0] == 0:
            batch_size = len(proposals)
            det_bbox = rois.new_zeros(0, 5)
            det_label = rois.new_zeros((0,), dtype=torch.long)
            if rcnn_test_cfg is None:
                det_bbox = det_bbox[:, :4]
                det_label = rois.new_zeros((0, self.bbox_head.fc_cls.out_features))
            return [det_bbox] * batch_size, [det_label] * batch_size
        bbox_results = self._bbox_forward(x, rois)
        img_shapes = tuple(meta['img_shape'] for meta in img_metas)
        scale_factors = tuple(meta['scale_factor'] for meta in img_metas)
        cls_score = self.fuse_scores(bbox_results['cls_score'], prior)
        bbox_pred = bbox_results['bbox_pred']
        num_per_img = tuple(len(p) for p in proposals)
        rois = rois.split(num_per_img, 0)
        cls_score = cls_score.split(num_per_img, 0)
        bbox_pred = bbox_pred.split(num_per_img, 0)
        det_bboxes, det_labels = [], []
        for i in range(len(proposals)):
            if rois[i].shape[0] == 0:
                det_bbox = rois[i].new_zeros(0, 5)
                det_label = rois[i].new_zeros((0,), dtype=torch.long)
                if rcnn_test_cfg is None:
                    det_bbox = det_bbox[:, :4]
                    det_label = rois[i].new_zeros((0, self.bbox_head.fc_cls.out_features))
            else:
                det_bbox, det_label = self.bbox_head.get_bboxes(
                    rois[i], cls_score[i], bbox_pred[i], img_shapes[i], scale_factors[i],
                    rescale=rescale, cfg=rcnn_test_cfg)
            det_bboxes.append(det_bbox)
            det_labels.append(det_label)
        return det_bboxes, det_labels

    def simple_test(self, x, proposal_list, img_metas, proposals=None, rescale=False):
        det_bboxes, det_labels = self.simple_test_bboxes(x, img_metas, proposal_list, self.test_cfg,
                                                         rescale=rescale)
        return [bbox2result(det_bboxes[i], det_labels[i], self.bbox_head.num_classes)
                for i in range(len(det_bboxes))]

    def simple_test_padded(self, feats_nhwc, dets, num, img_metas, rescale=False):
        """Device-resident second stage for the whole batch: `dets` (B,K,5) zero-padded RPN
        proposals with `num` (B,) valid rows.  Returns (det_bboxes (B,M,5), det_labels (B,M)
        long (-1 padded), num_dets (B,) int32), no host sync."""
        cfg = self.test_cfg
        nms_cfg = dict(cfg.nms)
        nms_type = nms_cfg.pop('type', 'nms')
        head = self.bbox_head
        B, K, _ = dets.shape
        C = head.num_classes
        device = dets.device
        assert nms_type in ('nms', 'soft_nms') and not nms_cfg.get('class_agnostic', False) and \
            not head.reg_class_agnostic
        split = K * C >= nms_cfg.get('split_thr', 10000)
        assert nms_type == 'nms' or split, 'soft-NMS below split_thr keeps pick order: per-image path'
        bidx = torch.arange(B, device=device, dtype=dets.dtype).view(B, 1, 1).expand(B, K, 1)
        rois = torch.cat([bidx, dets[..., :4]], -1).view(B * K, 5)
        prior = dets[..., 4].reshape(-1)
        roi_feats = self.bbox_roi_extractor.forward_nhwc(feats_nhwc, rois)
        stage_mark('roi_align')
        cls_score, bbox_pred = head.forward_nhwc(roi_feats)
        stage_mark('fc_head')
        # per-image clip border / rescale (image shapes are host metadata)
        max_shape = const_rows([m['img_shape'][:2] for m in img_metas], dets)
        coder = head.bbox_coder
        if not split and self.prob and type(self).fuse_scores is ProbRoIHead.fuse_scores and dets.is_cuda and \
                cls_score.dtype == torch.float32 and not getattr(coder, 'add_ctr_clamp', False) and \
                getattr(coder, 'clip_border', True) and hasattr(ops, 'rcnn_decode'):
            # score fusion, per-class decode, clip, rescale, threshold: one launch after the softmax
            sf = const_rows([list(m['scale_factor']) for m in img_metas], dets) if rescale else None
            bb, sc, lb, va = ops.rcnn_decode(cls_score.softmax(1), bbox_pred, dets, num, max_shape, sf, C,
                                             cfg.score_thr, coder.means, coder.stds)
            return batched_nms_images(bb, sc, lb, va, nms_cfg['iou_threshold'], cfg.max_per_img,
                                      nms_cfg.get('offset', 0))
        scores = self.fuse_scores(cls_score, prior).view(B, K, C + 1)
        bboxes = head.bbox_coder.decode(rois[:, 1:].view(B, K, 4), bbox_pred.view(B, K, 4 * C),
                                        max_shape=max_shape)
        if rescale:
            sf = const_rows([list(m['scale_factor']) for m in img_metas], dets)
            bboxes = (bboxes.view(B, K, C, 4) / sf.view(B, 1, 1, 4)).view(B, K, 4 * C)
        row_ok = torch.arange(K, device=device)[None, :] < num[:, None]
        s = scores[..., :C]
        valid = (s > cfg.score_thr) & row_ok[..., None]
        labels = torch.arange(C, device=device).view(1, 1, C).expand(B, K, C)
        if split:
            # mmcv's per-class branch (80-class COCO heads: 256 x 80 candidates): class-major
            # slots, one segmented (soft-)NMS launch over (image, class); same boxes, same order
            from .postprocess import batched_nms_images_by_level
            cm = lambda t: t.transpose(1, 2).reshape(B, C * K, *t.shape[3:]).contiguous()   # noqa: E731
            return batched_nms_images_by_level(cm(bboxes.view(B, K, C, 4)), cm(s), cm(labels), cm(valid),
                                               [K] * C, nms_cfg.get('iou_threshold', 0.3), cfg.max_per_img,
                                               nms_cfg.get('offset', 0), return_ids=True,
                                               soft=nms_cfg if nms_type == 'soft_nms' else None)
        det, lab, nd = batched_nms_images(bboxes.reshape(B, K * C, 4), s.reshape(B, K * C),
                                          labels.reshape(B, K * C), valid.reshape(B, K * C),
                                          nms_cfg['iou_threshold'], cfg.max_per_img,
                                          nms_cfg.get('offset', 0))
        return det, lab, nd


# ----------------------------------------------------------------------------- variants
@HEADS.register_module()
class BoostRoIHead(ProbRoIHead):
    """prob_roi_head.py:285-468: the variant fed by a first stage that scores every class --
    proposals are (n, 4 + P) rows [x1,y1,x2,y2,s_0..s_{P-1}].  Training priors are the (n, P+1)
    matrix [s | bg], bg = 0 for positives and max_p s_p for negatives, gathered at the sample's
    label; the boosted weights go STRAIGHT into the head loss as label weights (no norm_loss
    rescaling, default reduction).  Test-time fusion multiplies softmax column c by s_c and the
    background column by 1."""

    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels,
                      gt_bboxes_ignore=None, gt_masks=None):
        num_imgs = len(img_metas)
        if gt_bboxes_ignore is None:
            gt_bboxes_ignore = [None for _ in range(num_imgs)]
        sampling_results, priors, ious = [], [], []
        for i in range(num_imgs):
            assign_result = self.bbox_assigner.assign(proposal_list[i], gt_bboxes[i],
                                                      gt_bboxes_ignore[i], gt_labels[i])
            sampling_result = self.bbox_sampler.sample(assign_result, proposal_list[i],
                                                       gt_bboxes[i], gt_labels[i],
                                                       feats=[lvl[i][None] for lvl in x])
            sampling_results.append(sampling_result)
            num_gts = assign_result.num_gts
            pos_inds = sampling_result.pos_inds[num_gts:].clone() - num_gts
            neg_inds = sampling_result.neg_inds.clone() - num_gts
            neg_scores = proposal_list[i][neg_inds, 4:]
            prior = torch.cat((proposal_list[i][pos_inds, 4:], neg_scores), dim=0).clone()
            prior = torch.cat((prior, prior.new_zeros(prior.shape[0], 1)), dim=1)
            prior[pos_inds.shape[0]:, -1] = neg_scores.clone().max(-1)[0]
            gt_weights = prior.new_zeros(num_gts, prior.shape[1])
            if self.quality:
                pos_ious = assign_result.max_overlaps[sampling_result.pos_inds]
                neg_ious = 1 - assign_result.max_overlaps[sampling_result.neg_inds]
                ious.append(torch.cat([pos_ious, neg_ious], dim=0).detach())
            priors.append(torch.cat([gt_weights, prior], dim=0).detach())
        priors = torch.cat(priors, dim=0)
        ious = torch.cat(ious, dim=0) if self.quality else None
        losses = dict()
        if self.boost:
            bbox_results = self._bbox_forward_train_boost(x, sampling_results, gt_bboxes, gt_labels,
                                                          img_metas, priors, ious)
        else:
            bbox_results = self._bbox_forward_train(x, sampling_results, gt_bboxes, gt_labels,
                                                    img_metas)
        losses.update(bbox_results['loss_bbox'])
        return losses

    # ---- device-resident train step (whole batch) ------------------------------------------------
    def device_train_ok(self):
        """the one training case the reference runs (a single foreground class: its assigner rejects (n, 4 + P) rows
        for P > 1; `quality` broadcasts an (n, n) weight matrix -- left to the per-image chain): proposals are then
        (n, 5) rows like ProbRoIHead's, and the whole-batch kernels apply"""
        from .losses import L1Loss
        return bool(type(self) is BoostRoIHead and self._device_common_ok() and not self.quality and
                    self.bbox_head.num_classes == 1 and type(self.bbox_head.loss_bbox) in (L1Loss, SmoothL1Loss))

    def forward_train_device(self, feats_nhwc, img_metas, dets, num, gt_flat, overlap_work=None, proposal_stream=None):
        """forward_train below on the padded device proposals (P = 1): the prior gathered at the label is the score for
        a positive and ALSO for a negative (bg column = max_p s_p, :318-326) where ProbRoIHead uses 1 - score; the
        boosted weights are plain label weights of the head's loss (:438-468)"""
        from . import train_ops
        smp, extra = self.sample_device(dets, num, gt_flat, overlap_work, proposal_stream)
        stage_mark('sampler')
        if callable(feats_nhwc):
            feats_nhwc = feats_nhwc()
        roi_feats = self.bbox_roi_extractor.forward_nhwc(feats_nhwc, smp['rois'])
        stage_mark('roi_align')
        cls_score, bbox_pred = self.bbox_head.forward_nhwc(roi_feats)
        stage_mark('fc_head')
        h = self.bbox_head
        labels = smp['labels']
        priors = torch.where(labels >= h.num_classes, 1 - smp['priors'], smp['priors'])
        out3 = train_ops.boost_loss(cls_score, bbox_pred, labels, priors, smp['bbox_targets'], h.num_classes,
                                    self.gamma if self.boost else 0.0, self.alpha if self.boost else 0.0, None, 0.0,
                                    h.loss_cls.loss_weight, h.loss_bbox.loss_weight, 'bbox_num', h.reg_class_agnostic,
                                    plain_label_weights=True, smooth_l1_beta=getattr(h.loss_bbox, 'beta', 0.0))
        stage_mark('boost_loss')
        self.last_samples = smp
        return dict(loss_cls=out3[0], loss_bbox=out3[1], acc=out3[2].reshape(1)), extra

    def _bbox_forward_train_boost(self, x, sampling_results, gt_bboxes, gt_labels, img_metas,
                                  priors, ious=None):
        rois = bbox2roi([res.bboxes for res in sampling_results])
        bbox_results = self._bbox_forward(x, rois)
        labels, label_weights, bbox, bbox_weights = self.bbox_head.get_targets(
            sampling_results, gt_bboxes, gt_labels, self.train_cfg)
        priors = torch.gather(priors, 1, labels.reshape(-1, 1)).squeeze()
        label_weights_new = self.boost_weights(bbox_results['cls_score'], labels, priors, ious)
        loss_bbox = self.bbox_head.loss(bbox_results['cls_score'], bbox_results['bbox_pred'], rois,
                                        labels, label_weights_new, bbox, bbox_weights)
        bbox_results.update(loss_bbox=loss_bbox)
        return bbox_results

    def fuse_scores(self, cls_score, prior):
        """prob_roi_head.py:369-393"""
        if not self.prob:
            return cls_score
        return (cls_score.softmax(1) * prior) ** 0.5

    def simple_test_bboxes(self, x, img_metas, proposals, rcnn_test_cfg, rescale=False):
        # the parent's flow with the per-class prior matrix [s | 1] in place of the score column
        prior = torch.cat([boxes[:, 4:] for boxes in proposals], dim=0)
        prior = torch.cat((prior, prior.new_ones(prior.shape[0], 1)), dim=1)
        return self._simple_test_bboxes_with_prior(x, img_metas, proposals, prior, rcnn_test_cfg, rescale)

    device_test_path = False    # per-class priors: the per-image test path (simple_test) serves it


EPS = 1e-15


@HEADS.register_module()
class DyProbRoIHead(ProbRoIHead):
    """prob_roi_head.py:473-623: Dynamic R-CNN schedule on the boosting head.  Every iteration
    records the batch mean of the `iou_topk`-th largest proposal IoU and the `beta_topk`-th
    smallest mean |dx,dy| regression target; every `update_iter_interval` iterations the
    assigner's IoU thresholds become max(initial_iou, mean(history)) and the SmoothL1 beta
    min(initial_beta, median(history)).  The boosted weights are plain label weights here."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        assert isinstance(self.bbox_head.loss_bbox, SmoothL1Loss)
        self.iou_history = []
        self.beta_history = []

    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels,
                      gt_bboxes_ignore=None, gt_masks=None):
        num_imgs = len(img_metas)
        if gt_bboxes_ignore is None:
            gt_bboxes_ignore = [None for _ in range(num_imgs)]
        dyn = self.train_cfg.dynamic_rcnn
        sampling_results, priors, kth = [], [], []
        for i in range(num_imgs):
            assign_result = self.bbox_assigner.assign(proposal_list[i], gt_bboxes[i],
                                                      gt_bboxes_ignore[i], gt_labels[i])
            sampling_result = self.bbox_sampler.sample(assign_result, proposal_list[i],
                                                       gt_bboxes[i], gt_labels[i],
                                                       feats=[lvl[i][None] for lvl in x])
            iou_topk = min(dyn.iou_topk, len(assign_result.max_overlaps))
            kth.append(torch.topk(assign_result.max_overlaps, iou_topk)[0][-1])
            sampling_results.append(sampling_result)
            num_gts = assign_result.num_gts
            pos_inds = sampling_result.pos_inds[num_gts:].clone() - num_gts
            neg_inds = sampling_result.neg_inds.clone() - num_gts
            pos_prior = proposal_list[i][pos_inds, -1].clone()
            neg_prior = 1 - proposal_list[i][neg_inds, -1].clone()
            priors.append(torch.cat([neg_prior.new_zeros(num_gts), pos_prior, neg_prior], dim=0).detach())
        priors = torch.cat(priors, dim=0)
        # one host read for the whole batch (the reference reads one scalar per image)
        self.iou_history.append(np.mean(torch.stack(kth).tolist()))
        losses = dict()
        if self.boost:
            bbox_results = self._bbox_forward_train_boost(x, sampling_results, gt_bboxes, gt_labels,
                                                          img_metas, priors)
        else:
            bbox_results = self._bbox_forward_train(x, sampling_results, gt_bboxes, gt_labels,
                                                    img_metas)
        losses.update(bbox_results['loss_bbox'])
        if len(self.iou_history) % dyn.update_iter_interval == 0:
            self.update_hyperparameters()
        return losses

    # ---- device-resident train step (whole batch) ------------------------------------------------
    _needs_overlaps = True

    def device_train_ok(self):
        """DyProbRoIHead on the whole-batch kernels: the common conditions + SmoothL1Loss; boosted or not"""
        return bool(type(self) is DyProbRoIHead and self._device_common_ok() and not self.quality and
                    type(self.bbox_head.loss_bbox) is SmoothL1Loss)

    def forward_train_device(self, feats_nhwc, img_metas, dets, num, gt_flat, overlap_work=None, proposal_stream=None):
        """forward_train above on the padded device proposals: the same sampling / target kernels as ProbRoIHead's
        step, the Dynamic R-CNN statistics from device tensors (resolved on the host only when the schedule is due:
        every update_iter_interval iterations instead of the reference's per-image reads), the boosted weights as
        plain label weights and SmoothL1(beta) inside the boosting-loss kernels"""
        from . import train_ops
        dyn = self.train_cfg.dynamic_rcnn
        smp, extra = self.sample_device(dets, num, gt_flat, overlap_work, proposal_stream)
        stage_mark('sampler')
        # k-th largest overlap of [1.0 x ground truths (added by the sampler), proposals] per image (:515-517; the sampler's
        # add_gt_ has already extended assign_result.max_overlaps when the reference takes the top-k)
        mo = smp['max_overlaps']
        B, K = mo.shape
        valid = torch.arange(K, device=mo.device)[None, :] < num.reshape(B, 1)
        srt = torch.where(valid, mo, mo.new_full((), -1.0)).sort(dim=1, descending=True)[0]
        g_t = torch.tensor([g if self.bbox_sampler.add_gt_as_proposals else 0 for g in smp['num_gts_host']],
                           dtype=torch.int64, device=mo.device)
        k_t = torch.clamp(g_t + num.reshape(B).to(torch.int64), max=int(dyn.iou_topk))     # min(iou_topk, len(max_overlaps))
        idx = k_t - g_t - 1                                                                # < 0: the k-th is a ground truth's 1.0
        kth = torch.where(idx >= 0, srt.gather(1, idx.clamp(min=0).reshape(B, 1)).reshape(B), srt.new_ones(B))
        self.iou_history.append(kth.mean())
        if callable(feats_nhwc):
            feats_nhwc = feats_nhwc()
        roi_feats = self.bbox_roi_extractor.forward_nhwc(feats_nhwc, smp['rois'])
        stage_mark('roi_align')
        cls_score, bbox_pred = self.bbox_head.forward_nhwc(roi_feats)
        stage_mark('fc_head')
        h = self.bbox_head
        # beta statistic (:562-567): beta_topk-th smallest mean |dx|, |dy| target over the batch's positives
        labels, tg = smp['labels'], smp['bbox_targets']
        n_pos = int(smp['num_pos_host'])
        cur = torch.where(labels < h.num_classes, tg[:, :2].abs().mean(dim=1), tg.new_full((), float('inf')))
        beta_topk = min(dyn.beta_topk * B, n_pos)
        self.beta_history.append(cur.sort()[0][beta_topk - 1])
        out3 = train_ops.boost_loss(cls_score, bbox_pred, labels, smp['priors'], tg, h.num_classes,
                                    self.gamma if self.boost else 0.0, self.alpha if self.boost else 0.0, None, 0.0,
                                    h.loss_cls.loss_weight, h.loss_bbox.loss_weight, 'bbox_num', h.reg_class_agnostic,
                                    plain_label_weights=True, smooth_l1_beta=h.loss_bbox.beta)
        stage_mark('boost_loss')
        self.last_samples = smp
        if len(self.iou_history) % dyn.update_iter_interval == 0:
            self.update_hyperparameters()
        return dict(loss_cls=out3[0], loss_bbox=out3[1], acc=out3[2].reshape(1)), extra

    @staticmethod
    def _resolved(history):
        """history entries as host floats (the device path appends device scalars: one transfer here)"""
        if any(torch.is_tensor(v) for v in history):
            dev = [v for v in history if torch.is_tensor(v)]
            vals = iter(torch.stack([v.detach().float().reshape(()) for v in dev]).tolist())
            return [next(vals) if torch.is_tensor(v) else v for v in history]
        return list(history)

    def _record_beta(self, bbox, bbox_weights, num_imgs):
        pos_inds = bbox_weights[:, 0].nonzero().squeeze(1)
        cur_target = bbox[pos_inds, :2].abs().mean(dim=1)
        beta_topk = min(self.train_cfg.dynamic_rcnn.beta_topk * num_imgs, len(pos_inds))
        self.beta_history.append(torch.kthvalue(cur_target, beta_topk)[0].item())

    def _bbox_forward_train(self, x, sampling_results, gt_bboxes, gt_labels, img_metas):
        rois = bbox2roi([res.bboxes for res in sampling_results])
        bbox_results = self._bbox_forward(x, rois)
        bbox_targets = self.bbox_head.get_targets(sampling_results, gt_bboxes, gt_labels,
                                                  self.train_cfg)
        self._record_beta(bbox_targets[2], bbox_targets[3], len(img_metas))
        loss_bbox = self.bbox_head.loss(bbox_results['cls_score'], bbox_results['bbox_pred'], rois,
                                        *bbox_targets)
        bbox_results.update(loss_bbox=loss_bbox)
        return bbox_results

    def _bbox_forward_train_boost(self, x, sampling_results, gt_bboxes, gt_labels, img_metas,
                                  priors, ious=None):
        rois = bbox2roi([res.bboxes for res in sampling_results])
        bbox_results = self._bbox_forward(x, rois)
        labels, label_weights, bbox, bbox_weights = self.bbox_head.get_targets(
            sampling_results, gt_bboxes, gt_labels, self.train_cfg)
        self._record_beta(bbox, bbox_weights, len(img_metas))
        label_weights_new = self.boost_weights(bbox_results['cls_score'], labels, priors)
        loss_bbox = self.bbox_head.loss(bbox_results['cls_score'], bbox_results['bbox_pred'], rois,
                                        labels, label_weights_new, bbox, bbox_weights)
        bbox_results.update(loss_bbox=loss_bbox)
        return bbox_results

    def update_hyperparameters(self):
        dyn = self.train_cfg.dynamic_rcnn
        self.iou_history, self.beta_history = self._resolved(self.iou_history), self._resolved(self.beta_history)
        new_iou_thr = max(dyn.initial_iou, np.mean(self.iou_history))
        self.iou_history = []
        self.bbox_assigner.pos_iou_thr = new_iou_thr
        self.bbox_assigner.neg_iou_thr = new_iou_thr
        self.bbox_assigner.min_pos_iou = new_iou_thr
        if np.median(self.beta_history) < EPS:
            new_beta = self.bbox_head.loss_bbox.beta
        else:
            new_beta = min(dyn.initial_beta, np.median(self.beta_history))
        self.beta_history = []
        self.bbox_head.loss_bbox.beta = new_beta
        return new_iou_thr, new_beta
