"""RetinaRPN (`ATSSRPNHead`): shared 4-conv GN tower + objectness / box / IoU heads, anchor
target assignment + losses, and the proposal stage.

Mirrors `mmdet/models/dense_heads/atss_rpn_head.py:109-783` on top of
`anchor_head.py:37-268` and `rpn_head.py:16-34` (constructor arguments, parameter names
`rpn_convs.{i}.{conv,gn}`, `rpn_cls/rpn_reg/rpn_iou`, `scales.{i}.scale`, loss definitions,
proposal semantics).  Differences are in execution only:
  * the tower and heads run as NHWC implicit-GEMM convolutions (GroupNorm+ReLU fused kernel;
    the per-level `Scale` folded into the rpn_reg epilogue);
  * `get_bboxes` processes the whole batch at once on the device: fused
    sqrt(sigmoid*sigmoid) scoring, per-level top-k, on-the-fly anchor decode, one segmented
    NMS launch for all images, a single host sync at the end (the reference loops over
    images and levels in Python, atss_rpn_head.py:485,706).
"""
import copy

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .blocks import (ConvModule, PackedCache, Scale, bias_init_with_prob, pack_weight, to_nhwc)
from .postprocess import batched_nms_images
from .core import (anchor_inside_flags, bbox_overlaps, images_to_levels, multi_apply, unmap)
from .registry import (HEADS, build_anchor_generator, build_assigner, build_bbox_coder,
                       build_loss, build_sampler)

EPS = 1e-12


def cat_rows(feats):
    """the pyramid levels as one (rows, C) tensor, level-major: a view when the neck wrote them back to back"""
    from .autograd import cat_rows as _cr
    return _cr(list(feats))


def reduce_mean(tensor):
    """mean over ranks (mmdet/core/utils/dist_utils.py:67-73); identity when not distributed"""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor


class AnchorHead(nn.Module):
    """Target assignment shared by anchor-based heads (anchor_head.py:37-268)."""

    def __init__(self, num_classes, in_channels, feat_channels=256,
                 anchor_generator=dict(type='AnchorGenerator', scales=[8, 16, 32],
                                       ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64]),
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True,
                                 target_means=(.0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0)),
                 reg_decoded_bbox=False,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 train_cfg=None, test_cfg=None, init_cfg=None):
        super().__init__()
        self.in_channels, self.num_classes, self.feat_channels = in_channels, num_classes, feat_channels
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        self.sampling = loss_cls['type'] not in ['FocalLoss', 'GHMC', 'QualityFocalLoss']
        self.cls_out_channels = num_classes if self.use_sigmoid_cls else num_classes + 1
        if self.cls_out_channels <= 0:
            raise ValueError(f'num_classes={num_classes} is too small')
        self.reg_decoded_bbox = reg_decoded_bbox
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if self.train_cfg:
            self.assigner = build_assigner(self.train_cfg.assigner)
            if self.sampling and hasattr(self.train_cfg, 'sampler'):
                sampler_cfg = self.train_cfg.sampler
            else:
                sampler_cfg = dict(type='PseudoSampler')
            self.sampler = build_sampler(sampler_cfg, context=self)
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self._init_layers()

    def get_anchors(self, featmap_sizes, img_metas, device='cuda'):
        num_imgs = len(img_metas)
        multi_level_anchors = self.anchor_generator.grid_anchors(featmap_sizes, device)
        anchor_list = [multi_level_anchors for _ in range(num_imgs)]
        valid_flag_list = [self.anchor_generator.valid_flags(featmap_sizes, m['pad_shape'], device)
                           for m in img_metas]
        # remembered for get_targets: images whose flags are all ones by construction
        self._all_valid = [self.anchor_generator.all_valid(featmap_sizes, m['pad_shape']) for m in img_metas]
        return anchor_list, valid_flag_list

    def _get_targets_single(self, flat_anchors, valid_flags, gt_bboxes, gt_bboxes_ignore,
                            gt_labels, img_meta, all_valid=False, label_channels=1, unmap_outputs=True):
        # every anchor valid and no border test: the masked gather / scatter of the reference are
        # identities -- skip them (and the host syncs of `.any()` and boolean indexing)
        identity = bool(all_valid) and self.train_cfg.allowed_border < 0
        if identity:
            anchors = flat_anchors
            unmap_outputs = False
        else:
            inside_flags = anchor_inside_flags(flat_anchors, valid_flags, img_meta['img_shape'][:2],
                                               self.train_cfg.allowed_border)
            if not inside_flags.any():
                return (None,) * 7
            anchors = flat_anchors[inside_flags, :]
        assign_result = self.assigner.assign(anchors, gt_bboxes, gt_bboxes_ignore,
                                             None if self.sampling else gt_labels)
        sampling_result = self.sampler.sample(assign_result, anchors, gt_bboxes)
        num_valid = anchors.shape[0]
        bbox_targets = torch.zeros_like(anchors)
        bbox_weights = torch.zeros_like(anchors)
        labels = anchors.new_full((num_valid,), self.num_classes, dtype=torch.long)
        label_weights = anchors.new_zeros(num_valid, dtype=torch.float)
        pos_inds, neg_inds = sampling_result.pos_inds, sampling_result.neg_inds
        if len(pos_inds) > 0:
            if not self.reg_decoded_bbox:
                pos_bbox_targets = self.bbox_coder.encode(sampling_result.pos_bboxes,
                                                          sampling_result.pos_gt_bboxes)
            else:
                pos_bbox_targets = sampling_result.pos_gt_bboxes
            bbox_targets[pos_inds, :] = pos_bbox_targets
            bbox_weights[pos_inds, :] = 1.0
            if gt_labels is None:
                labels[pos_inds] = 0
            else:
                labels[pos_inds] = gt_labels[sampling_result.pos_assigned_gt_inds]
            label_weights[pos_inds] = 1.0 if self.train_cfg.pos_weight <= 0 \
                else self.train_cfg.pos_weight
        if len(neg_inds) > 0:
            label_weights[neg_inds] = 1.0
        if unmap_outputs:
            n = flat_anchors.size(0)
            labels = unmap(labels, n, inside_flags, fill=self.num_classes)
            label_weights = unmap(label_weights, n, inside_flags)
            bbox_targets = unmap(bbox_targets, n, inside_flags)
            bbox_weights = unmap(bbox_weights, n, inside_flags)
        return (labels, label_weights, bbox_targets, bbox_weights, pos_inds, neg_inds,
                sampling_result)


@HEADS.register_module()
class ATSSRPNHead(AnchorHead):
    def __init__(self, in_channels, num_classes=1, stacked_convs=4, conv_cfg=None, gamma=1,
                 atss=False, bridge=False, last_conv='norm', aug_reg_loss=None,
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                 loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=0.5),
                 init_cfg=None, num_convs=1, **kwargs):
        assert not atss, 'atss=True (ATSSAssigner targets) is outside the shipped boosting configs'
        assert not bridge and last_conv == 'norm', 'bridge / dcn / aspp variants are out of scope'
        self.stacked_convs, self.conv_cfg, self.norm_cfg = stacked_convs, conv_cfg, norm_cfg
        self.last_conv = last_conv
        super().__init__(num_classes, in_channels, init_cfg=init_cfg, **kwargs)
        self.loss_centerness = build_loss(loss_centerness)
        self.gamma, self.atss, self.bridge = gamma, atss, bridge
        self.with_aug_loss = aug_reg_loss is not None
        if self.with_aug_loss:
            self.aug_loss = build_loss(aug_reg_loss)
        self._head_caches = [PackedCache() for _ in range(8)]
        self._tower_caches = [PackedCache() for _ in range(stacked_convs)]
        self._fused_head_cache = PackedCache()
        self._base_anchor_cache = {}
        self.init_weights()

    def _init_layers(self):
        self.rpn_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            self.rpn_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                                             conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
        self.rpn_cls = nn.Conv2d(self.feat_channels, self.num_anchors * self.cls_out_channels, 3,
                                 padding=1)
        self.rpn_reg = nn.Conv2d(self.feat_channels, self.num_anchors * 4, 3, padding=1)
        self.rpn_iou = nn.Conv2d(self.feat_channels, self.num_anchors * 1, 3, padding=1)
        self.scales = nn.ModuleList([Scale(1.0) for _ in self.anchor_generator.strides])

    def init_weights(self):
        """Normal(0, 0.01) convs, rpn_cls bias for prior 0.01 (atss_rpn_head.py:123-131)"""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.rpn_cls.bias, bias_init_with_prob(0.01))

    # ------------------------------------------------------------------ forward
    def forward_single_nhwc(self, x, level):
        for conv in self.rpn_convs:
            x = conv.forward_nhwc(x)
        scale = self.scales[level].scale
        from .autograd import conv2d_nhwc_autograd, wants_grad
        if wants_grad(x, self.rpn_cls.weight, scale):
            # training: the three heads as one differentiable 54-channel conv
            heads = (self.rpn_cls, self.rpn_reg, self.rpn_iou)
            y = conv2d_nhwc_autograd(x, torch.cat([h.weight for h in heads], 0),
                                     torch.cat([h.bias for h in heads], 0), 1, self.rpn_cls.padding[0], out_f32=True).float()
            a, c = self.num_anchors, self.cls_out_channels
            return (y[..., :a * c], y[..., a * c:a * c + 4 * a] * scale, y[..., a * c + 4 * a:])
        cls = self._head_conv(x, self.rpn_cls, self._head_caches[0], None)
        reg = self._head_conv(x, self.rpn_reg, self._head_caches[1 + level], scale)
        iou = self._head_conv(x, self.rpn_iou, self._head_caches[7], None)
        return cls, reg, iou

    @staticmethod
    def _head_conv(x, conv, cache, scale):
        srcs = [conv.weight, conv.bias] + ([scale] if scale is not None else [])

        def builder():
            from .blocks import pack_weight
            w = pack_weight(conv.weight).to(x.dtype)       # bf16 mode: bf16 operands, fp32 result
            b = conv.bias.detach().float()
            if scale is None:
                return w, None, b.contiguous()
            s = scale.detach().float().expand(conv.out_channels).contiguous()
            return w, s, (b * s).contiguous()   # scale*(conv+b) = conv*s + b*s

        from .autograd import conv2d_nhwc_autograd, wants_grad
        if wants_grad(x, conv.weight, conv.bias, scale):
            y = conv2d_nhwc_autograd(x, conv.weight, conv.bias, 1, conv.padding[0], out_f32=True).float()
            return y * scale if scale is not None else y
        w, s, b = cache.get(srcs, builder)
        # fp32 result in either mode: the proposal stage scores / decodes in fp32
        return ops.conv2d_nhwc(x, w, s, b, None, False, 1, conv.padding[0], out_f32=True)

    def forward_nhwc(self, feats):
        """feats: list of (N,h,w,C) -> 3 lists of (N,h,w,A | 4A | A) NHWC head outputs"""
        from .autograd import wants_grad
        if feats[0].is_cuda and wants_grad(feats[0], self.rpn_cls.weight) and \
                all(isinstance(c.norm, nn.GroupNorm) and c.conv.bias is None and c.conv.groups == 1
                    for c in self.rpn_convs) and self.feat_channels <= 256 and self.feat_channels % 4 == 0:
            return self._forward_train_fused(feats)
        outs = [self.forward_single_nhwc(f, i) for i, f in enumerate(feats)]
        return tuple(map(list, zip(*outs)))

    def _forward_train_fused(self, feats):
        """Training forward with all pyramid levels in one launch per layer: the tower and the
        heads share their weights across levels (atss_rpn_head.py:296-297), so conv forward, data
        gradient and weight gradient of a layer each run once over the concatenated rows, and
        GroupNorm keeps its per-(level, image, group) statistics through the segment table."""
        from .autograd import GroupNormNHWCFunction, conv2d_nhwc_multi_autograd
        B = feats[0].shape[0]
        sizes = tuple(tuple(f.shape[1:3]) for f in feats)
        x = cat_rows(feats)        # (a view when the neck wrote the levels back to back: ops.output_into)
        for conv in self.rpn_convs:
            x = conv2d_nhwc_multi_autograd(x, conv.conv.weight, None, B, sizes, 1, conv.conv.padding[0])
            x = GroupNormNHWCFunction.apply(x, conv.norm.weight, conv.norm.bias, conv.norm.num_groups, B, sizes,
                                            conv.norm.eps, conv.with_activation)
        heads = (self.rpn_cls, self.rpn_reg, self.rpn_iou)
        y = conv2d_nhwc_multi_autograd(x, torch.cat([h.weight for h in heads], 0),
                                       torch.cat([h.bias for h in heads], 0), B, sizes, 1,
                                       self.rpn_cls.padding[0], out_f32=True).float()
        a, c = self.num_anchors, self.cls_out_channels
        cls, reg, iou, r0 = [], [], [], 0
        for lvl, (h, w) in enumerate(sizes):
            n = B * h * w
            t = y[r0:r0 + n].view(B, h, w, y.shape[1])
            cls.append(t[..., :a * c])
            reg.append(t[..., a * c:a * c + 4 * a] * self.scales[lvl].scale)
            iou.append(t[..., a * c + 4 * a:])
            r0 += n
        return cls, reg, iou

    def forward_fused(self, feats):
        """All pyramid levels in ONE launch per layer (the tower and the heads share their
        weights across levels, atss_rpn_head.py:296-297), and the three 3x3 heads as ONE
        54-channel conv (cls 9 | reg 36 | iou 9).  Returns per level a (N,h,w,54) tensor whose
        channel slices are the raw head outputs; the per-level `Scale` of rpn_reg is applied by
        the consumer (`scale(self.rpn_reg(x))` == raw * scale)."""
        B = feats[0].shape[0]
        sizes = [tuple(f.shape[1:3]) for f in feats]
        x = cat_rows(feats)        # (a view when the neck wrote the levels back to back: ops.output_into)
        for i, conv in enumerate(self.rpn_convs):
            assert isinstance(conv.norm, nn.GroupNorm) and conv.conv.bias is None
            w = self._tower_caches[i].get([conv.conv.weight],
                                          lambda c=conv: pack_weight(c.conv.weight).to(x.dtype))
            x, _ = ops.conv2d_nhwc_multi(x, w, B, sizes, None, None, None, False, 1, 1)
            x = ops.groupnorm_nhwc_multi(x, conv.norm.weight.detach(), conv.norm.bias.detach(),
                                         conv.norm.num_groups, B, sizes, conv.norm.eps,
                                         conv.with_activation)
        heads = (self.rpn_cls, self.rpn_reg, self.rpn_iou)

        def builder():
            return (torch.cat([pack_weight(h.weight) for h in heads], 0).to(x.dtype).contiguous(),
                    torch.cat([h.bias.detach().float() for h in heads], 0).contiguous())
        w, b = self._fused_head_cache.get([t for h in heads for t in (h.weight, h.bias)], builder)
        # fp32 result in either mode: the proposal stage scores / decodes in fp32
        y, _ = ops.conv2d_nhwc_multi(x, w, B, sizes, None, b, None, False, 1, 1, out_f32=True)
        outs, r0 = [], 0
        for (h, wd) in sizes:
            n = B * h * wd
            outs.append(y[r0:r0 + n].view(B, h, wd, y.shape[1]))
            r0 += n
        return outs

    def split_fused(self, fused):
        """(cls, reg_raw, iou) channel-slice views of the fused head outputs + per-level scales"""
        a, c = self.num_anchors, self.cls_out_channels
        cls = [f[..., :a * c] for f in fused]
        reg = [f[..., a * c:a * c + 4 * a] for f in fused]
        iou = [f[..., a * c + 4 * a:] for f in fused]
        return cls, reg, iou

    def forward(self, feats, bridge=False):
        """reference signature: list of (N,C,h,w) -> lists of (N,A,h,w), (N,4A,h,w), (N,A,h,w)"""
        cls, reg, iou = self.forward_nhwc([to_nhwc(f) for f in feats])
        v = lambda lst: [t.permute(0, 3, 1, 2) for t in lst]  # noqa: E731
        return v(cls), v(reg), v(iou)

    # ------------------------------------------------------------------ train
    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None,
                      proposal_cfg=None, **kwargs):
        outs = self(x)
        if gt_labels is None:
            loss_inputs = outs + (gt_bboxes, img_metas)
        else:
            loss_inputs = outs + (gt_bboxes, gt_labels, img_metas)
        losses = self.loss(*loss_inputs, gt_bboxes_ignore=gt_bboxes_ignore)
        if proposal_cfg is None:
            return losses
        return losses, self.get_bboxes(*outs, img_metas, cfg=proposal_cfg)

    def get_targets(self, anchor_list, valid_flag_list, gt_bboxes_list, img_metas,
                    gt_bboxes_ignore_list=None, gt_labels_list=None, label_channels=1,
                    unmap_outputs=True):
        num_imgs = len(img_metas)
        assert len(anchor_list) == len(valid_flag_list) == num_imgs
        concat_anchor_list = [torch.cat(a) for a in anchor_list]
        concat_valid_flag_list = [torch.cat(v) for v in valid_flag_list]
        if gt_bboxes_ignore_list is None:
            gt_bboxes_ignore_list = [None] * num_imgs
        if gt_labels_list is None:
            gt_labels_list = [None] * num_imgs
        all_valid = getattr(self, '_all_valid', None)
        if all_valid is None or len(all_valid) != num_imgs:
            all_valid = [False] * num_imgs
        (labels, label_weights, bbox_targets, bbox_weights, _, _, sampling_result) = multi_apply(
            self._get_targets_single, concat_anchor_list, concat_valid_flag_list, gt_bboxes_list,
            gt_bboxes_ignore_list, gt_labels_list, img_metas, all_valid, label_channels=1, unmap_outputs=True)
        self._all_valid = None
        if any(l is None for l in labels):
            return None
        pos_inds = [((0 <= l) & (l < self.num_classes)).nonzero().view(-1) for l in labels]
        gt_inds = [s.pos_assigned_gt_inds for s in sampling_result]
        return (labels, label_weights, bbox_targets, bbox_weights, pos_inds, gt_inds,
                concat_anchor_list, concat_valid_flag_list)

    def loss_single(self, anchors, cls_score, bbox_pred, iou_pred, labels, label_weights,
                    bbox_targets, level_pos_inds=None, num_total_samples=1.0):
        anchors = anchors.reshape(-1, 4)
        cls_score = cls_score.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels).contiguous()
        bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(-1, 4)
        iou_pred = iou_pred.permute(0, 2, 3, 1).reshape(-1)
        bbox_targets = bbox_targets.reshape(-1, 4)
        labels = labels.reshape(-1)
        label_weights = label_weights.reshape(-1)
        pos_inds = level_pos_inds if level_pos_inds is not None else \
            ((labels >= 0) & (labels < self.num_classes)).nonzero().squeeze(1)
        if len(pos_inds) > 0:
            pos_bbox_targets = bbox_targets[pos_inds]
            pos_bbox_pred = bbox_pred[pos_inds]
            pos_anchors = anchors[pos_inds]
            pos_ious = iou_pred[pos_inds]
            if self.reg_decoded_bbox:
                pos_decode_bbox_pred = self.bbox_coder.decode(pos_anchors, pos_bbox_pred)
                pos_encode_bbox_targets = self.bbox_coder.encode(pos_anchors, pos_bbox_targets)
                iou_target = bbox_overlaps(pos_decode_bbox_pred.detach(), pos_bbox_targets,
                                           is_aligned=True)
                if self.with_aug_loss:
                    w_aug = torch.ones_like(pos_bbox_pred) * (iou_target ** self.gamma)[:, None]
                    loss_bbox_aug = self.aug_loss(pos_bbox_pred, pos_encode_bbox_targets,
                                                  w_aug.clamp(min=EPS), avg_factor=1.0)
                bbox_weights = iou_target ** self.gamma
                loss_bbox = self.loss_bbox(pos_decode_bbox_pred, pos_bbox_targets,
                                           weight=bbox_weights.clamp(min=EPS), avg_factor=1.0)
                if self.with_aug_loss:
                    loss_bbox = (loss_bbox + loss_bbox_aug) * 0.5
            else:
                pos_decode_bbox_pred = self.bbox_coder.decode(pos_anchors, pos_bbox_pred)
                pos_decode_bbox_targets = self.bbox_coder.decode(pos_anchors, pos_bbox_targets)
                iou_target = bbox_overlaps(pos_decode_bbox_pred.detach(), pos_decode_bbox_targets,
                                           is_aligned=True)
                bbox_weights = torch.ones_like(pos_bbox_pred) * (iou_target ** self.gamma)[:, None]
                loss_bbox = self.loss_bbox(pos_bbox_pred, pos_bbox_targets,
                                           bbox_weights.clamp(min=EPS), avg_factor=1.0)
            avg_factor = iou_target.sum()
            loss_iou = self.loss_centerness(pos_ious, iou_target, avg_factor=num_total_samples)
        else:
            loss_bbox = bbox_pred.sum() * 0
            loss_iou = iou_pred.sum() * 0
            iou_target = bbox_targets.new_tensor(0.)
            avg_factor = iou_target.sum()
        from .losses import VarifocalLoss
        if isinstance(self.loss_cls, VarifocalLoss):      # atss_rpn_head.py:393-397
            cls_iou_targets = torch.zeros_like(cls_score)
            if len(pos_inds) > 0:
                cls_iou_targets[pos_inds] = iou_target.unsqueeze(-1)
            loss_cls = self.loss_cls(cls_score, cls_iou_targets, avg_factor=num_total_samples)
        else:
            loss_cls = self.loss_cls(cls_score, labels, label_weights, avg_factor=num_total_samples)
        return loss_cls, loss_bbox, loss_iou, avg_factor

    def loss(self, cls_scores, bbox_preds, iou_preds, gt_bboxes, img_metas, gt_bboxes_ignore=None):
        featmap_sizes = [f.size()[-2:] for f in cls_scores]
        assert len(featmap_sizes) == self.anchor_generator.num_levels
        device = cls_scores[0].device
        if device.type == 'cuda' and gt_bboxes_ignore is None and self.device_train_ok():
            # the reference signature on device tensors: repack the per-level NCHW outputs as the fused
            # (rows, 6A) layout (their Scale is already applied: unit scales) and run the HIP kernels
            sizes = tuple(tuple(int(v) for v in s_) for s_ in featmap_sizes)
            rows = [torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in (c_, r_, i_)], 1)
                    for c_, r_, i_ in zip(cls_scores, bbox_preds, iou_preds)]
            y = torch.cat(rows, 0).float().contiguous()
            return self.loss_fused(y, sizes, gt_bboxes, img_metas,
                                   scales=torch.ones(len(sizes), dtype=torch.float32, device=device))
        label_channels = self.cls_out_channels if self.use_sigmoid_cls else 1
        num_level_anchors = [int(h) * int(w) * self.num_anchors for h, w in featmap_sizes]

        def build_targets():
            # anchors and flags depend on shapes only; they are created on the stream the targets are
            # built on (a cache miss on another stream would race with the reads below)
            anchor_list, valid_flag_list = self.get_anchors(featmap_sizes, img_metas, device=device)
            targets = self.get_targets(anchor_list, valid_flag_list, gt_bboxes, img_metas,
                                       gt_bboxes_ignore_list=gt_bboxes_ignore, gt_labels_list=None,
                                       label_channels=label_channels)
            if targets is None:
                return None
            (labels_l, label_weights_l, bbox_targets_l, _, pos_inds, _, anchors_l, _) = targets
            lv_labels = images_to_levels(labels_l, num_level_anchors)
            # positives per level, in the flattened (image, anchor) order loss_single uses
            lv_pos = [((l.reshape(-1) >= 0) & (l.reshape(-1) < self.num_classes)).nonzero().squeeze(1)
                      for l in lv_labels]
            return (lv_labels,
                    images_to_levels(label_weights_l, num_level_anchors),
                    images_to_levels(anchors_l, num_level_anchors),
                    images_to_levels(bbox_targets_l, num_level_anchors),
                    lv_pos, sum(p.numel() for p in pos_inds))

        ev = getattr(self, '_inputs_ready', None)
        self._inputs_ready = None
        if ev is not None and device.type == 'cuda':
            # target assignment on the side stream opened before the backbone was queued
            # (detectors.forward_train): its host syncs do not wait for the main stream
            main = torch.cuda.current_stream()
            if getattr(self, '_side_stream', None) is None:
                self._side_stream = torch.cuda.Stream()
            side = self._side_stream
            side.wait_event(ev)
            with torch.cuda.stream(side):
                built = build_targets()
            main.wait_stream(side)
            if built is not None:
                for lst in built[:5]:
                    for t in lst:
                        t.record_stream(main)
        else:
            built = build_targets()
        if built is None:
            return None
        labels_list, label_weights_list, anchor_list, bbox_targets_list, level_pos, num_total_pos = built
        if dist.is_available() and dist.is_initialized():
            # rank mean of the positive count (atss_rpn_head.py:440-444), kept on the device
            num_total_samples = reduce_mean(
                torch.tensor(num_total_pos, dtype=torch.float, device=device)).clamp(min=1.0)
        else:
            num_total_samples = max(float(num_total_pos), 1.0)
        losses_cls, losses_bbox, losses_iou, bbox_avg_factor = multi_apply(
            self.loss_single, anchor_list, cls_scores, bbox_preds, iou_preds, labels_list,
            label_weights_list, bbox_targets_list, level_pos, num_total_samples=num_total_samples)
        bbox_avg_factor = sum(bbox_avg_factor)
        # stays a device scalar: reading it back would stall the host behind the whole forward pass
        bbox_avg_factor = reduce_mean(bbox_avg_factor).clamp(min=1).detach()
        losses_bbox = [x / bbox_avg_factor for x in losses_bbox]
        return dict(loss_rpn_cls=losses_cls, loss_rpn_bbox=losses_bbox, loss_rpn_iou=losses_iou)

    # ------------------------------------------------------------------ device-resident train step
    def _fused_loss_modes(self):
        """(cls_mode, reg_mode) of `brcnn_rpn_loss_*` for this head's loss configuration, or None when the
        kernels do not cover it.  Covered: FocalLoss / VarifocalLoss classification; decoded regression with
        IoULoss('log') [+ MSELoss aug] or raw-delta regression with CIoULoss (reg_decoded_bbox=False)."""
        from .losses import CIoULoss, FocalLoss, IoULoss, MSELoss, VarifocalLoss
        if type(self.loss_cls) is FocalLoss:
            cls_mode = 0
        elif type(self.loss_cls) is VarifocalLoss and self.loss_cls.use_sigmoid:
            cls_mode = 1 if self.loss_cls.iou_weighted else 2
        else:
            return None
        if self.loss_cls.reduction != 'mean' or self.loss_bbox.reduction != 'mean':
            return None
        if self.reg_decoded_bbox:
            if not (type(self.loss_bbox) is IoULoss and self.loss_bbox.mode == 'log'):
                return None
            if self.with_aug_loss and not (type(self.aug_loss) is MSELoss and self.aug_loss.reduction == 'mean'):
                return None
            return cls_mode, 0
        if type(self.loss_bbox) is CIoULoss and self.loss_bbox.eps == 1e-6:
            return cls_mode, 1
        return None

    def device_train_ok(self):
        """the loss / target configurations the fused target / loss kernels cover (all nine recipes):
        focal or varifocal classification, decoded IoU-log regression + MSE aug or CIoU on raw deltas,
        sigmoid-BCE IoU branch, MaxIoU assignment without ignore regions"""
        from .core import MaxIoUAssigner, PseudoSampler
        from .losses import CrossEntropyLoss
        tc = self.train_cfg
        return bool(
            tc is not None and self.use_sigmoid_cls and self.cls_out_channels == 1 and
            self._fused_loss_modes() is not None and
            type(self.loss_centerness) is CrossEntropyLoss and self.loss_centerness.use_sigmoid and
            self.loss_centerness.class_weight is None and self.loss_centerness.reduction == 'mean' and
            type(self.assigner) is MaxIoUAssigner and self.assigner.ignore_iof_thr <= 0 and
            self.assigner.gt_max_assign_all and type(self.sampler) is PseudoSampler and
            not getattr(self.bbox_coder, 'add_ctr_clamp', False))

    def _tower_fusable(self):
        return all(isinstance(c.norm, nn.GroupNorm) and c.conv.bias is None and c.conv.groups == 1
                   for c in self.rpn_convs) and self.feat_channels <= 256 and self.feat_channels % 4 == 0

    def forward_head_fused(self, feats):
        """training forward of tower + heads over all levels (one launch per layer); returns the fused
        head output y (rows, Cpad) fp32 [cls A | reg 4A (raw, before Scale) | iou A | zero padding],
        level-major rows, differentiable"""
        from .autograd import ConvNHWCFunction, GroupNormNHWCFunction, conv2d_nhwc_multi_autograd, fused_head_weights
        B = feats[0].shape[0]
        sizes = tuple(tuple(int(v) for v in f.shape[1:3]) for f in feats)
        x = cat_rows(feats)        # (a view when the neck wrote the levels back to back: ops.output_into)
        for conv in self.rpn_convs:
            x = conv2d_nhwc_multi_autograd(x, conv.conv.weight, None, B, sizes, 1, conv.conv.padding[0])
            x = GroupNormNHWCFunction.apply(x, conv.norm.weight, conv.norm.bias, conv.norm.num_groups, B, sizes,
                                            conv.norm.eps, conv.with_activation)
        heads = (self.rpn_cls, self.rpn_reg, self.rpn_iou)
        # (one differentiable node for cat + pad whose backward hands out views: the fused head's weight-gradient launch
        # leaves the main stream like the others)
        w, b = fused_head_weights(heads, 32 if x.dtype == torch.float32 else 64)
        y = ConvNHWCFunction.apply(x, w, b, B, sizes, 1, self.rpn_cls.padding[0], False, False, True)   # fp32 head output
        return y, sizes

    def _anchor_table(self, sizes, device):
        """all levels' grid anchors of ONE image as a cached (n, 4) tensor + the level geometry"""
        key = (tuple(sizes), str(device))
        cache = self.__dict__.setdefault('_anchor_tables', {})
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            anchors = torch.cat(self.anchor_generator.grid_anchors(list(sizes), device), 0).contiguous()
            starts = [0]
            for (h, w) in sizes:
                starts.append(starts[-1] + h * w * self.num_anchors)
            cache[key] = (anchors, (starts, [w for _, w in sizes], self.num_anchors))
        return cache[key]

    def assign_targets_device(self, sizes, img_metas, gts, gt_offsets, device):
        """AnchorHead._get_targets_single's assignment for the whole batch on the device: gt_inds
        (B, anchors per image) int32 (-1 ignore / outside, 0 negative, k matched to gt k-1)"""
        import numpy as np
        from . import train_ops
        from .core import const_rows
        anchors, geom = self._anchor_table(sizes, device)
        B = len(img_metas)
        valid_hw = None
        if not all(self.anchor_generator.all_valid(list(sizes), m['pad_shape']) for m in img_metas):
            rows = []
            for m in img_metas:
                h, w = m['pad_shape'][:2]
                r = []
                for (fh, fw), st in zip(sizes, self.anchor_generator.strides):
                    r += [min(int(np.ceil(h / st[1])), fh), min(int(np.ceil(w / st[0])), fw)]
                rows.append(r)
            valid_hw = const_rows(rows, anchors).to(torch.int32).view(B, len(sizes), 2).contiguous()
        border = self.train_cfg.allowed_border
        img_hw = const_rows([m['img_shape'][:2] for m in img_metas], anchors) if border >= 0 else None
        a = self.assigner
        return train_ops.assign_max_iou(anchors, gts, gt_offsets, a.pos_iou_thr, a.neg_iou_thr, a.min_pos_iou,
                                        a.match_low_quality, batch=B, geom=geom, valid_hw=valid_hw, img_hw=img_hw,
                                        allowed_border=border)

    def loss_fused(self, y, sizes, gt_bboxes, img_metas, gt_flat=None, scales=None):
        """ATSSRPNHead.loss on the fused head output: assignment + every loss term in HIP kernels, no
        host synchronisation.  Each loss comes back as a one-element list holding the sum over the
        pyramid levels (the reference returns the per-level terms and sums them in _parse_losses);
        `self.last_rpn_targets` keeps (gt_inds, per-level losses (3, L), [num_pos, sum iou_target])."""
        from . import train_ops
        device = y.device
        gts, _, offs = gt_flat if gt_flat is not None else train_ops.flatten_gts(gt_bboxes)
        gt_inds = self.assign_targets_device(sizes, img_metas, gts, offs, device)
        tc = self.train_cfg
        meta = train_ops.RPNLossMeta(
            len(img_metas), sizes, self.anchor_generator.strides,
            [self._base_anchors(l, device) for l in range(len(sizes))], self.num_anchors, offs, self.loss_cls.gamma,
            self.loss_cls.alpha, tc.pos_weight, self.gamma, self.bbox_coder.means, self.bbox_coder.stds, 16 / 1000,
            self.with_aug_loss, self.loss_cls.loss_weight, self.loss_bbox.loss_weight,
            self.aug_loss.loss_weight if self.with_aug_loss else 0.0, self.loss_centerness.loss_weight,
            *self._fused_loss_modes())
        if scales is None:
            scales = torch.stack([m.scale.reshape(()) for m in self.scales])
        losses3, per_level, totals = train_ops.rpn_loss(y, scales, gt_inds, gts, meta)
        self.last_rpn_targets = (gt_inds, per_level, totals)
        return dict(loss_rpn_cls=[losses3[0]], loss_rpn_bbox=[losses3[1]], loss_rpn_iou=[losses3[2]])

    def proposals_fused(self, y, sizes, img_metas, cfg):
        """proposal stage straight from the fused head output (no copies: channel-slice views), with
        the live Scale parameters read on the device; returns padded (dets (B,K,5), num (B,))"""
        yd = y.detach()
        B = len(img_metas)
        a, c = self.num_anchors, self.cls_out_channels
        cls, reg, iou, r0 = [], [], [], 0
        for (h, w) in sizes:
            n = B * h * w
            t = yd[r0:r0 + n].view(B, h, w, yd.shape[1])
            cls.append(t[..., :a * c])
            reg.append(t[..., a * c:a * c + 4 * a])
            iou.append(t[..., a * c + 4 * a:a * c + 5 * a])
            r0 += n
        scales = torch.stack([m.scale.detach().reshape(()) for m in self.scales]).float().contiguous()
        return self.get_bboxes_padded(cls, reg, iou, img_metas, cfg=cfg, reg_scales=scales)

    # ------------------------------------------------------------------ proposals
    def _base_anchors(self, level, device):
        key = (level, str(device))
        if key not in self._base_anchor_cache:
            self._base_anchor_cache[key] = self.anchor_generator.base_anchors[level].to(device)
        return self._base_anchor_cache[key]

    def get_bboxes_padded(self, cls_nhwc, reg_nhwc, iou_nhwc, img_metas, cfg=None, reg_scales=None):
        """Device-resident proposal stage for the whole batch (no host sync).

        Inputs are the NHWC head outputs per level: (B,h,w,A), (B,h,w,4A), (B,h,w,A) (dense or
        channel-slice views of the fused head output; then `reg_scales` carries the per-level
        Scale still to be applied to the raw deltas).
        Returns (dets (B, max_per_img, 5) zero-padded, num (B,) int32)."""
        cfg = copy.deepcopy(self.test_cfg if cfg is None else cfg)
        assert self.use_sigmoid_cls and self.cls_out_channels == 1
        nms_cfg = dict(cfg.nms)
        assert nms_cfg.pop('type', 'nms') == 'nms', 'RPN proposals use greedy NMS'
        B = cls_nhwc[0].shape[0]
        device = cls_nhwc[0].device
        shapes = {tuple(m['img_shape'][:2]) for m in img_metas}
        A = self.num_anchors
        sc_l, pr_l, va_l, id_l = [], [], [], []
        raw = [ops.rpn_score(cls_nhwc[lvl], iou_nhwc[lvl]).view(B, -1) for lvl in range(len(cls_nhwc))]
        if 0 < cfg.nms_pre <= 4096:
            # descending, ties by ascending index (the shared tie rule), all levels in one launch
            picked = ops.rpn_topk(raw, cfg.nms_pre)
        else:
            picked = []
            for score in raw:
                n = score.shape[1]
                if cfg.nms_pre > 0 and n > cfg.nms_pre:
                    ranked, rank_inds = score.sort(dim=1, descending=True, stable=True)
                    picked.append((ranked[:, :cfg.nms_pre], rank_inds[:, :cfg.nms_pre].contiguous()))
                else:
                    picked.append((score, torch.arange(n, device=device).expand(B, n).contiguous()))
        dev_scales = isinstance(reg_scales, torch.Tensor)
        if (len(shapes) == 1 or dev_scales) and cfg.min_bbox_size >= 0 and device.type == 'cuda':
            # every level decoded by one launch that also writes the validity and level-id columns of
            # the (B, T) candidate slots; device-resident scales (training) also take the per-image
            # clip border from a device table, so a batch of differently sized images stays one launch
            L = len(cls_nhwc)
            from .core import const_rows
            per_image = const_rows([m['img_shape'][:2] for m in img_metas], raw[0]) if dev_scales and len(shapes) > 1 \
                else None
            props, valid, ids = ops.rpn_decode_levels(
                [picked[l][1] for l in range(L)], [reg_nhwc[l] for l in range(L)],
                [self._base_anchors(l, device) for l in range(L)], [tuple(cls_nhwc[l].shape[1:3]) for l in range(L)],
                list(self.anchor_generator.strides), self.bbox_coder.means, self.bbox_coder.stds, next(iter(shapes)),
                cfg.min_bbox_size, pred_scales=None if reg_scales is None else (reg_scales if dev_scales else list(reg_scales)),
                max_shapes=per_image)
            sc_l = [picked[l][0] for l in range(L)]
            scores = torch.cat(sc_l, 1)
            return self._proposals_from_candidates(props, scores, ids, valid, sc_l, nms_cfg, cfg)
        for lvl in range(len(cls_nhwc)):
            h, w = cls_nhwc[lvl].shape[1:3]
            score, topk_inds = picked[lvl]
            stride = self.anchor_generator.strides[lvl]
            rs = 1.0 if reg_scales is None else reg_scales[lvl]
            if len(shapes) == 1:
                max_shape = next(iter(shapes))
                props, valid = ops.rpn_decode(topk_inds, reg_nhwc[lvl], self._base_anchors(lvl, device),
                                              (h, w), stride, self.bbox_coder.means,
                                              self.bbox_coder.stds, max_shape, cfg.min_bbox_size,
                                              pred_scale=rs)
            else:   # per-image clip border
                pl, vl = [], []
                for b in range(B):
                    p1, v1 = ops.rpn_decode(topk_inds[b:b + 1], reg_nhwc[lvl][b:b + 1],
                                            self._base_anchors(lvl, device), (h, w), stride,
                                            self.bbox_coder.means, self.bbox_coder.stds,
                                            img_metas[b]['img_shape'][:2], cfg.min_bbox_size,
                                            pred_scale=rs)
                    pl.append(p1)
                    vl.append(v1)
                props, valid = torch.cat(pl), torch.cat(vl)
            if cfg.min_bbox_size < 0:
                valid = torch.ones_like(valid)
            sc_l.append(score)
            pr_l.append(props)
            va_l.append(valid.bool())
            id_l.append(torch.full((B, score.shape[1]), lvl, dtype=torch.long, device=device))
        scores = torch.cat(sc_l, 1)            # (B, T)
        props = torch.cat(pr_l, 1)             # (B, T, 4)
        valid = torch.cat(va_l, 1)             # (B, T)  `proposals[valid_mask]` (:750-754)
        ids = torch.cat(id_l, 1)
        return self._proposals_from_candidates(props, scores, ids, valid, sc_l, nms_cfg, cfg)

    @staticmethod
    def _proposals_from_candidates(props, scores, ids, valid, sc_l, nms_cfg, cfg):
        if scores.shape[1] >= nms_cfg.get('split_thr', 10000):
            # mmcv batched_nms switches to its per-id (per-level) loop at >= split_thr candidates
            # (the training proposal cfg: 15 150 per image).  Both of its branches keep the same
            # boxes in the same order (the coordinate offsets already isolate the levels), so the
            # whole batch runs as one segmented launch over (image, level) segments.
            from .postprocess import batched_nms_images_by_level
            return batched_nms_images_by_level(props, scores, ids, valid, [s_.shape[1] for s_ in sc_l],
                                               nms_cfg['iou_threshold'], cfg.max_per_img,
                                               nms_cfg.get('offset', 0))
        dets, _, num = batched_nms_images(props, scores, ids, valid, nms_cfg['iou_threshold'],
                                          cfg.max_per_img, nms_cfg.get('offset', 0))
        return dets, num

    def get_bboxes(self, cls_scores, bbox_preds, iou_preds, img_metas, cfg=None, rescale=False,
                   with_nms=True):
        """reference signature (atss_rpn_head.py:466-503): per-level (N,A,h,w) / (N,4A,h,w) /
        (N,A,h,w) -> list of (k,5) proposals per image."""
        assert with_nms, '``with_nms`` in RPNHead should always True'
        assert len(cls_scores) == len(bbox_preds)
        f = lambda lst: [to_nhwc(t.detach()).contiguous() for t in lst]  # noqa: E731
        dets, num = self.get_bboxes_padded(f(cls_scores), f(bbox_preds), f(iou_preds), img_metas, cfg)
        num = num.tolist()   # the one host sync of the proposal stage
        return [dets[i, :num[i]] for i in range(len(img_metas))]

    def simple_test_rpn(self, x, img_metas):
        cls_scores, bbox_preds, iou_preds = self(x)
        return self.get_bboxes(cls_scores, bbox_preds, iou_preds, img_metas)
