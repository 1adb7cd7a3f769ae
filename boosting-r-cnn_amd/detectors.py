"""Detector shell: `FasterRCNN` (a `TwoStageDetector`) wiring backbone -> neck -> RetinaRPN
-> Boosting RoI head, with the reference's calling conventions.

Mirrors mmdet/models/detectors/{base.py:112-244, two_stage.py:18-182, faster_rcnn.py:24-43}:
`detector(img=..., img_metas=..., return_loss=..., **gt)`, `forward_test` (list-of-aug inputs,
sets `batch_input_shape`), `simple_test -> list[list[ndarray(k,5)]]`,
`train_step(data, optimizer) -> dict(loss, log_vars, num_samples)`, `_parse_losses` (sum of
every entry whose key contains 'loss'; log scalars averaged over ranks).

`simple_test` keeps the whole batch on the device between stages (NHWC feature maps, padded
proposals with counts) and synchronises with the host once, when results are copied out.
"""
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.nn as nn

from .blocks import to_nchw_view, to_nhwc
from .core import bbox2result
from .profiling import stage_mark
from .registry import DETECTORS, build_backbone, build_head, build_neck

# the RPN tower's weight-gradient launches wait for the end of the early RPN backward pass (autograd.held_weight_gradients)
HOLD_RPN_WGRAD = __import__('os').environ.get('BRCNN_HOLD_RPN_WGRAD', '1')       # '0' off, '1' until the RPN pass ends, '2' until the box head's forward pass is queued


class LazyLogVars(OrderedDict):
    """log_vars of `_parse_losses`: the reference reads every scalar back with `.item()` right after the
    forward pass (base.py:208), which stalls the host before backward is queued.  Same keys and float
    values, but the device->host copy is started asynchronously and only waited for on first access."""

    def __init__(self, names, vals):
        super().__init__()
        self._names = list(names)
        self._pending = None
        if vals.is_cuda:
            self._host = torch.empty(vals.shape, dtype=vals.dtype).pin_memory()
            self._host.copy_(vals, non_blocking=True)
            self._pending = torch.cuda.Event()
            self._pending.record()
        else:
            self._host = vals
        for k in self._names:
            OrderedDict.__setitem__(self, k, None)

    def _resolve(self):
        if self._host is not None:
            if self._pending is not None:
                self._pending.synchronize()
            for k, v in zip(self._names, self._host.tolist()):
                OrderedDict.__setitem__(self, k, v)
            self._host = self._pending = None

    def __getitem__(self, k):
        self._resolve()
        return OrderedDict.__getitem__(self, k)

    def get(self, k, default=None):
        self._resolve()
        return OrderedDict.get(self, k, default)

    def items(self):
        self._resolve()
        return OrderedDict.items(self)

    def values(self):
        self._resolve()
        return OrderedDict.values(self)

    def __repr__(self):
        self._resolve()
        return repr(dict(OrderedDict.items(self)))

    def __reduce__(self):
        self._resolve()
        return (OrderedDict, (list(OrderedDict.items(self)),))


class BaseDetector(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg
        self.fp16_enabled = False

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    @property
    def with_rpn(self):
        return hasattr(self, 'rpn_head') and self.rpn_head is not None

    @property
    def with_roi_head(self):
        return hasattr(self, 'roi_head') and self.roi_head is not None

    def init_weights(self):
        """mmcv BaseModule.init_weights: every submodule initialises itself at construction; what is left
        is the backbone's `init_cfg=dict(type='Pretrained', ...)` (configs/boosting_rcnn/*.py:15), loaded from
        the local model directory -- a missing file raises (see blocks.load_pretrained)"""
        bb = getattr(self, 'backbone', None)
        if bb is not None and hasattr(bb, 'init_weights'):
            bb.init_weights(pretrained=True)

    def forward_test(self, imgs, img_metas, **kwargs):
        for var, name in [(imgs, 'imgs'), (img_metas, 'img_metas')]:
            if not isinstance(var, list):
                raise TypeError(f'{name} must be a list, but got {type(var)}')
        num_augs = len(imgs)
        if num_augs != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) '
                             f'!= num of image meta ({len(img_metas)})')
        for img, img_meta in zip(imgs, img_metas):
            for img_id in range(len(img_meta)):
                img_meta[img_id]['batch_input_shape'] = tuple(img.size()[-2:])
        if num_augs == 1:
            if 'proposals' in kwargs:
                kwargs['proposals'] = kwargs['proposals'][0]
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        raise NotImplementedError('test-time augmentation is outside the hot path')

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def _parse_losses(self, losses):
        """base.py:185-210: every loss entry reduced to a scalar (`mean`, lists summed), `loss` = the sum of the entries
        whose name holds 'loss'.  The entries of the device-resident train step are one-element tensors already: they
        are stacked ONCE and the per-name sums and the total come out of one (names + 1, entries) selection-matrix
        product -- 3 small launches forward and as many backward where the reference's chain of per-entry `mean` /
        `sum` / `stack` calls costs ~35 (profiles/r06_notes.md); same values up to the order of a handful of fp32 adds."""
        names, groups, flat = [], [], []
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                vs = [value]
            elif isinstance(value, list):
                vs = list(value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
            names.append(name)
            groups.append(range(len(flat), len(flat) + len(vs)))
            for v in vs:
                v = v.reshape(()) if v.numel() == 1 else v.mean()
                flat.append(v if v.dtype == torch.float32 else v.float())
        if not flat:
            raise TypeError('no losses')
        key = (tuple(names), tuple(len(g) for g in groups), flat[0].device)
        sel = self.__dict__.get('_loss_sel')
        if sel is None or sel[0] != key:
            m = torch.zeros(len(names) + 1, len(flat))
            for r, (name, g) in enumerate(zip(names, groups)):
                for c in g:
                    m[r, c] = 1.0
                    if 'loss' in name:
                        m[len(names), c] = 1.0
            sel = self.__dict__['_loss_sel'] = (key, m.to(flat[0].device))
        named = torch.mv(sel[1], torch.stack(flat))             # (names + 1,): per-name sums, then the total loss
        loss = named[len(names)]
        names = names + ['loss']
        vals = named.detach()
        if dist.is_available() and dist.is_initialized():
            # the reference issues one all-reduce per scalar (base.py:202-207); same values,
            # one fused collective
            vals = vals.clone()
            dist.all_reduce(vals.div_(dist.get_world_size()))
        return loss, LazyLogVars(names, vals)

    def train_step(self, data, optimizer):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    def val_step(self, data, optimizer=None):
        return self.train_step(data, optimizer)


@DETECTORS.register_module()
class TwoStageDetector(BaseDetector):
    def __init__(self, backbone, neck=None, rpn_head=None, roi_head=None, train_cfg=None,
                 test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        self.backbone = build_backbone(backbone)
        if neck is not None:
            self.neck = build_neck(neck)
        if rpn_head is not None:
            rpn_train_cfg = train_cfg.rpn if train_cfg is not None else None
            rpn_head_ = rpn_head.copy()
            rpn_head_.update(train_cfg=rpn_train_cfg, test_cfg=test_cfg.rpn)
            self.rpn_head = build_head(rpn_head_)
        if roi_head is not None:
            rcnn_train_cfg = train_cfg.rcnn if train_cfg is not None else None
            roi_head = roi_head.copy()
            roi_head.update(train_cfg=rcnn_train_cfg)
            roi_head.update(test_cfg=test_cfg.rcnn)
            self.roi_head = build_head(roi_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self._rpn_scale_cache = None    # host copy of rpn_head.scales (freeze_for_inference)

    def set_compute_dtype(self, dtype):
        """'f32' (default: exact-fp32 MFMA, the parity path), 'bf16' or 'f16' (16-bit MFMA conv stack with
        fp32 accumulation; head outputs, proposal stage, losses and NMS stay fp32).  Training in 'f16' wants
        loss scaling (apis.train_detector wires the recipes' `fp16 = dict(loss_scale=512.)`)."""
        from . import blocks
        blocks.set_compute_dtype(dtype)
        return self

    def freeze_for_inference(self):
        """read the (tiny) host-side constants once so that `simple_test_device` issues no
        device->host copies: the five learnable rpn_reg scales."""
        self._rpn_scale_cache = [float(m.scale.detach().cpu()) for m in self.rpn_head.scales]
        return self

    # ---- features ----------------------------------------------------------------------
    def extract_feat_nhwc(self, img):
        # training: the neck's lateral convs run inside the backbone's stage loop (FPN.lateral_tap) so that the
        # gradient fan-in of a stage output (lateral conv + next stage) costs no aten add
        tap = None
        if self.with_neck and torch.is_grad_enabled() and hasattr(self.neck, 'lateral_tap') and \
                getattr(self.backbone, 'supports_tap', False):
            self.neck.__dict__.pop('_pre_laterals', None)
            tap = self.neck.lateral_tap
        if hasattr(self.backbone, 'forward_from_nchw'):
            x = self.backbone.forward_from_nchw(img, tap) if tap is not None else self.backbone.forward_from_nchw(img)
        else:
            x = self.backbone.forward_nhwc(to_nhwc(img), tap) if tap is not None else self.backbone.forward_nhwc(to_nhwc(img))
        stage_mark('backbone')
        if self.with_neck:
            x = self.neck.forward_nhwc(x)
            stage_mark('neck')
        return x

    def extract_feat(self, img):
        return tuple(to_nchw_view(f) for f in self.extract_feat_nhwc(img))

    # ---- train ---------------------------------------------------------------------------
    def _device_train_ok(self, img, gt_bboxes_ignore, proposals):
        return (img.is_cuda and self.with_rpn and proposals is None and gt_bboxes_ignore is None and
                getattr(self, 'device_train_path', True) and hasattr(self.rpn_head, 'device_train_ok') and
                self.rpn_head.device_train_ok() and self.rpn_head._tower_fusable() and
                hasattr(self.roi_head, 'device_train_ok') and self.roi_head.device_train_ok() and
                img.shape[0] <= 64)

    def forward_train_device(self, img, img_metas, gt_bboxes, gt_labels):
        """The train step with targets and losses on the device: RPN tower -> proposals -> second-stage
        assignment, then -- while the (B,2) sampler counts travel to the host -- the RPN assignment and
        loss kernels, then sampling, RoI head and the boosting loss.  One host synchronisation."""
        from . import train_ops
        from .graphs import trunk_features
        feats = trunk_features(self, img)           # backbone + neck: eager, or replayed from HIP graphs (graph_trunk)
        rpn = self.rpn_head
        if self._early_rpn_backward_ok(feats):
            return self._forward_train_device_early(feats, img_metas, gt_bboxes, gt_labels)
        y, sizes = rpn.forward_head_fused(list(feats))
        gt_flat = train_ops.flatten_gts(gt_bboxes, gt_labels)
        proposal_cfg = self.train_cfg.get('rpn_proposal', self.test_cfg.rpn)
        dets, num = rpn.proposals_fused(y, sizes, img_metas, proposal_cfg)
        roi_losses, rpn_losses = self.roi_head.forward_train_device(
            feats, img_metas, dets, num, gt_flat,
            overlap_work=lambda: rpn.loss_fused(y, sizes, gt_bboxes, img_metas, gt_flat=gt_flat))
        losses = dict()
        losses.update(rpn_losses)
        losses.update(roi_losses)
        return losses

    # ---- the RPN branch's backward pass inside the forward pass ------------------------------------
    # The proposal stage (top-k, decode, NMS, second-stage assignment) is a chain of latency-bound launches that
    # leaves most of the device idle for about a millisecond, and nothing else of the forward pass is independent of
    # it.  The backward pass of the RPN branch is: it needs the RPN losses only.  With `early_rpn_backward` set, the
    # proposal stage runs on a second stream while the main stream computes the RPN losses and back-propagates them
    # through the heads and the tower (parameter gradients are stored, the gradient w.r.t. the pyramid is kept and
    # added where the second stage's gradient arrives in the caller's backward()).  OPT-IN, because it changes what
    # the caller may do between forward and backward: gradients of the RPN parameters exist when forward_train
    # returns, so `optimizer.zero_grad()` has to come BEFORE the forward pass (mmcv's OptimizerHook calls it after:
    # apis.Runner and bench.py order it accordingly when they switch this on), the seed of backward() has to be
    # announced (`early_backward_scale`, the static loss scale of the fp16 recipes; 1 otherwise), and a forward
    # pass whose losses are never back-propagated still leaves gradients behind.
    early_rpn_backward = False
    early_backward_scale = 1.0
    # backbone + neck of the device-resident train step replayed from two HIP graphs (forward, backward) once an input
    # shape has come up twice: brcnn/graphs.py.  OPT-IN (the trunk's parameter gradients are assigned by the graph's
    # backward, not by AccumulateGrad nodes) and off everywhere by default: measured slower than the eager launches on
    # ROCm 7.2 unless the host is the bottleneck
    graph_trunk = False

    def _early_rpn_backward_ok(self, feats):
        return bool(self.early_rpn_backward) and torch.is_grad_enabled() and feats[0].is_cuda and \
            any(p.requires_grad for p in self.rpn_head.parameters())

    def _forward_train_device_early(self, feats, img_metas, gt_bboxes, gt_labels):
        from . import autograd as _A, train_ops
        rpn = self.rpn_head
        dev = feats[0].device
        main = torch.cuda.current_stream(dev)
        side = self.__dict__.get('_proposal_stream')
        if side is None or side.device != dev:
            side = self.__dict__['_proposal_stream'] = torch.cuda.Stream(dev)
        cut = [f.detach().requires_grad_(f.requires_grad) for f in feats]
        y, sizes = rpn.forward_head_fused(cut)
        stage_mark('rpn_tower')
        gt_flat = train_ops.flatten_gts(gt_bboxes, gt_labels)
        proposal_cfg = self.train_cfg.get('rpn_proposal', self.test_cfg.rpn)
        side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            dets, num = rpn.proposals_fused(y, sizes, img_metas, proposal_cfg)
        y.record_stream(side)
        rpn_grads = []

        def rpn_branch():
            losses = rpn.loss_fused(y, sizes, gt_bboxes, img_metas, gt_flat=gt_flat)
            terms = []
            for k, v in losses.items():
                if 'loss' in k:
                    terms += [e.reshape(()) if e.numel() == 1 else e.mean() for e in ([v] if isinstance(v, torch.Tensor) else v)]
            total = None
            if terms:       # (one stack + one sum instead of a mean and an add per entry)
                total = terms[0] if len(terms) == 1 else torch.stack([t.float() for t in terms]).sum()
            if total is not None and total.requires_grad:
                params = [p for p in rpn.parameters() if p.requires_grad]
                leaves = [c for c in cut if c.requires_grad]
                scale = float(self.early_backward_scale)
                with _A.deferred_side_stream_join(), _A.held_weight_gradients(HOLD_RPN_WGRAD != '0', HOLD_RPN_WGRAD != '2'):
                    grads = torch.autograd.grad(total * scale if scale != 1.0 else total, leaves + params,
                                                allow_unused=True)
                it = iter(grads[:len(leaves)])
                rpn_grads.extend(next(it) if c.requires_grad else None for c in cut)
                late = [(p, g) for p, g in zip(params, grads[len(leaves):]) if g is not None and p.grad is not None]
                if late:        # gradients left over from an earlier pass: accumulate as autograd would
                    _A.join_side_streams(dev)
                for p, g in zip(params, grads[len(leaves):]):
                    if g is None:
                        continue
                    if p.grad is None:
                        p.grad = g if g.dtype == p.dtype else g.to(p.dtype)
                    else:
                        p.grad = p.grad + g
            stage_mark('rpn_loss_and_backward')
            return {k: ([e.detach() for e in v] if isinstance(v, list) else v.detach()) for k, v in losses.items()}

        roi_losses, rpn_losses = self.roi_head.forward_train_device(
            lambda: _A.inject_gradients(feats, rpn_grads), img_metas, dets, num, gt_flat, overlap_work=rpn_branch,
            proposal_stream=side)
        losses = dict()
        losses.update(rpn_losses)
        losses.update(roi_losses)
        return losses

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None,
                      gt_masks=None, proposals=None, **kwargs):
        if self._device_train_ok(img, gt_bboxes_ignore, proposals):
            return self.forward_train_device(img, img_metas, gt_bboxes, gt_labels)
        if self.with_rpn and img.is_cuda:
            # the RPN targets depend on anchors and ground truth only: the head computes them on a
            # side stream that starts HERE, so the host syncs of the assignment (nonzero / unique)
            # wait for that stream instead of for the whole backbone queued on the main one
            ev = torch.cuda.Event()
            ev.record()
            self.rpn_head._inputs_ready = ev
        x = self.extract_feat(img)
        losses = dict()
        if self.with_rpn:
            proposal_cfg = self.train_cfg.get('rpn_proposal', self.test_cfg.rpn)
            rpn_losses, proposal_list = self.rpn_head.forward_train(
                x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=gt_bboxes_ignore,
                proposal_cfg=proposal_cfg, **kwargs)
            losses.update(rpn_losses)
        else:
            proposal_list = proposals
        roi_losses = self.roi_head.forward_train(x, img_metas, proposal_list, gt_bboxes, gt_labels,
                                                 gt_bboxes_ignore, gt_masks, **kwargs)
        losses.update(roi_losses)
        return losses

    # ---- test ----------------------------------------------------------------------------
    def simple_test_device(self, img, img_metas, rescale=False):
        """Whole inference pass with no host synchronisation: returns device tensors
        (det_bboxes (B,M,5), det_labels (B,M), num_dets (B,))."""
        feats = self.extract_feat_nhwc(img)
        rpn = self.rpn_head
        cls, reg, iou = rpn.split_fused(rpn.forward_fused(list(feats)))
        stage_mark('rpn_tower')
        scales = [float(s) for s in torch.stack([m.scale.detach() for m in rpn.scales]).tolist()] \
            if self._rpn_scale_cache is None else self._rpn_scale_cache
        dets, num = rpn.get_bboxes_padded(cls, reg, iou, img_metas, reg_scales=scales)
        stage_mark('rpn_postprocess')
        out = self.roi_head.simple_test_padded(feats, dets, num, img_metas, rescale=rescale)
        stage_mark('rcnn_decode_nms')
        return out

    def simple_test(self, img, img_metas, proposals=None, rescale=False):
        assert self.with_roi_head, 'Bbox head must be implemented.'
        if proposals is None and self._device_path_ok():
            det, lab, nd = self.simple_test_device(img, img_metas, rescale)
            det, lab, nd = det.cpu(), lab.cpu(), nd.tolist()   # the one host sync
            nc = self.roi_head.bbox_head.num_classes
            return [bbox2result(det[i, :nd[i]], lab[i, :nd[i]], nc) for i in range(len(img_metas))]
        x = self.extract_feat(img)
        proposal_list = self.rpn_head.simple_test_rpn(x, img_metas) if proposals is None else proposals
        return self.roi_head.simple_test(x, proposal_list, img_metas, rescale=rescale)

    def _device_path_ok(self):
        rc, rp = self.test_cfg.rcnn, self.test_cfg.rpn
        typ = rc.nms.get('type', 'nms')
        split = rp.max_per_img * self.roi_head.bbox_head.num_classes >= rc.nms.get('split_thr', 10000)
        return (typ == 'nms' or (typ == 'soft_nms' and split)) and rp.nms.get('type', 'nms') == 'nms' and \
            not rc.nms.get('class_agnostic', False) and getattr(self.roi_head, 'device_test_path', True)


@DETECTORS.register_module()
class FasterRCNN(TwoStageDetector):
    """mmdet/models/detectors/faster_rcnn.py:24-43 (the fork's DG/EMA research variants in the
    same file are outside the hot path)."""

    def __init__(self, backbone, rpn_head, roi_head, train_cfg, test_cfg, neck=None,
                 pretrained=None, init_cfg=None):
        super().__init__(backbone=backbone, neck=neck, rpn_head=rpn_head, roi_head=roi_head,
                         train_cfg=train_cfg, test_cfg=test_cfg, pretrained=pretrained,
                         init_cfg=init_cfg)
