"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE (/root/reference) in the authoring
container, on CPU, under the test-only mmcv shim (tests/golden/_mmcv_shim.py), with mmcv's
native ops served by this repo's C oracle.  The fixtures are data (inputs + expected
outputs); no reference source is stored.  Run:  python tests/golden/make_golden.py
"""
import copy
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _mmcv_shim  # noqa: E402

_mmcv_shim.install()
from tests import util  # noqa: E402
import brcnn  # noqa: E402,F401
from brcnn.config import Config  # noqa: E402

REF_CFG = '/root/reference/configs/boosting_rcnn/boosting_rcnn_r50_pafpn_1x_utdac.py'


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB')


def cfgdict(x):
    return _mmcv_shim.ConfigDict.wrap(x)


def g1_anchors():
    from mmdet.core.anchor.anchor_generator import AnchorGenerator
    ag = AnchorGenerator(strides=[8, 16, 32, 64, 128], ratios=[0.5, 1.0, 2.0], octave_base_scale=4,
                         scales_per_octave=3)
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    anchors = ag.grid_anchors(sizes, device='cpu')
    flags = ag.valid_flags(sizes, (800, 1344, 3), device='cpu')
    d = {f'base{i}': b for i, b in enumerate(ag.base_anchors)}
    for i, a in enumerate(anchors):
        d[f'head{i}'] = a[:64]
        d[f'tail{i}'] = a[-64:]
        d[f'sum{i}'] = a.double().sum(0)
        d[f'n{i}'] = a.shape[0]
        d[f'flags{i}'] = int(flags[i].sum())
    # a ragged pad_shape exercises valid_flags
    flags2 = ag.valid_flags(sizes, (790, 1300, 3), device='cpu')
    d['flags_ragged'] = np.array([int(f.sum()) for f in flags2])
    npz('g1_anchors', **d)


def g2_coder():
    from mmdet.core.bbox.coder.delta_xywh_bbox_coder import bbox2delta, delta2bbox
    g = torch.Generator().manual_seed(2)
    rois = util.rand_boxes(4096, seed=2)
    deltas = torch.randn(4096, 4, generator=g)
    deltas[:64] *= 6           # clamp edges
    deltas16 = torch.randn(4096, 16, generator=g)
    gts = util.rand_boxes(4096, seed=3)
    npz('g2_coder', rois=rois, deltas=deltas, deltas16=deltas16, gts=gts,
        dec=delta2bbox(rois, deltas, max_shape=(800, 1333, 3)),
        dec_noclip=delta2bbox(rois, deltas),
        dec16=delta2bbox(rois, deltas16, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2), (800, 1333, 3)),
        enc=bbox2delta(rois, gts, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2)))


def g3_overlaps():
    from mmdet.core.bbox.iou_calculators import bbox_overlaps
    a, b = util.rand_boxes(20, seed=4), util.clustered_boxes(4096, seed=5)
    c = util.clustered_boxes(4096, seed=6)
    npz('g3_overlaps', a=a, b=b, c=c, iou=bbox_overlaps(a, b), aligned=bbox_overlaps(b, c, is_aligned=True),
        giou=bbox_overlaps(b, c, mode='giou', is_aligned=True), iof=bbox_overlaps(a, b, mode='iof'))


def g4_assign_sample():
    from mmdet.core.bbox.assigners import MaxIoUAssigner
    from mmdet.core.bbox.samplers import RandomSampler
    boxes = util.clustered_boxes(3000, n_clusters=25, seed=7)
    gts = util.clustered_boxes(12, n_clusters=25, seed=7)
    labels = torch.arange(12) % 4
    d = dict(boxes=boxes, gts=gts, labels=labels)
    for name, kw in [('rpn', dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0, match_low_quality=True,
                                  ignore_iof_thr=-1)),
                     ('rcnn', dict(pos_iou_thr=0.6, neg_iou_thr=0.6, min_pos_iou=0.6,
                                   match_low_quality=False, ignore_iof_thr=-1))]:
        r = MaxIoUAssigner(**kw).assign(boxes, gts, None, labels)
        d[name + '_gt_inds'] = r.gt_inds
        d[name + '_max_overlaps'] = r.max_overlaps
        d[name + '_labels'] = r.labels
    props = torch.cat([boxes, torch.rand(3000, 1, generator=torch.Generator().manual_seed(8))], 1)
    r = MaxIoUAssigner(pos_iou_thr=0.6, neg_iou_thr=0.6, min_pos_iou=0.6, match_low_quality=False,
                       ignore_iof_thr=-1).assign(props, gts, None, labels)
    torch.manual_seed(1234)
    s = RandomSampler(num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True).sample(
        r, props, gts, labels)
    d.update(props=props, s_pos_inds=s.pos_inds, s_neg_inds=s.neg_inds, s_pos_is_gt=s.pos_is_gt,
             s_pos_assigned=s.pos_assigned_gt_inds, s_bboxes=s.bboxes)
    npz('g4_assign_sample', **d)


def _fake_rpn_head(cfg):
    """a reference ATSSRPNHead built from the UTDAC config"""
    from mmdet.models.dense_heads.atss_rpn_head import ATSSRPNHead
    c = copy.deepcopy(cfg.model.rpn_head.to_dict())
    c.pop('type')
    c.update(train_cfg=cfgdict(cfg.model.train_cfg.rpn.to_dict()), test_cfg=cfgdict(cfg.model.test_cfg.rpn.to_dict()))
    return ATSSRPNHead(**cfgdict(c))


def g5_rpn_get_bboxes(cfg):
    """ATSSRPNHead._get_bboxes_single at reduced map sizes (same 5-level structure), test and
    train proposal cfg, plus the pre-NMS candidates captured at the batched_nms call."""
    import mmdet.models.dense_heads.atss_rpn_head as M
    head = _fake_rpn_head(cfg)
    sizes = [(40, 64), (20, 32), (10, 16), (5, 8), (3, 4)]
    g = torch.Generator().manual_seed(5)
    B = 2
    cls = [torch.randn(B, 9, h, w, generator=g) * 2 for h, w in sizes]
    reg = [torch.randn(B, 36, h, w, generator=g) * 0.5 for h, w in sizes]
    iou = [torch.randn(B, 9, h, w, generator=g) * 2 for h, w in sizes]
    metas = [dict(img_shape=(320, 509, 3), scale_factor=np.ones(4, np.float32), pad_shape=(320, 512, 3))
             for _ in range(B)]
    d = {}
    for i in range(5):
        d[f'cls{i}'], d[f'reg{i}'], d[f'iou{i}'] = cls[i], reg[i], iou[i]
    captured = []
    orig = M.batched_nms

    def spy(boxes, scores, idxs, nms_cfg, class_agnostic=False):
        captured.append((boxes.clone(), scores.clone(), idxs.clone()))
        return orig(boxes, scores, idxs, nms_cfg, class_agnostic)
    M.batched_nms = spy
    try:
        for name, pc in [('test', cfg.model.test_cfg.rpn.to_dict()),
                         ('train', dict(cfg.model.train_cfg.rpn_proposal.to_dict(), nms_pre=1500, max_per_img=700)),
                         ('small', dict(nms_pre=300, max_per_img=64, nms=dict(type='nms', iou_threshold=0.7),
                                        min_bbox_size=0))]:
            captured.clear()
            res = head.get_bboxes(cls, reg, iou, metas, cfg=cfgdict(pc))
            d[name + '_cfg'] = json.dumps(pc)
            for b in range(B):
                d[f'{name}_props{b}'] = res[b]
                d[f'{name}_pre_boxes{b}'], d[f'{name}_pre_scores{b}'], d[f'{name}_pre_ids{b}'] = captured[b]
    finally:
        M.batched_nms = orig
    npz('g5_rpn_get_bboxes', **d)


def g6_rpn_loss(cfg):
    head = _fake_rpn_head(cfg)
    sizes = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
    g = torch.Generator().manual_seed(6)
    B = 2
    cls = [(torch.randn(B, 9, h, w, generator=g)).requires_grad_() for h, w in sizes]
    reg = [(torch.randn(B, 36, h, w, generator=g) * 0.3).requires_grad_() for h, w in sizes]
    iou = [(torch.randn(B, 9, h, w, generator=g)).requires_grad_() for h, w in sizes]
    _, metas, gts, _ = util.demo_inputs(B, 128, 192, seed=6)
    d = dict(gt0=gts[0], gt1=gts[1])
    for gamma in (0.5, 2):
        head.gamma = gamma
        losses = head.loss(cls, reg, iou, gts, metas)
        tot = sum(sum(v) for v in losses.values())
        grads = torch.autograd.grad(tot, cls + reg + iou)
        for k, v in losses.items():
            d[f'g{gamma}_{k}'] = torch.stack(v)
        for i in range(5):
            d[f'g{gamma}_dcls{i}'], d[f'g{gamma}_dreg{i}'], d[f'g{gamma}_diou{i}'] = \
                grads[i], grads[5 + i], grads[10 + i]
    for i in range(5):
        d[f'cls{i}'], d[f'reg{i}'], d[f'iou{i}'] = cls[i], reg[i], iou[i]
    npz('g6_rpn_loss', **d)


def g7_boost_loss(cfg):
    """ProbRoIHead._bbox_forward_train_boost tail + ProbConvFCBBoxHead.loss: values and the
    autograd gradient w.r.t. cls_score / bbox_pred (the only place the boosting weights show)."""
    from mmdet.models.roi_heads.bbox_heads.convfc_bbox_head import ProbConvFCBBoxHead
    from mmdet.models.roi_heads.prob_roi_head import ProbRoIHead
    hc = copy.deepcopy(cfg.model.roi_head.bbox_head.to_dict())
    hc.pop('type')
    head = ProbConvFCBBoxHead(**cfgdict(hc))
    g = torch.Generator().manual_seed(7)
    n, C = 1024, 4
    cls_score = (torch.randn(n, C + 1, generator=g) * 2).requires_grad_()
    bbox_pred = (torch.randn(n, 4 * C, generator=g) * 0.5).requires_grad_()
    labels = torch.randint(0, C + 1, (n,), generator=g)
    labels[n // 2:] = C
    label_weights = torch.ones(n)
    rois = torch.cat([torch.zeros(n, 1), util.rand_boxes(n, seed=70)], 1)
    bbox_targets = torch.randn(n, 4, generator=g)
    bbox_weights = (labels < C).float()[:, None].expand(n, 4).contiguous()
    priors = torch.rand(n, generator=g)
    priors[:8] = 0.0
    d = dict(cls_score=cls_score, bbox_pred=bbox_pred, labels=labels, label_weights=label_weights,
             rois=rois, bbox_targets=bbox_targets, bbox_weights=bbox_weights, priors=priors)
    for gamma in (0.5, 0.1):
        lw_new = (1 - priors) ** gamma
        lb = head.loss(cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_weights,
                       reduction_override='none')
        loss_cls = ProbRoIHead.norm_loss(None, lb['loss_cls'], lw_new, lw_new.shape[0])
        loss_bbox = lb['loss_bbox'].sum() / bbox_targets.size(0)
        gc, gb = torch.autograd.grad(loss_cls + loss_bbox, [cls_score, bbox_pred])
        d.update({f'g{gamma}_loss_cls': loss_cls, f'g{gamma}_loss_bbox': loss_bbox,
                  f'g{gamma}_acc': lb['acc'], f'g{gamma}_dcls': gc, f'g{gamma}_dbbox': gb})
    npz('g7_boost_loss', **d)


def g8_g9_test_head(cfg):
    from mmdet.models.roi_heads.bbox_heads.convfc_bbox_head import ProbConvFCBBoxHead
    from mmdet.models.roi_heads.roi_extractors import SingleRoIExtractor
    hc = copy.deepcopy(cfg.model.roi_head.bbox_head.to_dict())
    hc.pop('type')
    head = ProbConvFCBBoxHead(**cfgdict(hc))
    g = torch.Generator().manual_seed(8)
    n, C = 256, 4
    rois = torch.cat([torch.zeros(n, 1), util.clustered_boxes(n, n_clusters=12, seed=80)], 1)
    cls_score = torch.randn(n, C + 1, generator=g) * 2
    bbox_pred = torch.randn(n, 4 * C, generator=g) * 0.3
    prior = torch.rand(n, generator=g)
    fused = (cls_score.softmax(1) * prior.reshape(-1, 1)) ** 0.5
    sf = np.array([1.1, 1.2, 1.1, 1.2], np.float32)
    bboxes, scores = head.get_bboxes(rois, fused, bbox_pred, (800, 1333, 3), sf, rescale=True, cfg=None)
    det, lab = head.get_bboxes(rois, fused, bbox_pred, (800, 1333, 3), sf, rescale=True,
                               cfg=cfgdict(cfg.model.test_cfg.rcnn.to_dict()))
    ex = SingleRoIExtractor(roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=0),
                            out_channels=256, featmap_strides=[8, 16, 32, 64, 128])
    r10k = util.rand_rois(10000, 4, seed=9, min_size=2, max_size=1300)
    npz('g8_g9_test_head', rois=rois, cls_score=cls_score, bbox_pred=bbox_pred, prior=prior, fused=fused,
        bboxes=bboxes, scores=scores, det=det, lab=lab, r10k=r10k, lvls10k=ex.map_roi_levels(r10k, 5))


def g10_model(cfg):
    """Whole reference FasterRCNN (R50-PAFPN, UTDAC config) on a 2 x 3 x 128 x 192 batch with
    seeded synthetic weights: stage outputs + end-to-end detections, and the train losses."""
    from mmdet.models import build_detector
    m = build_detector(cfgdict(copy.deepcopy(cfg.model.to_dict())))
    m.load_state_dict(util.seeded_state_dict(m, seed=10))
    m.eval()
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10)
    d = dict(state_keys=np.array(sorted(m.state_dict().keys())))
    with torch.no_grad():
        c = m.backbone(img)
        p = m.neck(c)
        cls, reg, iou = m.rpn_head(p)
        for i, t in enumerate(c):
            d[f'c{i}_stat'] = np.array([t.double().mean(), t.double().std(), t.double().abs().max()])
            d[f'c{i}_slice'] = t[:, :8, :4, :4]
        for i, t in enumerate(p):
            d[f'p{i}'] = t[:, :16]
            d[f'p{i}_sum'] = t.double().sum((2, 3))
        for i in range(5):
            d[f'cls{i}'], d[f'reg{i}'], d[f'iou{i}'] = cls[i], reg[i], iou[i]
        props = m.rpn_head.get_bboxes(cls, reg, iou, metas)
        for b in range(2):
            d[f'props{b}'] = props[b]
        rois_all = torch.cat([torch.cat([torch.full((len(q), 1), float(b)), q[:, :4]], 1)
                              for b, q in enumerate(props)])
        feats = m.roi_head.bbox_roi_extractor(p, rois_all)
        d['roi_feats_sum'] = feats.double().sum((2, 3))
        cs, bp = m.roi_head.bbox_head(feats)
        d['cls_score'], d['bbox_pred'] = cs, bp
        res = m.simple_test(img, metas, rescale=True)
        for b in range(2):
            for c_ in range(4):
                d[f'res{b}_{c_}'] = res[b][c_]
    npz('g10_model', **d)
    # training losses of the same model (CPU reference path), seeded sampler
    m.train()
    torch.manual_seed(77)
    losses = m.forward_train(img, metas, gts, gls)
    npz('g10_train_losses', **{k: (torch.stack(v) if isinstance(v, list) else v) for k, v in losses.items()})


def g11_fpn_config():
    """boosting_rcnn_r50_fpn_1x_coco.py (FPN neck, CIoU on encoded deltas, 80 classes): train
    losses and test detections of the reference on the demo batch"""
    from mmdet.models import build_detector
    cfg = Config.fromfile('/root/reference/configs/boosting_rcnn/boosting_rcnn_r50_fpn_1x_coco.py')
    m = build_detector(cfgdict(copy.deepcopy(cfg.model.to_dict())))
    m.load_state_dict(util.seeded_state_dict(m, seed=11))
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, num_classes=80, seed=11)
    m.eval()
    d = {}
    with torch.no_grad():
        res = m.simple_test(img, metas, rescale=True)
    for b in range(2):
        d[f'det{b}'] = np.concatenate([np.concatenate([r, np.full((len(r), 1), c, np.float32)], 1)
                                       for c, r in enumerate(res[b])], 0)
    m.train()
    torch.manual_seed(78)
    losses = m.forward_train(img, metas, gts, gls)
    for k, v in losses.items():
        d['loss_' + k] = torch.stack(v) if isinstance(v, list) else v
    npz('g11_fpn_config', **d)


def kat():
    """known-answer vectors of the mmcv ops, re-derived here in float64"""
    boxes = [[6, 3, 8, 7], [3, 6, 9, 11], [3, 7, 10, 12], [1, 4, 13, 7]]
    scores = [0.6, 0.9, 0.7, 0.2]
    b, s = np.array(boxes, np.float64), np.array(scores, np.float64)

    def iou(x, y):
        w = max(0., min(x[2], y[2]) - max(x[0], y[0]))
        h = max(0., min(x[3], y[3]) - max(x[1], y[1]))
        i = w * h
        return i / ((x[2] - x[0]) * (x[3] - x[1]) + (y[2] - y[0]) * (y[3] - y[1]) - i)

    def soft(method, thr=0.3, sigma=0.5, min_score=1e-3):
        bb, ss, idx = b.copy(), s.copy(), list(range(4))
        out = []
        while len(idx):
            m = int(np.argmax(ss))
            out.append((idx[m], ss[m]))
            mb = bb[m]
            bb, ss, idx = np.delete(bb, m, 0), np.delete(ss, m), [x for j, x in enumerate(idx) if j != m]
            for j in range(len(idx)):
                o = iou(mb, bb[j])
                wgt = {'naive': 0. if o >= thr else 1., 'linear': 1 - o if o >= thr else 1.,
                       'gaussian': np.exp(-o * o / sigma)}[method]
                ss[j] *= wgt
            keep = ss >= min_score
            bb, ss, idx = bb[keep], ss[keep], [x for j, x in enumerate(idx) if keep[j]]
        return [o[0] for o in out], [float(o[1]) for o in out]
    order = np.argsort(-s)
    keep = []
    for i in order:
        if all(iou(b[i], b[j]) <= 0.3 for j in keep):
            keep.append(int(i))
    d = dict(boxes=boxes, scores=scores, nms_keep=keep)
    for m in ('naive', 'linear', 'gaussian'):
        ii, ss = soft(m)
        d[f'soft_{m}_inds'], d[f'soft_{m}_scores'] = ii, ss
    d['roialign'] = [
        dict(input=[[1, 2], [3, 4]], roi=[0, 0, 0, 1, 1], aligned=[[1.0, 1.25], [1.5, 1.75]],
             legacy=[[1.75, 2.25], [2.75, 3.25]]),
        dict(input=[[1, 2, 5, 6], [3, 4, 7, 8], [9, 10, 13, 14], [11, 12, 15, 16]], roi=[0, 0, 0, 3, 3],
             aligned=[[1.9375, 4.75], [7.5625, 10.375]], legacy=[[3.625, 6.875], [10.125, 13.375]])]
    # ---- edge vectors: exact-threshold IoU, duplicate boxes, score ties, degenerate
    # boxes.  All coordinates / areas / IoUs are exactly representable, so float64 brute force is THE answer.
    # mmcv semantics restated: NMS suppresses on `iou > thr` (strict), soft-NMS decays on `iou >= thr`;
    # 0/0 of two zero-area boxes is NaN and compares false (nothing suppressed).  Score ties: mmcv's sort is
    # unstable -- the order below is this repository's tie rule (descending score, ascending index).
    def nms_bf(bx, sc, thr):
        bx, sc = np.array(bx, np.float64), np.array(sc, np.float64)
        order = sorted(range(len(sc)), key=lambda i: (-sc[i], i))
        keep = []
        for i in order:
            ok = True
            for j in keep:
                w = max(0., min(bx[i][2], bx[j][2]) - max(bx[i][0], bx[j][0]))
                h = max(0., min(bx[i][3], bx[j][3]) - max(bx[i][1], bx[j][1]))
                inter = w * h
                un = (bx[i][2] - bx[i][0]) * (bx[i][3] - bx[i][1]) + (bx[j][2] - bx[j][0]) * (bx[j][3] - bx[j][1]) - inter
                if un != 0 and inter / un > thr:
                    ok = False
            if ok:
                keep.append(i)
        return keep
    half = [[0, 0, 3, 1], [1, 0, 4, 1], [10, 10, 13, 11]]             # IoU(0,1) = 2 / 4 = 0.5 exactly
    quarter = [[0, 0, 4, 4], [2, 2, 6, 6], [0, 8, 4, 12]]             # IoU(0,1) = 4 / 28 = 1/7; (0,2) disjoint
    edges = [
        dict(name='iou == thr is kept (strict >)', boxes=half, scores=[0.9, 0.8, 0.7], thr=0.5),
        dict(name='just below thr suppresses', boxes=half, scores=[0.9, 0.8, 0.7], thr=0.49),
        dict(name='duplicate boxes', boxes=[[5, 5, 25, 45], [5, 5, 25, 45], [5, 5, 25, 45], [100, 100, 120, 130]],
             scores=[0.3, 0.9, 0.5, 0.9], thr=0.7),
        dict(name='score ties, disjoint', boxes=[[0, 0, 10, 10], [20, 0, 30, 10], [40, 0, 50, 10], [60, 0, 70, 10]],
             scores=[0.5, 0.5, 0.5, 0.5], thr=0.3),
        dict(name='score ties, overlapping: lower index wins', boxes=[[0, 0, 10, 10], [1, 0, 11, 10], [2, 0, 12, 10]],
             scores=[0.75, 0.75, 0.75], thr=0.5),
        dict(name='zero-area boxes never suppress', boxes=[[5, 5, 5, 5], [5, 5, 5, 5], [0, 0, 10, 10]],
             scores=[0.9, 0.8, 0.7], thr=0.3),
        dict(name='one seventh', boxes=quarter, scores=[0.2, 0.6, 0.4], thr=0.125),
    ]
    for e in edges:
        e['keep'] = nms_bf(e['boxes'], e['scores'], e['thr'])
    d['nms_edges'] = edges
    # soft-NMS at the exact threshold: linear decay applies at iou >= thr -> 0.8 * (1 - 0.5)
    d['soft_edges'] = [dict(boxes=half, scores=[0.9, 0.8, 0.7], thr=0.5, method='linear', inds=[0, 2, 1],
                            scores_out=[0.9, 0.7, 0.4]),
                       dict(boxes=half, scores=[0.9, 0.8, 0.7], thr=0.5, method='naive', inds=[0, 2], scores_out=[0.9, 0.7])]
    # ---- hand-derived cv2.resize(INTER_LINEAR, uint8) vectors.  cv2 is not in this image, so
    # these are NOT outputs of OpenCV: they follow its published rule -- source coordinate (d + 0.5) * scale - 0.5,
    # clamped to the image, bilinear blend, round half up -- on cases small enough to check on paper
    # (weights 0.25 / 0.75 / 0.5 are exact in the 11-bit fixed point cv2 uses).  "parity unpinned against cv2".
    d['resize_hand'] = [
        dict(src=[[0, 255]], size_wh=[4, 1], out=[[0, 64, 191, 255]],
             derivation='x = -0.25 (clamped: 0), 0.25 -> 63.75 -> 64, 0.75 -> 191.25 -> 191, 1.25 (clamped: 255)'),
        dict(src=[[0, 100], [200, 255]], size_wh=[4, 4],
             out=[[0, 25, 75, 100], [50, 72, 117, 139], [150, 167, 200, 216], [200, 214, 241, 255]],
             derivation='separable weights (1, .75/.25, .25/.75, 1): e.g. centre (1,1) = .5625*0 + .1875*100 + .1875*200 + '
                        '.0625*255 = 72.19 -> 72; (1,2) = .1875*0 + .5625*100 + .0625*200 + .1875*255 = 116.56 -> 117'),
        dict(src=[[0, 16, 32, 48], [64, 80, 96, 112], [128, 144, 160, 176], [192, 208, 224, 240]], size_wh=[2, 2],
             out=[[40, 72], [168, 200]], derivation='scale 2: source coordinate 0.5 / 2.5 -> mean of each 2x2 block'),
    ]
    with open(os.path.join(HERE, 'kat_mmcv_ops.json'), 'w') as f:
        json.dump(d, f, indent=1)
    print('kat_mmcv_ops.json', d['nms_keep'], d['soft_linear_inds'], d['soft_linear_scores'])


from synth import synthetic_coco, synthetic_voc  # noqa: E402


def g12_pipeline(cfg):
    """The reference's Resize / RandomFlip / Normalize / Pad / DefaultFormatBundle / Collect /
    MultiScaleFlipAug on a synthetic image + boxes (the pixel arithmetic is this repo's
    restatement through the shim; scales, box maths, flip draws and metas are the reference's)."""
    from mmdet.datasets.pipelines import Compose
    rng = np.random.RandomState(5)
    d = {}
    train = [dict(type='Resize', img_scale=[(160, 96), (200, 128)], multiscale_mode='range', keep_ratio=True),
             dict(type='RandomFlip', flip_ratio=0.5, direction=['horizontal', 'vertical']),
             dict(type='Normalize', mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
             dict(type='Pad', size_divisor=32),
             dict(type='DefaultFormatBundle'),
             dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])]
    test = [dict(type='MultiScaleFlipAug', img_scale=(160, 96), flip=False,
                 transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                             dict(type='Normalize', mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375],
                                  to_rgb=True),
                             dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']),
                             dict(type='Collect', keys=['img'])])]
    img = rng.randint(0, 256, (75, 113, 3), dtype=np.uint8)
    boxes = np.array([[10, 12, 60, 50], [0, 0, 113, 75], [100, 60, 112.5, 74.2]], dtype=np.float32)
    labels = np.array([0, 3, 1], dtype=np.int64)
    d['img'], d['boxes'], d['labels'] = img, boxes, labels
    d['train_cfg'] = np.array(json.dumps(train))
    d['test_cfg'] = np.array(json.dumps(test))

    def fresh():
        return dict(img=img.copy(), img_shape=img.shape, ori_shape=img.shape, img_fields=['img'],
                    filename='x.npy', ori_filename='x.npy', gt_bboxes=boxes.copy(), gt_labels=labels.copy(),
                    bbox_fields=['gt_bboxes'])
    pipe = Compose(train)
    for s in range(6):
        np.random.seed(40 + s)
        out = pipe(fresh())
        meta = out['img_metas'].data
        t = out['img'].data
        d[f'tr{s}_img_shape'] = np.array(t.shape)
        d[f'tr{s}_img_crop'] = t[:, :24, :24]
        d[f'tr{s}_img_sum'] = t.double().sum((1, 2))
        d[f'tr{s}_img_tail'] = t[:, -40:, -40:]
        d[f'tr{s}_boxes'] = out['gt_bboxes'].data
        d[f'tr{s}_labels'] = out['gt_labels'].data
        d[f'tr{s}_meta'] = np.array(json.dumps(dict(
            img_shape=list(meta['img_shape']), pad_shape=list(meta['pad_shape']), ori_shape=list(meta['ori_shape']),
            scale_factor=[float(v) for v in meta['scale_factor']], flip=bool(meta['flip']),
            flip_direction=meta['flip_direction'])))
    out = Compose(test)(fresh())
    d['te_img_shape'] = np.array(out['img'][0].shape)
    d['te_img_crop'] = out['img'][0][:, :24, :24]
    d['te_img_sum'] = out['img'][0].double().sum((1, 2))
    m = out['img_metas'][0].data
    d['te_meta'] = np.array(json.dumps(dict(img_shape=list(m['img_shape']), pad_shape=list(m['pad_shape']),
                                            scale_factor=[float(v) for v in m['scale_factor']], flip=bool(m['flip']))))
    npz('g12_pipeline', **d)


def g13_samplers():
    """GroupSampler / DistributedGroupSampler / DistributedSampler index streams"""
    from mmdet.datasets.samplers import DistributedGroupSampler, DistributedSampler, GroupSampler

    class DS:
        def __init__(self, flag):
            self.flag = flag

        def __len__(self):
            return len(self.flag)
    flag = (np.random.RandomState(1).rand(37) > 0.35).astype(np.uint8)
    d = dict(flag=flag)
    np.random.seed(11)
    d['group_spg2'] = np.array(list(GroupSampler(DS(flag), samples_per_gpu=2)))
    np.random.seed(12)
    d['group_spg3'] = np.array(list(GroupSampler(DS(flag), samples_per_gpu=3)))
    for world in (2, 4):
        for rank in range(world):
            s = DistributedGroupSampler(DS(flag), samples_per_gpu=2, num_replicas=world, rank=rank, seed=7)
            s.set_epoch(3)
            d[f'dgroup_w{world}_r{rank}'] = np.array(list(s))
            t = DistributedSampler(DS(flag), num_replicas=world, rank=rank, shuffle=False)
            d[f'dtest_w{world}_r{rank}'] = np.array(list(t))
    npz('g13_samplers', **d)


def g14_coco_dataset():
    """the reference's CocoDataset (load_annotations / _filter_imgs / _parse_ann_info / flag /
    results2json records) on the synthetic dataset of `synthetic_coco`"""
    import tempfile
    from mmdet.datasets import CocoDataset
    with tempfile.TemporaryDirectory() as root:
        ann_file, prefix = synthetic_coco(root)
        classes = ('echinus', 'starfish', 'holothurian', 'scallop')
        d = {}
        for mode in ('train', 'test'):
            ds = CocoDataset(ann_file=ann_file, pipeline=[], classes=classes, img_prefix=prefix,
                             test_mode=(mode == 'test'))
            d[f'{mode}_img_ids'] = np.array(ds.img_ids)
            d[f'{mode}_len'] = np.array(len(ds))
            if mode == 'train':
                d['train_flag'] = ds.flag
            for i in range(len(ds)):
                a = ds.get_ann_info(i)
                d[f'{mode}_{i}_bboxes'] = a['bboxes']
                d[f'{mode}_{i}_labels'] = a['labels']
                d[f'{mode}_{i}_ignore'] = a['bboxes_ignore']
        rng = np.random.RandomState(9)
        results = [[np.concatenate([np.sort(rng.rand(k, 2) * 50, 0), 50 + rng.rand(k, 2) * 40, rng.rand(k, 1)], 1)
                    .astype(np.float32) for k in rng.randint(0, 3, 4)] for _ in range(len(ds))]
        js = ds._det2json(results)
        d['det_json'] = np.array(json.dumps(js))
        for i, r in enumerate(results):
            for c, x in enumerate(r):
                d[f'res_{i}_{c}'] = x
    npz('g14_coco_dataset', **d)


def g15_voc():
    """VOC recipe pieces through the reference: VarifocalLoss values + gradient, the ConvFC box
    head with cls FCs / GN reg convs (forward on seeded weights), RPN loss with VarifocalLoss,
    XMLDataset annotation parsing, eval_map (VOC07 11-point and area AP)."""
    import tempfile
    from mmdet.models.losses import VarifocalLoss
    from mmdet.models import build_head
    from mmdet.core.evaluation.mean_ap import eval_map
    from mmdet.datasets import VOCDataset
    d = {}
    g = torch.Generator().manual_seed(15)
    pred = torch.randn(300, 1, generator=g).requires_grad_()
    target = torch.zeros(300, 1)
    target[::7, 0] = torch.rand(43, generator=g)
    for iw in (True, False):
        loss = VarifocalLoss(use_sigmoid=True, alpha=0.75, gamma=2.0, iou_weighted=iw, loss_weight=1.5)(
            pred, target, avg_factor=37.0)
        grad, = torch.autograd.grad(loss, pred)
        d[f'vfl_{int(iw)}'], d[f'vfl_grad_{int(iw)}'] = loss.detach(), grad
    d['vfl_pred'], d['vfl_target'] = pred.detach(), target
    # box head of the VOC recipe
    head_cfg = dict(type='ProbConvFCBBoxHead', num_cls_fcs=2, num_reg_convs=4,
                    norm_cfg=dict(type='GN', num_groups=32, requires_grad=True), in_channels=256,
                    fc_out_channels=1024, roi_feat_size=7, num_classes=20,
                    bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.],
                                    target_stds=[0.1, 0.1, 0.2, 0.2]),
                    reg_class_agnostic=False, loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=2.0),
                    loss_bbox=dict(type='L1Loss', loss_weight=2.0))
    head = build_head(cfgdict(copy.deepcopy(head_cfg)))
    head.load_state_dict(util.seeded_state_dict(head, seed=15))
    head.eval()
    feats = torch.randn(12, 256, 7, 7, generator=torch.Generator().manual_seed(151))   # regenerated by the test
    with torch.no_grad():
        cs, bp = head(feats)
    d['head_cfg'] = np.array(json.dumps(head_cfg))
    d['head_keys'] = np.array(sorted(head.state_dict().keys()))
    d['head_cls'], d['head_reg'] = cs, bp
    # RPN loss with VarifocalLoss (VOC recipe's rpn_head / train_cfg.rpn)
    vcfg = Config.fromfile(os.path.join(ROOT, 'configs', 'boosting_rcnn', 'boosting_rcnn_r50_pafpn_1x_voc.py'))
    rc = vcfg.model.rpn_head.to_dict()
    rc.update(train_cfg=vcfg.model.train_cfg.rpn.to_dict(), test_cfg=vcfg.model.test_cfg.rpn.to_dict())
    rpn = build_head(cfgdict(copy.deepcopy(rc)))
    sizes = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
    cls = [torch.randn(2, 1, h, w, generator=g) for h, w in sizes]
    reg = [torch.randn(2, 4, h, w, generator=g) * 0.3 for h, w in sizes]
    iou = [torch.randn(2, 1, h, w, generator=g) for h, w in sizes]
    _, metas, gts, _ = util.demo_inputs(2, 128, 192, seed=15, num_gt=4)
    losses = rpn.loss(cls, reg, iou, gts, metas)
    for i in range(5):
        d[f'rpn_cls{i}'], d[f'rpn_reg{i}'], d[f'rpn_iou{i}'] = cls[i], reg[i], iou[i]
    for k, v in losses.items():
        d['rpn_' + k] = torch.stack([x.detach() for x in v])
    # dataset + evaluation
    with tempfile.TemporaryDirectory() as root:
        lst, prefix = synthetic_voc(root)
        ds = VOCDataset(ann_file=lst, img_prefix=prefix, pipeline=[], test_mode=True)
        d['voc_len'] = np.array(len(ds))
        rng = np.random.RandomState(3)
        results = []
        for i in range(len(ds)):
            a = ds.get_ann_info(i)
            for k in ('bboxes', 'labels', 'bboxes_ignore', 'labels_ignore'):
                d[f'voc_{i}_{k}'] = a[k]
            per = []
            for c in range(20):
                gt = a['bboxes'][a['labels'] == c]
                det = [np.concatenate([b + rng.uniform(-6, 6, 4), [rng.rand()]]) for b in gt if rng.rand() < 0.8]
                det += [np.concatenate([np.sort(rng.uniform(0, 90, 2)), np.sort(rng.uniform(0, 90, 2))])[[0, 2, 1, 3]].tolist()
                        + [rng.rand()] for _ in range(rng.randint(0, 2))]
                per.append(np.array(det, dtype=np.float32).reshape(-1, 5))
            results.append(per)
        for i, per in enumerate(results):
            for c, x in enumerate(per):
                d[f'res_{i}_{c}'] = x
        anns = [ds.get_ann_info(i) for i in range(len(ds))]
        for name, kw in (('voc07', dict(dataset='voc07', use_legacy_coordinate=True)),
                         ('area', dict(dataset=None, use_legacy_coordinate=False)),
                         ('thr75', dict(dataset='voc07', use_legacy_coordinate=True, iou_thr=0.75))):
            m, res = eval_map(results, anns, nproc=1, **kw)
            d[f'map_{name}'] = np.array(m)
            d[f'aps_{name}'] = np.array([r['ap'] for r in res])
        d['voc_eval'] = np.array(json.dumps(ds.evaluate(results, metric='mAP')))
    npz('g15_voc', **d)


def g16_resnext_autoaug():
    """the reference's ResNeXt (grouped 3x3 convs; 32x4d, depth 50: group widths 4/8/16/32) on a
    seeded input, and its AutoAugment / RandomCrop pipeline logic on a synthetic sample"""
    from mmdet.models import build_backbone
    from mmdet.datasets.pipelines import Compose
    d = {}
    bcfg = dict(type='ResNeXt', depth=50, groups=32, base_width=4, num_stages=4, out_indices=(0, 1, 2, 3),
                frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=True), style='pytorch')
    m = build_backbone(cfgdict(copy.deepcopy(bcfg)))
    m.load_state_dict(util.seeded_state_dict(m, seed=16))
    m.eval()
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(161))
    with torch.no_grad():
        outs = m(x)
    d['backbone_cfg'] = np.array(json.dumps(bcfg))
    d['keys'] = np.array(sorted(m.state_dict().keys()))
    for i, t in enumerate(outs):
        d[f'c{i}_stat'] = np.array([t.double().mean(), t.double().std(), t.double().abs().max()])
        d[f'c{i}_slice'] = t[:, :16, :4, :6]
        d[f'c{i}_sum'] = t.double().sum((2, 3))
    # AutoAugment with the x101 recipe's two policies (sizes scaled down)
    pol = [dict(type='RandomFlip', flip_ratio=0.5),
           dict(type='AutoAugment', policies=[
               [dict(type='Resize', img_scale=[(96, 200), (112, 200), (128, 200)], multiscale_mode='value',
                     keep_ratio=True)],
               [dict(type='Resize', img_scale=[(80, 420), (100, 420), (120, 420)], multiscale_mode='value',
                     keep_ratio=True),
                dict(type='RandomCrop', crop_type='absolute_range', crop_size=(64, 100), allow_negative_crop=True),
                dict(type='Resize', img_scale=[(96, 200), (112, 200), (128, 200)], multiscale_mode='value',
                     override=True, keep_ratio=True)]]),
           dict(type='Normalize', mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
           dict(type='Pad', size_divisor=1),
           dict(type='DefaultFormatBundle'),
           dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])]
    rng = np.random.RandomState(6)
    img = rng.randint(0, 256, (90, 140, 3), dtype=np.uint8)
    boxes = np.array([[10, 12, 60, 50], [0, 0, 140, 90], [100, 60, 139.5, 89.2], [70, 5, 90, 30]], dtype=np.float32)
    labels = np.array([0, 3, 1, 2], dtype=np.int64)
    d['img'], d['boxes'], d['labels'] = img, boxes, labels
    d['pipe_cfg'] = np.array(json.dumps(pol))
    pipe = Compose(pol)
    for s_ in range(8):
        np.random.seed(70 + s_)
        out = pipe(dict(img=img.copy(), img_shape=img.shape, ori_shape=img.shape, img_fields=['img'], filename='x',
                        ori_filename='x', gt_bboxes=boxes.copy(), gt_labels=labels.copy(), bbox_fields=['gt_bboxes']))
        t = out['img'].data
        d[f'aa{s_}_shape'] = np.array(t.shape)
        d[f'aa{s_}_sum'] = t.double().sum((1, 2))
        d[f'aa{s_}_boxes'] = out['gt_bboxes'].data
        d[f'aa{s_}_labels'] = out['gt_labels'].data
        m_ = out['img_metas'].data
        d[f'aa{s_}_meta'] = np.array(json.dumps(dict(img_shape=list(m_['img_shape']), pad_shape=list(m_['pad_shape']),
                                                     scale_factor=[float(v) for v in m_['scale_factor']],
                                                     flip=bool(m_['flip']))))
    npz('g16_resnext_autoaug', **d)


def g17_res2net():
    """the reference's Res2Net-50 26w x 4s (v1d: deep stem, avg_down shortcuts, hierarchical
    Bottle2neck; no DCN -- mmcv's deformable op is not importable) on a seeded input"""
    from mmdet.models import build_backbone
    d = {}
    bcfg = dict(type='Res2Net', depth=50, scales=4, base_width=26, num_stages=4, out_indices=(0, 1, 2, 3),
                frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch')
    m = build_backbone(cfgdict(copy.deepcopy(bcfg)))
    m.load_state_dict(util.seeded_state_dict(m, seed=17))
    m.eval()
    x = torch.randn(2, 3, 72, 100, generator=torch.Generator().manual_seed(171))     # odd sizes: ceil_mode pooling
    with torch.no_grad():
        outs = m(x)
    d['backbone_cfg'] = np.array(json.dumps(bcfg))
    d['keys'] = np.array(sorted(m.state_dict().keys()))
    for i, t in enumerate(outs):
        d[f'c{i}_shape'] = np.array(t.shape)
        d[f'c{i}_slice'] = t[:, :16, :4, :6]
        d[f'c{i}_sum'] = t.double().sum((2, 3))
    npz('g17_res2net', **d)


def _variant_head_cfg(cfg, typ, num_classes, **over):
    rc = copy.deepcopy(cfg.model.roi_head.to_dict())
    rc['type'] = typ
    rc['bbox_head']['num_classes'] = num_classes
    rc['train_cfg'] = copy.deepcopy(cfg.model.train_cfg.rcnn.to_dict())
    rc['test_cfg'] = copy.deepcopy(cfg.model.test_cfg.rcnn.to_dict())
    rc.update(over)
    return rc


def g18_boost_variants(cfg):
    """BoostRoIHead / DyProbRoIHead (prob_roi_head.py:285-623) run through the reference on CPU:
    train losses (seeded sampler), the multi-column test-time fusion, and the Dynamic R-CNN
    threshold / beta schedule over 4 iterations."""
    from mmdet.models import build_head
    d = {}
    # --- BoostRoIHead, single foreground class (the only case the reference's assigner accepts)
    for tag, over in (('q', dict(boost=True, quality=True, iou_gamma=0.5, gamma=0.5)),
                      ('p', dict(boost=True, quality=False, gamma=0.5, alpha=0.75))):
        rc = _variant_head_cfg(cfg, 'BoostRoIHead', 1, **over)
        head = build_head(cfgdict(rc))
        head.load_state_dict(util.seeded_state_dict(head, seed=18))
        head.train()
        feats, metas, gts, gls, props = util.variant_inputs(1, 1, 18)
        torch.manual_seed(5)
        losses = head.forward_train(feats, metas, props, gts, gls)
        for k, v in losses.items():
            d[f'boost_{tag}_{k}'] = v
    # --- BoostRoIHead test path, 3 classes, proposals carry one score per class
    rc = _variant_head_cfg(cfg, 'BoostRoIHead', 3, boost=True)
    head = build_head(cfgdict(rc))
    head.load_state_dict(util.seeded_state_dict(head, seed=19))
    head.eval()
    feats, metas, gts, gls, props = util.variant_inputs(3, 3, 19)
    with torch.no_grad():
        det, lab = head.simple_test_bboxes(feats, metas, props, head.test_cfg, rescale=True)
    for b in range(2):
        d[f'boost_det{b}'], d[f'boost_lab{b}'] = det[b], lab[b]
    # --- DyProbRoIHead: 4 iterations, update every 2
    rc = _variant_head_cfg(cfg, 'DyProbRoIHead', 4, boost=True, gamma=0.5)
    rc['bbox_head']['loss_bbox'] = dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)
    rc['train_cfg']['dynamic_rcnn'] = dict(iou_topk=40, beta_topk=6, update_iter_interval=2,
                                           initial_iou=0.4, initial_beta=1.0)
    head = build_head(cfgdict(rc))
    head.load_state_dict(util.seeded_state_dict(head, seed=20))
    head.train()
    sched = []
    for it in range(4):
        feats, metas, gts, gls, props = util.variant_inputs(4, 1, 30 + it)
        torch.manual_seed(50 + it)
        losses = head.forward_train(feats, metas, props, gts, gls)
        for k, v in losses.items():
            d[f'dy{it}_{k}'] = v
        sched.append([head.bbox_assigner.pos_iou_thr, head.bbox_assigner.neg_iou_thr,
                      head.bbox_assigner.min_pos_iou, head.bbox_head.loss_bbox.beta,
                      len(head.iou_history), len(head.beta_history)])
    d['dy_sched'] = np.array(sched, np.float64)
    npz('g18_boost_variants', **d)


def g19_coco_pafpn_train():
    """BASELINE configs[2]: boosting_rcnn_r50_pafpn_mstrain_2x_coco.py (the COCO-PAFPN recipe: 80 classes,
    RPN gamma 2, IoU / MSE loss weights 2) -- train losses, the total loss' gradient norm and a few
    parameter gradients of the reference (fp32, CPU) on the demo batch with seeded weights / sampler"""
    from mmdet.models import build_detector
    cfg = Config.fromfile('/root/reference/configs/boosting_rcnn/boosting_rcnn_r50_pafpn_mstrain_2x_coco.py')
    m = build_detector(cfgdict(copy.deepcopy(cfg.model.to_dict())))
    m.load_state_dict(util.seeded_state_dict(m, seed=19))
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, num_classes=80, seed=19)
    m.train()
    torch.manual_seed(79)
    losses = m.forward_train(img, metas, gts, gls)
    d = {}
    for k, v in losses.items():
        d['loss_' + k] = torch.stack(v) if isinstance(v, list) else v
    loss, _ = m._parse_losses(losses)
    loss.backward()
    sq = 0.0
    for k, p in m.named_parameters():
        if p.grad is not None:
            sq += float(p.grad.double().pow(2).sum())
    d['grad_norm'] = np.float64(sq ** 0.5)
    for k in ('roi_head.bbox_head.fc_cls.weight', 'roi_head.bbox_head.fc_reg.bias', 'rpn_head.rpn_cls.weight',
              'rpn_head.rpn_reg.bias', 'rpn_head.scales.0.scale', 'rpn_head.scales.2.scale',
              'neck.lateral_convs.2.conv.bias', 'backbone.layer4.2.bn3.weight', 'backbone.layer2.0.bn1.bias'):
        d['grad_' + k] = dict(m.named_parameters())[k].grad
    d['grad_backbone.layer4.2.conv3.weight_slice'] = dict(m.named_parameters())['backbone.layer4.2.conv3.weight'].grad[:16, :16]
    d['grad_rpn_head.rpn_convs.3.conv.weight_slice'] = dict(m.named_parameters())['rpn_head.rpn_convs.3.conv.weight'].grad[:8, :8]
    npz('g19_coco_pafpn_train', **d)


def g23_coco_second_stage_inputs():
    """what the reference's SECOND stage saw in the g19 run (same model, weights, inputs, seed): the proposals the RPN
    handed to `ProbRoIHead.forward_train` per image, and what the sampler drew from them (the sampled boxes and their
    labels, positives first).  Lets a 16-bit run of the second stage be compared on the SAME RoIs -- the
    proposals a 16-bit trunk produces differ in which boxes survive top-k / NMS, which moved the g19 comparison of the
    second-stage losses to 10 %."""
    from mmdet.models import build_detector
    cfg = Config.fromfile('/root/reference/configs/boosting_rcnn/boosting_rcnn_r50_pafpn_mstrain_2x_coco.py')
    m = build_detector(cfgdict(copy.deepcopy(cfg.model.to_dict())))
    m.load_state_dict(util.seeded_state_dict(m, seed=19))
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, num_classes=80, seed=19)
    m.train()
    d = {}
    head = m.roi_head
    orig_ft = head.forward_train
    orig_sample = head.bbox_sampler.sample
    samples = []

    def ft(x, img_metas, proposal_list, *a, **k):
        for i, p_ in enumerate(proposal_list):
            d[f'props{i}'] = p_.detach().clone()
        return orig_ft(x, img_metas, proposal_list, *a, **k)

    def sample(assign_result, bboxes, gt_bboxes, gt_labels=None, **k):
        r = orig_sample(assign_result, bboxes, gt_bboxes, gt_labels, **k)
        samples.append(r)
        return r
    head.forward_train = ft
    head.bbox_sampler.sample = sample
    torch.manual_seed(79)
    losses = m.forward_train(img, metas, gts, gls)
    for i, r in enumerate(samples):
        d[f'sampled_bboxes{i}'] = r.bboxes
        d[f'sampled_pos{i}'] = np.int64(r.pos_bboxes.shape[0])
        d[f'sampled_pos_gt_labels{i}'] = r.pos_gt_labels
    for k, v in losses.items():
        d['loss_' + k] = torch.stack(v) if isinstance(v, list) else v
    npz('g23_coco_second_stage_inputs', **d)


def g20_fullsize(cfg):
    """BASELINE configs[1] at FULL size (batch 8, 201 600 anchors / image): the reference's proposal stage
    (top-k anchor indices per level, batched-NMS keep indices, proposals) from seeded head outputs, and its
    second-stage multiclass NMS (keep indices into the (roi, class) candidates, detections) from seeded
    box-head outputs on those proposals"""
    import mmdet.models.dense_heads.atss_rpn_head as M
    from mmdet.core.post_processing import bbox_nms as BN
    head = _fake_rpn_head(cfg)
    B = 8
    sizes, cls, reg, iou = util.fullsize_head_outputs(20, B)
    metas = [dict(img_shape=(800, 1333, 3), scale_factor=np.ones(4, np.float32), pad_shape=(800, 1344, 3))
             for _ in range(B)]
    captured, kept = [], []
    orig = M.batched_nms

    def spy(boxes, scores, idxs, nms_cfg, class_agnostic=False):
        captured.append((boxes.clone(), scores.clone(), idxs.clone()))
        dets, keep = orig(boxes, scores, idxs, nms_cfg, class_agnostic)
        kept.append(keep.clone())
        return dets, keep
    M.batched_nms = spy
    try:
        props = head.get_bboxes(cls, reg, iou, metas, cfg=cfgdict(cfg.model.test_cfg.rpn.to_dict()))
    finally:
        M.batched_nms = orig
    d = {}
    for b in range(B):
        d[f'props{b}'] = props[b]
        d[f'keep{b}'] = kept[b].to(torch.int32)
        # the anchors the reference picked, per level, in its own order (scores.sort(descending=True)[:nms_pre]);
        # checked against the scores it handed to batched_nms
        inds, o = [], 0
        for l in range(5):
            s_ = (cls[l][b].permute(1, 2, 0).reshape(-1).sigmoid() * iou[l][b].permute(1, 2, 0).reshape(-1).sigmoid()).sqrt()
            k = min(1000, s_.numel())
            rk = s_.sort(descending=True)[1][:k] if s_.numel() > 1000 else torch.arange(s_.numel())
            inds.append(rk)
            o += k
        inds_cat = torch.cat(inds)
        # candidates with w<=0 or h<=0 after the border clip were dropped before NMS (atss_rpn_head.py:747-754):
        # the scores handed to batched_nms are the order-preserving subsequence of the picked anchors' scores
        sc_all = torch.cat([(cls[l][b].permute(1, 2, 0).reshape(-1).sigmoid() *
                             iou[l][b].permute(1, 2, 0).reshape(-1).sigmoid()).sqrt()[inds[l]] for l in range(5)])
        cap = captured[b][1]
        mask, j = torch.zeros(sc_all.numel(), dtype=torch.bool), 0
        for i in range(sc_all.numel()):
            if j < cap.numel() and sc_all[i] == cap[j]:
                mask[i] = True
                j += 1
        assert j == cap.numel(), (j, cap.numel())
        d[f'topk{b}'] = inds_cat.to(torch.int32)
        d[f'valid{b}'] = mask
    # second stage on the reference's proposals: seeded box-head outputs -> score fusion, decode, multiclass NMS
    from mmdet.models.roi_heads.prob_roi_head import ProbRoIHead  # noqa: F401
    from mmdet.models import build_head
    rc = copy.deepcopy(cfg.model.roi_head.to_dict())
    rc.update(train_cfg=None, test_cfg=cfgdict(cfg.model.test_cfg.rcnn.to_dict()))
    rh = build_head(cfgdict(rc))
    g = torch.Generator().manual_seed(21)
    keep2 = []
    orig2 = BN.batched_nms

    def spy2(boxes, scores, idxs, nms_cfg, class_agnostic=False):
        dets, keep = orig2(boxes, scores, idxs, nms_cfg, class_agnostic)
        keep2.append((keep.clone(), boxes.shape[0]))
        return dets, keep
    BN.batched_nms = spy2
    try:
        for b in range(B):
            n = props[b].shape[0]
            cs, bp = util.fullsize_box_head_outputs(g, n)
            prior = props[b][:, -1]
            fused = (cs.softmax(1) * prior[:, None]) ** 0.5
            rois = torch.cat([torch.zeros(n, 1), props[b][:, :4]], 1)
            det, lab = rh.bbox_head.get_bboxes(rois, fused, bp, (800, 1333, 3), np.ones(4, np.float32), rescale=True,
                                               cfg=rh.test_cfg)
            d[f'det{b}'], d[f'lab{b}'] = det, lab.to(torch.int32)
            d[f'keep2_{b}'] = keep2[-1][0][:100].to(torch.int32)
            d[f'ncand2_{b}'] = np.int32(keep2[-1][1])
    finally:
        BN.batched_nms = orig2
    npz('g20_fullsize', **d)


def g21_fullsize_softnms():
    """BASELINE configs[4] at FULL size: 8 images x 2000 proposals x 80 classes through the reference's
    `multiclass_nms` (bbox_nms.py:8-95) with the soft-NMS test settings of
    boosting_rcnn_r2_101_dcn_pafpn_mstrain_3x_coco.py:24-28 (score_thr 1e-4, soft_nms linear, iou 0.7, min_score 0,
    200 per image): detections in pick order with their decayed scores, and labels"""
    from mmdet.core.post_processing.bbox_nms import multiclass_nms
    cfg = Config.fromfile(os.path.join(os.path.dirname(REF_CFG), 'boosting_rcnn_r2_101_dcn_pafpn_mstrain_3x_coco.py'))
    tc = cfg.model.test_cfg.rcnn
    assert tc.score_thr == 0.0001 and tc.nms.type == 'soft_nms' and tc.max_per_img == 200
    d = {}
    for b in range(8):
        boxes, scores = util.fullsize_softnms_candidates(b)
        det, lab = multiclass_nms(boxes, scores, tc.score_thr, cfgdict(tc.nms.to_dict()), tc.max_per_img)
        d[f'det{b}'], d[f'lab{b}'] = det, lab.to(torch.int32)
    npz('g21_fullsize_softnms', **d)


def g22_recalls():
    """eval_recalls (mmdet/core/evaluation/recall.py:12-113, what CocoDataset.evaluate's 'proposal_fast' metric runs,
    datasets/coco.py:311-333,425-434) through the reference: recalls per (proposal_num, IoU threshold) for scored and
    unscored proposals.  Every image holds the same number of boxes: the reference stacks the per-image IoU matrices with
    np.array, which numpy >= 1.24 refuses for ragged shapes."""
    from mmdet.core.evaluation.recall import eval_recalls
    rng = np.random.RandomState(22)

    def boxes(n, jitter_of=None):
        if jitter_of is not None:
            b = jitter_of[rng.randint(0, len(jitter_of), n)] + rng.randn(n, 4) * 6
        else:
            xy = rng.rand(n, 2) * 300
            b = np.concatenate([xy, xy + 10 + rng.rand(n, 2) * 120], axis=1)
        b = b.astype(np.float32)
        b[:, 2:] = np.maximum(b[:, 2:], b[:, :2] + 1)
        return b
    d = {}
    gts = [boxes(7) for _ in range(5)]
    props = [np.concatenate([np.concatenate([boxes(60, g), boxes(40)]), rng.rand(100, 1).astype(np.float32)], axis=1) for g in gts]
    thrs = np.linspace(.5, .95, 10)
    for i, (g, p_) in enumerate(zip(gts, props)):
        d[f'gt{i}'], d[f'prop{i}'] = g, p_
    d['rec_scored'] = eval_recalls(gts, props, (10, 30, 100), thrs)
    d['rec_single_thr'] = eval_recalls(gts, props, 50, 0.5)
    d['rec_unscored'] = eval_recalls(gts, [p_[:, :4] for p_ in props], (10, 100), [0.5, 0.75])
    d['rec_legacy'] = eval_recalls(gts, props, (100,), [0.5, 0.7], use_legacy_coordinate=True)
    npz('g22_recalls', **d)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'recalls':
        g22_recalls()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'soft':
        g21_fullsize_softnms()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'kat':
        kat()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'coco':
        g19_coco_pafpn_train()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'coco2':
        g23_coco_second_stage_inputs()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'full':
        g20_fullsize(Config.fromfile(REF_CFG))
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'variants':
        g18_boost_variants(Config.fromfile(REF_CFG))
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'r2':
        g17_res2net()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'x101':
        g16_resnext_autoaug()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'voc':
        g15_voc()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'data':
        g12_pipeline(None)
        g13_samplers()
        g14_coco_dataset()
        return
    torch.set_num_threads(8)
    cfg = Config.fromfile(REF_CFG)
    kat()
    g1_anchors()
    g2_coder()
    g3_overlaps()
    g4_assign_sample()
    g5_rpn_get_bboxes(cfg)
    g6_rpn_loss(cfg)
    g7_boost_loss(cfg)
    g8_g9_test_head(cfg)
    g10_model(cfg)
    g11_fpn_config()


if __name__ == '__main__':
    main()
