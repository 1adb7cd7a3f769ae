"""-m gpu: the whole-batch train-step kernels (target assignment, RoI sampling, fused RPN loss,
boosting loss, per-level batched NMS) against the golden fixtures produced by the imported
reference (g4 / g6 / g7 / g10) and against this repository's CPU restatement of the reference
(`brcnn.core`, itself pinned to the same fixtures by tests/test_host_cpu.py) on seeded random
inputs.  Bars: assignment / sampling indices and IoUs bit-exact; losses 1e-5 relative; gradients
1e-4 relative (fp32 round-off of a different summation order)."""
import os

import numpy as np
import pytest
import torch

import brcnn  # noqa: F401
from brcnn import Config, build_detector, core, ops, train_ops
from tests import util
from tests.test_host_cpu import CFG, T, _rpn_head, load

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

RPN_KW = dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0, match_low_quality=True, ignore_iof_thr=-1)
RCNN_KW = dict(pos_iou_thr=0.6, neg_iou_thr=0.6, min_pos_iou=0.6, match_low_quality=False, ignore_iof_thr=-1)


def _assign_dev(boxes, gts_list, kw, **extra):
    gts, _, offs = train_ops.flatten_gts([g.to(DEV) for g in gts_list])
    return train_ops.assign_max_iou(boxes, gts, offs, kw['pos_iou_thr'], kw['neg_iou_thr'], kw['min_pos_iou'],
                                    kw['match_low_quality'], **extra)


def test_assign_golden():
    """the reference's own MaxIoUAssigner outputs (fixture g4), both parameterisations"""
    g = load('g4_assign_sample')
    boxes, gts = T(g['boxes']).to(DEV), T(g['gts'])
    for name, kw in (('rpn', RPN_KW), ('rcnn', RCNN_KW)):
        gi, mo, cnt = _assign_dev(boxes.contiguous(), [gts], kw, batch=1, want_overlaps=True, want_counts=True)
        ref = T(g[name + '_gt_inds'])
        assert torch.equal(gi[0].cpu().long(), ref), name
        assert torch.equal(mo[0].cpu(), T(g[name + '_max_overlaps'])), name
        assert cnt.cpu().tolist() == [[int((ref > 0).sum()), int((ref == 0).sum())]]


def test_assign_reference_known_answers():
    """tests/test_utils/test_assigner.py:16-61"""
    bb = torch.tensor([[0, 0, 10, 10], [10, 10, 20, 20], [5, 5, 15, 15], [32, 32, 38, 42.]], device=DEV)
    gg = torch.tensor([[0, 0, 10, 9], [0, 10, 10, 19.]])
    kw = dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.0, match_low_quality=True)
    assert _assign_dev(bb, [gg], kw, batch=1)[0].tolist() == [1, 0, 2, 0]
    assert _assign_dev(bb, [torch.empty(0, 4)], kw, batch=1)[0].tolist() == [0, 0, 0, 0]


def test_bbox_overlaps_and_assigner_reference_api_on_device():
    """the reference-signature entry points on device tensors run the HIP kernels: `bbox_overlaps` (all modes,
    bit-exact vs the reference's own values, fixture g3) and `MaxIoUAssigner.assign` (fixture g4)"""
    g = load('g3_overlaps')
    a, b, c = T(g['a']).to(DEV), T(g['b']).to(DEV), T(g['c']).to(DEV)
    assert torch.equal(core.bbox_overlaps(a, b).cpu(), T(g['iou']))
    assert torch.equal(core.bbox_overlaps(b, c, is_aligned=True).cpu(), T(g['aligned']))
    assert torch.equal(core.bbox_overlaps(b, c, mode='giou', is_aligned=True).cpu(), T(g['giou']))
    assert torch.equal(core.bbox_overlaps(a, b, mode='iof').cpu(), T(g['iof']))
    b1 = torch.tensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42.]], device=DEV)
    b2 = torch.tensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20.]], device=DEV)
    gi = core.bbox_overlaps(b1, b2, 'giou', is_aligned=True, eps=1e-6)          # tests/test_metrics/test_box_overlap.py:84-100
    assert np.allclose(gi.cpu().numpy().round(4), [0.5, -0.05, -0.8214], atol=1e-4)
    g = load('g4_assign_sample')
    boxes, gts, labels = T(g['boxes']).to(DEV), T(g['gts']).to(DEV), T(g['labels']).to(DEV)
    for name, kw in (('rpn', RPN_KW), ('rcnn', RCNN_KW)):
        r = core.MaxIoUAssigner(**kw).assign(boxes, gts, None, labels)
        assert torch.equal(r.gt_inds.cpu(), T(g[name + '_gt_inds']))
        assert torch.equal(r.max_overlaps.cpu(), T(g[name + '_max_overlaps']))
        assert torch.equal(r.labels.cpu(), T(g[name + '_labels']))
    r = core.MaxIoUAssigner(**RCNN_KW).assign(T(g['props']).to(DEV), gts, None, labels)        # 5-column proposals
    torch.manual_seed(1234)
    smp = core.RandomSampler(num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True).sample(
        r, T(g['props']).to(DEV), gts, labels)
    assert torch.equal(smp.pos_inds.cpu(), T(g['s_pos_inds'])) and torch.equal(smp.neg_inds.cpu(), T(g['s_neg_inds']))
    assert torch.equal(smp.bboxes.cpu(), T(g['s_bboxes']))
    r0 = core.MaxIoUAssigner(**RPN_KW).assign(boxes, torch.empty(0, 4, device=DEV), None, torch.empty(0, dtype=torch.long, device=DEV))
    assert (r0.gt_inds == 0).all() and (r0.labels == -1).all()


def _anchor_case(seed, sizes, pad_shapes, strides, num_gts, border=-1):
    """RPN anchors of a small pyramid, per-image padded shapes and ground truth; returns the device
    result and the CPU restatement (anchor_head.py:199-262 via brcnn.core)"""
    gen = core.AnchorGenerator(strides=strides, ratios=[0.5, 1.0, 2.0], octave_base_scale=4, scales_per_octave=3)
    anchors = torch.cat(gen.grid_anchors(sizes, 'cpu'), 0)
    W, H = sizes[0][1] * strides[0], sizes[0][0] * strides[0]
    g = torch.Generator().manual_seed(seed)
    gts = []
    for n in num_gts:
        b = util.rand_boxes(n, W, H, seed=seed + n, min_size=8, max_size=min(W, H) * 0.8) if n else torch.empty(0, 4)
        if n >= 4:
            b[1] = b[0]                      # duplicate ground truth: equal IoU columns (tie on argmax)
            b[2, 2:] = b[2, :2]              # zero-area ground truth: gt_max 0 (the min_pos_iou=0 quirk)
        gts.append(b)
    starts = [0]
    for (h, w) in sizes:
        starts.append(starts[-1] + h * w * 9)
    geom = (starts, [w for _, w in sizes], 9)
    valid_hw = torch.tensor([[[min(int(np.ceil(ps[0] / s)), h), min(int(np.ceil(ps[1] / s)), w)]
                              for (h, w), s in zip(sizes, strides)] for ps in pad_shapes], dtype=torch.int32)
    img_hw = torch.tensor([[ps[0], ps[1] - 3] for ps in pad_shapes], dtype=torch.float32)
    gi, mo = _assign_dev(anchors.to(DEV).contiguous(), gts, RPN_KW, batch=len(num_gts), geom=geom,
                         valid_hw=valid_hw.to(DEV), img_hw=img_hw.to(DEV) if border >= 0 else None,
                         allowed_border=border, want_overlaps=True)
    ref_gi, ref_mo = [], []
    a = core.MaxIoUAssigner(**RPN_KW)
    for b, ps in enumerate(pad_shapes):
        flags = torch.cat(gen.valid_flags(sizes, ps, 'cpu'))
        inside = core.anchor_inside_flags(anchors, flags, (ps[0], ps[1] - 3), border)
        r = a.assign(anchors[inside], gts[b], None, None)
        ref_gi.append(core.unmap(r.gt_inds, anchors.shape[0], inside, fill=-1))
        ref_mo.append(core.unmap(r.max_overlaps, anchors.shape[0], inside, fill=0))
    return gi.cpu().long(), mo.cpu(), torch.stack(ref_gi), torch.stack(ref_mo)


@pytest.mark.parametrize('border', [-1, 0])
def test_assign_anchors_batch_vs_cpu(border):
    sizes = [(32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]
    pads = [(256, 384), (200, 384), (256, 300), (160, 200)]
    gi, mo, rgi, rmo = _anchor_case(3, sizes, pads, [8, 16, 32, 64, 128], [7, 0, 300, 5], border)
    assert torch.equal(gi, rgi)
    assert torch.equal(mo, rmo)
    assert (gi > 0).any() and (gi == -1).any()


def test_assign_full_size_properties():
    """BASELINE size: 8 images x 201 600 anchors x 20 ground truths -- every ground truth owns at least one
    anchor (low-quality rule), positives overlap their ground truth by >= the threshold or are its best match"""
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    gen = core.AnchorGenerator(strides=[8, 16, 32, 64, 128], ratios=[0.5, 1.0, 2.0], octave_base_scale=4,
                               scales_per_octave=3)
    anchors = torch.cat(gen.grid_anchors(sizes, 'cpu'), 0).to(DEV).contiguous()
    assert anchors.shape[0] == 201600
    gts = [util.rand_boxes(20, 1333, 800, seed=40 + b, min_size=16, max_size=600) for b in range(8)]
    gi, mo = _assign_dev(anchors, gts, RPN_KW, batch=8, want_overlaps=True)
    gi, mo = gi.cpu().long(), mo.cpu()
    for b in range(8):
        assert set(range(1, 21)) <= set(gi[b].unique().tolist())
        iou = core.bbox_overlaps(gts[b], anchors.cpu())
        assert torch.equal(mo[b], iou.max(0)[0])
        pos = gi[b] > 0
        own = iou[gi[b][pos] - 1, pos.nonzero().squeeze(1)]
        assert ((own >= 0.5) | (own == iou.max(1)[0][gi[b][pos] - 1])).all()
        assert ((gi[b] == 0) == ((mo[b] < 0.5) & ~pos)).all()


def _cpu_rcnn_chain(props_list, gts, gls, seed, quality=False, num=512):
    """ProbRoIHead.forward_train's sampling block on the CPU restatement (prob_roi_head.py:23-69)"""
    cfg = Config.fromfile(CFG)
    head = brcnn.build_head(cfg.model.roi_head.bbox_head)
    a = core.MaxIoUAssigner(**RCNN_KW)
    sp = core.RandomSampler(num=num, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)
    torch.manual_seed(seed)
    srs, priors, ious = [], [], []
    for p, g, l in zip(props_list, gts, gls):
        r = a.assign(p, g, None, l)
        s = sp.sample(r, p, g, l)
        srs.append(s)
        n = r.num_gts
        pi, ni = s.pos_inds[n:].clone() - n, s.neg_inds.clone() - n
        priors.append(torch.cat([p.new_zeros(n), p[pi, -1], 1 - p[ni, -1]]))
        ious.append(torch.cat([r.max_overlaps[s.pos_inds], 1 - r.max_overlaps[s.neg_inds]]))
    rois = core.bbox2roi([s.bboxes for s in srs])
    labels, _, tgt, _ = head.get_targets(srs, gts, gls, cfg.model.train_cfg.rcnn)
    return rois, labels, tgt, torch.cat(priors), torch.cat(ious), srs


def _dev_rcnn_chain(props_list, gts, gls, seed, num=512):
    B = len(props_list)
    K = max(p.shape[0] for p in props_list)
    dets = torch.zeros(B, K, 5)
    for b, p in enumerate(props_list):
        dets[b, :p.shape[0]] = p
    nums = torch.tensor([p.shape[0] for p in props_list], dtype=torch.int32)
    dets, nums = dets.to(DEV), nums.to(DEV)
    gflat, lflat, offs = train_ops.flatten_gts([g.to(DEV) for g in gts], [l.to(DEV) for l in gls])
    gi, mo, cnt = train_ops.assign_max_iou(dets, gflat, offs, 0.6, 0.6, 0.6, False, num_boxes=nums, want_overlaps=True,
                                           want_counts=True)
    counts = [(p_ + offs[b + 1] - offs[b], n_) for b, (p_, n_) in enumerate(cnt.cpu().tolist())]
    torch.manual_seed(seed)
    perm, rows = train_ops.draw_sampler_perms(counts, num, int(num * 0.25), -1)
    out = train_ops.rcnn_sample(dets, nums, gi, mo, gflat, lflat, offs, perm.to(DEV), rows, num, int(num * 0.25), -1, 4,
                                (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2), want_ious=True, want_pos_flags=True)
    return out, rows


def test_rcnn_sample_golden_and_cpu_chain():
    g = load('g4_assign_sample')
    props, gts, labels = T(g['props']), T(g['gts']), T(g['labels'])
    out, rows = _dev_rcnn_chain([props], [gts], [labels], 1234)
    assert rows == [0, 512]
    assert torch.equal(out['rois'][:, 1:].cpu(), T(g['s_bboxes']))         # the reference's sampled boxes, in order
    assert torch.equal(out['labels'][:128].cpu(), labels[T(g['s_pos_assigned'])])
    rois, lab, tgt, pri, iou, _ = _cpu_rcnn_chain([props], [gts], [labels], 1234)
    assert torch.equal(out['rois'].cpu(), rois) and torch.equal(out['labels'].cpu(), lab)
    assert torch.equal(out['priors'].cpu(), pri) and torch.equal(out['ious'].cpu(), iou)
    assert torch.allclose(out['bbox_targets'].cpu(), tgt, rtol=1e-5, atol=1e-6)
    assert out['pos_flags'].cpu().tolist() == [1] * 128 + [0] * 384


def test_rcnn_sample_ragged_batch_vs_cpu():
    """images with many / few / no positives, no ground truth, fewer candidates than the sampler wants"""
    gts, gls, props = [], [], []
    spec = [(6, 900, 40), (3, 60, 2), (0, 500, 0), (10, 2000, 80), (2, 300, 0)]
    for b, (ng, nrand, jit) in enumerate(spec):
        g = util.rand_boxes(ng, 1333, 800, seed=70 + b, min_size=24, max_size=500) if ng else torch.empty(0, 4)
        gen = torch.Generator().manual_seed(90 + b)
        near = (g.repeat(jit, 1) + torch.randn(ng * jit, 4, generator=gen) * 4) if ng and jit else torch.empty(0, 4)
        bx = torch.cat([near, util.rand_boxes(nrand, 1333, 800, seed=80 + b)], 0)
        bx = torch.cat([torch.min(bx[:, :2], bx[:, 2:]), torch.max(bx[:, :2], bx[:, 2:]) + 1], 1)
        bx = bx[torch.randperm(bx.shape[0], generator=gen)]
        props.append(torch.cat([bx, torch.rand(bx.shape[0], 1, generator=gen)], 1))
        gts.append(g)
        gls.append(torch.randint(0, 4, (ng,), generator=gen))
    out, rows = _dev_rcnn_chain(props, gts, gls, 99)
    rois, lab, tgt, pri, iou, srs = _cpu_rcnn_chain(props, gts, gls, 99)
    assert rows[-1] == rois.shape[0]
    assert [rows[b + 1] - rows[b] for b in range(5)] == [len(s.pos_inds) + len(s.neg_inds) for s in srs]
    assert torch.equal(out['rois'].cpu(), rois) and torch.equal(out['labels'].cpu(), lab)
    assert torch.equal(out['priors'].cpu(), pri) and torch.equal(out['ious'].cpu(), iou)
    assert torch.allclose(out['bbox_targets'].cpu(), tgt, rtol=1e-5, atol=1e-6)
    assert any(len(s.pos_inds) == 128 for s in srs) and any(len(s.pos_inds) < 128 for s in srs)


@pytest.mark.parametrize('gamma', [0.5, 2])
def test_rpn_loss_golden(gamma):
    """ATSSRPNHead.loss through its reference signature on device tensors == the fused kernels:
    per-level loss values and the gradients of every head output against the reference's (fixture g6)"""
    g = load('g6_rpn_loss')
    head, _ = _rpn_head()
    head = head.to(DEV)
    head.gamma = gamma
    assert head.device_train_ok()
    cls = [T(g[f'cls{i}']).to(DEV).requires_grad_() for i in range(5)]
    reg = [T(g[f'reg{i}']).to(DEV).requires_grad_() for i in range(5)]
    iou = [T(g[f'iou{i}']).to(DEV).requires_grad_() for i in range(5)]
    _, metas, _, _ = util.demo_inputs(2, 128, 192, seed=6)
    gts = [T(g['gt0']).to(DEV), T(g['gt1']).to(DEV)]
    out = head.loss(cls, reg, iou, gts, metas)
    per_level = head.last_rpn_targets[1].cpu()
    for r, k in enumerate(('loss_rpn_cls', 'loss_rpn_bbox', 'loss_rpn_iou')):
        ref = T(g[f'g{gamma}_{k}'])
        assert torch.allclose(per_level[r], ref, rtol=1e-5, atol=1e-6), (k, per_level[r], ref)
        assert torch.allclose(out[k][0].cpu(), ref.sum(), rtol=1e-5, atol=1e-6)
    tot = sum(sum(v) for v in out.values())
    grads = torch.autograd.grad(tot, cls + reg + iou)
    for i in range(5):
        for j, nm in enumerate(('dcls', 'dreg', 'diou')):
            ref = T(g[f'g{gamma}_{nm}{i}'])
            got = grads[5 * j + i].cpu()
            assert torch.allclose(got, ref, rtol=1e-4, atol=1e-7), (nm, i, (got - ref).abs().max())


def test_rpn_loss_scale_gradient_and_padding():
    """the production layout: raw deltas + learnable per-level Scale, 64-channel padded rows; gradient
    w.r.t. the scales and the raw head output against autograd of the CPU restatement"""
    g = load('g6_rpn_loss')
    head, _ = _rpn_head()
    sizes = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
    _, metas, _, _ = util.demo_inputs(2, 128, 192, seed=6)
    gts = [T(g['gt0']), T(g['gt1'])]
    sc = torch.tensor([1.3, 0.7, 1.1, 0.9, 1.5])
    cls = [T(g[f'cls{i}']) for i in range(5)]
    raw = [T(g[f'reg{i}']).clone().requires_grad_() for i in range(5)]
    iou = [T(g[f'iou{i}']) for i in range(5)]
    scp = sc.clone().requires_grad_()
    out = head.loss(cls, [r * scp[i] for i, r in enumerate(raw)], iou, gts, metas)          # CPU restatement
    tot = sum(sum(v) for v in out.values())
    ref_g = torch.autograd.grad(tot, raw + [scp])
    rows = [torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in (c_, r_, i_)], 1)
            for c_, r_, i_ in zip(cls, [r.detach() for r in raw], iou)]
    y = torch.cat(rows, 0)
    y = torch.cat([y, torch.full((y.shape[0], 10), 7.0)], 1).to(DEV).requires_grad_()      # 54 -> 64 channels
    head = head.to(DEV)
    scd = sc.to(DEV).requires_grad_()
    o2 = head.loss_fused(y, tuple(sizes), [t.to(DEV) for t in gts], metas, scales=scd)
    t2 = sum(sum(v) for v in o2.values())
    assert torch.allclose(t2.cpu(), tot.detach(), rtol=1e-5)
    gy, gs = torch.autograd.grad(t2, [y, scd])
    assert torch.allclose(gs.cpu(), ref_g[5], rtol=1e-4, atol=1e-7), (gs, ref_g[5])
    assert (gy[:, 54:] == 0).all()
    r0 = 0
    for i, (h, w) in enumerate(sizes):
        n = 2 * h * w
        got = gy[r0:r0 + n, 9:45].view(2, h, w, 36).permute(0, 3, 1, 2).cpu()
        assert torch.allclose(got, ref_g[i], rtol=1e-4, atol=1e-7), i
        r0 += n


def _variant_head(recipe):
    from tests.test_host_cpu import ROOT
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'boosting_rcnn', recipe))
    c = cfg.model.rpn_head.copy()
    c.update(train_cfg=cfg.model.train_cfg.rpn, test_cfg=cfg.model.test_cfg.rpn)
    return brcnn.build_head(c)


@pytest.mark.parametrize('recipe,modes', [('boosting_rcnn_r50_fpn_1x_coco.py', (0, 1)),        # CIoU on raw deltas
                                          ('boosting_rcnn_r50_pafpn_1x_voc.py', (1, 0))])      # VarifocalLoss
def test_rpn_loss_variants_equal_cpu_chain(recipe, modes):
    """the FPN recipe's reg_decoded_bbox=False + CIoULoss and the VOC recipe's VarifocalLoss on the fused
    kernels: loss values, gradients w.r.t. the raw head output and the per-level Scale against autograd of
    the CPU restatement (pinned to the reference by fixtures g11 / g15)"""
    g = load('g6_rpn_loss')
    head = _variant_head(recipe)
    assert head._fused_loss_modes() == modes and head.device_train_ok()
    sizes = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
    _, metas, _, _ = util.demo_inputs(2, 128, 192, seed=6)
    gts = [T(g['gt0']), T(g['gt1'])]
    sc = torch.tensor([1.3, 0.7, 1.1, 0.9, 1.5])
    A = head.num_anchors
    gen = torch.Generator().manual_seed(21)
    cls = [(torch.randn(2, A, h, w, generator=gen) - 2).requires_grad_() for h, w in sizes]
    raw = [(torch.randn(2, 4 * A, h, w, generator=gen) * 0.3).requires_grad_() for h, w in sizes]
    iou = [torch.randn(2, A, h, w, generator=gen).requires_grad_() for h, w in sizes]
    scp = sc.clone().requires_grad_()
    out = head.loss(cls, [r * scp[i] for i, r in enumerate(raw)], iou, gts, metas)          # CPU restatement
    assert sum(out['loss_rpn_bbox']).item() > 0
    tot = sum(sum(v) for v in out.values())
    ref_g = torch.autograd.grad(tot, cls + raw + iou + [scp])
    rows = [torch.cat([t.detach().permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in (c_, r_, i_)], 1)
            for c_, r_, i_ in zip(cls, raw, iou)]
    y = torch.cat(rows, 0)
    y = torch.cat([y, torch.zeros((y.shape[0], 64 - 6 * A))], 1).to(DEV).requires_grad_()   # 6A -> 64 channels
    head = head.to(DEV)
    scd = sc.to(DEV).requires_grad_()
    o2 = head.loss_fused(y, tuple(sizes), [t.to(DEV) for t in gts], metas, scales=scd)
    for k in o2:
        assert torch.allclose(o2[k][0].cpu(), sum(out[k]).detach(), rtol=2e-5, atol=1e-6), (k, o2[k][0], sum(out[k]))
    t2 = sum(sum(v) for v in o2.values())
    gy, gs = torch.autograd.grad(t2, [y, scd])
    assert torch.allclose(gs.cpu(), ref_g[15], rtol=2e-4, atol=1e-7), (gs, ref_g[15])
    r0 = 0
    for i, (h, w) in enumerate(sizes):
        n = 2 * h * w
        for lo, hi, ref in ((0, A, ref_g[i]), (A, 5 * A, ref_g[5 + i]), (5 * A, 6 * A, ref_g[10 + i])):
            got = gy[r0:r0 + n, lo:hi].view(2, h, w, hi - lo).permute(0, 3, 1, 2).cpu()
            tol = 2e-4 * ref.abs().max().item() + 1e-7
            assert (got - ref).abs().max().item() <= tol, (recipe, i, lo, (got - ref).abs().max().item(), tol)
        r0 += n


def test_rpn_loss_varifocal_reference_golden():
    """the VOC recipe's RPN loss (VarifocalLoss) on the fused kernels against the reference's own per-level
    values (fixture g15: rpn.loss of the imported reference on the same seeded head outputs)"""
    g = load('g15_voc')
    head = _variant_head('boosting_rcnn_r50_pafpn_1x_voc.py').to(DEV)
    cls = [T(g[f'rpn_cls{i}']).to(DEV) for i in range(5)]
    reg = [T(g[f'rpn_reg{i}']).to(DEV) for i in range(5)]
    iou = [T(g[f'rpn_iou{i}']).to(DEV) for i in range(5)]
    _, metas, gts, _ = util.demo_inputs(2, 128, 192, seed=15, num_gt=4)
    out = head.loss(cls, reg, iou, [b.to(DEV) for b in gts], metas)
    per_level = head.last_rpn_targets[1].cpu()
    for r, k in enumerate(('loss_rpn_cls', 'loss_rpn_bbox', 'loss_rpn_iou')):
        ref = T(g['rpn_' + k])
        assert torch.allclose(per_level[r], ref, rtol=1e-5, atol=1e-6), (k, per_level[r], ref)
        assert torch.allclose(out[k][0].cpu(), ref.sum(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('gamma', [0.5, 0.1])
def test_boost_loss_golden(gamma):
    """loss values, accuracy and both gradients of the reference's boosting loss (fixture g7)"""
    g = load('g7_boost_loss')
    cls = T(g['cls_score']).to(DEV).requires_grad_()
    bb = T(g['bbox_pred']).to(DEV).requires_grad_()
    out3 = train_ops.boost_loss(cls, bb, T(g['labels']).to(DEV), T(g['priors']).to(DEV), T(g['bbox_targets']).to(DEV),
                                4, gamma, loss_cls_weight=2.0, loss_bbox_weight=2.0)
    gc, gb = torch.autograd.grad(out3[0] + out3[1], [cls, bb])
    assert torch.allclose(out3[0].cpu(), T(g[f'g{gamma}_loss_cls']), rtol=1e-5)
    assert torch.allclose(out3[1].cpu(), T(g[f'g{gamma}_loss_bbox']), rtol=1e-5)
    assert torch.allclose(out3[2].cpu(), T(g[f'g{gamma}_acc'])[0])
    assert torch.allclose(gc.cpu(), T(g[f'g{gamma}_dcls']), rtol=1e-4, atol=1e-9)
    assert torch.allclose(gb.cpu(), T(g[f'g{gamma}_dbbox']), rtol=1e-5, atol=1e-9)


def test_boost_loss_80_classes_quality_vs_cpu():
    """COCO head width (81 logits, two lanes rounds), the `quality` and `alpha` factors, reg_norm='mean',
    a non-unit upstream gradient"""
    from brcnn.roi_heads import ProbRoIHead
    gen = torch.Generator().manual_seed(5)
    n, C = 700, 80
    cls = torch.randn(n, C + 1, generator=gen) * 2
    bb = torch.randn(n, 4 * C, generator=gen)
    labels = torch.randint(0, C + 1, (n,), generator=gen)
    labels[::3] = C
    pri, iou = torch.rand(n, generator=gen), torch.rand(n, generator=gen)
    tgt = torch.randn(n, 4, generator=gen)
    c1, b1 = cls.clone().requires_grad_(), bb.clone().requires_grad_()
    L = 2.0 * torch.nn.functional.cross_entropy(c1, labels, reduction='none')
    p = torch.gather(c1.softmax(1).detach(), 1, labels[:, None]).squeeze(1)
    w = ((iou - p).abs() ** 0.7 * (1 - pri) ** 0.3) * 1.5
    loss_cls = ProbRoIHead.norm_loss(L, w, n)
    pos = labels < C
    l1 = 2.0 * (b1.view(n, C, 4)[pos, labels[pos]] - tgt[pos]).abs()
    loss_bbox = l1.mean()
    rc, rb = torch.autograd.grad(3.0 * loss_cls + 0.5 * loss_bbox, [c1, b1])
    c2, b2 = cls.to(DEV).requires_grad_(), bb.to(DEV).requires_grad_()
    out3 = train_ops.boost_loss(c2, b2, labels.to(DEV), pri.to(DEV), tgt.to(DEV), C, 0.3, alpha=1.5, ious=iou.to(DEV),
                                iou_gamma=0.7, loss_cls_weight=2.0, loss_bbox_weight=2.0, reg_norm='mean')
    gc, gb = torch.autograd.grad(3.0 * out3[0] + 0.5 * out3[1], [c2, b2])
    assert torch.allclose(out3[0].cpu(), loss_cls.detach(), rtol=1e-5)
    assert torch.allclose(out3[1].cpu(), loss_bbox.detach(), rtol=1e-5)
    acc = (cls.argmax(1) == labels).float().mean() * 100
    assert torch.allclose(out3[2].cpu(), acc, rtol=1e-6)
    assert torch.allclose(gc.cpu(), rc, rtol=2e-4, atol=1e-8)
    assert torch.allclose(gb.cpu(), rb, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('soft', [None, dict(method='linear', sigma=0.5, min_score=1e-3)])
def test_batched_nms_by_level_fused_equals_torch_chain(soft):
    """mmcv batched_nms above split_thr for a whole batch: the three-launch device form against the
    torch-op chain it replaces (itself tested against the per-image reference path)"""
    from brcnn.postprocess import batched_nms_images_by_level
    B, sizes = 3, [4000, 4000, 4000, 2457, 693]
    T_ = sum(sizes)
    boxes = torch.stack([util.clustered_boxes(T_, 300, seed=b) for b in range(B)]).to(DEV)
    scores = torch.stack([util.tie_free_scores(T_, seed=10 + b) for b in range(B)]).to(DEV)
    ids = torch.cat([torch.full((n,), l) for l, n in enumerate(sizes)]).expand(B, T_).contiguous().to(DEV)
    valid = (torch.rand(B, T_, generator=torch.Generator().manual_seed(3)) > 0.1).to(DEV)
    valid[2, 4000:8000] = False                      # an empty (image, level) segment
    if soft is not None:                             # soft-NMS segments are sequential: keep them short
        sizes = [500] * 8
        T_ = 4000
        boxes, scores, valid = boxes[:, :T_].contiguous(), scores[:, :T_].contiguous(), valid[:, :T_].contiguous()
        ids = torch.cat([torch.full((n,), l) for l, n in enumerate(sizes)]).expand(B, T_).contiguous().to(DEV)
    a = batched_nms_images_by_level(boxes, scores, ids, valid, sizes, 0.7, 2000, 0, return_ids=True, soft=soft, fused=True)
    b = batched_nms_images_by_level(boxes, scores, ids, valid, sizes, 0.7, 2000, 0, return_ids=True, soft=soft, fused=False)
    assert torch.equal(a[2], b[2].to(a[2].dtype))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def _model(cfg_path=CFG, seed=10):
    cfg = Config.fromfile(cfg_path)
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=seed))
    return m.to(DEV).train()


def test_train_step_device_path_equals_reference_chain():
    """the device-resident train step (whole-batch kernels) against the per-image / per-level chain of the
    reference's structure (device_train_path=False) on the same weights, inputs and sampler seed: every
    loss and every parameter gradient"""
    m = _model()
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10)
    args = (img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
    res = {}
    for mode in (True, False):
        m.device_train_path = mode
        m.zero_grad(set_to_none=True)
        torch.manual_seed(77)
        losses = m.forward_train(*args)
        loss, log_vars = m._parse_losses(losses)
        loss.backward()
        res[mode] = (dict(log_vars), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    assert res[True][0].keys() == res[False][0].keys()
    for k in res[True][0]:
        assert np.isclose(res[True][0][k], res[False][0][k], rtol=2e-4, atol=1e-6), (k, res[True][0][k], res[False][0][k])
    assert res[True][1].keys() == res[False][1].keys()
    for k, ga in res[True][1].items():
        gb = res[False][1][k]
        scale = gb.abs().max().item() + 1e-12
        assert (ga - gb).abs().max().item() <= 2e-3 * scale, (k, (ga - gb).abs().max().item(), scale)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_dyprob_detector_trains_on_the_device_path(dtype):
    """the detector with a DyProbRoIHead (Dynamic R-CNN schedule on the boosting head, prob_roi_head.py:473-623; no shipped
    recipe) takes the whole-batch device path: losses and parameter gradients of three iterations equal the
    per-image chain (device_train_path=False) on the same sampler seed, and the schedule moves the same thresholds"""
    import copy
    from brcnn import blocks
    cfg = Config.fromfile(CFG)
    mc = copy.deepcopy(cfg.model)
    mc.roi_head.type = 'DyProbRoIHead'
    mc.roi_head.bbox_head.loss_bbox = dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)
    mc.train_cfg.rcnn.dynamic_rcnn = dict(iou_topk=30, beta_topk=4, update_iter_interval=2, initial_iou=0.4, initial_beta=1.0)
    models = []
    for _ in range(2):
        m = build_detector(copy.deepcopy(mc))
        m.load_state_dict(util.seeded_state_dict(m, seed=10))
        models.append(m.to(DEV).train())
    dev_m, ref_m = models
    ref_m.device_train_path = False
    try:
        for m in models:
            m.set_compute_dtype(dtype)
        assert dev_m.roi_head.device_train_ok() and dev_m._device_train_ok(torch.zeros(1, device=DEV), None, None)
        tol = 2e-3 if dtype == 'f32' else 6e-2
        for it in range(3):
            img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=20 + it)
            args = (img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
            out = []
            for m in models:
                m.zero_grad(set_to_none=True)
                torch.manual_seed(77 + it)
                loss, log_vars = m._parse_losses(m.forward_train(*args))
                loss.backward()
                out.append((dict(log_vars), {k: p.grad.float().clone() for k, p in m.named_parameters() if p.grad is not None}))
            for k in out[0][0]:
                assert np.isclose(out[0][0][k], out[1][0][k], rtol=tol, atol=1e-4), (it, k, out[0][0][k], out[1][0][k])
            if dtype == 'f32':
                assert out[0][1].keys() == out[1][1].keys()
                for k, ga in out[0][1].items():
                    gb = out[1][1][k]
                    scale = gb.abs().max().item() + 1e-12
                    assert (ga - gb).abs().max().item() <= tol * scale, (it, k, (ga - gb).abs().max().item(), scale)
            a, b = dev_m.roi_head, ref_m.roi_head
            assert len(a.iou_history) == len(b.iou_history) and len(a.beta_history) == len(b.beta_history)
            assert np.isclose(a.bbox_assigner.pos_iou_thr, b.bbox_assigner.pos_iou_thr, rtol=1e-3 if dtype == 'f32' else 5e-2)
            assert np.isclose(a.bbox_head.loss_bbox.beta, b.bbox_head.loss_bbox.beta, rtol=1e-3 if dtype == 'f32' else 5e-2)
        assert dev_m.roi_head.bbox_assigner.pos_iou_thr != 0.5 or dev_m.roi_head.bbox_head.loss_bbox.beta != 1.0
    finally:
        blocks.set_compute_dtype('f32')


def test_wgrad_side_stream_gives_the_same_gradients():
    """weight-gradient launches on the second HIP stream (autograd.WGRAD_SIDE_STREAM): every parameter gradient of a
    bf16 train step equals the single-stream run up to the order of the fp32 atomics, over several steps that reuse
    the allocator's blocks (a missing dependency or a recycled buffer shows up as a wrong gradient)"""
    from brcnn import autograd as A, blocks
    m = _model()
    blocks.conv_weights_channels_last(m)
    m.set_compute_dtype('bf16')
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10)
    args = (img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
    saved = A.WGRAD_SIDE_STREAM
    res = {}
    try:
        for mode in (False, True, True, True):
            A.WGRAD_SIDE_STREAM = mode
            m.zero_grad(set_to_none=True)
            torch.manual_seed(77)
            loss, _ = m._parse_losses(m.forward_train(*args))
            loss.backward()
            torch.cuda.current_stream().synchronize()       # the join at the end of backward orders the side stream
            grads = {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}
            if not mode:
                res['ref'] = grads
            else:
                assert grads.keys() == res['ref'].keys()
                for k, ga in grads.items():
                    gb = res['ref'][k]
                    scale = gb.abs().max().item() + 1e-12
                    assert (ga - gb).abs().max().item() <= 1e-4 * scale, (k, (ga - gb).abs().max().item(), scale)
    finally:
        A.WGRAD_SIDE_STREAM = saved
        blocks.set_compute_dtype('f32')


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
def test_deferred_slab_reductions_give_the_same_bits(dtype):
    """autograd.set_wgrad_defer (BRCNN_WGRAD_DEFER=1, csrc/wgrad_defer.hip): the slab reductions of the weight-gradient
    launches on the second stream batched into table-driven launches -- every parameter gradient of a 16-bit train step
    (early RPN backward on: two backward passes, the first without a join) equals the per-layer form BIT FOR BIT, over
    several steps; the batched launches really ran; switching off again releases the arena"""
    import ctypes
    from brcnn import autograd as A, blocks, lib
    L = lib.load()
    m = _model()
    blocks.conv_weights_channels_last(m)
    m.set_compute_dtype(dtype)
    img, metas, gts, gls = util.demo_inputs(2, 256, 320, seed=12)
    args = (img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
    scale = 512.0 if dtype == 'f16' else 1.0
    saved = (A.WGRAD_DEFER, A.WGRAD_DEFER_ITEMS, m.early_rpn_backward, m.early_backward_scale)
    ref = None
    try:
        m.early_rpn_backward, m.early_backward_scale = True, scale
        for mode in (False, True, True, True, False):
            A.set_wgrad_defer(mode)
            m.zero_grad(set_to_none=True)
            A.grad_arena.new_step()
            torch.manual_seed(77)
            loss, _ = m._parse_losses(m.forward_train(*args))
            (loss * scale).backward()
            torch.cuda.current_stream().synchronize()
            grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
            if ref is None:
                ref = grads
                continue
            assert grads.keys() == ref.keys()
            for k, ga in grads.items():
                assert torch.equal(ga, ref[k]), (mode, k, float((ga.float() - ref[k].float()).abs().max()))
            if mode:
                side = A._side_streams[('cuda', 0)]
                fl, it = ctypes.c_longlong(0), ctypes.c_longlong(0)
                L.brcnn_wgrad_defer_stats(side.cuda_stream, ctypes.byref(fl), ctypes.byref(it))
                assert it.value >= 10 and fl.value >= 1 and it.value > 2 * fl.value, (fl.value, it.value)
                assert L.brcnn_wgrad_defer_pending(side.cuda_stream) == 0
        assert not A._defer_arenas
    finally:
        A.WGRAD_DEFER_ITEMS = saved[1]
        A.set_wgrad_defer(saved[0])
        m.early_rpn_backward, m.early_backward_scale = saved[2], saved[3]
        blocks.set_compute_dtype('f32')


@pytest.mark.parametrize('dtype,scale', [('f32', 1.0), ('bf16', 1.0), ('f16', 512.0)])
def test_early_rpn_backward_gives_the_same_step(dtype, scale):
    """`early_rpn_backward`: the RPN branch back-propagated inside the forward pass (proposal stage on a second
    stream) + the second stage's backward() afterwards = the one backward pass over both branches: same losses, and
    every parameter gradient equal up to the order of the weight-gradient atomics in fp32 (the two pyramid-gradient terms
    are added in the other order: a + b == b + a).  In the 16-bit modes the early form adds the RPN branch's pyramid
    gradient INSIDE the RoIAlign gradient gather (fp32 sum, one rounding at the store: round 6) where the one-pass form
    rounds the gather's result and then the autograd add's -- the pyramid gradients differ by one 16-bit rounding and
    the parameter gradients behind them by a fraction of that.  Over several steps that reuse the allocator's blocks
    across the two streams"""
    from brcnn import blocks
    m = _model()
    blocks.conv_weights_channels_last(m)
    m.set_compute_dtype(dtype)
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10)
    args = (img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
    res = {}
    try:
        for mode in (False, True, True, True):
            m.early_rpn_backward, m.early_backward_scale = mode, scale
            m.zero_grad(set_to_none=True)
            torch.manual_seed(77)
            loss, log_vars = m._parse_losses(m.forward_train(*args))
            if mode:        # the RPN parameters have their gradients already
                assert m.rpn_head.rpn_convs[0].conv.weight.grad is not None
            (loss * scale).backward()
            torch.cuda.synchronize()
            grads = {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}
            if not mode:
                res['ref'] = (dict(log_vars), grads)
            else:
                assert dict(log_vars) == res['ref'][0]
                assert grads.keys() == res['ref'][1].keys()
                for k, ga in grads.items():
                    gb = res['ref'][1][k]
                    # (equal up to the order of the fp32 atomics of the fp32 weight-gradient kernel)
                    scl = gb.abs().max().item() + 1e-12
                    # (16-bit: the two forms differ by ONE rounding of the pyramid gradient per level, and a parameter
                    # gradient sums many such terms: two roundings of its largest entry bound what was measured, 1.1)
                    tol = {'f32': 1e-4, 'bf16': 2 ** -7, 'f16': 2 ** -10}[dtype]
                    assert (ga - gb).abs().max().item() <= tol * scl, (k, (ga - gb).abs().max().item(), scl)
    finally:
        m.early_rpn_backward, m.early_backward_scale = False, 1.0
        blocks.set_compute_dtype('f32')


def test_wgrad_side_stream_with_a_weight_used_twice():
    """one conv applied to two inputs in one graph (the per-level RPN fallback shares its ConvModules across pyramid
    levels): autograd sums the two weight gradients on the main stream as soon as the second arrives, so only the first
    may use the side stream (autograd.py side-stream rule)"""
    from brcnn import autograd as A
    from brcnn.autograd import ConvNHWCFunction
    torch.manual_seed(11)
    saved = A.WGRAD_SIDE_STREAM
    res = {}
    try:
        for kh in (1, 3):
            w0 = (torch.randn(128, 64, kh, kh) / 16).to(DEV)
            if kh == 3:
                w0 = w0.contiguous(memory_format=torch.channels_last)
            b0 = torch.randn(128).to(DEV)
            xs = [torch.randn(2 * 40 * 56, 64, device=DEV).to(torch.bfloat16), torch.randn(2 * 24 * 32, 64, device=DEV).to(torch.bfloat16)]
            for mode in (False, True, True, True):
                A.WGRAD_SIDE_STREAM = mode
                w = w0.clone().requires_grad_(True)
                if kh == 3:
                    w = w0.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
                b = b0.clone().requires_grad_(True)
                ya = ConvNHWCFunction.apply(xs[0].clone().requires_grad_(True), w, b, 2, [(40, 56)], 1, kh // 2)
                yb = ConvNHWCFunction.apply(xs[1].clone().requires_grad_(True), w, b, 2, [(24, 32)], 1, kh // 2)
                (ya.float().pow(2).sum() + yb.float().pow(2).sum() * 3).backward()
                torch.cuda.current_stream().synchronize()
                if not mode:
                    res[kh] = (w.grad.clone(), b.grad.clone())
                else:
                    for got, ref in zip((w.grad, b.grad), res[kh]):
                        scale = ref.abs().max().item()
                        assert (got - ref).abs().max().item() <= 1e-4 * scale, (kh, (got - ref).abs().max().item(), scale)
    finally:
        A.WGRAD_SIDE_STREAM = saved


def test_fused_sgd_first_step_skipped_leaves_a_defined_momentum_buffer():
    """the very first step carries an inf gradient (fp16 static loss scaling overflow): the device skips it, and the
    next step must behave like torch's first step (optim.py momentum buffer)"""
    from brcnn.optim import FusedSGD
    torch.manual_seed(4)
    a = torch.nn.Linear(33, 17).to(DEV)
    b = torch.nn.Linear(33, 17).to(DEV)
    b.load_state_dict(a.state_dict())
    oa = FusedSGD(a.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    ob = torch.optim.SGD(b.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    before = [p.detach().clone() for p in a.parameters()]
    for p in a.parameters():
        p.grad = torch.full_like(p, float('nan'))
    a.weight.grad[0, 0] = float('inf')
    ctl = oa.step(max_norm=35, loss_scale=512.)
    assert float(ctl[2]) == 1.0 and all(torch.equal(x, y) for x, y in zip(a.parameters(), before))
    for it in range(3):
        for x, y in zip(a.parameters(), b.parameters()):
            g = torch.randn_like(x)
            x.grad, y.grad = g.clone(), g.clone()
        ctl = oa.step(max_norm=35)
        torch.nn.utils.clip_grad_norm_(list(b.parameters()), max_norm=35, norm_type=2)
        ob.step()
        assert float(ctl[2]) == 0.0
        for x, y in zip(a.parameters(), b.parameters()):
            assert torch.allclose(x, y, rtol=2e-6, atol=1e-7), it
            assert torch.allclose(oa.state[x]['momentum_buffer'], ob.state[y]['momentum_buffer'], rtol=2e-6, atol=1e-7)
    # without a loss scale nothing is skipped: the inf reaches the weights exactly where clip_grad_norm_ + SGD put it
    for x, y in zip(a.parameters(), b.parameters()):
        g = torch.randn_like(x)
        x.grad, y.grad = g.clone(), g.clone()
    a.weight.grad[1, 2] = float('inf')
    b.weight.grad[1, 2] = float('inf')
    ctl = oa.step(max_norm=35)
    torch.nn.utils.clip_grad_norm_(list(b.parameters()), max_norm=35, norm_type=2)
    ob.step()
    assert float(ctl[2]) == 0.0 and not torch.isfinite(a.weight[1, 2])
    for x, y in zip(a.parameters(), b.parameters()):
        assert torch.equal(torch.isfinite(x), torch.isfinite(y))
        fin = torch.isfinite(y)
        assert torch.allclose(x[fin], y[fin], rtol=2e-6, atol=1e-7)


def test_train_step_device_path_mixed_shapes_and_empty_gt():
    """a batch whose images differ in img_shape / pad_shape (validity flags, per-image clip border) and
    hold 0 ground truths in one image"""
    m = _model()
    img, metas, gts, gls = util.demo_inputs(3, 128, 192, seed=12)
    metas[1] = dict(metas[1], img_shape=(100, 150, 3), pad_shape=(128, 160, 3))
    gts[1] = torch.tensor([[10., 12., 90., 80.], [60., 20., 140., 95.]])
    gls[1] = torch.tensor([1, 3])
    gts[2], gls[2] = torch.empty(0, 4), torch.empty(0, dtype=torch.long)
    args = (img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
    vals = {}
    for mode in (True, False):
        m.device_train_path = mode
        torch.manual_seed(5)
        loss, log_vars = m._parse_losses(m.forward_train(*args))
        vals[mode] = dict(log_vars)
    for k in vals[True]:
        assert np.isclose(vals[True][k], vals[False][k], rtol=2e-4, atol=1e-6), (k, vals[True][k], vals[False][k])


@pytest.mark.parametrize('channels_last', [False, True])
def test_fused_sgd_matches_torch_sgd_with_clipping(channels_last):
    """FusedSGD (clip + SGD momentum / weight decay in one pass, per-group lr / decay) against torch.optim.SGD +
    clip_grad_norm_ over several steps; a non-finite gradient skips the step (the GradScaler rule) and loss-scaled
    gradients are unscaled inside; the conv operands it writes for the next step equal the per-layer packing"""
    from brcnn.optim import FusedSGD
    from brcnn import lib
    from brcnn.ops import _ptr, _stream
    torch.manual_seed(3)

    def make():
        net = torch.nn.Sequential(torch.nn.Conv2d(64, 128, 3, bias=True), torch.nn.BatchNorm2d(128), torch.nn.Conv2d(128, 64, 1),
                                  torch.nn.Linear(7, 5)).to(DEV)
        return net
    a, b = make(), make()
    b.load_state_dict(a.state_dict())
    groups = lambda n: [dict(params=[n[0].weight, n[2].weight]), dict(params=[n[0].bias, n[2].bias, n[3].bias], lr=0.04, weight_decay=0.0),  # noqa: E731
                        dict(params=[n[1].weight, n[1].bias, n[3].weight], weight_decay=0.0)]
    if channels_last:       # the 3x3 weight stored in the weight-gradient kernel's layout (blocks.conv_weights_channels_last)
        from brcnn import blocks
        assert blocks.conv_weights_channels_last(a) == 1 and not a[0].weight.is_contiguous()
    oa = FusedSGD(groups(a), lr=0.02, momentum=0.9, weight_decay=1e-4)
    ob = torch.optim.SGD(groups(b), lr=0.02, momentum=0.9, weight_decay=1e-4)
    assert oa.register_conv_weights(a, torch.bfloat16) == 2
    pa, pb = list(a.parameters()), list(b.parameters())
    gen = torch.Generator().manual_seed(5)
    for it in range(5):
        scale = 512.0 if it == 3 else 1.0
        for x, y in zip(pa, pb):
            g = torch.randn(x.shape, generator=gen).to(DEV) * (30.0 if it % 2 else 0.01)     # clipped / not clipped
            x.grad, y.grad = g * scale, g.clone()
        ctl = oa.step(max_norm=35, loss_scale=scale)
        gn = torch.nn.utils.clip_grad_norm_(pb, max_norm=35, norm_type=2)
        ob.step()
        assert torch.allclose(ctl[0], gn, rtol=1e-5) and float(ctl[2]) == 0.0
        for x, y in zip(pa, pb):
            assert torch.allclose(x, y, rtol=2e-6, atol=1e-7), it
        for x, y in zip(pa, pb):
            assert torch.allclose(oa.state[x]['momentum_buffer'], ob.state[y]['momentum_buffer'], rtol=2e-6, atol=1e-7)
    # packed operands of the (updated) conv weights == the per-layer packing kernel
    for conv in (a[0], a[2]):
        w = conv.weight
        ver, dt, f, d = w._brcnn_pack
        assert ver == w._version and dt == torch.bfloat16
        co, ci, kh, kw = w.shape
        f2 = torch.empty((co, kh, kw, ci), dtype=torch.bfloat16, device=DEV)
        d2 = torch.empty((ci, kh, kw, co), dtype=torch.bfloat16, device=DEV)
        st = lib.load().brcnn_pack_conv_weights(_ptr(w.detach().contiguous()), _ptr(f2), _ptr(d2), co, ci, kh, kw, 1, _stream())
        assert st == 0 and torch.equal(f, f2) and torch.equal(d, d2)
        assert torch.equal(f.float(), w.detach().permute(0, 2, 3, 1).to(torch.bfloat16).float())
    # a non-finite gradient: nothing moves, the packed operands stay valid
    before = [x.detach().clone() for x in pa]
    for x in pa:
        x.grad = torch.ones_like(x)
    pa[0].grad[0, 0, 0, 0] = float('inf')
    ctl = oa.step(max_norm=35, skip_nonfinite=True)
    assert float(ctl[2]) == 1.0 and all(torch.equal(x, y) for x, y in zip(pa, before))
    assert set(oa.state_dict()['state'][0].keys()) == {'momentum_buffer'}


@pytest.mark.parametrize('dtype,early', [('bf16', True), ('bf16', False), ('f16', True)])
def test_graphed_trunk_trains_like_the_eager_trunk(dtype, early):
    """`graph_trunk`: backbone + neck, forward and backward, replayed from two HIP graphs (brcnn/graphs.py) against the
    eager launches of the same kernels -- two models from the same seed, FusedSGD + clipping, five steps each on the
    same inputs and sampler seeds: the losses of every step and every parameter after the last one agree (16-bit conv
    stacks: bit for bit -- the slab sums of their weight gradients are fixed-order and the chained stream-K schedule,
    off under capture, only moves work between workgroups; fp32: up to the order of the weight-gradient atomics).  The
    first step of the graphed model runs eagerly (a key is captured the second time it comes up), the capture happens
    once, steps 2..5 replay it; the early RPN backward injects its pyramid gradients into the replayed backward."""
    from brcnn import blocks
    from brcnn.optim import FusedSGD
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10)
    args = (img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
    scale = 512.0 if dtype == 'f16' else 1.0
    out = {}
    try:
        for graphed in (False, True):
            m = _model()
            blocks.conv_weights_channels_last(m)
            m.set_compute_dtype(dtype)
            opt = FusedSGD([p for p in m.parameters() if p.requires_grad], lr=2e-3, momentum=0.9, weight_decay=1e-4)
            opt.register_conv_weights(m, blocks.compute_dtype())
            m.graph_trunk = graphed
            m.early_rpn_backward, m.early_backward_scale = early, scale
            losses = []
            for it in range(5):
                opt.zero_grad(set_to_none=True)
                torch.manual_seed(100 + it)
                loss, log_vars = m._parse_losses(m.forward_train(*args))
                (loss * scale).backward()
                opt.step(max_norm=35, loss_scale=scale)
                losses.append(dict(log_vars))
            torch.cuda.synchronize()
            out[graphed] = (losses, {k: p.detach().clone() for k, p in m.named_parameters()})
            if graphed:
                gt = m.__dict__['_graphed_trunk']
                assert gt.disabled_reason is None and gt.captures == 1, (gt.disabled_reason, gt.captures)
                assert sum(c.replays for c in gt.caps.values()) == 4
            else:
                assert '_graphed_trunk' not in m.__dict__
    finally:
        blocks.set_compute_dtype('f32')
    for it, (a, b) in enumerate(zip(out[False][0], out[True][0])):
        for k in a:
            if dtype == 'f32':
                assert abs(a[k] - b[k]) <= 2e-4 * max(1.0, abs(a[k])), (it, k, a[k], b[k])     # (atomics order, fed back through five SGD steps)
            else:
                assert a[k] == b[k], (it, k, a[k], b[k])
    for k, pa in out[False][1].items():
        pb = out[True][1][k]
        if dtype == 'f32':
            assert (pa - pb).abs().max().item() <= 1e-4 * (pa.abs().max().item() + 1e-12), k
        else:
            assert torch.equal(pa, pb), (k, (pa - pb).abs().max().item())


def test_graphed_trunk_fp32_step_equals_the_eager_step():
    """fp32: the weight-gradient kernel accumulates with atomics, so two runs of the SAME step differ by ~2e-7 of a
    gradient's largest entry, and a trajectory of SGD steps over random weights amplifies that through the proposal
    stage's thresholds -- the comparison is per step instead: from the same parameters, inputs and sampler seed the step
    with the replayed trunk gives the eager step's losses (1e-6) and gradients (2e-5 of the largest entry), three
    replays in a row"""
    m = _model()
    from brcnn import blocks
    blocks.conv_weights_channels_last(m)
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10)
    args = (img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
    m.early_rpn_backward = True

    def step(graphed):
        m.graph_trunk = graphed
        m.zero_grad(set_to_none=True)
        from brcnn import autograd as A
        A.grad_arena.new_step()
        torch.manual_seed(77)
        loss, log_vars = m._parse_losses(m.forward_train(*args))
        loss.backward()
        A.join_side_streams()
        torch.cuda.synchronize()
        return dict(log_vars), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    try:
        ref = step(False)
        step(True)                      # first sight of the key: still eager
        for _ in range(3):
            cur = step(True)
            for k, v in ref[0].items():
                assert abs(v - cur[0][k]) <= 1e-6 * max(1.0, abs(v)), (k, v, cur[0][k])
            assert cur[1].keys() == ref[1].keys()
            for k, g in ref[1].items():
                d = (cur[1][k] - g).abs().max().item()
                assert d <= 2e-5 * (g.abs().max().item() + 1e-12), (k, d)
        gt = m.__dict__['_graphed_trunk']
        assert gt.captures == 1 and sum(c.replays for c in gt.caps.values()) == 3 and gt.disabled_reason is None
    finally:
        m.graph_trunk, m.early_rpn_backward = False, False


def test_graphed_trunk_keys_per_input_shape_and_falls_back():
    """`graph_trunk` bookkeeping: a capture per input shape, taken the second time the shape comes up (a shape seen once
    never pays for one), kept for the shapes that recur; a parameter whose storage moves (here: a weight re-laid out
    channels-last after the capture) changes the key, so the stale graphs are not replayed; `BRCNN_GRAPH_TRUNK=0`-style
    veto (`graphs.ENABLED`) and no-grad mode run the eager trunk"""
    from brcnn import blocks, graphs
    m = _model()
    m.set_compute_dtype('bf16')
    try:
        m.graph_trunk = True
        data = {}
        for h, w in ((128, 192), (160, 160)):
            img, metas, gts, gls = util.demo_inputs(2, h, w, seed=h)
            data[(h, w)] = (img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])

        def step(key):
            m.zero_grad(set_to_none=True)
            torch.manual_seed(5)
            loss, log_vars = m._parse_losses(m.forward_train(*data[key]))
            loss.backward()
            torch.cuda.synchronize()
            return dict(log_vars)
        a, b = (128, 192), (160, 160)
        first = {k: step(k) for k in (a, b)}                     # seen once each: eager
        gt = m.__dict__['_graphed_trunk']
        assert gt.captures == 0
        for k in (a, b, a, b):
            assert step(k) == first[k]                           # bf16: replay = eager, bit for bit
        assert gt.captures == 2 and sorted(c.replays for c in gt.caps.values()) == [2, 2]
        with torch.no_grad():                                    # no gradients wanted: the eager trunk
            assert not gt.usable(data[a][0])
        saved, graphs.ENABLED = graphs.ENABLED, False
        try:
            assert step(a) == first[a] and gt.captures == 2
        finally:
            graphs.ENABLED = saved
        assert blocks.conv_weights_channels_last(m) > 0          # parameter storage moved: new keys
        step(a)
        assert gt.captures == 2                                  # first sight of the new key: eager
        step(a)
        assert gt.captures == 3
    finally:
        m.graph_trunk = False
        blocks.set_compute_dtype('f32')
