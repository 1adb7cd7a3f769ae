"""-m gpu: the device path of the host modules against the golden fixtures produced by the
imported reference (tests/golden/make_golden.py): RPN proposal stage, box-head post-processing
and the whole R50-PAFPN detector with seeded synthetic weights."""
import json
import os

import numpy as np
import pytest
import torch

import brcnn  # noqa: F401
from brcnn import Config, build_detector, ops
from tests import util
from tests.test_host_cpu import CFG, T, load

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _match_dets(got, ref, box_tol=1e-2, score_tol=1e-3):
    """fraction of reference detections that have a counterpart (same place, same score)"""
    if len(ref) == 0:
        return 1.0 if len(got) == 0 else 0.0
    if len(got) == 0:
        return 0.0
    d = np.abs(ref[:, None, :4] - got[None, :, :4]).max(-1)
    s = np.abs(ref[:, None, 4] - got[None, :, 4])
    ok = ((d < box_tol) & (s < score_tol)).any(1)
    return ok.mean()


def _close(a, b, tol=2e-4):
    """max |a-b| relative to the magnitude of the reference tensor (fp32 accumulation error of
    a K~2304 dot product scales with the tensor's magnitude, not with each element's)"""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return (a - b).abs().max().item() <= tol * max(b.abs().max().item(), 1e-6)


def _canon(p):
    """rows with EXACTLY equal scores may come in either order (the reference's sort is not
    stable, ours breaks ties by index): order such runs by coordinates before comparing"""
    p = p.clone()
    i = 0
    while i < len(p):
        j = i + 1
        while j < len(p) and p[j, 4] == p[i, 4]:
            j += 1
        if j - i > 1:
            blk = p[i:j]
            key = blk[:, 0] * 1e6 + blk[:, 1]
            p[i:j] = blk[torch.argsort(key)]
        i = j
    return p


def test_rpn_get_bboxes_golden():
    g = load('g5_rpn_get_bboxes')
    cfg = Config.fromfile(CFG)
    c = cfg.model.rpn_head.copy()
    c.update(train_cfg=cfg.model.train_cfg.rpn, test_cfg=cfg.model.test_cfg.rpn)
    head = brcnn.build_head(c).to(DEV)
    cls = [T(g[f'cls{i}']).to(DEV) for i in range(5)]
    reg = [T(g[f'reg{i}']).to(DEV) for i in range(5)]
    iou = [T(g[f'iou{i}']).to(DEV) for i in range(5)]
    metas = [dict(img_shape=(320, 509, 3), scale_factor=np.ones(4, np.float32), pad_shape=(320, 512, 3))
             for _ in range(2)]
    from brcnn.config import ConfigDict
    for name in ('test', 'train', 'small'):
        pc = ConfigDict(json.loads(str(g[name + '_cfg'])))
        res = head.get_bboxes(cls, reg, iou, metas, cfg=pc)
        for b in range(2):
            ref = _canon(T(g[f'{name}_props{b}']))
            got = _canon(res[b].cpu())
            assert got.shape == ref.shape, (name, got.shape, ref.shape)
            # identical proposals in identical order (up to exact score ties); device expf may differ from the host's
            # by an ulp inside sigmoid / exp(dw): allow 1e-4 px and 1e-6 score
            assert torch.allclose(got[:, :4], ref[:, :4], rtol=0, atol=2e-4), (name, b)
            assert torch.allclose(got[:, 4], ref[:, 4], rtol=0, atol=1e-6), (name, b)


def test_box_head_postprocess_golden():
    g = load('g8_g9_test_head')
    cfg = Config.fromfile(CFG)
    rh = brcnn.build_head(dict(cfg.model.roi_head, train_cfg=None, test_cfg=cfg.model.test_cfg.rcnn)).to(DEV)
    fused = rh.fuse_scores(T(g['cls_score']).to(DEV), T(g['prior']).to(DEV))
    assert torch.allclose(fused.cpu(), T(g['fused']), atol=1e-6)
    sf = np.array([1.1, 1.2, 1.1, 1.2], np.float32)
    # use the golden fused scores so the NMS input is bit-identical
    det, lab = rh.bbox_head.get_bboxes(T(g['rois']).to(DEV), T(g['fused']).to(DEV), T(g['bbox_pred']).to(DEV),
                                       (800, 1333, 3), sf, rescale=True, cfg=cfg.model.test_cfg.rcnn)
    assert torch.equal(lab.cpu(), T(g['lab']))
    assert torch.allclose(det.cpu(), T(g['det']), atol=2e-4)
    lv = rh.bbox_roi_extractor.map_roi_levels(T(g['r10k']).to(DEV), 5)
    assert (lv.cpu() != T(g['lvls10k'])).sum().item() == 0


@pytest.fixture(scope='module')
def model():
    cfg = Config.fromfile(CFG)
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=10))
    return m.to(DEV).eval()


def test_model_stages_golden(model):
    g = load('g10_model')
    img, metas, _, _ = util.demo_inputs(2, 128, 192, seed=10)
    with torch.no_grad():
        c = model.backbone(img.to(DEV))
        p = model.neck(c)
        cls, reg, iou = model.rpn_head(p)
    for i, t in enumerate(c):
        t = t.cpu()
        stat = np.array([t.double().mean(), t.double().std(), t.double().abs().max()])
        assert np.allclose(stat, g[f'c{i}_stat'], rtol=1e-4), (i, stat, g[f'c{i}_stat'])
        assert _close(t[:, :8, :4, :4], T(g[f'c{i}_slice']))
    for i, t in enumerate(p):
        t = t.cpu()
        assert _close(t[:, :16], T(g[f'p{i}'])), i
        assert _close(t.double().sum((2, 3)), T(g[f'p{i}_sum']))
    for i in range(5):
        assert _close(cls[i], T(g[f'cls{i}'])), i
        assert _close(reg[i], T(g[f'reg{i}'])), i
        assert _close(iou[i], T(g[f'iou{i}'])), i
    # second stage on the GOLDEN proposals: RoI features + FC head
    props = [T(g[f'props{b}']).to(DEV) for b in range(2)]
    rois = torch.cat([torch.cat([torch.full((len(q), 1), float(b), device=DEV), q[:, :4]], 1)
                      for b, q in enumerate(props)])
    with torch.no_grad():
        feats = model.roi_head.bbox_roi_extractor(p, rois)
        cs, bp = model.roi_head.bbox_head(feats)
    assert _close(feats.double().sum((2, 3)), T(g['roi_feats_sum']))
    assert _close(cs, T(g['cls_score']))
    assert _close(bp, T(g['bbox_pred']))


def test_model_end_to_end_golden(model):
    g = load('g10_model')
    img, metas, _, _ = util.demo_inputs(2, 128, 192, seed=10)
    with torch.no_grad():
        res = model(return_loss=False, rescale=True, img=[img.to(DEV)], img_metas=[metas])
        # the reference-signature path (per-image python flow) must agree with the device path
        x = model.extract_feat(img.to(DEV))
        pl = model.rpn_head.simple_test_rpn(x, metas)
        res2 = model.roi_head.simple_test(x, pl, metas, rescale=True)
    n_ref = sum(len(g[f'res{b}_{c}']) for b in range(2) for c in range(4))
    assert n_ref > 20
    for b in range(2):
        # RPN proposals (score-ordered): the bulk must coincide with the reference's
        ref_p, got_p = g[f'props{b}'], pl[b].cpu().numpy()
        assert _match_dets(got_p, ref_p, 5e-2, 1e-3) > 0.97
        for c in range(4):
            ref = g[f'res{b}_{c}']
            assert res[b][c].dtype == np.float32 and res[b][c].shape[1] == 5
            assert _match_dets(res[b][c], ref) >= 0.99, (b, c, len(ref), len(res[b][c]))
            assert _match_dets(ref, res[b][c]) >= 0.99
            assert _match_dets(res[b][c], res2[b][c]) >= 0.95 and _match_dets(res2[b][c], res[b][c]) >= 0.95


def test_fused_rpn_head_equals_per_level(model):
    img, metas, _, _ = util.demo_inputs(2, 128, 192, seed=10)
    with torch.no_grad():
        feats = model.extract_feat_nhwc(img.to(DEV))
        cls, reg, iou = model.rpn_head.forward_nhwc(list(feats))
        fc, fr, fi = model.rpn_head.split_fused(model.rpn_head.forward_fused(list(feats)))
    for l in range(5):
        s = float(model.rpn_head.scales[l].scale.detach())
        assert torch.allclose(fc[l], cls[l], rtol=1e-5, atol=1e-5)
        assert torch.allclose(fi[l], iou[l], rtol=1e-5, atol=1e-5)
        assert torch.allclose(fr[l] * s, reg[l], rtol=1e-5, atol=1e-5)


def test_train_step_losses_golden():
    """forward_train of the whole detector on the device (HIP conv fwd, focal loss, RoIAlign)
    against the reference's CPU losses on the same seeded weights / inputs / sampler seed, then a
    backward pass through the HIP dgrad / wgrad kernels."""
    g = load('g10_train_losses')
    cfg = Config.fromfile(CFG)
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=10))
    m = m.to(DEV).train()
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10)
    torch.manual_seed(77)
    losses = m.forward_train(img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
    for k, ref in g.items():
        # the device path returns the sum over the pyramid levels of the RPN terms as a one-element
        # list (the reference lists them per level and _parse_losses adds them up)
        got = torch.stack([v.reshape(()) for v in losses[k]]).sum() if isinstance(losses[k], list) else losses[k]
        want = T(ref).float().sum() if isinstance(losses[k], list) else T(ref).float()
        assert torch.allclose(got.detach().cpu().float().reshape(want.shape), want, rtol=2e-3, atol=1e-4), \
            (k, got.detach().cpu(), ref)
    per_level = m.rpn_head.last_rpn_targets[1].cpu()
    for r, k in enumerate(('loss_rpn_cls', 'loss_rpn_bbox', 'loss_rpn_iou')):
        assert torch.allclose(per_level[r], T(g[k]).float(), rtol=2e-3, atol=1e-4), (k, per_level[r], g[k])
    out = m.train_step(dict(img=img.to(DEV), img_metas=metas, gt_bboxes=[b.to(DEV) for b in gts],
                            gt_labels=[l.to(DEV) for l in gls]), None)
    out['loss'].backward()
    assert out['num_samples'] == 2 and 'loss_rpn_cls' in out['log_vars']
    n_grad = sum(1 for p in m.parameters() if p.requires_grad and p.grad is not None)
    n_train = sum(1 for p in m.parameters() if p.requires_grad)
    assert n_grad == n_train, (n_grad, n_train)
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    assert m.backbone.conv1.weight.grad is None and m.backbone.layer1[0].conv1.weight.grad is None


@pytest.mark.parametrize('cfg_name', ['boosting_rcnn_r101_pafpn_softnms_coco.py',
                                      'boosting_rcnn_r50_pafpn_1x_coco.py',
                                      'boosting_rcnn_r50_pafpn_1x_voc.py'])
def test_other_configs_device_vs_cpu_oracle_pipeline(cfg_name):
    """BASELINE configs #3/#5 at test time: COCO heads (80 classes) and the ResNet-101 +
    soft-NMS recipe (2000 proposals, score_thr 1e-4) -- device path against the CPU oracle
    pipeline (same module graph, PyTorch-CPU convs + C oracle RoIAlign/NMS/soft-NMS)."""
    import os
    from oracle import cpu_pipeline
    cfg = Config.fromfile(os.path.join(os.path.dirname(CFG), cfg_name))
    img, metas, _, _ = util.demo_inputs(2, 128, 192, seed=3)
    with cpu_pipeline.patched():
        m = build_detector(cfg.model)
        sd = util.seeded_state_dict(m, seed=5)
        m.load_state_dict(sd)
        m.eval()
        with torch.no_grad():
            ref = m(return_loss=False, rescale=True, img=[img], img_metas=[[dict(x) for x in metas]])
    m = build_detector(cfg.model)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    with torch.no_grad():
        got = m(return_loss=False, rescale=True, img=[img.to(DEV)], img_metas=[metas])
    n_ref = sum(len(r) for b in ref for r in b)
    assert n_ref > 10
    hit = tot = 0
    for b in range(2):
        for c in range(len(ref[b])):
            r, g_ = ref[b][c], got[b][c]
            tot += len(r)
            if len(r) and len(g_):
                d = np.abs(r[:, None, :4] - g_[None, :, :4]).max(-1)
                s = np.abs(r[:, None, 4] - g_[None, :, 4])
                hit += int(((d < 1e-2) & (s < 1e-3)).any(1).sum())
    assert hit >= 0.99 * tot, (hit, tot)


def test_fpn_ciou_config_golden():
    """the FPN / CIoU / reg_decoded_bbox=False boosting config against the
    reference's own train losses and detections (golden g11)."""
    import os
    g = load('g11_fpn_config')
    cfg = Config.fromfile(os.path.join(os.path.dirname(CFG), 'boosting_rcnn_r50_fpn_1x_coco.py'))
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=11))
    m = m.to(DEV)
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, num_classes=80, seed=11)
    m.eval()
    with torch.no_grad():
        res = m(return_loss=False, rescale=True, img=[img.to(DEV)], img_metas=[metas])
    for b in range(2):
        got = np.concatenate([np.concatenate([r, np.full((len(r), 1), c, np.float32)], 1)
                              for c, r in enumerate(res[b])], 0)
        ref = g[f'det{b}']
        assert len(ref) > 5
        d = np.abs(ref[:, None, :4] - got[None, :, :4]).max(-1)
        s = np.abs(ref[:, None, 4] - got[None, :, 4])
        same_cls = ref[:, None, 5] == got[None, :, 5]
        assert ((d < 1e-2) & (s < 1e-3) & same_cls).any(1).mean() >= 0.99
    m.train()
    torch.manual_seed(78)
    losses = m.forward_train(img.to(DEV), metas, [x.to(DEV) for x in gts], [x.to(DEV) for x in gls])
    # the device-resident step returns each RPN loss as the sum over the levels and keeps the per-level terms
    # (the reference's list) in rpn_head.last_rpn_targets: reg_decoded_bbox=False + CIoULoss on the fused kernels
    per_level = m.rpn_head.last_rpn_targets[1].cpu()
    for r, k in enumerate(('loss_rpn_cls', 'loss_rpn_bbox', 'loss_rpn_iou')):
        assert torch.allclose(per_level[r], T(g['loss_' + k]).float(), rtol=2e-3, atol=1e-4), (k, per_level[r])
    for k, v in losses.items():
        got = torch.stack(v) if isinstance(v, list) else v
        ref = T(g['loss_' + k]).float()
        if k.startswith('loss_rpn') and got.numel() == 1:
            ref = ref.sum().reshape(1)
        # the R-CNN terms depend on WHICH 512 proposals the seeded RandomSampler draws: one
        # proposal flipping across the 0.6 IoU threshold (fp32 round-off) shifts them by ~1/1024
        tol = 1e-2 if k in ('loss_cls', 'loss_bbox', 'acc') else 2e-3
        assert torch.allclose(got.detach().cpu().float().reshape(ref.shape), ref, rtol=tol, atol=1e-4), k


def test_device_path_edge_cases(model):
    """empty / ragged situations the reference tests in test_forward.py:260-340: no proposal
    survives, different image shapes in one batch, batch of one."""
    img, metas, _, _ = util.demo_inputs(2, 128, 192, seed=21)
    # (a) images of different valid shapes in one padded batch -> per-image clip borders
    metas[1] = dict(metas[1], img_shape=(100, 150, 3), ori_shape=(100, 150, 3))
    with torch.no_grad():
        res = model(return_loss=False, rescale=True, img=[img.to(DEV)], img_metas=[metas])
        x = model.extract_feat(img.to(DEV))
        ref = model.roi_head.simple_test(x, model.rpn_head.simple_test_rpn(x, metas), metas, rescale=True)
    for b in range(2):
        for c in range(4):
            assert _match_dets(res[b][c], ref[b][c]) >= 0.95 and _match_dets(ref[b][c], res[b][c]) >= 0.95
            if len(res[b][c]):
                lim = np.array(metas[b]['img_shape'][:2][::-1] * 2) / metas[b]['scale_factor']
                assert (res[b][c][:, :4] <= lim + 1e-3).all()
    # (b) batch of one
    with torch.no_grad():
        r1 = model(return_loss=False, rescale=True, img=[img[:1].to(DEV)], img_metas=[[metas[0]]])
    assert len(r1) == 1 and len(r1[0]) == 4
    # (c) nothing passes the score threshold -> empty (0,5) float32 arrays per class
    old = model.roi_head.test_cfg.score_thr
    model.roi_head.test_cfg.score_thr = 2.0
    try:
        with torch.no_grad():
            r0 = model(return_loss=False, rescale=True, img=[img.to(DEV)], img_metas=[metas])
    finally:
        model.roi_head.test_cfg.score_thr = old
    assert all(a.shape == (0, 5) and a.dtype == np.float32 for b in r0 for a in b)
    # (d) no proposals at all: min_bbox_size larger than any box
    from brcnn.config import ConfigDict
    cfg = ConfigDict(dict(model.rpn_head.test_cfg.to_dict(), min_bbox_size=1e6))
    with torch.no_grad():
        cls, reg, iou = model.rpn_head(model.extract_feat(img.to(DEV)))
        props = model.rpn_head.get_bboxes(cls, reg, iou, metas, cfg=cfg)
    assert all(p.shape == (0, 5) for p in props)
    out = model.roi_head.simple_test(x, props, metas, rescale=True)
    assert all(a.shape == (0, 5) for b in out for a in b)


def test_train_step_empty_gt_image():
    """an image without ground truth in the batch (test_forward.py empty-GT case)"""
    cfg = Config.fromfile(CFG)
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=10))
    m = m.to(DEV).train()
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10)
    gts[1], gls[1] = gts[1][:0], gls[1][:0]
    torch.manual_seed(1)
    losses = m.forward_train(img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
    loss, log_vars = m._parse_losses(losses)
    assert torch.isfinite(loss) and loss.item() > 0
    loss.backward()


def test_voc_box_head_golden_and_train_step():
    """VOC recipe: ConvFC box head with 2 cls FCs + 4 GroupNorm reg convs on
    the HIP path against the reference's own forward (golden g15), and one train step of the
    whole VOC detector (VarifocalLoss RPN) with finite gradients on every trainable tensor."""
    import os
    g = load('g15_voc')
    head = brcnn.build_head(json.loads(str(g['head_cfg'])))
    head.load_state_dict(util.seeded_state_dict(head, seed=15))
    head = head.to(DEV).eval()
    feats = torch.randn(12, 256, 7, 7, generator=torch.Generator().manual_seed(151)).to(DEV)
    with torch.no_grad():
        cs, bp = head(feats)
    assert _close(cs, T(g['head_cls'])) and _close(bp, T(g['head_reg']))
    cfg = Config.fromfile(os.path.join(os.path.dirname(CFG), 'boosting_rcnn_r50_pafpn_1x_voc.py'))
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=4))
    m = m.to(DEV).train()
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, num_classes=20, seed=4)
    torch.manual_seed(3)
    losses = m.forward_train(img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
    loss, log_vars = m._parse_losses(losses)
    assert np.isfinite(log_vars['loss']) and 'loss_rpn_cls' in log_vars
    loss.backward()
    n = 0
    for k, p in m.named_parameters():
        if p.requires_grad and not k.startswith(('backbone.conv1', 'backbone.bn1', 'backbone.layer1')):
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
            n += 1
    assert n > 100


def test_grouped_conv_and_resnext_golden():
    """ResNeXt (x101 recipe): the grouped 3x3 convolution on the MFMA kernel
    (block-diagonal 64-channel tiles) against torch's grouped conv in fp64, and the whole
    ResNeXt-50 32x4d backbone against the reference's own forward (golden g16)."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(8)
    for (n, c, h, w, groups, stride, res) in [(2, 128, 20, 30, 32, 1, False), (1, 256, 17, 23, 32, 2, False),
                                              (2, 256, 9, 14, 64, 1, True), (1, 1024, 7, 9, 32, 1, False),
                                              (1, 512, 10, 12, 64, 2, True)]:
        x = torch.randn(n, c, h, w, generator=gen)
        wt = torch.randn(c, c // groups, 3, 3, generator=gen) / np.sqrt(9 * c / groups)
        sc, sh = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen)
        ref = F.conv2d(x.double(), wt.double(), None, stride, 1, groups=groups) * sc.double().view(1, -1, 1, 1) + \
            sh.double().view(1, -1, 1, 1)
        r = torch.randn(ref.shape, generator=gen) if res else None
        if res:
            ref = ref + r.double()
        ref = ref.relu()
        wp, window = ops.pack_grouped_weight(wt.to(DEV), groups)
        assert window == 64 * (c // groups) // (c // groups) and wp.shape == (c, 3, 3, window)
        y = ops.conv2d_nhwc_grouped(x.permute(0, 2, 3, 1).contiguous().to(DEV), wp, window, sc.to(DEV), sh.to(DEV),
                                    r.permute(0, 2, 3, 1).contiguous().to(DEV) if res else None, True, stride, 1)
        y = y.permute(0, 3, 1, 2).cpu().double()
        assert y.shape == ref.shape and (y - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    g = load('g16_resnext_autoaug')
    m = brcnn.build_backbone(json.loads(str(g['backbone_cfg'])))
    assert sorted(m.state_dict().keys()) == g['keys'].tolist()
    m.load_state_dict(util.seeded_state_dict(m, seed=16))
    m = m.to(DEV).eval()
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(161)).to(DEV)
    with torch.no_grad():
        outs = m(x)
    for i, t in enumerate(outs):
        assert _close(t[:, :16, :4, :6], T(g[f'c{i}_slice']))
        assert _close(t.double().sum((2, 3)), T(g[f'c{i}_sum']), tol=1e-3)
    # the x101 recipe builds and runs end to end on the device
    import os
    cfg = Config.fromfile(os.path.join(os.path.dirname(CFG), 'boosting_rcnn_x101_pafpn_mstrain_3x_coco.py'))
    det = build_detector(cfg.model)
    assert type(det.backbone).__name__ == 'ResNeXt' and det.backbone.layer3[0].conv2.groups == 64
    det.load_state_dict(util.seeded_state_dict(det, seed=2))
    det = det.to(DEV).eval()
    img, metas, _, _ = util.demo_inputs(1, 128, 192, seed=2)
    with torch.no_grad():
        res = det(return_loss=False, rescale=True, img=[img.to(DEV)], img_metas=[metas])
    assert len(res) == 1 and len(res[0]) == 80 and all(r.shape[1] == 5 for r in res[0])
    # bf16 mode (the recipe's `fp16` key maps to it): 1x1 convs on bf16 MFMA, grouped 3x3 widened to fp32
    from brcnn import blocks
    try:
        with torch.no_grad():
            f32 = [f.float() for f in det.extract_feat_nhwc(img.to(DEV))]
            det.set_compute_dtype('bf16')
            f16 = det.extract_feat_nhwc(img.to(DEV))
        assert all(f.dtype == torch.bfloat16 for f in f16)
        for a, b in zip(f16, f32):
            assert (a.float() - b).abs().max().item() < 0.08 * b.abs().max().item()
    finally:
        blocks.set_compute_dtype('f32')


def _deform_ref(x, om, w, stride, pad):
    """modulated deformable 3x3 conv in fp64 torch, straight from the published algorithm
    (mmcv modulated_deform_conv: taps displaced by (dy, dx), zero outside (-1, H) x (-1, W),
    bilinear with per-corner validity, times sigmoid(mask))"""
    n, c, h, wd = x.shape
    ho, wo = om.shape[2], om.shape[3]
    x, om, w = x.double(), om.double(), w.double()
    out = torch.zeros(n, w.shape[0], ho, wo, dtype=torch.float64)
    ys = (torch.arange(ho) * stride - pad).view(1, ho, 1).double()
    xs = (torch.arange(wo) * stride - pad).view(1, 1, wo).double()
    for t in range(9):
        i, j = t // 3, t % 3
        hy = ys + i + om[:, 2 * t]
        wx = xs + j + om[:, 2 * t + 1]
        mask = torch.sigmoid(om[:, 18 + t])
        inside = (hy > -1) & (wx > -1) & (hy < h) & (wx < wd)
        hl, wl = torch.floor(hy), torch.floor(wx)
        lh, lw = hy - hl, wx - wl
        val = torch.zeros(n, c, ho, wo, dtype=torch.float64)
        for (hh, ww, wt) in ((hl, wl, (1 - lh) * (1 - lw)), (hl, wl + 1, (1 - lh) * lw),
                             (hl + 1, wl, lh * (1 - lw)), (hl + 1, wl + 1, lh * lw)):
            ok = inside & (hh >= 0) & (hh <= h - 1) & (ww >= 0) & (ww <= wd - 1)
            hi, wi = hh.clamp(0, h - 1).long(), ww.clamp(0, wd - 1).long()
            g = x[torch.arange(n).view(n, 1, 1), :, hi, wi].permute(0, 3, 1, 2)      # (n,c,ho,wo)
            val += g * (wt * ok).unsqueeze(1)
        val = val * mask.unsqueeze(1)
        out += torch.einsum('nchw,oc->nohw', val, w[:, :, i, j])
    return out


def test_avgpool_deform_res2net_golden():
    """Res2Net / DCNv2 rows: AvgPool2d variants against torch, the modulated
    deformable conv (HIP im2col + MFMA GEMM) against an fp64 restatement of the published
    algorithm, the Res2Net-50 backbone against the reference's own forward (golden g17), and the
    r2_101 DCN recipe end to end."""
    import os
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(12)
    for (h, w, k, s, p, ceil, cip) in [(36, 50, 3, 2, 1, False, True), (36, 50, 2, 2, 0, True, False),
                                       (37, 51, 2, 2, 0, True, False), (9, 13, 3, 1, 1, False, True),
                                       (18, 25, 3, 2, 1, False, True)]:
        x = torch.randn(2, 32, h, w, generator=gen)
        ref = F.avg_pool2d(x, k, s, p, ceil_mode=ceil, count_include_pad=cip)
        y = ops.avgpool_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), k, s, p, ceil, cip).permute(0, 3, 1, 2).cpu()
        assert y.shape == ref.shape and torch.allclose(y, ref, rtol=1e-6, atol=1e-6), (h, w, k, s)
    for (c, h, w, stride) in [(32, 14, 19, 1), (64, 15, 22, 2), (128, 9, 11, 1)]:
        x = torch.randn(2, c, h, w, generator=gen)
        ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
        om = torch.randn(2, 27, ho, wo, generator=gen) * 2.0
        om[:, :18] += (torch.rand(2, 18, ho, wo, generator=gen) > 0.9).float() * 30      # some taps far outside
        wt = torch.randn(c, c, 3, 3, generator=gen) / np.sqrt(9 * c)
        ref = _deform_ref(x, om, wt, stride, 1)
        col, (ho2, wo2) = ops.deform_im2col_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV),
                                                 om.permute(0, 2, 3, 1).contiguous().to(DEV), 3, stride, 1, 1)
        assert (ho2, wo2) == (ho, wo)
        wp = wt.permute(0, 2, 3, 1).reshape(c, 1, 1, 9 * c).contiguous().to(DEV)
        y = ops.conv2d_nhwc(col.view(2 * ho * wo, 1, 1, 9 * c), wp).view(2, ho, wo, c).permute(0, 3, 1, 2).cpu().double()
        assert (y - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item()), (c, stride)
    g = load('g17_res2net')
    m = brcnn.build_backbone(json.loads(str(g['backbone_cfg'])))
    assert sorted(m.state_dict().keys()) == g['keys'].tolist()
    m.load_state_dict(util.seeded_state_dict(m, seed=17))
    m = m.to(DEV).eval()
    x = torch.randn(2, 3, 72, 100, generator=torch.Generator().manual_seed(171)).to(DEV)
    with torch.no_grad():
        outs = m(x)
    for i, t in enumerate(outs):
        assert list(t.shape) == g[f'c{i}_shape'].tolist()
        assert _close(t[:, :16, :4, :6], T(g[f'c{i}_slice']))
        assert _close(t.double().sum((2, 3)), T(g[f'c{i}_sum']), tol=1e-3)
    # DCN recipe end to end; with the zero-initialised conv_offset a DCN conv is an ordinary conv
    # with mask 0.5, which the seeded (non-zero) offsets of this test then move away from
    cfg = Config.fromfile(os.path.join(os.path.dirname(CFG), 'boosting_rcnn_r2_101_dcn_pafpn_mstrain_3x_coco.py'))
    det = build_detector(cfg.model)
    blk = det.backbone.layer2[1]
    assert type(det.backbone).__name__ == 'Res2Net' and blk.with_dcn and not det.backbone.layer1[0].with_dcn
    det.load_state_dict(util.seeded_state_dict(det, seed=3))
    det = det.to(DEV).eval()
    img, metas, _, _ = util.demo_inputs(1, 128, 192, seed=3)
    with torch.no_grad():
        res = det(return_loss=False, rescale=True, img=[img.to(DEV)], img_metas=[metas])
        # one DCN conv of the network against the fp64 restatement on its real input
        xs = torch.randn(1, 40, 56, 64, generator=torch.Generator().manual_seed(9)).to(DEV)     # NHWC, wp = 64
        xs[..., 52:] = 0
        d = blk._packed()
        y = blk._conv_i(d, 0, xs)
        conv, bn = blk.convs[0], blk.bns[0]
        xr = xs[..., :52].permute(0, 3, 1, 2).cpu()
        omr = F.conv2d(xr, conv.conv_offset.weight.cpu(), conv.conv_offset.bias.cpu(), 1, 1)
        ref = _deform_ref(xr, omr, conv.weight.detach().cpu(), 1, 1)
        sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().cpu().double()
        ref = (ref * sc.view(1, -1, 1, 1) + (bn.bias - bn.running_mean * sc.float().to(DEV)).detach().cpu().double()
               .view(1, -1, 1, 1)).relu()
    assert len(res[0]) == 80
    got = y[..., :52].permute(0, 3, 1, 2).cpu().double()
    assert (got - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    assert float(y[..., 52:].abs().max()) == 0.0            # pad channels stay exactly zero


def test_x101_recipe_train_step():
    """the ResNeXt recipe trains: one step through the grouped-conv forward / dgrad / wgrad kernels"""
    import os
    cfg = Config.fromfile(os.path.join(os.path.dirname(CFG), 'boosting_rcnn_x101_pafpn_mstrain_3x_coco.py'))
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=6))
    m = m.to(DEV).train()
    img, metas, gts, gls = util.demo_inputs(1, 128, 192, num_classes=80, seed=6)
    torch.manual_seed(1)
    losses = m.forward_train(img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
    loss, log_vars = m._parse_losses(losses)
    loss.backward()
    assert np.isfinite(log_vars['loss'])
    g2 = m.backbone.layer3[5].conv2.weight.grad
    assert g2 is not None and g2.shape == (1024, 16, 3, 3) and torch.isfinite(g2).all() and g2.abs().max() > 0
    assert m.backbone.layer1[0].conv2.weight.grad is None          # frozen stage


def test_deform_conv_backward_and_res2net_train_step():
    """DCNv2 backward (HIP col2im: dx by atomics, offset / mask gradients by wave reductions) against
    torch autograd through the fp64 restatement, then one train step of the Res2Net-DCN recipe"""
    import os
    from brcnn.autograd import deform_im2col_autograd, linear_autograd
    gen = torch.Generator().manual_seed(14)
    for (c, h, w, stride) in [(32, 11, 13, 1), (64, 12, 17, 2)]:
        x = torch.randn(2, c, h, w, generator=gen)
        ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
        om = torch.randn(2, 27, ho, wo, generator=gen) * 1.5
        om[:, :18] += (torch.rand(2, 18, ho, wo, generator=gen) > 0.9).float() * 20
        wt = torch.randn(c, c, 3, 3, generator=gen) / np.sqrt(9 * c)
        xr, omr, wr = x.double().requires_grad_(), om.double().requires_grad_(), wt.double().requires_grad_()
        ref = _deform_ref(xr, omr, wr, stride, 1)
        go = torch.randn(ref.shape, generator=gen)
        ref.backward(go.double())
        xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_()
        og = om.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_()
        wg = wt.to(DEV).requires_grad_()
        col = deform_im2col_autograd(xg, og, stride, 1)
        y = linear_autograd(col, wg.permute(0, 2, 3, 1).reshape(c, 9 * c), None).view(2, ho, wo, c)
        y.backward(go.permute(0, 2, 3, 1).contiguous().to(DEV))
        rel = lambda a, b: (a - b).abs().max().item() / max(1.0, b.abs().max().item())   # noqa: E731
        assert rel(y.detach().permute(0, 3, 1, 2).cpu().double(), ref.detach()) < 3e-5
        assert rel(xg.grad.permute(0, 3, 1, 2).cpu().double(), xr.grad) < 1e-4
        assert rel(og.grad.permute(0, 3, 1, 2).cpu().double(), omr.grad) < 1e-4
        assert rel(wg.grad.cpu().double(), wr.grad) < 2e-4
    cfg = Config.fromfile(os.path.join(os.path.dirname(CFG), 'boosting_rcnn_r2_101_dcn_pafpn_mstrain_3x_coco.py'))
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=8))
    m = m.to(DEV).train()
    img, metas, gts, gls = util.demo_inputs(1, 128, 192, num_classes=80, seed=8)
    torch.manual_seed(2)
    losses = m.forward_train(img.to(DEV), metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls])
    loss, log_vars = m._parse_losses(losses)
    loss.backward()
    assert np.isfinite(log_vars['loss'])
    blk = m.backbone.layer3[4]
    for p in (blk.convs[1].weight, blk.convs[1].conv_offset.weight, blk.convs[1].conv_offset.bias, blk.conv1.weight,
              blk.conv3.weight, blk.bns[0].weight, m.backbone.layer2[0].downsample[1].weight):
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0
    assert m.backbone.layer1[0].conv1.weight.grad is None


@pytest.mark.parametrize('cfg_name', ['boosting_rcnn_r101_pafpn_softnms_coco.py', 'boosting_rcnn_r50_pafpn_1x_coco.py'])
def test_batched_postprocess_equals_per_image_reference_path(cfg_name):
    """the device-resident whole-batch second stage (one segmented NMS / soft-NMS launch over (image,
    class)) against the per-image restatement of the reference's simple_test_bboxes ->
    multiclass_nms -> mmcv batched_nms loop, on the same features and proposals"""
    import os
    cfg = Config.fromfile(os.path.join(os.path.dirname(CFG), cfg_name))
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=5))
    m = m.to(DEV).eval()
    img, metas, _, _ = util.demo_inputs(2, 128, 192, seed=3)
    assert m._device_path_ok()
    with torch.no_grad():
        fast = m.simple_test(img.to(DEV), metas, rescale=True)
        x = m.extract_feat(img.to(DEV))
        props = m.rpn_head.simple_test_rpn(x, metas)
        slow = m.roi_head.simple_test(x, props, metas, rescale=True)
    n = sum(len(c) for r in slow for c in r)
    assert n > 10 and sum(len(c) for r in fast for c in r) == n
    for b in range(2):
        for c in range(80):
            assert len(fast[b][c]) == len(slow[b][c]), (b, c)
            if len(slow[b][c]):
                assert _match_dets(fast[b][c], slow[b][c], 1e-3, 1e-5) == 1.0, (b, c)


# ---- remaining boosting variants: BoostRoIHead / DyProbRoIHead ---------------
def _variant_head(typ, num_classes, seed, **over):
    import copy
    from brcnn.registry import build_head
    cfg = Config.fromfile(CFG)
    rc = copy.deepcopy(cfg.model.roi_head)
    rc['type'] = typ
    rc['bbox_head']['num_classes'] = num_classes
    rc['train_cfg'] = copy.deepcopy(cfg.model.train_cfg.rcnn)
    rc['test_cfg'] = copy.deepcopy(cfg.model.test_cfg.rcnn)
    dyn = over.pop('dynamic_rcnn', None)
    if dyn:
        rc['train_cfg']['dynamic_rcnn'] = dyn
    lb = over.pop('loss_bbox', None)
    if lb:
        rc['bbox_head']['loss_bbox'] = lb
    rc.update(over)
    head = build_head(rc)
    head.load_state_dict(util.seeded_state_dict(head, seed=seed))
    return head.to(DEV)


def _dev_inputs(num_classes, cols, seed):
    feats, metas, gts, gls, props = util.variant_inputs(num_classes, cols, seed)
    return ([f.to(DEV) for f in feats], metas, [b.to(DEV) for b in gts], [l.to(DEV) for l in gls],
            [p.to(DEV) for p in props])


def _padded_proposals(props):
    B, K = len(props), max(len(p) for p in props)
    dets = torch.zeros((B, K, 5), dtype=torch.float32, device=DEV)
    for b, p in enumerate(props):
        dets[b, :len(p)] = p
    return dets, torch.tensor([len(p) for p in props], dtype=torch.int32, device=DEV)


def _loss_close(got, ref):
    return torch.allclose(got.detach().cpu().float().reshape(-1), T(ref).float().reshape(-1), rtol=2e-3, atol=1e-4)


@pytest.mark.parametrize('tag,over', [('q', dict(boost=True, quality=True, iou_gamma=0.5, gamma=0.5)),
                                      ('p', dict(boost=True, quality=False, gamma=0.5, alpha=0.75))])
def test_boost_roi_head_train_golden(tag, over):
    """BoostRoIHead.forward_train (prob_roi_head.py:290-359,438-468) on the device against the
    reference's CPU losses: per-class prior matrix gathered at the label, weights passed as label
    weights (incl. the reference's (n,n) broadcast when `quality` is on)"""
    g = load('g18_boost_variants')
    head = _variant_head('BoostRoIHead', 1, 18, **over).train()
    feats, metas, gts, gls, props = _dev_inputs(1, 1, 18)
    torch.manual_seed(5)
    losses = head.forward_train(feats, metas, props, gts, gls)
    for k in ('loss_cls', 'loss_bbox', 'acc'):
        assert _loss_close(losses[k], g[f'boost_{tag}_{k}']), (k, losses[k], g[f'boost_{tag}_{k}'])
    (losses['loss_cls'] + losses['loss_bbox']).backward()
    assert all(torch.isfinite(p.grad).all() for p in head.parameters() if p.grad is not None)


@pytest.mark.parametrize('smooth', [False, True])
def test_boost_roi_head_whole_batch_device_path_golden(smooth):
    """BoostRoIHead.forward_train_device (P = 1, the case the reference trains): the whole-batch kernels against the
    REFERENCE's losses (g18 'p': gamma 0.5, alpha 0.75, L1Loss) and, with a SmoothL1 box loss, against the per-image chain;
    gradients of both paths agree"""
    from brcnn import train_ops
    g = load('g18_boost_variants')
    over = dict(boost=True, quality=False, gamma=0.5, alpha=0.75)
    if smooth:
        over['loss_bbox'] = dict(type='SmoothL1Loss', beta=0.5, loss_weight=1.0)
    head = _variant_head('BoostRoIHead', 1, 18, **over).train()
    ref_head = _variant_head('BoostRoIHead', 1, 18, **over).train()
    assert head.device_train_ok()
    feats, metas, gts, gls, props = _dev_inputs(1, 1, 18)
    dets, num = _padded_proposals(props)
    torch.manual_seed(5)
    losses, _ = head.forward_train_device([f.permute(0, 2, 3, 1).contiguous() for f in feats], metas, dets, num,
                                          train_ops.flatten_gts(gts, gls))
    torch.manual_seed(5)
    ref = ref_head.forward_train(feats, metas, props, gts, gls)
    for k in ('loss_cls', 'loss_bbox', 'acc'):
        if not smooth:
            assert _loss_close(losses[k], g[f'boost_p_{k}']), (k, losses[k], g[f'boost_p_{k}'])
        assert torch.allclose(losses[k].reshape(-1), ref[k].reshape(-1), rtol=2e-3, atol=1e-4), (k, losses[k], ref[k])
    (losses['loss_cls'] + losses['loss_bbox']).backward()
    (ref['loss_cls'] + ref['loss_bbox']).backward()
    for (k, pa), (_, pb) in zip(head.named_parameters(), ref_head.named_parameters()):
        if pb.grad is None:
            continue
        scale = float(pb.grad.abs().max()) + 1e-12
        assert float((pa.grad - pb.grad).abs().max()) <= 2e-3 * scale, (k, float((pa.grad - pb.grad).abs().max()), scale)


def test_boost_roi_head_test_golden():
    """per-class test-time fusion sqrt(softmax * [s_0..s_{C-1}, 1]) (prob_roi_head.py:361-436)"""
    g = load('g18_boost_variants')
    head = _variant_head('BoostRoIHead', 3, 19, boost=True).eval()
    feats, metas, gts, gls, props = _dev_inputs(3, 3, 19)
    with torch.no_grad():
        det, lab = head.simple_test_bboxes(feats, metas, props, head.test_cfg, rescale=True)
    for b in range(2):
        ref, rl = g[f'boost_det{b}'], g[f'boost_lab{b}']
        got, gl = det[b].cpu().numpy(), lab[b].cpu().numpy()
        assert len(ref) > 5 and abs(len(got) - len(ref)) <= max(2, len(ref) // 20)
        for c in range(3):
            assert _match_dets(got[gl == c], ref[rl == c], 5e-2, 1e-3) >= 0.9
            assert _match_dets(ref[rl == c], got[gl == c], 5e-2, 1e-3) >= 0.9


def test_dyprob_roi_head_schedule_golden():
    """DyProbRoIHead (prob_roi_head.py:473-623): losses of 4 iterations and the Dynamic R-CNN
    IoU-threshold / SmoothL1-beta updates after iterations 2 and 4"""
    g = load('g18_boost_variants')
    head = _variant_head('DyProbRoIHead', 4, 20, boost=True, gamma=0.5,
                         loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0),
                         dynamic_rcnn=dict(iou_topk=40, beta_topk=6, update_iter_interval=2,
                                           initial_iou=0.4, initial_beta=1.0)).train()
    for it in range(4):
        feats, metas, gts, gls, props = _dev_inputs(4, 1, 30 + it)
        torch.manual_seed(50 + it)
        losses = head.forward_train(feats, metas, props, gts, gls)
        for k in ('loss_cls', 'loss_bbox', 'acc'):
            assert _loss_close(losses[k], g[f'dy{it}_{k}']), (it, k, losses[k], g[f'dy{it}_{k}'])
        sched = [head.bbox_assigner.pos_iou_thr, head.bbox_assigner.neg_iou_thr,
                 head.bbox_assigner.min_pos_iou, head.bbox_head.loss_bbox.beta,
                 len(head.iou_history), len(head.beta_history)]
        assert np.allclose(sched, g['dy_sched'][it], rtol=1e-4, atol=1e-6), (it, sched, g['dy_sched'][it])
    assert g['dy_sched'][1][3] != 1.0 or g['dy_sched'][1][0] != 0.6   # the schedule moved something


@pytest.mark.parametrize('boost', [True, False])
def test_dyprob_roi_head_whole_batch_device_path_golden(boost):
    """DyProbRoIHead.forward_train_device (whole batch: brcnn_assign_max_iou + brcnn_rcnn_sample + the boosting-loss
    kernels with plain label weights and SmoothL1(beta), Dynamic R-CNN statistics from device tensors) against the
    REFERENCE's losses and schedule of the same four iterations (g18; boost=True), and against the per-image path of
    this repo for the un-boosted branch (prob_roi_head.py:473-623); gradients of both paths agree"""
    from brcnn import train_ops
    g = load('g18_boost_variants')
    kw = dict(boost=boost, gamma=0.5, loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0),
              dynamic_rcnn=dict(iou_topk=40, beta_topk=6, update_iter_interval=2, initial_iou=0.4, initial_beta=1.0))
    head = _variant_head('DyProbRoIHead', 4, 20, **kw).train()
    ref_head = _variant_head('DyProbRoIHead', 4, 20, **kw).train()
    assert head.device_train_ok()
    for it in range(4):
        feats, metas, gts, gls, props = _dev_inputs(4, 1, 30 + it)
        dets, num = _padded_proposals(props)
        nhwc = [f.permute(0, 2, 3, 1).contiguous() for f in feats]
        torch.manual_seed(50 + it)
        losses, _ = head.forward_train_device(nhwc, metas, dets, num, train_ops.flatten_gts(gts, gls))
        torch.manual_seed(50 + it)
        ref = ref_head.forward_train(feats, metas, props, gts, gls)
        for k in ('loss_cls', 'loss_bbox', 'acc'):
            if boost:
                assert _loss_close(losses[k], g[f'dy{it}_{k}']), (it, k, losses[k], g[f'dy{it}_{k}'])
            assert torch.allclose(losses[k].reshape(-1), ref[k].reshape(-1), rtol=2e-3, atol=1e-4), (it, k, losses[k], ref[k])
        head.zero_grad(set_to_none=True); ref_head.zero_grad(set_to_none=True)
        (losses['loss_cls'] + losses['loss_bbox']).backward()
        (ref['loss_cls'] + ref['loss_bbox']).backward()
        for (k, pa), (_, pb) in zip(head.named_parameters(), ref_head.named_parameters()):
            if pb.grad is None:
                assert pa.grad is None or float(pa.grad.abs().max()) == 0.0, k
                continue
            scale = float(pb.grad.abs().max()) + 1e-12
            assert float((pa.grad - pb.grad).abs().max()) <= 2e-3 * scale, (it, k, float((pa.grad - pb.grad).abs().max()), scale)
        sched = [head.bbox_assigner.pos_iou_thr, head.bbox_assigner.neg_iou_thr, head.bbox_assigner.min_pos_iou,
                 head.bbox_head.loss_bbox.beta, len(head.iou_history), len(head.beta_history)]
        rsched = [ref_head.bbox_assigner.pos_iou_thr, ref_head.bbox_assigner.neg_iou_thr, ref_head.bbox_assigner.min_pos_iou,
                  ref_head.bbox_head.loss_bbox.beta, len(ref_head.iou_history), len(ref_head.beta_history)]
        assert np.allclose(sched, rsched, rtol=1e-4, atol=1e-6), (it, sched, rsched)
        if boost:
            assert np.allclose(sched, g['dy_sched'][it], rtol=1e-4, atol=1e-6), (it, sched, g['dy_sched'][it])


@pytest.mark.parametrize('dt,ratio', [('f32', 0.1), ('bf16', 0.35)])
def test_train_steps_reduce_the_loss(dt, ratio):
    """40 SGD steps (clip 35, the reference's optimizer hooks) on two fixed images through the whole
    HIP train path -- conv / GroupNorm / RoIAlign / focal forward and backward -- bring the loss
    down by an order of magnitude: the gradients do not only match element-wise, they train"""
    cfg = Config.fromfile(CFG)
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=10))
    m = m.to(DEV).train()
    m.set_compute_dtype(dt)
    try:
        img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10)
        data = dict(img=img.to(DEV), img_metas=metas, gt_bboxes=[b.to(DEV) for b in gts],
                    gt_labels=[l.to(DEV) for l in gls])
        params = [p for p in m.parameters() if p.requires_grad]
        opt = torch.optim.SGD(params, lr=0.002, momentum=0.9, weight_decay=1e-4)
        torch.manual_seed(3)
        ls = []
        for _ in range(40):
            opt.zero_grad(set_to_none=True)
            out = m.train_step(data, None)
            out['loss'].backward()
            torch.nn.utils.clip_grad_norm_(params, 35.0)
            opt.step()
            ls.append(float(out['loss'].detach()))
        assert all(np.isfinite(ls))
        assert sum(ls[-5:]) / 5 < ratio * sum(ls[:5]) / 5, ls[::4]
    finally:
        m.set_compute_dtype('f32')
