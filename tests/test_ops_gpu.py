"""-m gpu parity tests: the HIP operators (called through the C ABI via brcnn.ops) against
the CPU oracle (oracle/orc.py) on identical seeded inputs.  Bars: bit-exact for indices and
for RoIAlign values (same fp32 operation order, no FMA contraction); 1e-5/1e-6 where a
transcendental (expf/logf/powf) differs between device and host libm."""
import os

import numpy as np
import pytest
import torch

import brcnn  # noqa: F401
from brcnn import ops
from oracle import orc
from tests import util

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


# --------------------------------------------------------------------------- RoIAlign
def _feat(n, c, h, w, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, c, h, w, generator=g)


@pytest.mark.parametrize('aligned', [True, False])
@pytest.mark.parametrize('sampling_ratio', [0, 2])
def test_roi_align_nchw_bit_exact(aligned, sampling_ratio):
    x = _feat(2, 16, 50, 84)
    rois = util.rand_rois(300, 2, 84 * 16.0, 50 * 16.0, seed=1)
    # adversarial: outside the map, tiny (grid 1), huge (grid >= 15), zero-size
    rois = torch.cat([rois, torch.tensor([[0, -200., -200., -100., -100.], [1, 10., 10., 11., 11.],
                                          [0, 0., 0., 1343., 799.], [1, 100., 100., 100., 100.],
                                          [0, 1300., 700., 1500., 900.]])])
    ref = orc.roi_align_forward(x, rois, 7, 1 / 16., sampling_ratio, 'avg', aligned)
    out = ops.roi_align(x.to(DEV), rois.to(DEV), 7, 1 / 16., sampling_ratio, 'avg', aligned)
    assert out.shape == ref.shape
    assert torch.equal(out.cpu(), ref)


def test_roi_align_nhwc_bit_exact_c256():
    x = _feat(2, 256, 25, 42, seed=3)
    rois = util.rand_rois(256, 2, 42 * 32.0, 25 * 32.0, seed=4)
    ref = orc.roi_align_forward(x, rois, 7, 1 / 32., 0, 'avg', True)
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last)
    from brcnn import lib
    try:        # exact-order kernel: bit-identical to the reference's CPU accumulation order
        lib.load().brcnn_roi_align_set_exact(1)
        out = ops.roi_align(xg, rois.to(DEV), 7, 1 / 32., 0, 'avg', True)
    finally:
        lib.load().brcnn_roi_align_set_exact(0)
    assert out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out.cpu().contiguous(), ref)
    # default footprint-form kernel: same value up to fp32 re-association (a bin sums <= ~40 products)
    out2 = ops.roi_align(xg, rois.to(DEV), 7, 1 / 32., 0, 'avg', True).cpu().contiguous()
    assert (out2 - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    f64 = orc.roi_align_f64(x, rois, 7, 1 / 32., 0, True)
    assert (out2.double() - f64).abs().max().item() <= (ref.double() - f64).abs().max().item() * 2 + 1e-6


def test_roi_align_max_mode():
    x = _feat(1, 8, 20, 20, seed=5)
    rois = util.rand_rois(40, 1, 320., 320., seed=6)
    ref = orc.roi_align_forward(x, rois, (3, 5), 1 / 16., 2, 'max', True)
    out = ops.roi_align(x.to(DEV), rois.to(DEV), (3, 5), 1 / 16., 2, 'max', True)
    assert torch.equal(out.cpu(), ref)


def test_roi_align_known_answers():
    # SURVEY 8c / mmcv's published unit vectors
    x = torch.tensor([[[[1., 2.], [3., 4.]]]])
    rois = torch.tensor([[0., 0., 0., 1., 1.]])
    out = ops.roi_align(x.to(DEV), rois.to(DEV), 2, 1.0, 2, 'avg', True).cpu()
    assert torch.allclose(out, torch.tensor([[[[1.0, 1.25], [1.5, 1.75]]]]))
    x = torch.tensor([[[[1., 2., 5., 6.], [3., 4., 7., 8.], [9., 10., 13., 14.], [11., 12., 15., 16.]]]])
    rois = torch.tensor([[0., 0., 0., 3., 3.]])
    out = ops.roi_align(x.to(DEV), rois.to(DEV), 2, 1.0, 2, 'avg', True).cpu()
    assert torch.allclose(out, torch.tensor([[[[1.9375, 4.75], [7.5625, 10.375]]]]))
    out = ops.roi_align(x.to(DEV), rois.to(DEV), 2, 1.0, 2, 'avg', False).cpu()
    assert torch.allclose(out, torch.tensor([[[[3.625, 6.875], [10.125, 13.375]]]]))


def test_roi_align_empty_and_errors():
    x = _feat(1, 4, 8, 8).to(DEV)
    out = ops.roi_align(x, torch.zeros(0, 5, device=DEV), 7, 1.0, 0, 'avg', True)
    assert out.shape == (0, 4, 7, 7)
    with pytest.raises(AssertionError):
        ops.roi_align(x, torch.zeros(3, 4, device=DEV), 7, 1.0, 0, 'avg', True)
    with pytest.raises(RuntimeError):
        ops.roi_align(x.cpu(), torch.zeros(3, 5), 7, 1.0, 0, 'avg', True)


@pytest.mark.parametrize('nhwc', [False, True])
def test_roi_align_backward(nhwc):
    x = _feat(2, 32, 25, 42, seed=8)
    rois = util.rand_rois(64, 2, 42 * 32.0, 25 * 32.0, seed=9)
    g = torch.Generator().manual_seed(10)
    go = torch.randn(64, 32, 7, 7, generator=g)
    ref = orc.roi_align_backward(go, rois, x.shape, 7, 1 / 32., 0, True)
    xg = x.to(DEV)
    if nhwc:
        xg = xg.contiguous(memory_format=torch.channels_last)
    xg.requires_grad_(True)
    out = ops.roi_align(xg, rois.to(DEV), 7, 1 / 32., 0, 'avg', True)
    out.backward(go.to(DEV))
    # atomics reorder the fp32 sums: tolerance, not bit equality
    assert torch.allclose(xg.grad.cpu().contiguous(), ref, rtol=1e-5, atol=1e-5)


def test_roi_extract_fused_matches_per_level():
    strides = [8, 16, 32, 64, 128]
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    feats = [_feat(2, 256, h, w, seed=20 + i) for i, (h, w) in enumerate(sizes)]
    rois = util.rand_rois(512, 2, 1333., 800., seed=21, min_size=8., max_size=1200.)
    # reference: map_roi_levels + per-level oracle RoIAlign (single_level_roi_extractor.py:36-115)
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lvls = torch.floor(torch.log2(scale / 56 + 1e-6)).clamp(min=0, max=4).long()
    ref = torch.zeros(512, 256, 7, 7)
    for i in range(5):
        inds = (lvls == i).nonzero(as_tuple=False).squeeze(1)
        if inds.numel():
            ref[inds] = orc.roi_align_forward(feats[i], rois[inds], 7, 1. / strides[i], 0, 'avg', True)
    fg = [f.to(DEV).permute(0, 2, 3, 1).contiguous() for f in feats]
    from brcnn import lib
    try:
        lib.load().brcnn_roi_align_set_exact(1)
        out, lv = ops.roi_extract(fg, rois.to(DEV), 7, strides, 56, 0)
    finally:
        lib.load().brcnn_roi_align_set_exact(0)
    assert torch.equal(lv.cpu().long(), lvls)
    assert torch.equal(out.permute(0, 3, 1, 2).cpu().contiguous(), ref)
    # default (footprint form): RoIs from 8 px to 1200 px -> grids up to 22 x 22 samples per bin and
    # the > 64-sample fallback; equal to fp32 round-off of a sum of up to ~500 products
    out2, lv2 = ops.roi_extract(fg, rois.to(DEV), 7, strides, 56, 0)
    assert torch.equal(lv2, lv)
    err = (out2.permute(0, 3, 1, 2).cpu() - ref).abs().max().item()
    assert err <= 1e-5 * max(1.0, ref.abs().max().item()), err
    # the other selectable forms of the footprint kernel (2: per-bin loop, round-robin rows; 3: column streaming,
    # round-robin rows): same bound; the XCD-contiguous row order of the default changes no value
    try:
        for mode in (2, 3):
            lib.load().brcnn_roi_align_set_exact(mode)
            o_m, lv_m = ops.roi_extract(fg, rois.to(DEV), 7, strides, 56, 0)
            assert torch.equal(lv_m, lv)
            err_m = (o_m.permute(0, 3, 1, 2).cpu() - ref).abs().max().item()
            assert err_m <= 1e-5 * max(1.0, ref.abs().max().item()), (mode, err_m)
            if mode == 3:
                assert torch.equal(o_m, out2)
    finally:
        lib.load().brcnn_roi_align_set_exact(0)
    huge = torch.tensor([[0., -500., -300., 2500., 1900.], [1., 10., 10., 1300., 790.]])   # bins > 64 px
    o1, _ = ops.roi_extract(fg, huge.to(DEV), 7, strides, 56, 0)
    r1 = orc.roi_align_forward(feats[4], huge, 7, 1. / 128, 0, 'avg', True)
    assert (o1.permute(0, 3, 1, 2).cpu() - r1).abs().max().item() <= 1e-5


def test_roi_extract_configs1_size_real_proposals_vs_oracle():
    """BASELINE configs[1] as it is worded (batch 8, 1333 x 800 inputs, inference-only RoIAlign vs CPU): the REFERENCE's own
    256 proposals per image (g20 `props*`, produced by its RPN on the full-size seeded batch) through the shipped
    `roi_extract` -- the default footprint form, the one the headline runs -- on a seeded full-size fp32 pyramid, against
    `map_roi_levels` + the per-level C oracle (mmcv's sample-order RoIAlign).  2048 RoIs, the recipe's five levels, 8 x 100 x 168 ... maps.
    Level map identical; values within fp32 round-off of a sum of up to ~500 products (2e-6 of the map scale); the exact
    sample-order form is bit-identical."""
    from tests.test_host_cpu import load
    g = load('g20_fullsize')
    strides = [8, 16, 32, 64, 128]
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    B = 8
    feats = [_feat(B, 256, h, w, seed=120 + i) for i, (h, w) in enumerate(sizes)]
    rois = torch.cat([torch.cat([torch.full((256, 1), float(b)), torch.from_numpy(g[f'props{b}'][:, :4]).float()], 1)
                      for b in range(B)], 0)
    assert rois.shape == (2048, 5)
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lvls = torch.floor(torch.log2(scale / 56 + 1e-6)).clamp(min=0, max=4).long()
    ref = torch.zeros(2048, 256, 7, 7)
    for i in range(5):
        inds = (lvls == i).nonzero(as_tuple=False).squeeze(1)
        if inds.numel():
            ref[inds] = orc.roi_align_forward(feats[i], rois[inds], 7, 1. / strides[i], 0, 'avg', True)
    assert len(set(lvls.tolist())) >= 3             # the reference's proposals spread over the pyramid
    fg = [f.to(DEV).permute(0, 2, 3, 1).contiguous() for f in feats]
    out, lv = ops.roi_extract(fg, rois.to(DEV), 7, strides, 56, 0)
    assert torch.equal(lv.cpu().long(), lvls)
    err = (out.permute(0, 3, 1, 2).cpu() - ref).abs().max().item()
    assert err <= 2e-6 * max(1.0, ref.abs().max().item()), err
    from brcnn import lib
    try:
        lib.load().brcnn_roi_align_set_exact(1)
        out_e, _ = ops.roi_extract(fg, rois.to(DEV), 7, strides, 56, 0)
    finally:
        lib.load().brcnn_roi_align_set_exact(0)
    assert torch.equal(out_e.permute(0, 3, 1, 2).cpu().contiguous(), ref)


# --------------------------------------------------------------------------- NMS
def test_nms_known_answer():
    boxes = torch.tensor([[6., 3., 8., 7.], [3., 6., 9., 11.], [3., 7., 10., 12.], [1., 4., 13., 7.]])
    scores = torch.tensor([0.6, 0.9, 0.7, 0.2])
    dets, inds = ops.nms(boxes.to(DEV), scores.to(DEV), 0.3)
    assert inds.cpu().tolist() == [1, 0, 3]
    assert torch.equal(dets.cpu(), torch.cat([boxes[[1, 0, 3]], scores[[1, 0, 3], None]], 1))


@pytest.mark.parametrize('n,thr,offset', [(1, 0.5, 0), (63, 0.5, 0), (64, 0.5, 1), (65, 0.7, 0),
                                          (1000, 0.7, 0), (4693, 0.7, 0), (9999, 0.5, 1)])
def test_nms_indices_bit_exact(n, thr, offset):
    boxes = util.clustered_boxes(n, seed=n)
    scores = util.tie_free_scores(n, seed=n + 1)
    _, ref = orc.nms(boxes, scores, thr, offset)
    dets, inds = ops.nms(boxes.to(DEV), scores.to(DEV), thr, offset)
    assert torch.equal(inds.cpu(), ref)
    assert 0 < ref.numel() < n or n == 1


def test_nms_ties_follow_index_order():
    boxes = util.clustered_boxes(500, seed=3)
    scores = torch.full((500,), 0.5)
    scores[::7] = 0.9
    _, ref = orc.nms(boxes, scores, 0.6)
    _, inds = ops.nms(boxes.to(DEV), scores.to(DEV), 0.6)
    assert torch.equal(inds.cpu(), ref)


def test_nms_score_threshold_max_num_empty():
    boxes = util.clustered_boxes(800, seed=5)
    scores = util.tie_free_scores(800, seed=6)
    dref, ref = orc.nms(boxes, scores, 0.5, 0, 0.3, 20)
    dets, inds = ops.nms(boxes.to(DEV), scores.to(DEV), 0.5, 0, 0.3, 20)
    assert torch.equal(inds.cpu(), ref) and torch.equal(dets.cpu(), dref)
    dets, inds = ops.nms(torch.zeros(0, 4, device=DEV), torch.zeros(0, device=DEV), 0.5)
    assert dets.shape == (0, 5) and inds.shape == (0,) and inds.dtype == torch.int64


def test_nms_segments():
    lens = [4693, 1, 0, 2500, 64]
    seg = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    n = int(seg[-1])
    boxes = util.clustered_boxes(n, seed=11)
    scores = util.tie_free_scores(n, seed=12)
    keep, num = ops.nms_segments(boxes.to(DEV), scores.to(DEV), seg.to(DEV), max(lens), 0.7, 0, 256)
    keep, num = keep.cpu(), num.cpu()
    for s, ln in enumerate(lens):
        b, e = int(seg[s]), int(seg[s + 1])
        _, ref = orc.nms(boxes[b:e], scores[b:e], 0.7, 0, 0, 256)
        assert int(num[s]) == ref.numel()
        assert torch.equal(keep[b:b + ref.numel()], ref + b)


@pytest.mark.parametrize('n', [3000, 12000])   # below / above split_thr=10000
def test_batched_nms_matches_oracle(n):
    boxes = util.clustered_boxes(n, seed=n)
    scores = util.tie_free_scores(n, seed=n + 3)
    g = torch.Generator().manual_seed(n)
    ids = torch.randint(0, 5, (n,), generator=g)
    cfg = dict(type='nms', iou_threshold=0.7)
    dref, kref = orc.batched_nms(boxes, scores, ids, cfg)
    dets, keep = ops.batched_nms(boxes.to(DEV), scores.to(DEV), ids.to(DEV), cfg)
    assert torch.equal(keep.cpu(), kref)
    assert torch.equal(dets.cpu(), dref)


# --------------------------------------------------------------------------- soft-NMS
def test_soft_nms_known_answers():
    boxes = torch.tensor([[6., 3., 8., 7.], [3., 6., 9., 11.], [3., 7., 10., 12.], [1., 4., 13., 7.]])
    scores = torch.tensor([0.6, 0.9, 0.7, 0.2])
    b, s = boxes.to(DEV), scores.to(DEV)
    dets, inds = ops.soft_nms(b, s, 0.3, 0.5, 1e-3, 'naive')
    assert inds.cpu().tolist() == [1, 0, 3]
    dets, inds = ops.soft_nms(b, s, 0.3, 0.5, 1e-3, 'linear')
    assert inds.cpu().tolist() == [1, 0, 2, 3]
    assert np.allclose(dets[:, 4].cpu().numpy(), [0.9, 0.6, 0.29024392, 0.2], atol=1e-6)
    dets, inds = ops.soft_nms(b, s, 0.3, 0.5, 1e-3, 'gaussian')
    assert inds.cpu().tolist() == [1, 0, 2, 3]
    assert np.allclose(dets[:, 4].cpu().numpy(), [0.9, 0.59630775, 0.35275510, 0.18650459], atol=1e-6)


@pytest.mark.parametrize('method', ['linear', 'naive'])
@pytest.mark.parametrize('n,min_score', [(300, 0.0), (2000, 0.0), (2000, 0.05), (777, 0.3)])
def test_soft_nms_bit_exact(method, n, min_score):
    boxes = util.clustered_boxes(n, seed=n + 1)
    scores = util.tie_free_scores(n, seed=n + 2)
    dref, iref = orc.soft_nms(boxes, scores, 0.7, 0.5, min_score, method)
    dets, inds = ops.soft_nms(boxes.to(DEV), scores.to(DEV), 0.7, 0.5, min_score, method)
    assert torch.equal(inds.cpu(), iref)
    assert torch.equal(dets.cpu(), dref)


def test_soft_nms_gaussian_close():
    n = 1500
    boxes = util.clustered_boxes(n, seed=31)
    scores = util.tie_free_scores(n, seed=32)
    dref, iref = orc.soft_nms(boxes, scores, 0.3, 0.5, 1e-3, 'gaussian')
    dets, inds = ops.soft_nms(boxes.to(DEV), scores.to(DEV), 0.3, 0.5, 1e-3, 'gaussian')
    # device expf vs glibc expf: <= 1 ulp per decay step; the pick sets must agree and the
    # scores must agree to 1e-6
    assert inds.numel() == iref.numel()
    assert set(inds.cpu().tolist()) == set(iref.tolist())
    order = torch.argsort(inds.cpu())
    oref = torch.argsort(iref)
    assert torch.allclose(dets.cpu()[order, 4], dref[oref, 4], atol=1e-6)


def test_soft_nms_ties_and_segments():
    lens = [900, 0, 1, 1300]
    seg = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    n = int(seg[-1])
    boxes = util.clustered_boxes(n, seed=41)
    g = torch.Generator().manual_seed(42)
    scores = (torch.randint(1, 20, (n,), generator=g).float() / 20.0)   # many exact ties
    dets, inds, num = ops.soft_nms_segments(boxes.to(DEV), scores.to(DEV), seg.to(DEV), 0.5, 0.5,
                                            0.1, 1, 0)
    dets, inds, num = dets.cpu(), inds.cpu(), num.cpu()
    for s, ln in enumerate(lens):
        b, e = int(seg[s]), int(seg[s + 1])
        dref, iref = orc.soft_nms(boxes[b:e], scores[b:e], 0.5, 0.5, 0.1, 'linear')
        assert int(num[s]) == iref.numel()
        assert torch.equal(inds[b:b + iref.numel()], iref + b)
        assert torch.equal(dets[b:b + iref.numel()], dref)


# --------------------------------------------------------------------------- focal loss
@pytest.mark.parametrize('c', [1, 4, 80])
def test_sigmoid_focal_loss_fwd_bwd(c):
    n = 5000
    g = torch.Generator().manual_seed(c)
    x = torch.randn(n, c, generator=g) * 3
    t = torch.randint(0, c + 1, (n,), generator=g)
    ref = orc.sigmoid_focal_loss_forward(x, t, 2.0, 0.25)
    gref = orc.sigmoid_focal_loss_backward(x, t, 2.0, 0.25)
    xg = x.to(DEV).requires_grad_(True)
    out = ops.sigmoid_focal_loss(xg, t.to(DEV), 2.0, 0.25, None, 'none')
    assert torch.allclose(out.cpu(), ref, rtol=1e-5, atol=1e-7)
    out.sum().backward()
    assert torch.allclose(xg.grad.cpu(), gref, rtol=1e-5, atol=1e-7)
    # python form of the reference CPU path (focal_loss.py:12-57)
    import torch.nn.functional as F
    tt = F.one_hot(t, c + 1)[:, :c].float()
    p = x.sigmoid()
    pt = (1 - p) * tt + p * (1 - tt)
    py = F.binary_cross_entropy_with_logits(x, tt, reduction='none') * (0.25 * tt + 0.75 * (1 - tt)) * pt.pow(2.0)
    # closed form vs python form: 1-sigmoid(x) cancellation at large |x| -> 1e-3 (north_star fp32 bar)
    assert torch.allclose(out.detach().cpu(), py, rtol=1e-2, atol=1e-5)
    assert abs(out.sum().item() - py.sum().item()) < 1e-4 * abs(py.sum().item())


# --------------------------------------------------------------------------- conv stack
def _conv_ref(x_nchw, w, scale, shift, res, relu, stride, pad):
    import torch.nn.functional as F
    y = F.conv2d(x_nchw.double(), w.double(), None, stride, pad)
    if scale is not None:
        y = y * scale.double().view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.double().view(1, -1, 1, 1)
    if res is not None:
        y = y + res.double()
    if relu:
        y = y.relu()
    return y


@pytest.mark.parametrize('cfg', [
    # (N, Cin, H, W, Cout, k, stride, pad, scale, shift, residual, relu)
    (2, 64, 24, 40, 64, 1, 1, 0, True, True, False, True),
    (2, 64, 24, 40, 256, 1, 1, 0, True, True, True, True),
    (1, 128, 30, 31, 128, 3, 1, 1, True, True, False, True),
    (2, 256, 25, 42, 256, 3, 2, 1, False, True, False, False),
    (1, 512, 13, 21, 1024, 1, 2, 0, True, True, False, False),
    (2, 3, 64, 96, 64, 7, 2, 3, True, True, False, True),
    (1, 256, 13, 21, 9, 3, 1, 1, False, True, False, False),
    (1, 256, 13, 21, 36, 3, 1, 1, False, True, False, False),
    (300, 12544 // 49, 7, 7, 130, 7, 1, 0, False, True, False, True),   # FC as a 7x7 "valid" conv
])
def test_conv2d_nhwc_vs_float64(cfg):
    n, cin, h, w, cout, k, stride, pad, has_scale, has_shift, has_res, relu = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 1000)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    scale = torch.rand(cout, generator=g) + 0.5 if has_scale else None
    shift = torch.randn(cout, generator=g) if has_shift else None
    ho, wo = ops.conv_out_size(h, w, k, k, stride, pad)
    res = torch.randn(n, cout, ho, wo, generator=g) if has_res else None
    ref = _conv_ref(x, wt, scale, shift, res, relu, stride, pad)
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wg = wt.permute(0, 2, 3, 1).contiguous().to(DEV)
    rg = res.permute(0, 2, 3, 1).contiguous().to(DEV) if has_res else None
    y = ops.conv2d_nhwc(xg, wg, scale.to(DEV) if has_scale else None,
                        shift.to(DEV) if has_shift else None, rg, relu, stride, pad)
    y = y.permute(0, 3, 1, 2).cpu().double()
    assert y.shape == ref.shape
    err = (y - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


def test_linear_large_k():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(257, 12544, generator=g)
    w = torch.randn(1024, 12544, generator=g) / 112.0
    b = torch.randn(1024, generator=g)
    ref = (x.double() @ w.double().t() + b.double()).relu()
    y = ops.linear_nhwc(x.to(DEV), w.to(DEV), b.to(DEV), True).cpu().double()
    assert (y - ref).abs().max().item() < 5e-5


def test_small_nhwc_kernels():
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, 33, 47, generator=g)
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = ops.maxpool3x3s2_nhwc(xg).permute(0, 3, 1, 2).cpu()
    assert torch.equal(y, F.max_pool2d(x, 3, 2, 1))
    # group norm + relu
    x = torch.randn(2, 256, 25, 42, generator=g) * 2 + 0.3
    gamma, beta = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    ref = F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5).relu()
    y = ops.groupnorm_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), gamma.to(DEV), beta.to(DEV),
                           32, 1e-5, True).permute(0, 3, 1, 2).cpu().double()
    assert (y - ref).abs().max().item() < 1e-5
    # nearest upsample + add (exact 2x and ragged)
    for (hd, wd, hs, ws) in [(50, 84, 25, 42), (13, 21, 7, 11), (25, 41, 13, 21)]:
        d = torch.randn(2, 256, hd, wd, generator=g)
        s = torch.randn(2, 256, hs, ws, generator=g)
        ref = d + F.interpolate(s, size=(hd, wd), mode='nearest')
        dg = d.permute(0, 2, 3, 1).contiguous().to(DEV)
        ops.upsample_nearest_add_nhwc_(dg, s.permute(0, 2, 3, 1).contiguous().to(DEV))
        assert torch.equal(dg.permute(0, 3, 1, 2).cpu(), ref)
    # layout shuffles
    x = torch.randn(3, 37, 19, 23, generator=g)
    assert torch.equal(ops.nchw_to_nhwc(x.to(DEV)).cpu(), x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_nchw(x.permute(0, 2, 3, 1).contiguous().to(DEV)).cpu(), x)


# --------------------------------------------------------------------------- RPN stage
def test_rpn_topk_matches_stable_sort():
    """bit-exact indices against torch's stable descending sort on the host (the tie rule):
    distinct scores, heavy ties (quantised scores), all-equal rows, +-0, n <= k pass-through"""
    g = torch.Generator().manual_seed(21)
    B = 3
    levels = [torch.rand(B, 151200, generator=g), torch.rand(B, 37800, generator=g),
              (torch.rand(B, 9450, generator=g) * 50).floor() / 50,          # ~190 ties per value
              torch.full((B, 2394), 0.25), torch.rand(B, 630, generator=g)]
    levels[1][:, ::7] = 0.0
    levels[1][:, 1::14] = -0.0
    levels[0][0, 5000:9000] = levels[0][0, 4999]                             # a 4000-long run of ties
    levels[0][1, 18000:20000] = 0.99999                                      # ties across a part boundary (18900)
    levels[0][2, 37700:37900] = 2.0                                          # the winners straddle two parts
    for k in (1000, 2000, 37):
        got = ops.rpn_topk([l.to(DEV) for l in levels], k)
        for l, (gs, gi) in zip(levels, got):
            n = l.shape[1]
            if n > k:
                rs, ri = l.sort(dim=1, descending=True, stable=True)
                rs, ri = rs[:, :k], ri[:, :k]
            else:
                rs, ri = l, torch.arange(n).expand(B, n)
            assert torch.equal(gi.cpu(), ri), (n, k)
            assert torch.equal(gs.cpu(), rs)
    with pytest.raises(Exception):
        ops.rpn_topk([levels[0].to(DEV)], 5000)


def test_rpn_score_and_decode():
    g = torch.Generator().manual_seed(77)
    cls, iou = torch.randn(2, 13, 21, 9, generator=g) * 3, torch.randn(2, 13, 21, 9, generator=g) * 3
    ref = (cls.sigmoid() * iou.sigmoid()).sqrt()
    out = ops.rpn_score(cls.to(DEV), iou.to(DEV)).cpu()
    assert util.ulp_diff(out, ref).max().item() <= 2
    # decode against the reference formulas (anchor grid + delta2bbox), CPU torch
    A, H, W, stride = 9, 13, 21, 64
    base = torch.randn(A, 4, generator=g) * 100
    base = torch.stack([-base[:, 0].abs(), -base[:, 1].abs(), base[:, 2].abs(), base[:, 3].abs()], 1)
    deltas = torch.randn(2, H, W, 4 * A, generator=g)
    deltas[0, 0, 0, :8] = torch.tensor([0., 0., 9., -9., 50., -50., 0.3, 0.2])   # clamp edges
    inds = torch.stack([torch.randperm(H * W * A, generator=g)[:500] for _ in range(2)])
    sx = torch.arange(W) * stride
    sy = torch.arange(H) * stride
    yy, xx = torch.meshgrid(sy, sx, indexing='ij')
    shifts = torch.stack([xx, yy, xx, yy], -1).reshape(-1, 1, 4).float()
    anchors = (base[None] + shifts).reshape(-1, 4)
    props, valid = ops.rpn_decode(inds.to(DEV), deltas.to(DEV), base.to(DEV), (H, W), stride,
                                  (0., 0., 0., 0.), (1., 1., 1., 1.), (800, 1333, 3), 0.0)
    for b in range(2):
        a = anchors[inds[b]]
        d = deltas[b].reshape(-1, 4)[inds[b]]
        px, py = (a[:, 0] + a[:, 2]) * 0.5, (a[:, 1] + a[:, 3]) * 0.5
        pw, ph = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
        mr = abs(np.log(16 / 1000))
        dw, dh = d[:, 2].clamp(-mr, mr), d[:, 3].clamp(-mr, mr)
        gw, gh = pw * dw.exp(), ph * dh.exp()
        gx, gy = px + pw * d[:, 0], py + ph * d[:, 1]
        ref = torch.stack([gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5], -1)
        ref[:, 0::2] = ref[:, 0::2].clamp(0, 1333)
        ref[:, 1::2] = ref[:, 1::2].clamp(0, 800)
        got = props[b].cpu()
        assert torch.allclose(got, ref, rtol=1e-6, atol=1e-3)
        v = ((ref[:, 2] - ref[:, 0]) > 0) & ((ref[:, 3] - ref[:, 1]) > 0)
        agree = (valid[b].cpu().bool() == v)
        assert agree.float().mean().item() > 0.995


def test_multi_level_conv_and_groupnorm_match_per_level():
    sizes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    B, C = 2, 64
    g = torch.Generator().manual_seed(3)
    feats = [torch.randn(B, h, w, C, generator=g).to(DEV) for h, w in sizes]
    w = (torch.randn(54, 3, 3, C, generator=g) / 24).to(DEV)
    b = torch.randn(54, generator=g).to(DEV)
    x = torch.cat([f.reshape(-1, C) for f in feats], 0)
    y, osz = ops.conv2d_nhwc_multi(x, w, B, sizes, None, b, None, False, 1, 1)
    assert osz == sizes
    r0 = 0
    for f, (h, wd) in zip(feats, sizes):
        ref = ops.conv2d_nhwc(f, w, None, b, None, False, 1, 1)
        n = B * h * wd
        assert torch.equal(y[r0:r0 + n].view(B, h, wd, 54), ref)
        r0 += n
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    z = ops.groupnorm_nhwc_multi(x, gamma, beta, 32, B, sizes, 1e-5, True)
    r0 = 0
    for f, (h, wd) in zip(feats, sizes):
        ref = ops.groupnorm_nhwc(f, gamma, beta, 32, 1e-5, True)
        n = B * h * wd
        assert torch.allclose(z[r0:r0 + n].view(B, h, wd, C), ref, atol=1e-6)
        r0 += n
    # strided (channel-slice) inputs of the rpn kernels
    full = torch.randn(2, 5, 7, 54, generator=g).to(DEV)
    s1 = ops.rpn_score(full[..., :9], full[..., 45:])
    s2 = ops.rpn_score(full[..., :9].contiguous(), full[..., 45:].contiguous())
    assert torch.equal(s1, s2)


# --------------------------------------------------------------------------- conv backward
@pytest.mark.parametrize('cfg', [
    # (N, Cin, H, W, Cout, k, stride, pad, bias)
    (2, 64, 20, 28, 64, 3, 1, 1, True),
    (2, 128, 17, 23, 256, 1, 1, 0, False),
    (2, 64, 20, 28, 128, 3, 2, 1, True),
    (1, 256, 13, 21, 512, 1, 2, 0, False),
    (2, 256, 13, 21, 54, 3, 1, 1, True),      # padded output channels
    (2, 32, 15, 16, 32, 3, 2, 1, True),       # odd H: output_padding case of the data gradient
])
def test_conv_autograd_matches_torch(cfg):
    from brcnn.autograd import conv2d_nhwc_autograd
    import torch.nn.functional as F
    n, cin, h, w, cout, k, stride, pad, has_bias = cfg
    g = torch.Generator().manual_seed(sum(cfg[:8]))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g) if has_bias else None
    xr = x.double().requires_grad_(); wr = wt.double().requires_grad_()
    br = b.double().requires_grad_() if has_bias else None
    yr = F.conv2d(xr, wr, br, stride, pad)
    go = torch.randn(yr.shape, generator=g)
    yr.backward(go.double())
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_()
    wg = wt.to(DEV).requires_grad_()
    bg = b.to(DEV).requires_grad_() if has_bias else None
    y = conv2d_nhwc_autograd(xg, wg, bg, stride, pad)
    y.backward(go.permute(0, 2, 3, 1).contiguous().to(DEV))
    def rel(a, ref):
        return (a.double().cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
    assert rel(y.detach().permute(0, 3, 1, 2), yr.detach()) < 2e-5
    assert rel(xg.grad.permute(0, 3, 1, 2), xr.grad) < 5e-5
    assert rel(wg.grad, wr.grad) < 5e-5
    if has_bias:
        assert rel(bg.grad, br.grad) < 5e-5


def test_linear_and_roi_extract_autograd():
    from brcnn.autograd import linear_autograd, roi_extract_autograd
    g = torch.Generator().manual_seed(4)
    x = torch.randn(300, 1024, generator=g); w = torch.randn(21, 1024, generator=g) / 32; b = torch.randn(21, generator=g)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = xr @ wr.t() + br
    go = torch.randn(300, 21, generator=g)
    yr.backward(go.double())
    xg, wg, bg = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = linear_autograd(xg, wg, bg)
    y.backward(go.to(DEV))
    for a, r in ((y.detach(), yr.detach()), (xg.grad, xr.grad), (wg.grad, wr.grad), (bg.grad, br.grad)):
        assert (a.double().cpu() - r).abs().max().item() < 5e-5 * max(r.abs().max().item(), 1.0)
    # RoI extraction gradient vs the oracle's per-level RoIAlign backward
    strides = [8, 16, 32, 64, 128]
    sizes = [(40, 64), (20, 32), (10, 16), (5, 8), (3, 4)]
    feats = [torch.randn(2, 16, h, w_, generator=g) for h, w_ in sizes]
    rois = util.rand_rois(100, 2, 512., 320., seed=5, min_size=8., max_size=600.)
    go = torch.randn(100, 7, 7, 16, generator=g)
    fg = [f.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_() for f in feats]
    out = roi_extract_autograd(fg, rois.to(DEV), 7, strides, 56, 0)
    out.backward(go.to(DEV))
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lvls = torch.floor(torch.log2(scale / 56 + 1e-6)).clamp(min=0, max=4).long()
    for i in range(5):
        inds = (lvls == i).nonzero(as_tuple=False).squeeze(1)
        ref = orc.roi_align_backward(go[inds].permute(0, 3, 1, 2).contiguous(), rois[inds], feats[i].shape, 7,
                                     1. / strides[i], 0, True) if inds.numel() else torch.zeros_like(feats[i])
        assert torch.allclose(fg[i].grad.permute(0, 3, 1, 2).cpu(), ref, rtol=1e-4, atol=1e-5), i


def test_roi_extract_backward_gather_form():
    """the gather-form RoI feature gradient (no atomics: bit-reproducible run to run) against the C oracle's
    per-level RoIAlign backward and the scatter form, at C = 256 / 320 (a second channel chunk), with RoIs that hang
    over the border, sit outside the map, are tiny (many bins on one pixel) or huge"""
    from brcnn import autograd as ag
    from brcnn.autograd import roi_extract_autograd
    g = torch.Generator().manual_seed(14)
    strides = [8, 16, 32, 64, 128]
    sizes = [(50, 84), (25, 42), (13, 21), (7, 11), (4, 6)]
    for C, B, K in ((256, 3, 700), (320, 2, 150), (8, 1, 60)):
        feats = [torch.randn(B, h, w_, C, generator=g) for h, w_ in sizes]
        rois = util.rand_rois(K, B, 672., 400., seed=5 + C, min_size=4., max_size=900.)
        extra = torch.tensor([[0, -80., -60., 30., 20.], [0, 650., 380., 720., 460.], [B - 1, -500., -500., -300., -300.],
                              [0, 100., 100., 101., 101.], [B - 1, 0., 0., 672., 400.], [0, 300., 10., 310., 390.]])
        rois = torch.cat([rois, extra])
        go = torch.randn(rois.shape[0], 7, 7, C, generator=g)
        res = {}
        for mode in (True, False, True):
            ag.ROI_BACKWARD_GATHER = mode
            try:
                fg = [f.clone().to(DEV).requires_grad_() for f in feats]
                roi_extract_autograd(fg, rois.to(DEV), 7, strides, 56, 0).backward(go.to(DEV))
            finally:
                ag.ROI_BACKWARD_GATHER = True
            grads = [f.grad.clone() for f in fg]
            if mode and True in res:
                assert all(torch.equal(a, b_) for a, b_ in zip(grads, res[True])), 'gather form is deterministic'
            res[mode] = grads
        scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
        lvls = torch.floor(torch.log2(scale / 56 + 1e-6)).clamp(min=0, max=4).long()
        for i in range(5):
            mag = max(1.0, res[False][i].abs().max().item())
            assert (res[True][i] - res[False][i]).abs().max().item() <= 2e-5 * mag, (C, i)
            if C <= 8:      # the CPU oracle is slow: small case only
                inds = (lvls == i).nonzero(as_tuple=False).squeeze(1)
                ref = orc.roi_align_backward(go[inds].permute(0, 3, 1, 2).contiguous(), rois[inds],
                                             feats[i].permute(0, 3, 1, 2).shape, 7, 1. / strides[i], 0, True) \
                    if inds.numel() else torch.zeros_like(feats[i].permute(0, 3, 1, 2))
                assert torch.allclose(res[True][i].permute(0, 3, 1, 2).cpu(), ref, rtol=1e-4, atol=1e-5), i


def test_stem_vector_path_matches_float64():
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(12)
    for (n, h, w) in [(2, 64, 96), (1, 75, 83), (2, 800 // 4, 1344 // 4)]:
        img = torch.randn(n, 3, h, w, generator=g)
        wt = torch.randn(64, 3, 7, 7, generator=g) / 12
        sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
        ref = (F.conv2d(img.double(), wt.double(), None, 2, 3) * sc.double().view(1, -1, 1, 1) +
               sh.double().view(1, -1, 1, 1)).relu()
        y = ops.stem7x7s2_nchw(img.to(DEV), ops.pack_stem_weight(wt.to(DEV)), sc.to(DEV), sh.to(DEV), True)
        y = y.permute(0, 3, 1, 2).cpu().double()
        assert y.shape == ref.shape
        assert (y - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_fused_stem_and_pool_matches_float64():
    """conv 7x7/s2 + BN + ReLU + max-pool 3x3/s2 in one launch (resnet.py:631-636) vs the float64 chain: sizes with partial
    tiles on both axes, odd conv / pool extents, one map smaller than a tile, the full-size aspect"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(13)
    for (n, h, w) in [(2, 64, 96), (1, 75, 83), (1, 9, 7), (3, 29, 131), (2, 800 // 4, 1344 // 4)]:
        img = torch.randn(n, 3, h, w, generator=g)
        wt = torch.randn(64, 3, 7, 7, generator=g) / 12
        sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
        ref = F.max_pool2d((F.conv2d(img.double(), wt.double(), None, 2, 3) * sc.double().view(1, -1, 1, 1) +
                            sh.double().view(1, -1, 1, 1)).relu(), 3, 2, 1)
        y = ops.stem7x7s2_pool_nchw(img.to(DEV), ops.pack_stem_pool_weight(wt.to(DEV)), sc.to(DEV), sh.to(DEV))
        y = y.permute(0, 3, 1, 2).cpu().double()
        assert y.shape == ref.shape
        assert (y - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
        # the raw C ABI refuses what the kernel does not cover (another channel count, an unknown dtype, a NULL image)
        from brcnn import lib
        L = lib.load()
        wq = ops.pack_stem_pool_weight(wt.to(DEV))
        yq = torch.empty((n, y.shape[2], y.shape[3], 64), device=DEV)
        imgd = img.to(DEV)
        args = lambda cout, dt, im: (im, wq.data_ptr(), None, None, yq.data_ptr(), n, h, w, cout, dt, lib.raw_stream_handle())
        assert L.brcnn_stem7x7s2_pool_nchw(*args(32, 0, imgd.data_ptr())) == -22
        assert L.brcnn_stem7x7s2_pool_nchw(*args(64, 7, imgd.data_ptr())) == -22
        assert L.brcnn_stem7x7s2_pool_nchw(*args(64, 0, None)) == -22
        # ... and the two-launch form it replaces, to fp32 round-off (another summation order)
        y2 = ops.maxpool3x3s2_nhwc(ops.stem7x7s2_nchw(img.to(DEV), ops.pack_stem_weight(wt.to(DEV)), sc.to(DEV), sh.to(DEV), True))
        assert (y2.permute(0, 3, 1, 2).cpu().double() - y).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_fused_bottleneck_tail_equals_the_two_launches_bit_for_bit():
    """conv2 3x3 + bn2 + relu + conv3 1x1 + bn3 + identity + relu of a frozen stage-1 block in one launch
    (csrc/bottleneck_tail_f32.hip): same K order, same MFMA sequence, same epilogue arithmetic as the two conv launches, the
    intermediate an fp32 tile either way -> torch.equal; image borders, several images, one tile per image row"""
    g = torch.Generator().manual_seed(77)
    for (n, h, w) in [(1, 8, 8), (2, 16, 24), (3, 8, 40), (2, 50, 64)]:
        x = torch.randn(n, h, w, 64, generator=g).to(DEV)
        idn = torch.randn(n, h, w, 256, generator=g).to(DEV)
        w2 = (torch.randn(64, 3, 3, 64, generator=g) / 24).to(DEV)
        w3 = (torch.randn(256, 1, 1, 64, generator=g) / 8).to(DEV)
        s2, b2 = (torch.rand(64, generator=g) + 0.5).to(DEV), torch.randn(64, generator=g).to(DEV)
        s3, b3 = (torch.rand(256, generator=g) + 0.5).to(DEV), torch.randn(256, generator=g).to(DEV)
        t = ops.conv2d_nhwc(x, w2, s2, b2, None, True, 1, 1)
        ref = ops.conv2d_nhwc(t, w3, s3, b3, idn, True, 1, 0)
        assert ops.bottleneck_tail_supported(x, w2, w3, idn)
        y = ops.bottleneck_tail_nhwc(x, w2, s2, b2, w3, s3, b3, idn)
        assert torch.equal(y, ref), (n, h, w, (y - ref).abs().max().item())
        # no BatchNorm operands: plain convs
        ref0 = ops.conv2d_nhwc(ops.conv2d_nhwc(x, w2, None, None, None, True, 1, 1), w3, None, None, idn, True, 1, 0)
        assert torch.equal(ops.bottleneck_tail_nhwc(x, w2, None, None, w3, None, None, idn), ref0)
    assert not ops.bottleneck_tail_supported(torch.zeros(1, 5, 5, 64, device=DEV), w2, w3, torch.zeros(1, 5, 5, 256, device=DEV))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_bn_act_fused_forward_backward(dtype):
    """fused eval-BN affine (+residual) + ReLU and its backward against the torch ops it replaces"""
    from brcnn.autograd import bn_act_autograd
    g = torch.Generator().manual_seed(31)
    for (rows, c, has_res, relu) in [(1000, 64, False, True), (777, 256, True, True), (130, 2048, True, True),
                                     (513, 128, False, False), (64, 1024, True, False)]:
        z = torch.randn(rows, c, generator=g).to(DEV, dtype)
        res = torch.randn(rows, c, generator=g).to(DEV, dtype) if has_res else None
        scale = (torch.rand(c, generator=g) + 0.5).to(DEV).requires_grad_()
        shift = torch.randn(c, generator=g).to(DEV).requires_grad_()
        go = torch.randn(rows, c, generator=g).to(DEV, dtype)
        z1 = z.clone().requires_grad_()
        r1 = res.clone().requires_grad_() if has_res else None
        out = bn_act_autograd(z1, scale, shift, r1, relu)
        out.backward(go)
        # reference in fp64 on the same (dtype-representable) inputs
        z2 = z.double().requires_grad_()
        r2 = res.double().requires_grad_() if has_res else None
        s2, h2 = scale.detach().double().requires_grad_(), shift.detach().double().requires_grad_()
        ref = z2 * s2 + h2
        if has_res:
            ref = ref + r2
        if relu:
            ref = ref.relu()
        ref.backward(go.double())
        tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8
        mag = lambda t: max(1.0, t.abs().max().item())   # noqa: E731
        assert (out.double() - ref).abs().max().item() <= tol * mag(ref) * 1.01
        assert (z1.grad.double() - z2.grad).abs().max().item() <= tol * mag(z2.grad) * 1.01
        if has_res:
            assert (r1.grad.double() - r2.grad).abs().max().item() <= tol * mag(r2.grad) * 1.01
        assert (scale.grad.double() - s2.grad).abs().max().item() <= 2e-5 * mag(s2.grad) * (rows ** 0.5)
        assert (shift.grad.double() - h2.grad).abs().max().item() <= 2e-5 * mag(h2.grad) * (rows ** 0.5)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_bn_eval_act_fused_forward_backward(dtype):
    """the same tail with the eval-mode BatchNorm parameters themselves (scale / shift formed inside the
    kernels, dgamma / dbeta returned directly) against the torch expression it replaces, in fp64"""
    from brcnn.autograd import bn_eval_act_autograd
    g = torch.Generator().manual_seed(32)
    for (rows, c, has_res, relu) in [(1000, 64, False, True), (777, 256, True, True), (130, 2048, True, True),
                                     (513, 128, False, False), (20000, 512, True, True)]:
        bn = torch.nn.BatchNorm2d(c).to(DEV).eval()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(c, generator=g))
            bn.running_mean.copy_(torch.randn(c, generator=g))
            bn.running_var.copy_(torch.rand(c, generator=g) + 0.3)
        z = torch.randn(rows, c, generator=g).to(DEV, dtype)
        res = torch.randn(rows, c, generator=g).to(DEV, dtype) if has_res else None
        go = torch.randn(rows, c, generator=g).to(DEV, dtype)
        z1 = z.clone().requires_grad_()
        r1 = res.clone().requires_grad_() if has_res else None
        out = bn_eval_act_autograd(z1, bn, r1, relu)
        out.backward(go)
        w2, b2 = bn.weight.detach().double().requires_grad_(), bn.bias.detach().double().requires_grad_()
        z2 = z.double().requires_grad_()
        r2 = res.double().requires_grad_() if has_res else None
        scale = w2 / torch.sqrt(bn.running_var.double() + bn.eps)
        ref = z2 * scale + (b2 - bn.running_mean.double() * scale)
        if has_res:
            ref = ref + r2
        if relu:
            ref = ref.relu()
        ref.backward(go.double())
        tol = 2e-6 if dtype == torch.float32 else 2.0 ** -8
        mag = lambda t: max(1.0, t.abs().max().item())   # noqa: E731
        assert (out.double() - ref).abs().max().item() <= tol * mag(ref) * 1.01
        assert (z1.grad.double() - z2.grad).abs().max().item() <= tol * mag(z2.grad) * 1.01
        if has_res:
            assert (r1.grad.double() - r2.grad).abs().max().item() <= tol * mag(r2.grad) * 1.01
        assert (bn.weight.grad.double() - w2.grad).abs().max().item() <= 2e-5 * mag(w2.grad) * (rows ** 0.5)
        assert (bn.bias.grad.double() - b2.grad).abs().max().item() <= 2e-5 * mag(b2.grad) * (rows ** 0.5)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_maxpool_backward_matches_torch(dtype):
    """max-pool 3x3/s2/p1 gradient (a trainable stem) against torch's max_pool2d autograd, ties included
    (bf16 inputs and a plateau repeat values inside windows: the first maximum takes the gradient)"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(8)
    for (n, h, w, c) in [(2, 17, 23, 64), (1, 64, 96, 8), (3, 5, 4, 128)]:
        x = torch.randn(n, h, w, c, generator=g)
        x[:, 2:4, 1:4] = 1.5                                  # a plateau of exact ties
        x = x.to(DEV, dtype).requires_grad_()
        y = ops.maxpool3x3s2_nhwc(x)
        gy = torch.randn(y.shape, generator=g).to(DEV, dtype)
        y.backward(gy)
        x2 = x.detach().float().permute(0, 3, 1, 2).contiguous().requires_grad_()
        y2 = F.max_pool2d(x2, 3, 2, 1)
        y2.backward(gy.float().permute(0, 3, 1, 2))
        assert torch.equal(y.float(), y2.permute(0, 2, 3, 1))
        ref = x2.grad.permute(0, 2, 3, 1)
        tol = 1e-6 if dtype == torch.float32 else 2.0 ** -7
        assert (x.grad.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


def test_nms_edge_vectors_on_device():
    """the float64-derived edge vectors (exact threshold, duplicates, ties, zero area) through the HIP NMS /
    soft-NMS"""
    import json
    kat = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'kat_mmcv_ops.json')))
    for e in kat['nms_edges']:
        _, inds = ops.nms(torch.tensor(e['boxes'], dtype=torch.float32, device=DEV), torch.tensor(e['scores'], device=DEV),
                          e['thr'])
        assert inds.tolist() == e['keep'], e['name']
    for e in kat['soft_edges']:
        dets, inds = ops.soft_nms(torch.tensor(e['boxes'], dtype=torch.float32, device=DEV),
                                  torch.tensor(e['scores'], device=DEV), e['thr'], 0.5, 1e-3, e['method'])
        assert inds.tolist() == e['inds'] and np.allclose(dets[:, 4].cpu().numpy(), e['scores_out'], atol=1e-7), e


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_upsample_add_autograd_and_colsum(dtype):
    """FPN top-down step (dst + nearest_upsample(src, size=dst)) forward / both gradients against
    F.interpolate + add, for the non-integer scale factors of the pyramid (13 -> 25, 7 -> 13); bias-gradient
    column sums against torch"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(12)
    for (n, hd, wd, hs, ws, c) in [(2, 25, 42, 13, 21, 256), (1, 13, 21, 7, 11, 64), (2, 100, 168, 50, 84, 8)]:
        dst = torch.randn(n, hd, wd, c, generator=g).to(DEV, dtype).requires_grad_()
        src = torch.randn(n, hs, ws, c, generator=g).to(DEV, dtype).requires_grad_()
        out = ops.upsample_nearest_add_nhwc(dst, src)
        go = torch.randn(out.shape, generator=g).to(DEV, dtype)
        out.backward(go)
        d2, s2 = dst.detach().float().requires_grad_(), src.detach().float().requires_grad_()
        ref = d2 + F.interpolate(s2.permute(0, 3, 1, 2), size=(hd, wd), mode='nearest').permute(0, 2, 3, 1)
        ref.backward(go.float())
        tol = 1e-6 if dtype == torch.float32 else 2.0 ** -7
        mag = lambda t: max(1.0, t.abs().max().item())   # noqa: E731
        assert (out.float() - ref).abs().max().item() <= tol * mag(ref)
        assert torch.equal(dst.grad, go)
        assert (src.grad.float() - s2.grad).abs().max().item() <= tol * mag(s2.grad)
    for rows, c in [(1000, 64), (33000, 256), (7, 1024), (2048, 32), (513, 2048)]:
        x = torch.randn(rows, c, generator=g).to(DEV, dtype)
        got = ops.colsum(x)
        ref = x.double().sum(0)
        assert got.dtype == torch.float32
        assert (got.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()) * rows ** 0.5


def test_grouped_conv_autograd_matches_torch():
    """forward / dgrad / wgrad of the grouped conv (ResNeXt conv2) against torch's grouped conv in fp64"""
    import torch.nn.functional as F
    from brcnn.autograd import grouped_conv_autograd
    gen = torch.Generator().manual_seed(41)
    for (n, c, h, w, groups, stride) in [(2, 128, 20, 30, 32, 1), (1, 256, 17, 23, 32, 2), (2, 256, 9, 14, 64, 1),
                                         (1, 512, 10, 12, 64, 2), (1, 1024, 7, 9, 32, 1)]:
        x = torch.randn(n, c, h, w, generator=gen)
        wt = torch.randn(c, c // groups, 3, 3, generator=gen) / np.sqrt(9 * c / groups)
        xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
        ref = F.conv2d(xr, wr, None, stride, 1, groups=groups)
        go = torch.randn(ref.shape, generator=gen)
        ref.backward(go.double())
        xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_()
        wg = wt.to(DEV).requires_grad_()
        y = grouped_conv_autograd(xg, wg, groups, stride, 1)
        y.backward(go.permute(0, 2, 3, 1).contiguous().to(DEV))
        tol = lambda t: 3e-5 * max(1.0, t.abs().max().item())    # noqa: E731
        assert (y.detach().permute(0, 3, 1, 2).cpu().double() - ref.detach()).abs().max().item() < tol(ref)
        assert (xg.grad.permute(0, 3, 1, 2).cpu().double() - xr.grad).abs().max().item() < tol(xr.grad)
        assert wg.grad.shape == wt.shape
        assert (wg.grad.cpu().double() - wr.grad).abs().max().item() < 2e-4 * max(1.0, wr.grad.abs().max().item())


@pytest.mark.parametrize('dtype,relu', [(torch.float32, True), (torch.float32, False), (torch.bfloat16, True)])
def test_groupnorm_multi_backward_matches_torch(dtype, relu):
    """HIP GroupNorm(+ReLU) backward over two NHWC segments against torch's fp64 group_norm
    autograd per (segment): dx, dgamma, dbeta"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(11)
    B, C, G = 2, 256, 32
    sizes = [(13, 21), (7, 11)]
    xs = [torch.randn(B, h, w, C, generator=g) * 1.5 + 0.3 for h, w in sizes]
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    gos = [torch.randn(B, h, w, C, generator=g) for h, w in sizes]
    if dtype == torch.bfloat16:
        xs = [t.bfloat16().float() for t in xs]
        gos = [t.bfloat16().float() for t in gos]
    # reference
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    refs = []
    for x, go in zip(xs, gos):
        xr = x.double().permute(0, 3, 1, 2).requires_grad_()
        y = F.group_norm(xr, G, gr, br, 1e-5)
        y = y.relu() if relu else y
        y.backward(go.double().permute(0, 3, 1, 2))
        refs.append((y.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1)))
    # HIP
    from brcnn.autograd import GroupNormNHWCFunction
    x_cat = torch.cat([x.reshape(-1, C) for x in xs]).to(dtype).to(DEV).requires_grad_()
    gg, bg = gamma.to(DEV).requires_grad_(), beta.to(DEV).requires_grad_()
    y = GroupNormNHWCFunction.apply(x_cat, gg, bg, G, B, tuple(sizes), 1e-5, relu)
    y.backward(torch.cat([go.reshape(-1, C) for go in gos]).to(dtype).to(DEV))
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    r0 = 0
    for (h, w), (yr, dxr) in zip(sizes, refs):
        n = B * h * w
        yy = y[r0:r0 + n].detach().float().cpu().view(B, h, w, C).double()
        dx = x_cat.grad[r0:r0 + n].float().cpu().view(B, h, w, C).double()
        assert (yy - yr).abs().max().item() <= tol * max(yr.abs().max().item(), 1.0)
        assert (dx - dxr).abs().max().item() <= tol * max(dxr.abs().max().item(), 1.0), (dtype, relu)
        r0 += n
    assert (gg.grad.double().cpu() - gr.grad).abs().max().item() <= tol * gr.grad.abs().max().item() + 1e-4
    assert (bg.grad.double().cpu() - br.grad).abs().max().item() <= tol * br.grad.abs().max().item() + 1e-4


def test_batched_nms_images_fused_equals_torch_formulation():
    """brcnn_nms_prepare / brcnn_nms_collect (two launches around the segmented NMS) against the
    ~30-op torch formulation of the same steps: compaction, max-coordinate offsets, ranges,
    survivor gather -- bit for bit, incl. an empty image and an all-valid one"""
    from brcnn.postprocess import batched_nms_images
    g = torch.Generator().manual_seed(21)
    B, T = 4, 1500
    boxes = util.clustered_boxes(B * T, n_clusters=60, seed=21).view(B, T, 4)
    scores = torch.rand(B, T, generator=g)
    ids = torch.randint(0, 5, (B, T), generator=g)
    valid = torch.rand(B, T, generator=g) > 0.3
    valid[1] = False
    valid[2] = True
    args = [t.to(DEV) for t in (boxes, scores, ids, valid)]
    for max_keep in (100, -1):
        a = batched_nms_images(*args, 0.7, max_keep, fused=True)
        b = batched_nms_images(*args, 0.7, max_keep, fused=False)
        assert torch.equal(a[2], b[2]) and int(a[2][1]) == 0 and int(a[2].sum()) > 50
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), max_keep


def test_rcnn_decode_fused_equals_torch_chain():
    """brcnn_rcnn_decode (score fusion + per-class delta2bbox + clip + rescale + threshold in one
    launch) against the torch chain of ProbRoIHead.simple_test_padded on the same inputs"""
    from brcnn.core import DeltaXYWHBBoxCoder
    g = torch.Generator().manual_seed(31)
    B, K, C = 3, 200, 4
    dets = torch.cat([util.rand_boxes(B * K, seed=31), torch.rand(B * K, 1, generator=g)], 1).view(B, K, 5)
    num = torch.tensor([200, 57, 0], dtype=torch.int32)
    cls_score = torch.randn(B * K, C + 1, generator=g) * 2
    bbox_pred = torch.randn(B * K, 4 * C, generator=g) * 0.5
    bbox_pred[0, :4] = torch.tensor([0., 0., 60., -60.])        # clamp edges
    max_shape = torch.tensor([[800., 1333.], [750., 1200.], [640., 640.]])
    sf = torch.tensor([[1.1, 1.2, 1.1, 1.2], [1., 1., 1., 1.], [0.5, 0.5, 0.5, 0.5]])
    coder = DeltaXYWHBBoxCoder(target_means=(0., 0., 0., 0.), target_stds=(0.1, 0.1, 0.2, 0.2))
    d = lambda t: t.to(DEV)   # noqa: E731
    probs = d(cls_score).softmax(1)
    bb, sc, lb, va = ops.rcnn_decode(probs, d(bbox_pred), d(dets), d(num), d(max_shape), d(sf), C, 0.05,
                                     coder.means, coder.stds)
    # torch chain
    scores = ((probs * d(dets)[..., 4].reshape(-1, 1)) ** 0.5).view(B, K, C + 1)
    boxes = coder.decode(d(dets)[..., :4], d(bbox_pred).view(B, K, 4 * C), max_shape=d(max_shape))
    boxes = (boxes.view(B, K, C, 4) / d(sf).view(B, 1, 1, 4)).reshape(B, K * C, 4)
    row_ok = torch.arange(K, device=DEV)[None, :] < d(num)[:, None]
    s = scores[..., :C]
    valid = ((s > 0.05) & row_ok[..., None]).reshape(B, K * C)
    assert torch.equal(sc, s.reshape(B, K * C)) and torch.equal(va, valid)
    assert torch.equal(lb, torch.arange(C, device=DEV).view(1, 1, C).expand(B, K, C).reshape(B, K * C))
    assert util.ulp_diff(bb.cpu(), boxes.cpu()).max().item() <= 2, util.ulp_diff(bb.cpu(), boxes.cpu()).max()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_multi_level_conv_autograd_matches_torch(dtype):
    """conv2d_nhwc_multi_autograd: one forward / dgrad / wgrad launch over several pyramid levels
    that share the weights (the training form of the RPN tower) against torch conv2d per level with
    the weight gradients summed -- maps wider and narrower than the wgrad tile step (incremental
    row decode vs full decode), segment boundaries inside a tile"""
    import torch.nn.functional as F
    from brcnn.autograd import conv2d_nhwc_multi_autograd
    g = torch.Generator().manual_seed(41)
    B, Cin, Cout = 2, 64, 128
    sizes = [(37, 70), (19, 35), (10, 18), (5, 9)]
    xs = [torch.randn(B, Cin, h, w, generator=g) for h, w in sizes]
    wt = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    gos = [torch.randn(B, Cout, h, w, generator=g) for h, w in sizes]
    if dtype == torch.bfloat16:
        xs = [t.bfloat16().float() for t in xs]
        gos = [t.bfloat16().float() for t in gos]
        wt = wt.bfloat16().float()
    wr, br = wt.double().requires_grad_(), bias.double().requires_grad_()
    xr = [t.double().requires_grad_() for t in xs]
    for x, go in zip(xr, gos):
        F.conv2d(x, wr, br, 1, 1).backward(go.double())
    x_cat = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cin) for t in xs]).to(dtype).to(DEV).requires_grad_()
    wg, bg = wt.to(DEV).requires_grad_(), bias.to(DEV).requires_grad_()
    y = conv2d_nhwc_multi_autograd(x_cat, wg, bg, B, tuple(sizes), 1, 1)
    y.backward(torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cout) for t in gos]).to(dtype).to(DEV))
    tol = 5e-5 if dtype == torch.float32 else 1.5e-2
    def rel(a, ref):
        return (a.double().cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
    r0 = 0
    for (h, w), x in zip(sizes, xr):
        n = B * h * w
        dx = x_cat.grad[r0:r0 + n].float().view(B, h, w, Cin).permute(0, 3, 1, 2)
        assert rel(dx, x.grad) < tol, (dtype, h, w)
        r0 += n
    assert rel(wg.grad, wr.grad) < tol and rel(bg.grad, br.grad) < tol


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_roi_extract_16bit_training_path(dtype):
    """the extractor on a 16-bit pyramid in training (no widening casts): features and the pyramid gradient equal the
    fp32 path on the same (16-bit-representable) values, rounded once to the 16-bit dtype"""
    from brcnn.autograd import roi_extract_autograd
    g = torch.Generator().manual_seed(15)
    strides = [8, 16, 32, 64]
    sizes = [(50, 84), (25, 42), (13, 21), (7, 11)]
    B, C, K = 2, 256, 300
    feats = [torch.randn(B, h, w_, C, generator=g).to(dtype) for h, w_ in sizes]
    rois = util.rand_rois(K, B, 672., 400., seed=7, min_size=4., max_size=900.)
    go = torch.randn(K, 7, 7, C, generator=g).to(dtype)
    res = {}
    for dt in (torch.float32, dtype):
        fg = [f.to(DEV, dt).requires_grad_() for f in feats]
        out = roi_extract_autograd(fg, rois.to(DEV), 7, strides, 56, 0)
        assert out.dtype == dt
        out.backward(go.to(DEV, dt))
        assert all(f.grad.dtype == dt for f in fg)
        res[dt] = (out.detach().float(), [f.grad.float() for f in fg])
    assert torch.equal(res[dtype][0], res[torch.float32][0].to(dtype).float())
    for a, b in zip(res[dtype][1], res[torch.float32][1]):
        # same fp32 accumulation order in both runs; the 16-bit run rounds once at the store
        assert torch.equal(a, b.to(dtype).float())



@pytest.mark.parametrize('cfg', [
    # N, H, W, Cin, Cout, k, stride, pad, residual
    (2, 50, 84, 256, 256, 3, 1, 1, False),        # 72 K tiles of 32
    (2, 50, 84, 96, 256, 3, 1, 1, True),          # 27 K tiles (odd), residual
    (3, 25, 42, 160, 512, 1, 1, 0, False),        # 5 K tiles, ragged row tiles
    (2, 51, 85, 64, 256, 3, 2, 1, False),         # stride 2
    (8, 50, 84, 256, 1024, 1, 1, 0, True),        # 528 tiles: the chained stream-K schedule in the heuristic
])
def test_f32_eight_phase_kernel_is_bit_identical(cfg):
    """the fp32 256 x 256 eight-phase kernel (conv_pp_f32.hip) keeps the two-buffer kernel's K order per accumulator:
    identical results (plain and stream-K launches, K-tile count parities, strides, residual), launch after launch"""
    from brcnn import lib as _lib
    L = _lib.load()
    n, h, w_, ci, co, k, stride, pad, res = cfg
    g = torch.Generator().manual_seed(35)
    x = torch.randn(n, h, w_, ci, generator=g).to(DEV)
    w = (torch.randn(co, k, k, ci, generator=g) * 0.05).to(DEV)
    sc = (torch.rand(co, generator=g) + 0.5).to(DEV)
    sh = torch.randn(co, generator=g).to(DEV)
    ho, wo = ops.conv_out_size(h, w_, k, k, stride, pad)
    r = torch.randn(n, ho, wo, co, generator=g).to(DEV) if res else None
    try:
        assert L.brcnn_conv_set_tile(-2, 0) == 0
        ref = ops.conv2d_nhwc(x, w, sc, sh, r, True, stride, pad)
        for rows in (256, 128):                  # both tile heights forced
            assert L.brcnn_conv_set_tile(-2, rows) == 0
            for sk in (-3, -5, -4):
                assert L.brcnn_conv_set_tile_bf16(sk) == 0
                for rep in range(2):
                    out = ops.conv2d_nhwc(x, w, sc, sh, r, True, stride, pad)
                    assert torch.equal(out, ref), (rows, sk, rep, (out - ref).abs().max().item())
    finally:
        L.brcnn_conv_set_tile(-2, 1)
        L.brcnn_conv_set_tile_bf16(-4)


def test_f32_eight_phase_kernel_data_gradient():
    """the zero-stuffed data gradient of a stride-2 conv (DIL form) and the five-level launch on the fp32 eight-phase kernel"""
    from brcnn import lib as _lib
    from brcnn.autograd import conv2d_nhwc_autograd
    L = _lib.load()
    g = torch.Generator().manual_seed(36)
    sizes = [(40, 64), (20, 32), (10, 16), (5, 8), (3, 4)]
    B, C = 2, 256
    xc = torch.cat([torch.randn(B, h, w, C, generator=g).reshape(-1, C) for h, w in sizes], 0).to(DEV)
    wt = (torch.randn(C, 3, 3, C, generator=g) / 48).to(DEV)
    dy = torch.randn(2, 25, 42, 256, generator=g).to(DEV)
    w2 = (torch.randn(256, 256, 3, 3, generator=g) / 48).to(DEV)
    try:
        outs = {}
        for mode in (0, 256, 128):
            assert L.brcnn_conv_set_tile(-2, mode) == 0
            y, _ = ops.conv2d_nhwc_multi(xc, wt, B, sizes, None, None, None, False, 1, 1)
            x = torch.randn(2, 50, 84, 256, generator=torch.Generator().manual_seed(37)).to(DEV).requires_grad_(True)
            z = conv2d_nhwc_autograd(x, w2.clone().requires_grad_(True), None, 2, 1)
            z.backward(dy)
            outs[mode] = (y, x.grad.clone())
        for mode in (256, 128):
            assert torch.equal(outs[0][0], outs[mode][0]) and torch.equal(outs[0][1], outs[mode][1]), mode
    finally:
        L.brcnn_conv_set_tile(-2, 1)


def test_f32_split_k_option_is_reproducible_and_within_round_off():
    """the optional split-K of few-tile launches (brcnn_conv_set_tile_bf16(-9); off by default because its result is
    not the unsplit chain's bits): pieces of a tile summed at the end in a fixed order -- the same bits run to run,
    the unsplit result to fp32 round-off, and untouched launches (enough tiles) keep their bits"""
    from brcnn import lib as _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(41)
    x = torch.randn(8, 25, 42, 512, generator=g).to(DEV)
    w = (torch.randn(512, 3, 3, 512, generator=g) / 68).to(DEV)
    xb = torch.randn(8, 50, 84, 256, generator=g).to(DEV)
    wb = (torch.randn(256, 3, 3, 256, generator=g) / 48).to(DEV)
    sc = (torch.rand(512, generator=g) + 0.5).to(DEV)
    sh = torch.randn(512, generator=g).to(DEV)
    try:
        ref = ops.conv2d_nhwc(x, w, sc, sh, None, True, 1, 1)
        refb = ops.conv2d_nhwc(xb, wb, None, None, None, False, 1, 1)
        assert L.brcnn_conv_set_tile_bf16(-9) == 0
        a = ops.conv2d_nhwc(x, w, sc, sh, None, True, 1, 1)
        b = ops.conv2d_nhwc(x, w, sc, sh, None, True, 1, 1)
        assert torch.equal(a, b)
        assert not torch.equal(a, ref)          # the layer IS split (66 x 2 tiles of 128 x 256 on 256 CUs)
        assert (a - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
        assert torch.equal(ops.conv2d_nhwc(xb, wb, None, None, None, False, 1, 1), refb)
    finally:
        L.brcnn_conv_set_tile_bf16(-8)


def test_roi_extract_visiting_order_and_rows_per_wave_keep_the_bits():
    """brcnn_roi_extract_forward_ordered: the band-ordered visit (counting sort by image / level / row band, from 12 288
    RoIs on by itself; forced here) and the all-rows-per-wave form (hook 17) give the unordered one-row-per-wave kernel's output and levels
    bit for bit; the order workspace is the caller's"""
    import ctypes
    from brcnn import lib
    L = lib.load()
    B = 3
    strides = [8, 16, 32, 64, 128]
    sizes = [(50, 84), (25, 42), (13, 21), (7, 11), (4, 6)]
    g = torch.Generator().manual_seed(12)
    feats = [torch.randn(B, h, w, 256, generator=g).to(DEV) for h, w in sizes]
    rois = util.rand_rois(7000, B, 672., 400., seed=4, min_size=4., max_size=500.).to(DEV)     # unsorted images, tiny and huge RoIs
    try:
        L.brcnn_roi_align_set_exact(30)      # the footprint form itself (round 5's prepared records off)
        L.brcnn_roi_align_set_exact(11); L.brcnn_roi_align_set_exact(20)
        ref, lref = ops.roi_extract(feats, rois, 7, strides, 56, 0)
        for rpw in (11, 17):
            for od in (20, 22):
                L.brcnn_roi_align_set_exact(rpw); L.brcnn_roi_align_set_exact(od)
                out, lv = ops.roi_extract(feats, rois, 7, strides, 56, 0)
                assert torch.equal(out, ref) and torch.equal(lv, lref), (rpw, od)
        # raw C ABI: the order scratch is written (a permutation of 0..n-1) only when given
        L.brcnn_roi_align_set_exact(22)
        n = rois.shape[0]
        order = torch.full((n,), -1, dtype=torch.int32, device=DEV)
        out = torch.empty_like(ref)
        ptrs = (ctypes.c_void_p * 5)(*[f.data_ptr() for f in feats])
        hs = (ctypes.c_int * 5)(*[h for h, _ in sizes]); ws = (ctypes.c_int * 5)(*[w for _, w in sizes])
        sc = (ctypes.c_float * 5)(*[1.0 / s for s in strides])
        st = L.brcnn_roi_extract_forward_ordered(ptrs, hs, ws, sc, 5, rois.data_ptr(), out.data_ptr(), None, B, 256, n, 7, 7, 0,
                                                 56.0, 0, order.data_ptr(), lib.stream_handle())
        assert st == 0
        torch.cuda.synchronize()
        assert torch.equal(out, ref) and torch.equal(order.sort().values, torch.arange(n, dtype=torch.int32, device=DEV))
    finally:
        L.brcnn_roi_align_set_exact(10); L.brcnn_roi_align_set_exact(21); L.brcnn_roi_align_set_exact(0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_roi_extract_prepared_records_equal_the_footprint_form(dtype):
    """round 5: level mapping / geometry / axis weights once per RoI (roi_prep_kernel -> a 576-byte record), one wave
    per bin row starting from the record.  Same arithmetic on the bin rows of the streaming form: bit-identical to the
    footprint kernel; the other bin rows (bins beyond ~3 px: forced here with RoIs far larger than their level expects)
    run the reference's sample loop -- equal to fp32 round-off.  Levels identical; unsorted images, tiny, huge, degenerate
    and out-of-image RoIs; with and without the visiting order; NULL scratch falls back; against the C oracle."""
    import ctypes
    from oracle import orc
    from brcnn import lib
    L = lib.load()
    B = 3
    strides = [8, 16, 32, 64]
    sizes = [(50, 84), (25, 42), (13, 21), (7, 11)]
    g = torch.Generator().manual_seed(12)
    feats32 = [torch.randn(B, h, w, 256, generator=g) for h, w in sizes]
    feats = [f.to(DEV, dtype) for f in feats32]
    rois = util.rand_rois(5000, B, 672., 400., seed=4, min_size=4., max_size=500.)
    extra = torch.tensor([[0, -50., -40., 700., 500.], [1, 10., 10., 10., 10.], [2, 660., 390., 672., 400.],
                          [1, 300., 200., 280., 180.], [0, 0., 0., 671., 2.], [2, 0., 0., 3., 399.]])
    rois = torch.cat([rois, extra]).to(DEV)
    try:
        L.brcnn_roi_align_set_exact(30)
        ref, lref = ops.roi_extract(feats, rois, 7, strides, 56, 0)
        for od in (20, 22):
            L.brcnn_roi_align_set_exact(31); L.brcnn_roi_align_set_exact(od)
            out, lv = ops.roi_extract(feats, rois, 7, strides, 56, 0)
            assert torch.equal(lv, lref)
            differ = (out != ref).flatten(1).any(1)
            assert differ.float().mean().item() < 0.2, differ.float().mean().item()      # most RoIs: the streaming form, same bits
            tol = 2e-6 if dtype == torch.float32 else 8e-3
            err = (out.float() - ref.float()).abs().max().item()
            assert err <= tol * max(1.0, ref.float().abs().max().item()), err
        # against the C oracle (fp32 pyramid), per level as SingleRoIExtractor does
        if dtype == torch.float32:
            cpu_rois = rois.cpu()
            for lvl in range(4):
                idx = (lref.cpu() == lvl).nonzero().squeeze(1)[:300]
                want = orc.roi_align_forward(feats32[lvl].permute(0, 3, 1, 2).contiguous(), cpu_rois[idx], 7, 1.0 / strides[lvl], 0,
                                             'avg', True)
                got = out[idx.to(DEV)].permute(0, 3, 1, 2).cpu()
                assert (got - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item()), lvl
        # raw C ABI: NULL scratch = the footprint form; a scratch that is too small or misaligned is refused / ignored
        n = rois.shape[0]
        ptrs = (ctypes.c_void_p * 4)(*[f.data_ptr() for f in feats])
        hs = (ctypes.c_int * 4)(*[h for h, _ in sizes]); ws = (ctypes.c_int * 4)(*[w for _, w in sizes])
        sc = (ctypes.c_float * 4)(*[1.0 / s for s in strides])
        dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 3}[dtype]
        o2 = torch.empty_like(ref)
        st = L.brcnn_roi_extract_forward_prepared(ptrs, hs, ws, sc, 4, rois.data_ptr(), o2.data_ptr(), None, B, 256, n, 7, 7, 0,
                                                  56.0, dt, None, None, 0, lib.raw_stream_handle())
        torch.cuda.synchronize()
        assert st == 0 and torch.equal(o2, ref)
        nb = L.brcnn_roi_extract_prep_workspace_bytes(n)
        assert nb == n * 576
        L.brcnn_roi_align_set_exact(30)
        assert L.brcnn_roi_extract_prep_workspace_bytes(n) == 0         # off (the default): the caller need not allocate
        L.brcnn_roi_align_set_exact(31)
        wsb = torch.empty(nb + 256, dtype=torch.uint8, device=DEV)
        st = L.brcnn_roi_extract_forward_prepared(ptrs, hs, ws, sc, 4, rois.data_ptr(), o2.data_ptr(), None, B, 256, n, 7, 7, 0,
                                                  56.0, dt, None, wsb.data_ptr() + 4, nb, lib.raw_stream_handle())
        assert st == -22                                         # misaligned scratch
        assert L.brcnn_roi_extract_order_min_rois() == 12288
        # no RoIs at all
        e, le = ops.roi_extract(feats, rois[:0], 7, strides, 56, 0)
        assert e.shape == (0, 7, 7, 256) and le.numel() == 0
    finally:
        L.brcnn_roi_align_set_exact(21); L.brcnn_roi_align_set_exact(30)


@pytest.mark.parametrize('cfg', [
    # n, h, w, ci, co, k, stride, pad, residual
    (2, 100, 168, 128, 128, 3, 1, 1, False),     # stage-2 3x3: 132 tiles of 256 x 128, even K-tile count
    (2, 100, 168, 256, 128, 3, 2, 1, False),     # strided
    (2, 101, 83, 512, 128, 1, 1, 0, False),      # 1x1, ragged M (last row tile partly out of range)
    (1, 64, 96, 96, 384, 3, 1, 1, True),         # Cout = 3 x 128 (three column tiles), Cin % 32 == 0 only, residual, odd K tiles
])
def test_f32_eight_phase_256x128_tile_is_bit_identical(cfg):
    """the 256 x 128 form of the fp32 eight-phase kernel (four row groups x two column strips; layers with 128 output
    channels) against the 64 x 64 two-buffer kernel: identical bits, plain and stream-K launches"""
    from brcnn import lib as _lib
    L = _lib.load()
    n, h, w_, ci, co, k, stride, pad, res = cfg
    g = torch.Generator().manual_seed(43)
    x = torch.randn(n, h, w_, ci, generator=g).to(DEV)
    w = (torch.randn(co, k, k, ci, generator=g) * 0.05).to(DEV)
    sc = (torch.rand(co, generator=g) + 0.5).to(DEV)
    sh = torch.randn(co, generator=g).to(DEV)
    ho, wo = ops.conv_out_size(h, w_, k, k, stride, pad)
    r = torch.randn(n, ho, wo, co, generator=g).to(DEV) if res else None
    try:
        assert L.brcnn_conv_set_tile(-3, 0) == 0
        ref = ops.conv2d_nhwc(x, w, sc, sh, r, True, stride, pad)
        y64 = ref.double()
        xd, wd = x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2)
        want = torch.nn.functional.conv2d(xd, wd, None, stride, pad).permute(0, 2, 3, 1) * sc.double() + sh.double()
        if r is not None:
            want = want + r.double()
        assert (y64 - want.relu()).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
        assert L.brcnn_conv_set_tile(-3, 2) == 0
        for sk in (-3, -5, -4):
            assert L.brcnn_conv_set_tile_bf16(sk) == 0
            for rep in range(2):
                out = ops.conv2d_nhwc(x, w, sc, sh, r, True, stride, pad)
                assert torch.equal(out, ref), (sk, rep, (out - ref).abs().max().item())
    finally:
        L.brcnn_conv_set_tile(-3, 1)
        L.brcnn_conv_set_tile_bf16(-4)
