"""-m gpu: the fp16 mode (`fp16 = dict(loss_scale=512.)` of the x101 recipe and BASELINE configs[4];
mmdet/apis/train.py:115-119): v_mfma_f32_32x32x16_f16 instantiations of the 16-bit conv / wgrad kernels
and the fp16 element type of every NHWC kernel, against fp64 references of the same fp16-rounded
operands (fp32 accumulation + ONE rounding: |err| <= 2^-11 |ref|), and the detector / train step in
fp16 with static loss scaling against the fp32 run."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import brcnn  # noqa: F401
from brcnn import Config, blocks, build_detector, ops
from brcnn.autograd import bn_eval_act_autograd, conv2d_nhwc_autograd, groupnorm_nhwc_autograd
from tests import util
from tests.test_host_cpu import CFG

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
H = torch.float16
HALF_ULP = 2.0 ** -11


def _rne_close(y, ref, acc_tol=2e-5, ulp=HALF_ULP):
    mag = max(1.0, ref.abs().max().item())
    bad = (y - ref).abs() > ulp * ref.abs() * 1.001 + acc_tol * mag
    return not bad.any().item()


def _h(x):
    return x.to(H).float()


@pytest.mark.parametrize('cfg', [
    # (N, Cin, H, W, Cout, k, stride, pad, scale, residual, relu, out_f32)
    (2, 64, 24, 40, 64, 1, 1, 0, True, False, True, False),
    (2, 64, 24, 40, 256, 1, 1, 0, True, True, True, False),
    (1, 128, 30, 31, 128, 3, 1, 1, True, False, True, False),
    (2, 256, 25, 42, 256, 3, 2, 1, False, False, False, False),
    (8, 256, 50, 84, 256, 3, 1, 1, False, False, True, False),      # the 8-wave / 128x128 tiles
    (1, 256, 13, 21, 54, 3, 1, 1, False, False, False, True),
    (300, 256, 7, 7, 130, 7, 1, 0, False, False, True, True),
])
def test_conv2d_nhwc_f16(cfg):
    n, cin, h, w, cout, k, stride, pad, has_scale, has_res, relu, out_f32 = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 1000)
    x = _h(torch.randn(n, cin, h, w, generator=g))
    wt = _h(torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k))
    scale = torch.rand(cout, generator=g) + 0.5 if has_scale else None
    shift = torch.randn(cout, generator=g)
    ho, wo = ops.conv_out_size(h, w, k, k, stride, pad)
    res = _h(torch.randn(n, cout, ho, wo, generator=g)) if has_res else None
    ref = F.conv2d(x.double(), wt.double(), None, stride, pad)
    if has_scale:
        ref = ref * scale.double().view(1, -1, 1, 1)
    ref = ref + shift.double().view(1, -1, 1, 1)
    if has_res:
        ref = ref + res.double()
    if relu:
        ref = ref.relu()
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV, H)
    wg = wt.permute(0, 2, 3, 1).contiguous().to(DEV, H)
    rg = res.permute(0, 2, 3, 1).contiguous().to(DEV, H) if has_res else None
    y = ops.conv2d_nhwc(xg, wg, scale.to(DEV) if has_scale else None, shift.to(DEV), rg, relu, stride, pad,
                        out_f32=out_f32)
    assert y.dtype == (torch.float32 if out_f32 else H)
    y = y.permute(0, 3, 1, 2).cpu().double()
    if out_f32:
        assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    else:
        assert _rne_close(y, ref)


def test_small_kernels_f16():
    g = torch.Generator().manual_seed(9)
    x = _h(torch.randn(2, 64, 33, 47, generator=g))
    y = ops.maxpool3x3s2_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV, H))
    assert y.dtype == H and torch.equal(y.float().permute(0, 3, 1, 2).cpu(), F.max_pool2d(x, 3, 2, 1))
    x = _h(torch.randn(2, 256, 25, 42, generator=g) * 2 + 0.3)
    gamma, beta = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    ref = F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5).relu()
    y = ops.groupnorm_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV, H), gamma.to(DEV), beta.to(DEV), 32, 1e-5, True)
    assert y.dtype == H and _rne_close(y.float().permute(0, 3, 1, 2).cpu().double(), ref)
    for (hd, wd, hs, ws) in [(50, 84, 25, 42), (13, 21, 7, 11)]:
        d = _h(torch.randn(2, 256, hd, wd, generator=g))
        s = _h(torch.randn(2, 256, hs, ws, generator=g))
        ref = (d + F.interpolate(s, size=(hd, wd), mode='nearest')).to(H)
        dg = d.permute(0, 2, 3, 1).contiguous().to(DEV, H)
        ops.upsample_nearest_add_nhwc_(dg, s.permute(0, 2, 3, 1).contiguous().to(DEV, H))
        assert torch.equal(dg.permute(0, 3, 1, 2).cpu(), ref)
    # stem (7x7/s2 on the NCHW image) and RoI extraction on fp16 maps
    img = torch.randn(2, 3, 64, 96, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) / 12
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    ref = (F.conv2d(_h(img).double(), _h(wt).double(), None, 2, 3) * sc.double().view(1, -1, 1, 1) +
           sh.double().view(1, -1, 1, 1)).relu()
    y = ops.stem7x7s2_nchw(img.to(DEV), ops.pack_stem_weight(wt.to(DEV), H), sc.to(DEV), sh.to(DEV), True)
    assert y.dtype == H and _rne_close(y.float().permute(0, 3, 1, 2).cpu().double(), ref)
    yp = ops.stem7x7s2_pool_nchw(img.to(DEV), ops.pack_stem_pool_weight(wt.to(DEV), H), sc.to(DEV), sh.to(DEV))
    assert yp.dtype == H and _rne_close(yp.float().permute(0, 3, 1, 2).cpu().double(), F.max_pool2d(ref, 3, 2, 1))
    strides = [4, 8, 16, 32]
    feats = [_h(torch.randn(2, 200 // s, 304 // s, 256, generator=g)).to(DEV) for s in strides]
    rois = util.rand_rois(300, 2, 304.0, 200.0, seed=3).to(DEV)
    o32, l32 = ops.roi_extract(feats, rois, 7, strides, 56, 0)
    o16, l16 = ops.roi_extract([f.to(H) for f in feats], rois, 7, strides, 56, 0)
    assert o16.dtype == H and torch.equal(l16, l32) and torch.equal(o16, o32.to(H))


@pytest.mark.parametrize('cfg', [(2, 64, 20, 28, 128, 3, 1, 1), (2, 128, 21, 17, 64, 1, 1, 0), (1, 256, 16, 24, 256, 3, 2, 1),
                                 (4, 256, 50, 84, 256, 3, 1, 1)])
def test_conv_autograd_f16(cfg):
    """forward, data gradient and weight gradient (fp32 accumulation / fp32 dW) of a trainable conv in fp16"""
    n, cin, h, w, cout, k, stride, pad = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = _h(torch.randn(n, h, w, cin, generator=g))
    wt = _h(torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k))
    xg = x.to(DEV, H).requires_grad_()
    wg = wt.to(DEV).requires_grad_()                         # fp32 master weight, cast per step
    y = conv2d_nhwc_autograd(xg, wg, None, stride, pad)
    assert y.dtype == H
    gy = _h(torch.randn(y.shape, generator=g))
    y.backward(gy.to(DEV, H))
    x2 = x.double().permute(0, 3, 1, 2).requires_grad_()
    w2 = wt.double().requires_grad_()
    y2 = F.conv2d(x2, w2, None, stride, pad)
    y2.backward(gy.double().permute(0, 3, 1, 2))
    assert _rne_close(y.float().cpu().double().permute(0, 3, 1, 2), y2.detach())
    assert _rne_close(xg.grad.float().cpu().double().permute(0, 3, 1, 2), x2.grad, acc_tol=5e-5)
    assert (wg.grad.cpu().double() - w2.grad).abs().max().item() <= 1e-4 * max(1.0, w2.grad.abs().max().item())


def test_norm_layers_autograd_f16():
    """eval-BN(+residual+ReLU) and GroupNorm(+ReLU) forward / backward on fp16 activations"""
    g = torch.Generator().manual_seed(3)
    bn = torch.nn.BatchNorm2d(256).to(DEV).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(256, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(256, generator=g))
        bn.running_mean.copy_(torch.randn(256, generator=g))
        bn.running_var.copy_(torch.rand(256, generator=g) + 0.3)
    z = _h(torch.randn(1000, 256, generator=g))
    res = _h(torch.randn(1000, 256, generator=g))
    go = _h(torch.randn(1000, 256, generator=g))
    z1, r1 = z.to(DEV, H).requires_grad_(), res.to(DEV, H).requires_grad_()
    out = bn_eval_act_autograd(z1, bn, r1, True)
    out.backward(go.to(DEV, H))
    z2, r2 = z.double().requires_grad_(), res.double().requires_grad_()
    w2, b2 = bn.weight.detach().cpu().double().requires_grad_(), bn.bias.detach().cpu().double().requires_grad_()
    sc = w2 / torch.sqrt(bn.running_var.cpu().double() + bn.eps)
    ref = (z2 * sc + (b2 - bn.running_mean.cpu().double() * sc) + r2).relu()
    ref.backward(go.double())
    assert _rne_close(out.float().cpu().double(), ref.detach())
    assert _rne_close(z1.grad.float().cpu().double(), z2.grad) and _rne_close(r1.grad.float().cpu().double(), r2.grad)
    assert (bn.weight.grad.cpu().double() - w2.grad).abs().max().item() <= 1e-3 * max(1.0, w2.grad.abs().max().item())
    assert (bn.bias.grad.cpu().double() - b2.grad).abs().max().item() <= 1e-3 * max(1.0, b2.grad.abs().max().item())
    x = _h(torch.randn(2, 25, 42, 256, generator=g) * 2 + 0.3)
    gamma = (torch.rand(256, generator=g) + 0.5).to(DEV).requires_grad_()
    beta = torch.randn(256, generator=g).to(DEV).requires_grad_()
    xg = x.to(DEV, H).requires_grad_()
    y = groupnorm_nhwc_autograd(xg, gamma, beta, 32, 1e-5, True)
    gy = _h(torch.randn(y.shape, generator=g))
    y.backward(gy.to(DEV, H))
    x2 = x.double().permute(0, 3, 1, 2).requires_grad_()
    g2, b2 = gamma.detach().cpu().double().requires_grad_(), beta.detach().cpu().double().requires_grad_()
    y2 = F.group_norm(x2, 32, g2, b2, 1e-5).relu()
    y2.backward(gy.double().permute(0, 3, 1, 2))
    assert _rne_close(y.float().cpu().double().permute(0, 3, 1, 2), y2.detach())
    assert _rne_close(xg.grad.float().cpu().double().permute(0, 3, 1, 2), x2.grad, acc_tol=1e-4)
    assert (gamma.grad.cpu().double() - g2.grad).abs().max().item() <= 2e-3 * max(1.0, g2.grad.abs().max().item())


def test_detector_f16_close_to_fp32():
    """whole detector in fp16 mode (no 8-bit-significand loss as in bf16: much closer to fp32)"""
    cfg = Config.fromfile(CFG)
    img, metas, _, _ = util.demo_inputs(2, 128, 192, seed=10)
    out = {}
    try:
        for mode in ('f32', 'f16'):
            m = build_detector(cfg.model)
            m.load_state_dict(util.seeded_state_dict(m, seed=10))
            m = m.to(DEV).eval()
            m.set_compute_dtype(mode)
            with torch.no_grad():
                feats = m.extract_feat_nhwc(img.to(DEV))
                out[mode] = ([f.float() for f in feats], m.simple_test(img.to(DEV), metas, rescale=True))
    finally:
        blocks.set_compute_dtype('f32')
    for a, b in zip(*[out[k][0] for k in ('f32', 'f16')]):
        assert (a - b).abs().max().item() <= 0.01 * a.abs().max().item()
    hit = tot = 0
    for b in range(2):
        for c in range(4):
            ra, rb = out['f32'][1][b][c], out['f16'][1][b][c]
            tot += len(ra)
            if len(ra) and len(rb):
                d = np.abs(ra[:, None, :4] - rb[None, :, :4]).max(-1)
                hit += int(((d < 1.0) & (np.abs(ra[:, None, 4] - rb[None, :, 4]) < 0.02)).any(1).sum())
    assert tot > 0 and hit >= 0.9 * tot, (hit, tot)


def test_train_step_f16_loss_scaling_close_to_fp32():
    """full train step in fp16 with the recipes' static loss scale 512 (scaled backward, unscale): losses
    within 1 % (RPN) / 5 % (second stage) of fp32, every parameter gradient finite and aligned with the fp32 gradient"""
    cfg = Config.fromfile(CFG)
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10)
    res = {}
    try:
        for mode in ('f32', 'f16'):
            m = build_detector(cfg.model)
            m.load_state_dict(util.seeded_state_dict(m, seed=10))
            m = m.to(DEV).train()
            m.set_compute_dtype(mode)
            torch.manual_seed(77)
            losses = m.forward_train(img.to(DEV), metas, [g_.to(DEV) for g_ in gts], [l.to(DEV) for l in gls])
            loss, log_vars = m._parse_losses(losses)
            scale = 512.0 if mode == 'f16' else 1.0
            (loss * scale).backward()
            grads = {k: p.grad.detach().float() / scale for k, p in m.named_parameters() if p.grad is not None}
            res[mode] = (dict(log_vars), grads)
    finally:
        blocks.set_compute_dtype('f32')
    for k, v in res['f32'][0].items():
        # RPN terms see the same anchors: 1 %; the second stage samples from slightly shifted proposals: 5 %
        tol = 0.01 if 'rpn' in k else 0.05
        assert abs(res['f16'][0][k] - v) <= tol * max(abs(v), 0.05), (k, res['f16'][0][k], v)
    assert res['f16'][1].keys() == res['f32'][1].keys()
    cos = []
    for k, g32 in res['f32'][1].items():
        g16 = res['f16'][1][k]
        assert torch.isfinite(g16).all(), k
        if g32.numel() > 16 and g32.abs().max() > 0:
            cos.append(torch.nn.functional.cosine_similarity(g16.flatten().double(), g32.flatten().double(), dim=0).item())
    assert min(cos) > 0.97 and np.mean(cos) > 0.995, (min(cos), np.mean(cos))


def test_fused_bottleneck_tail_16bit_equals_the_two_launches():
    """csrc/bottleneck_tail_bf16.hip: conv2 3x3 + bn2 + relu + conv3 1x1 + bn3 + identity + relu of a frozen stage-1 block in
    one launch -- the intermediate rounded to the 16-bit type as the two-launch form stores it, same K order and epilogue
    arithmetic: equal outputs"""
    g = torch.Generator().manual_seed(78)
    for (n, h, w) in [(1, 8, 16), (2, 16, 24), (1, 32, 40), (2, 50, 64)]:
        x = torch.randn(n, h, w, 64, generator=g).to(DEV, H)
        idn = torch.randn(n, h, w, 256, generator=g).to(DEV, H)
        w2 = (torch.randn(64, 3, 3, 64, generator=g) / 24).to(DEV, H)
        w3 = (torch.randn(256, 1, 1, 64, generator=g) / 8).to(DEV, H)
        s2, b2 = (torch.rand(64, generator=g) + 0.5).to(DEV), torch.randn(64, generator=g).to(DEV)
        s3, b3 = (torch.rand(256, generator=g) + 0.5).to(DEV), torch.randn(256, generator=g).to(DEV)
        t = ops.conv2d_nhwc(x, w2, s2, b2, None, True, 1, 1)
        ref = ops.conv2d_nhwc(t, w3, s3, b3, idn, True, 1, 0)
        assert ops.bottleneck_tail_supported(x, w2, w3, idn)
        y = ops.bottleneck_tail_nhwc(x, w2, s2, b2, w3, s3, b3, idn)
        assert y.dtype == ref.dtype and torch.equal(y, ref), (n, h, w, (y.float() - ref.float()).abs().max().item())
