"""-m gpu: the optional bf16 MFMA inference mode (north_star: "bf16 MFMA backbone").  The bf16
kernels multiply bf16 operands exactly and accumulate in fp32, so against an fp64 reference fed
the SAME bf16-rounded operands the only differences are fp32 accumulation order and the final
round-to-nearest-even to bf16 (relative 2^-9).  Tolerances below say exactly that.  The fp32
path stays the parity path; bf16 is checked against it at the detector level with the loose
tolerance bf16 activations warrant."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import brcnn  # noqa: F401
from brcnn import Config, build_detector, blocks, ops
from tests import util
from tests.test_host_cpu import CFG

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
BF = torch.bfloat16
HALF_ULP = 2.0 ** -8     # bf16 keeps 8 significand bits: round-to-nearest-even moves a value by <= 2^-8 |v|


def _rne_close(y, ref, acc_tol=2e-5):
    """elementwise: |y - ref| <= 2^-8 |ref| (the one rounding to bf16) + fp32 accumulation slack"""
    mag = max(1.0, ref.abs().max().item())
    bad = (y - ref).abs() > HALF_ULP * ref.abs() * 1.001 + acc_tol * mag
    return not bad.any().item()


def _bf(x):
    return x.to(BF).float()


@pytest.mark.parametrize('cfg', [
    # (N, Cin, H, W, Cout, k, stride, pad, scale, residual, relu, out_f32)
    (2, 64, 24, 40, 64, 1, 1, 0, True, False, True, False),
    (2, 64, 24, 40, 256, 1, 1, 0, True, True, True, False),
    (1, 128, 30, 31, 128, 3, 1, 1, True, False, True, False),
    (2, 256, 25, 42, 256, 3, 2, 1, False, False, False, False),
    (1, 512, 13, 21, 1024, 1, 2, 0, True, False, False, False),
    (1, 256, 13, 21, 54, 3, 1, 1, False, False, False, True),
    (300, 256, 7, 7, 130, 7, 1, 0, False, False, True, True),
])
def test_conv2d_nhwc_bf16(cfg):
    n, cin, h, w, cout, k, stride, pad, has_scale, has_res, relu, out_f32 = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 1000)
    x = _bf(torch.randn(n, cin, h, w, generator=g))
    wt = _bf(torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k))
    scale = torch.rand(cout, generator=g) + 0.5 if has_scale else None
    shift = torch.randn(cout, generator=g)
    ho, wo = ops.conv_out_size(h, w, k, k, stride, pad)
    res = _bf(torch.randn(n, cout, ho, wo, generator=g)) if has_res else None
    ref = F.conv2d(x.double(), wt.double(), None, stride, pad)
    if has_scale:
        ref = ref * scale.double().view(1, -1, 1, 1)
    ref = ref + shift.double().view(1, -1, 1, 1)
    if has_res:
        ref = ref + res.double()
    if relu:
        ref = ref.relu()
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV, BF)
    wg = wt.permute(0, 2, 3, 1).contiguous().to(DEV, BF)
    rg = res.permute(0, 2, 3, 1).contiguous().to(DEV, BF) if has_res else None
    y = ops.conv2d_nhwc(xg, wg, scale.to(DEV) if has_scale else None, shift.to(DEV), rg, relu, stride, pad,
                        out_f32=out_f32)
    assert y.dtype == (torch.float32 if out_f32 else BF)
    y = y.permute(0, 3, 1, 2).cpu().double()
    if out_f32:
        assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    else:
        assert _rne_close(y, ref)


def test_small_kernels_bf16():
    g = torch.Generator().manual_seed(9)
    x = _bf(torch.randn(2, 64, 33, 47, generator=g))
    y = ops.maxpool3x3s2_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV, BF))
    assert y.dtype == BF and torch.equal(y.float().permute(0, 3, 1, 2).cpu(), F.max_pool2d(x, 3, 2, 1))
    # group norm (fp32 statistics of the bf16 tensor) + relu
    x = _bf(torch.randn(2, 256, 25, 42, generator=g) * 2 + 0.3)
    gamma, beta = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    ref = F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5).relu()
    y = ops.groupnorm_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV, BF), gamma.to(DEV), beta.to(DEV),
                           32, 1e-5, True)
    assert y.dtype == BF
    y = y.float().permute(0, 3, 1, 2).cpu().double()
    assert _rne_close(y, ref)
    # upsample + add: fp32 add of two bf16 values, one RNE
    for (hd, wd, hs, ws) in [(50, 84, 25, 42), (13, 21, 7, 11)]:
        d = _bf(torch.randn(2, 256, hd, wd, generator=g))
        s = _bf(torch.randn(2, 256, hs, ws, generator=g))
        ref = (d + F.interpolate(s, size=(hd, wd), mode='nearest')).to(BF)
        dg = d.permute(0, 2, 3, 1).contiguous().to(DEV, BF)
        ops.upsample_nearest_add_nhwc_(dg, s.permute(0, 2, 3, 1).contiguous().to(DEV, BF))
        assert torch.equal(dg.permute(0, 3, 1, 2).cpu(), ref)


def test_multi_level_bf16_matches_per_level():
    g = torch.Generator().manual_seed(4)
    sizes = [(40, 64), (20, 32), (10, 16), (5, 8), (3, 4)]
    B, C = 2, 256
    feats = [torch.randn(B, h, w, C, generator=g).to(DEV, BF) for h, w in sizes]
    wt = (torch.randn(C, 3, 3, C, generator=g) / 48).to(DEV, BF)
    b = torch.randn(C, generator=g).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    xc = torch.cat([f.reshape(-1, C) for f in feats], 0)
    y, osz = ops.conv2d_nhwc_multi(xc, wt, B, sizes, None, b, None, False, 1, 1)
    yn = ops.groupnorm_nhwc_multi(y, gamma, beta, 32, B, osz, 1e-5, True)
    o = 0
    for f, (h, w) in zip(feats, sizes):
        m = B * h * w
        r = ops.conv2d_nhwc(f, wt, None, b, None, False, 1, 1)
        assert torch.equal(y[o:o + m].reshape(B, h, w, C), r)
        rn = ops.groupnorm_nhwc(r, gamma, beta, 32, 1e-5, True)
        assert torch.equal(yn[o:o + m].reshape(B, h, w, C), rn)
        o += m


def test_stem_and_roi_extract_bf16():
    g = torch.Generator().manual_seed(12)
    for (n, h, w) in [(2, 64, 96), (1, 75, 83)]:
        img = torch.randn(n, 3, h, w, generator=g)
        wt = torch.randn(64, 3, 7, 7, generator=g) / 12
        sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
        ref = (F.conv2d(_bf(img).double(), _bf(wt).double(), None, 2, 3) * sc.double().view(1, -1, 1, 1) +
               sh.double().view(1, -1, 1, 1)).relu()
        y = ops.stem7x7s2_nchw(img.to(DEV), ops.pack_stem_weight(wt.to(DEV), BF), sc.to(DEV), sh.to(DEV), True)
        assert y.dtype == BF
        y = y.float().permute(0, 3, 1, 2).cpu().double()
        assert _rne_close(y, ref)
        # ... and the one-launch stem + max-pool (max of correctly rounded values = the rounded max)
        yp = ops.stem7x7s2_pool_nchw(img.to(DEV), ops.pack_stem_pool_weight(wt.to(DEV), BF), sc.to(DEV), sh.to(DEV))
        assert yp.dtype == BF
        assert _rne_close(yp.float().permute(0, 3, 1, 2).cpu().double(), F.max_pool2d(ref, 3, 2, 1))
    # RoI extract on bf16 maps: fp32 bilinear blend of bf16 samples, one RNE => equals the fp32
    # kernel run on the same (bf16-representable) maps, rounded once
    strides = [4, 8, 16, 32]
    feats = [_bf(torch.randn(2, 200 // s, 304 // s, 256, generator=g)).to(DEV) for s in strides]
    rois = util.rand_rois(300, 2, 304.0, 200.0, seed=3).to(DEV)
    o32, l32 = ops.roi_extract(feats, rois, 7, strides, 56, 0)
    o16, l16 = ops.roi_extract([f.to(BF) for f in feats], rois, 7, strides, 56, 0)
    assert o16.dtype == BF and torch.equal(l16, l32)
    assert torch.equal(o16, o32.to(BF))


def test_detector_bf16_close_to_fp32():
    cfg = Config.fromfile(CFG)
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=10))
    m = m.to(DEV).eval()
    img, metas, _, _ = util.demo_inputs(2, 128, 192, seed=10)
    try:
        with torch.no_grad():
            f32 = [f.float() for f in m.extract_feat_nhwc(img.to(DEV))]
            r32 = m.simple_test(img.to(DEV), metas)
            m.set_compute_dtype('bf16')
            f16 = m.extract_feat_nhwc(img.to(DEV))
            assert all(f.dtype == BF for f in f16)
            r16 = m.simple_test(img.to(DEV), metas)
    finally:
        blocks.set_compute_dtype('f32')
    # pyramid features: ~50 stacked bf16 roundings; a few percent of the map's magnitude
    for a, b in zip(f16, f32):
        rel = (a.float() - b).abs().max().item() / b.abs().max().item()
        assert rel < 0.05, rel
    # detections: same count scale and most boxes found again within a pixel / 0.05 score
    n32 = sum(len(c) for r in r32 for c in r)
    n16 = sum(len(c) for r in r16 for c in r)
    assert n32 > 0 and abs(n16 - n32) <= 0.2 * n32 + 5, (n16, n32)
    d = np.concatenate([c for r in r32 for c in r]), np.concatenate([c for r in r16 for c in r])
    dist = np.abs(d[0][:, None, :4] - d[1][None, :, :4]).max(-1)
    ds = np.abs(d[0][:, None, 4] - d[1][None, :, 4])
    assert ((dist < 2.0) & (ds < 0.05)).any(1).mean() > 0.7


@pytest.mark.parametrize('cfg', [
    # (N, Cin, H, W, Cout, k, stride, pad)
    (2, 64, 24, 40, 64, 1, 1, 0),
    (2, 64, 24, 40, 256, 3, 1, 1),
    (1, 128, 30, 31, 128, 3, 2, 1),
    (2, 256, 25, 42, 512, 1, 2, 0),
    (1, 256, 13, 21, 54, 3, 1, 1),         # ragged Cout: padded to 64 inside
    (3, 512, 9, 14, 128, 3, 1, 1),
])
def test_conv_autograd_bf16(cfg):
    """bf16 forward / dgrad (bf16 MFMA, zero-stuffed dy for stride 2) / wgrad (transposing LDS
    reads) against the fp64 gradients of the same bf16-rounded operands"""
    from brcnn.autograd import conv2d_nhwc_autograd
    n, cin, h, w, cout, k, stride, pad = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 997)
    x = _bf(torch.randn(n, cin, h, w, generator=g))
    wt = _bf(torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g)
    xr = x.double().requires_grad_()
    wr = wt.double().requires_grad_()
    br = b.double().requires_grad_()
    ref = F.conv2d(xr, wr, br, stride, pad)
    go = _bf(torch.randn(ref.shape, generator=g))
    ref.backward(go.double())
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV, BF).requires_grad_()
    wg = wt.to(DEV).requires_grad_()          # fp32 master weight whose values are bf16-representable
    bg = b.to(DEV).requires_grad_()
    y = conv2d_nhwc_autograd(xg, wg, bg, stride, pad)
    assert y.dtype == BF and xg.grad is None
    y.backward(go.permute(0, 2, 3, 1).contiguous().to(DEV, BF))
    assert _rne_close(y.detach().float().permute(0, 3, 1, 2).cpu().double(), ref.detach())
    assert xg.grad.dtype == BF and wg.grad.dtype == torch.float32
    assert _rne_close(xg.grad.float().permute(0, 3, 1, 2).cpu().double(), xr.grad)
    dw, db = wg.grad.cpu().double(), bg.grad.cpu().double()
    assert (dw - wr.grad).abs().max().item() <= 2e-4 * max(1.0, wr.grad.abs().max().item())
    assert (db - br.grad).abs().max().item() <= 2e-4 * max(1.0, br.grad.abs().max().item())


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cfg', [
    # (N, Cin, H, W, Cout, k, stride, residual, relu, with_skip)
    (2, 64, 20, 28, 64, 3, 1, False, True, False), (2, 128, 14, 18, 256, 1, 1, True, True, False),
    (1, 64, 17, 23, 128, 3, 2, False, True, False), (2, 256, 9, 11, 64, 1, 1, False, True, True),
    (1, 128, 12, 10, 128, 1, 2, False, False, False), (8, 256, 50, 84, 1024, 1, 1, True, True, False)])
def test_conv_bn_act_one_launch_training_forward(cfg, dtype):
    """conv -> eval-BN [-> + residual] [-> ReLU] with the dual-store epilogue (`brcnn_conv2d_bn_act_nhwc_multi`):
    bit-identical to the conv kernel followed by the bn_act kernel (the affine is applied to the stored, rounded z)
    in the forward and in every gradient (input, weight, gamma, beta, residual, the identity alias); the two-kernel
    path itself is checked against float64 in test_conv_autograd_bf16 / test_bn_eval_act"""
    from brcnn.autograd import conv_bn_eval_act_autograd, conv_bn_eval_act_fusable
    N, Cin, H, W, Cout, k, stride, res, relu, with_skip = cfg
    g = torch.Generator().manual_seed(31)
    conv = torch.nn.Conv2d(Cin, Cout, k, stride, k // 2, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(Cout).eval().to(DEV)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(k * k * Cin))
        bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cout, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.2)
        bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV, dtype)
    r = torch.randn(N, Ho, Wo, Cout, generator=g).to(DEV, dtype) if res else None
    go = torch.randn(N, Ho, Wo, Cout, generator=g).to(DEV, dtype)
    gs = torch.randn(N, H, W, Cin, generator=g).to(DEV, dtype) if with_skip else None
    got = {}
    for fused in (True, False):
        blocks.FUSE_CONV_BN_TRAIN = fused
        try:
            conv.zero_grad(); bn.zero_grad()
            xd = x.clone().requires_grad_()
            rd = r.clone().requires_grad_() if res else None
            assert conv_bn_eval_act_fusable(xd, conv, bn, rd)
            out = blocks.conv_bn_act_nhwc(xd, conv, bn, None, relu, rd, with_skip)
            outs = [out[0], out[1]] if with_skip else [out]
            torch.autograd.backward(outs, [go, gs] if with_skip else [go])
            got[fused] = [outs[0].detach(), xd.grad, conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()] + \
                ([rd.grad] if res else [])
        finally:
            blocks.FUSE_CONV_BN_TRAIN = True
    assert got[True][0].dtype == dtype
    names = ['out', 'dx', 'dgamma', 'dbeta', 'dres']
    for i, nm in zip((0, 1, 3, 4, 5), names):
        if i < len(got[True]):
            assert torch.equal(got[True][i], got[False][i]), nm
    # the weight gradient accumulates with fp32 atomics (order varies from launch to launch)
    dw_a, dw_b = got[True][2], got[False][2]
    assert (dw_a - dw_b).abs().max().item() <= 1e-5 * max(1.0, dw_b.abs().max().item())


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cfg', [(2, 256, 64, 1, 20, 28, False), (2, 256, 128, 2, 21, 30, True),
                                 (8, 512, 128, 1, 50, 84, False), (1, 1024, 512, 2, 13, 17, True)])
def test_bottleneck_bn_backward_inside_data_gradient_launch(cfg, dtype):
    """a whole Bottleneck with bn1's / bn2's backward folded into conv2's / conv3's data-gradient launches
    (`brcnn_conv2d_dgrad_bn_backward_nhwc`) against the same block with the separate bn_act backward launches:
    input gradient bit for bit (dz is the same arithmetic on the same rounded values), dgamma / dbeta up to the
    summation order, weight gradients up to the order of their atomics"""
    from brcnn import autograd as A
    from brcnn.backbones import Bottleneck
    N, inplanes, planes, stride, H, W, down = cfg
    torch.manual_seed(41)
    ds = None
    if down:
        ds = torch.nn.Sequential(torch.nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), torch.nn.BatchNorm2d(planes * 4))
    blk = Bottleneck(inplanes, planes if down else inplanes // 4, stride if down else 1, ds).to(DEV)
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    blk.eval()          # norm_eval=True: BatchNorm uses its running statistics, gamma / beta still train
    x = torch.randn(N, H, W, inplanes, device=DEV).to(dtype)
    got = {}
    for fused in (True, False):
        A.FUSE_BN_BACKWARD_INTO_DGRAD = fused
        try:
            blk.zero_grad()
            xd = x.clone().requires_grad_()
            out = blk.forward_nhwc(xd)
            go = torch.randn(out.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(3)).to(dtype)
            out.backward(go)
            got[fused] = (out.detach(), xd.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()})
        finally:
            A.FUSE_BN_BACKWARD_INTO_DGRAD = True
    assert torch.equal(got[True][0], got[False][0]) and torch.equal(got[True][1], got[False][1])
    for k, ga in got[True][2].items():
        gb = got[False][2][k]
        assert (ga - gb).abs().max().item() <= 2e-5 * max(1.0, gb.abs().max().item()), k


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cfg', [(2, 128, 64, 2, 3, 21, 30), (8, 512, 128, 1, 4, 50, 84), (1, 512, 256, 2, 2, 13, 17)])
def test_res_layer_residual_bn_backward_inside_next_block(cfg, dtype):
    """a whole stage (ResLayer): bn3's backward (residual + ReLU) of every block but the last runs inside the NEXT
    block's conv1 data-gradient launch, bn1 / bn2 inside conv2 / conv3 -- against the same stage with every
    BatchNorm backward as its own launch: outputs and the input gradient bit for bit, parameter gradients up to the
    summation / atomics order"""
    from brcnn import autograd as A
    from brcnn.backbones import Bottleneck, ResLayer
    N, inplanes, planes, stride, nblocks, H, W = cfg
    torch.manual_seed(47)
    layer = ResLayer(Bottleneck, inplanes, planes, nblocks, stride).to(DEV)
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    layer.eval()
    x = torch.randn(N, H, W, inplanes, device=DEV).to(dtype)
    got = {}
    saved = (A.FUSE_BN_BACKWARD_INTO_DGRAD, A.FUSE_RESIDUAL_BN_BACKWARD)
    for fused in (True, False, True, False):
        A.FUSE_BN_BACKWARD_INTO_DGRAD = A.FUSE_RESIDUAL_BN_BACKWARD = fused
        try:
            layer.zero_grad()
            xd = x.clone().requires_grad_()
            out = layer.forward_nhwc(xd)
            go = torch.randn(out.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(3)).to(dtype)
            out.backward(go)
            cur = (out.detach(), xd.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()})
        finally:
            A.FUSE_BN_BACKWARD_INTO_DGRAD, A.FUSE_RESIDUAL_BN_BACKWARD = saved
        if fused in got:        # the BatchNorm gradients are fixed-order sums: a repeat reproduces them bit for bit
            assert torch.equal(cur[1], got[fused][1]), ('dx differs between two identical passes', fused)
            for k, g_ in cur[2].items():
                if g_.dim() == 1:
                    assert torch.equal(g_, got[fused][2][k]), ('not reproducible', fused, k, (g_ - got[fused][2][k]).abs().max().item())
        got[fused] = cur
    assert torch.equal(got[True][0], got[False][0]) and torch.equal(got[True][1], got[False][1])
    for k, ga in got[True][2].items():
        gb = got[False][2][k]
        # two different fixed orders of an fp32 sum over up to 33 600 x 8 products of magnitude ~1-10 (the per-tile
        # partials of the data-gradient epilogue against the row strips of bn_act_bwd): 1e-4 of the largest entry
        assert (ga - gb).abs().max().item() <= 1e-4 * max(1.0, gb.abs().max().item()), k


@pytest.mark.parametrize('down', [False, True])
def test_bottleneck_against_plain_torch_float64(down, dtype='f32'):
    """a Bottleneck in fp32 (identity alias through conv1, downsample branch through the same alias) against the
    textbook module graph in float64 on the CPU (resnet.py Bottleneck.forward:263-302 with eval-mode BatchNorm):
    output and every gradient.  (The 16-bit block with its fused BatchNorm launches is checked bit for bit against
    the same block with separate launches above; against float64 it differs by the ReLU masks that flip when a
    pre-activation rounds across zero -- ~5 % in the Frobenius norm on random data.)"""
    from brcnn.backbones import Bottleneck
    torch.manual_seed(43)
    inplanes, planes, stride = (128, 64, 2) if down else (256, 64, 1)
    ds = None
    if down:
        ds = torch.nn.Sequential(torch.nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), torch.nn.BatchNorm2d(planes * 4))
    blk = Bottleneck(inplanes, planes, stride, ds)
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    blk.eval()
    x = torch.randn(2, inplanes, 18, 22)
    go = torch.randn(2, planes * 4, 18 // stride, 22 // stride)
    if dtype == 'bf16':     # operands the 16-bit path represents exactly
        x, go = x.bfloat16().float(), go.bfloat16().float()
        with torch.no_grad():
            for m in blk.modules():
                if isinstance(m, torch.nn.Conv2d):
                    m.weight.copy_(m.weight.bfloat16().float())
    # float64 reference: plain torch modules
    import copy
    ref = copy.deepcopy(blk).double()
    xr = x.double().requires_grad_()

    def bn(mod, t):
        return F.batch_norm(t, mod.running_mean, mod.running_var, mod.weight, mod.bias, False, 0.0, mod.eps)
    o = bn(ref.bn1, ref.conv1(xr)).relu()
    o = bn(ref.bn2, ref.conv2(o)).relu()
    o = bn(ref.bn3, ref.conv3(o))
    idt = bn(ref.downsample[1], ref.downsample[0](xr)) if down else xr
    out_r = (o + idt).relu()
    out_r.backward(go.double())
    # device
    blk = blk.to(DEV)
    blocks.set_compute_dtype(dtype)
    try:
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        if dtype == 'bf16':
            xd = xd.bfloat16()
        xd.requires_grad_()
        out = blk.forward_nhwc(xd)
        out.backward(go.permute(0, 2, 3, 1).contiguous().to(DEV).to(out.dtype))
    finally:
        blocks.set_compute_dtype('f32')
    def close(a, b):
        a, b = a.double().cpu(), b.double()
        if dtype == 'f32':
            return (a - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item())
        # bf16: three layers of activations / gradients rounded to 8 bits, and a ReLU whose pre-activation rounds
        # across zero passes or blocks a whole gradient element: judged in the Frobenius norm
        return (a - b).norm().item() <= 2e-2 * b.norm().item() + 1e-6
    assert close(out.detach().float().permute(0, 3, 1, 2), out_r.detach())
    assert close(xd.grad.float().permute(0, 3, 1, 2), xr.grad)
    rp = dict(ref.named_parameters())
    for k, p_ in blk.named_parameters():
        assert p_.grad is not None and close(p_.grad, rp[k].grad), k


def test_fused_training_entries_reject_bad_arguments():
    """error behaviour of the fused training entry points: a status code, never a launch on inconsistent pointers"""
    import ctypes
    from brcnn import lib as L
    from brcnn.ops import DT_BF16, DT_F32, _ptr
    lib = L.load()
    n, h, w, ci, co = 1, 8, 8, 64, 64
    x = torch.zeros(n * h * w, ci, device=DEV, dtype=BF)
    wt = torch.zeros(co, 1, 1, ci, device=DEV, dtype=BF)
    v = torch.ones(co, device=DEV)
    z, y = torch.empty(n * h * w, co, device=DEV, dtype=BF), torch.empty(n * h * w, co, device=DEV, dtype=BF)
    hs, ws = (ctypes.c_int * 1)(h), (ctypes.c_int * 1)(w)
    ok = lib.brcnn_conv2d_bn_act_nhwc_multi(_ptr(x), _ptr(wt), _ptr(v), _ptr(v), _ptr(v), _ptr(v), 1e-5, None, _ptr(z), _ptr(y), n, 1,
                                            hs, ws, ci, co, 1, 1, 1, 0, 1, DT_BF16, None)
    assert ok == 0
    # fp32 is not a dual-store dtype; z_out is required; mean without var is inconsistent
    assert lib.brcnn_conv2d_bn_act_nhwc_multi(_ptr(x), _ptr(wt), _ptr(v), _ptr(v), _ptr(v), _ptr(v), 1e-5, None, _ptr(z), _ptr(y), n, 1,
                                              hs, ws, ci, co, 1, 1, 1, 0, 1, DT_F32, None) != 0
    assert lib.brcnn_conv2d_bn_act_nhwc_multi(_ptr(x), _ptr(wt), _ptr(v), _ptr(v), _ptr(v), _ptr(v), 1e-5, None, None, _ptr(y), n, 1,
                                              hs, ws, ci, co, 1, 1, 1, 0, 1, DT_BF16, None) != 0
    assert lib.brcnn_conv2d_bn_act_nhwc_multi(_ptr(x), _ptr(wt), _ptr(v), _ptr(v), _ptr(v), None, 1e-5, None, _ptr(z), _ptr(y), n, 1,
                                              hs, ws, ci, co, 1, 1, 1, 0, 1, DT_BF16, None) != 0
    # data gradient + BatchNorm backward: workspace too small; residual-producer pointers must come together
    dg, db = torch.empty(ci, device=DEV), torch.empty(ci, device=DEV)
    nb = lib.brcnn_conv2d_dgrad_bn_backward_workspace_bytes(n, h, w, ci)
    wsp = torch.empty(nb, dtype=torch.uint8, device=DEV)
    args = lambda dskip, prev, dres, nbytes: (_ptr(y), _ptr(wt), _ptr(x), _ptr(v), _ptr(v), _ptr(v), _ptr(v), 1e-5, 1, dskip, prev, dres,   # noqa: E731
                                              _ptr(torch.empty_like(x)), _ptr(dg), _ptr(db), _ptr(wsp), nbytes, n, h, w, h, w, ci, co,
                                              1, 1, 1, 0, DT_BF16, None)
    assert lib.brcnn_conv2d_dgrad_bn_backward_nhwc(*args(None, None, None, nb)) == 0
    assert lib.brcnn_conv2d_dgrad_bn_backward_nhwc(*args(None, None, None, nb - 4)) != 0
    assert lib.brcnn_conv2d_dgrad_bn_backward_nhwc(*args(_ptr(x), None, None, nb)) != 0
    assert lib.brcnn_conv2d_dgrad_bn_backward_nhwc(*args(_ptr(x), _ptr(x), None, nb)) != 0
    torch.cuda.synchronize()


def test_wgrad_bf16_tiles_and_multi_level_agree():
    """64x64, 128x128 and 256x256 (16-wave) output tiles, one multi-level launch vs per-level launches"""
    from brcnn import lib
    from brcnn.autograd import ConvNHWCFunction
    g = torch.Generator().manual_seed(6)
    sizes = [(20, 32), (10, 16), (5, 8)]
    B, C = 2, 256
    feats = [torch.randn(B * h * w, C, generator=g).to(DEV, BF) for h, w in sizes]
    wt = (torch.randn(C, C, 3, 3, generator=g) / 48).to(DEV)
    outs = {}
    for tile in (1, 2, 4):
        lib.load().brcnn_conv_set_tile_wgrad_bf16(tile)
        w1 = wt.clone().requires_grad_()
        xc = torch.cat(feats, 0).requires_grad_()
        y = ConvNHWCFunction.apply(xc, w1, None, B, tuple(sizes), 1, 1)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).to(DEV, BF)
        y.backward(gy)
        outs[tile] = (w1.grad.clone(), xc.grad.clone())
    lib.load().brcnn_conv_set_tile_wgrad_bf16(0)
    assert torch.allclose(outs[1][0], outs[2][0], rtol=1e-4, atol=1e-3)      # atomics order differs
    assert torch.allclose(outs[4][0], outs[2][0], rtol=1e-4, atol=1e-3)
    assert torch.equal(outs[1][1], outs[2][1]) and torch.equal(outs[4][1], outs[2][1])
    acc, o = torch.zeros_like(wt), 0
    for f, (h, w) in zip(feats, sizes):
        w2 = wt.clone().requires_grad_()
        x2 = f.clone().requires_grad_()
        y2 = ConvNHWCFunction.apply(x2, w2, None, B, ((h, w),), 1, 1)
        y2.backward(gy[o:o + B * h * w])
        assert torch.equal(x2.grad, outs[2][1][o:o + B * h * w])
        acc += w2.grad
        o += B * h * w
    assert torch.allclose(acc, outs[2][0], rtol=1e-4, atol=1e-3)


def test_train_step_bf16_close_to_fp32():
    cfg = Config.fromfile(CFG)
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10)
    res = {}
    try:
        for mode in ('f32', 'bf16'):
            m = build_detector(cfg.model)
            m.load_state_dict(util.seeded_state_dict(m, seed=10))
            m = m.to(DEV).train()
            m.set_compute_dtype(mode)
            torch.manual_seed(77)
            losses = m.forward_train(img.to(DEV), metas, [g_.to(DEV) for g_ in gts], [l.to(DEV) for l in gls])
            loss, log_vars = m._parse_losses(losses)
            loss.backward()
            grads = {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}
            res[mode] = (log_vars, grads)
    finally:
        blocks.set_compute_dtype('f32')
    lf, lb = res['f32'][0], res['bf16'][0]
    # the RPN terms are functions of the head outputs alone: 5 %.  The second-stage terms are taken over the sampled
    # proposals, and which proposals survive top-k / NMS (and are then drawn) flips with 16-bit rounding on these
    # random-weight score maps -- a handful of positives per image, so a changed draw moves loss_bbox by ~10 %: 20 %
    for k, tol in (('loss_rpn_cls', 0.05), ('loss_rpn_bbox', 0.05), ('loss_rpn_iou', 0.05), ('loss_bbox', 0.2),
                   ('loss', 0.08)):
        assert abs(lb[k] - lf[k]) <= tol * abs(lf[k]) + 1e-3, (k, lb[k], lf[k])
    gf, gb = res['f32'][1], res['bf16'][1]
    assert set(gf) == set(gb)
    for k in ('backbone.layer2.0.conv1.weight', 'backbone.layer4.2.conv3.weight', 'neck.lateral_convs.0.conv.weight',
              'rpn_head.rpn_convs.3.conv.weight', 'rpn_head.rpn_cls.weight', 'backbone.layer3.0.bn1.weight'):
        a, b = gf[k].flatten(), gb[k].flatten()
        assert torch.isfinite(b).all()
        cos = torch.dot(a, b) / (a.norm() * b.norm() + 1e-20)
        assert cos > 0.9, (k, cos.item())


def test_bf16_tile_variants_agree():
    """every bf16 tile shape (64x64, 128x64, 128x128 with 4 waves, 256x128 with 8 waves) computes the
    same K order per output element -> bit-identical results"""
    from brcnn import lib
    g = torch.Generator().manual_seed(5)
    L = lib.load()
    for (n, cin, h, w, cout, k, stride, res) in [(2, 128, 37, 41, 256, 3, 1, True), (1, 256, 50, 84, 128, 1, 1, False),
                                                 (3, 64, 29, 31, 512, 3, 2, True)]:
        x = torch.randn(n, h, w, cin, generator=g).to(DEV, BF)
        wt = (torch.randn(cout, k, k, cin, generator=g) / np.sqrt(cin * k * k)).to(DEV, BF)
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(DEV), torch.randn(cout, generator=g).to(DEV)
        ho, wo = ops.conv_out_size(h, w, k, k, stride, k // 2)
        r = torch.randn(n, ho, wo, cout, generator=g).to(DEV, BF) if res else None
        outs = []
        try:
            for tile in (11, 21, 22, 42, 82, 81, 164):
                assert L.brcnn_conv_set_tile_bf16(tile) == 0
                outs.append(ops.conv2d_nhwc(x, wt, sc, sh, r, True, stride, k // 2))
        finally:
            L.brcnn_conv_set_tile_bf16(0)
        for o in outs[1:]:
            assert torch.equal(o, outs[0])


def test_grouped_conv_bf16_forward_backward():
    """bf16 grouped conv (ResNeXt conv2): forward, dgrad and tiled wgrad against the fp64 gradients of the
    same bf16-rounded operands"""
    from brcnn.autograd import grouped_conv_autograd
    gen = torch.Generator().manual_seed(43)
    for (n, c, h, w, groups, stride) in [(2, 128, 20, 30, 32, 1), (1, 256, 17, 23, 64, 2), (2, 512, 9, 14, 64, 1),
                                         (1, 1024, 7, 9, 32, 2)]:
        x = _bf(torch.randn(n, c, h, w, generator=gen))
        wt = _bf(torch.randn(c, c // groups, 3, 3, generator=gen) / np.sqrt(9 * c / groups))
        xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
        ref = F.conv2d(xr, wr, None, stride, 1, groups=groups)
        go = _bf(torch.randn(ref.shape, generator=gen))
        ref.backward(go.double())
        xg = x.permute(0, 2, 3, 1).contiguous().to(DEV, BF).requires_grad_()
        wg = wt.to(DEV).requires_grad_()
        y = grouped_conv_autograd(xg, wg, groups, stride, 1)
        assert y.dtype == BF
        y.backward(go.permute(0, 2, 3, 1).contiguous().to(DEV, BF))
        assert _rne_close(y.detach().float().permute(0, 3, 1, 2).cpu().double(), ref.detach())
        assert xg.grad.dtype == BF and _rne_close(xg.grad.float().permute(0, 3, 1, 2).cpu().double(), xr.grad)
        assert wg.grad.dtype == torch.float32 and wg.grad.shape == wt.shape
        assert (wg.grad.cpu().double() - wr.grad).abs().max().item() <= 2e-4 * max(1.0, wr.grad.abs().max().item())


def test_scatter2_conv_bf16_equals_strided_copy():
    """brcnn_conv2d_nhwc_scatter2 (one parity class of a stride-2 data gradient) in bf16: the
    scattered rows equal the dense bf16 conv written through a strided copy, bit for bit, and
    pixels of the other parities keep their previous contents"""
    from brcnn import lib as _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(5)
    n, ho, wo, c_in, c_out, H, W = 2, 9, 11, 128, 64, 17, 22
    dy = torch.randn(n, ho, wo, c_in, generator=g).bfloat16().to(DEV)
    for (kh, kw, ph, pw) in [(1, 1, 0, 0), (2, 1, 1, 0), (1, 2, 0, 1), (2, 2, 1, 1)]:
        wt = (torch.randn(c_out, kh, kw, c_in, generator=g) * 0.05).bfloat16().to(DEV)
        dense = ops.conv2d_nhwc(dy, wt, None, None, None, False, 1, 1)
        na, nb = (H - ph + 1) // 2, (W - pw + 1) // 2
        ref = torch.full((n, H, W, c_out), 7.0, dtype=torch.bfloat16, device=DEV)
        ref[:, ph::2, pw::2] = dense[:, 1:1 + na, 1:1 + nb]
        out = torch.full((n, H, W, c_out), 7.0, dtype=torch.bfloat16, device=DEV)
        st = L.brcnn_conv2d_nhwc_scatter2(dy.data_ptr(), wt.data_ptr(), out.data_ptr(), n, ho, wo, c_in, c_out,
                                          kh, kw, 1, H, W, ph, pw, 1, 1, None)
        assert st == 0
        torch.cuda.synchronize()
        assert torch.equal(out, ref), (kh, kw, ph, pw)


def test_bf16_tile_shapes_agree_bit_for_bit():
    """every workgroup tile shape of the bf16 conv kernel (64x64 ... 256x256, 4 / 8 / 16 waves, the LDS
    ring variants) accumulates each output over K in the same order: on a layer large enough for all
    of them (incl. the 256x256 tile the heuristic picks for the fused RPN tower) the results are
    identical"""
    from brcnn import lib as _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 100, 168, 256, generator=g).bfloat16().to(DEV)
    w = (torch.randn(256, 3, 3, 256, generator=g) * 0.05).bfloat16().to(DEV)
    sc = (torch.rand(256, generator=g) + 0.5).to(DEV)
    sh = torch.randn(256, generator=g).to(DEV)
    try:
        outs = {}
        for t in (11, 21, 22, 81, 82, 164, 42, 2244, 2144, 382, 342, 8844, 8842):
            assert L.brcnn_conv_set_tile_bf16(t) == 0
            outs[t] = ops.conv2d_nhwc(x, w, sc, sh, None, True, 1, 1)
        torch.cuda.synchronize()
        for t, y in outs.items():
            assert torch.equal(y, outs[11]), t
    finally:
        L.brcnn_conv_set_tile_bf16(0)


@pytest.mark.parametrize('cfg', [
    # N, H, W, Cin, Cout, k, stride, pad, residual, out_f32
    (2, 50, 84, 256, 256, 3, 1, 1, False, False),      # 36 K tiles
    (2, 50, 84, 64, 256, 3, 1, 1, True, False),        # 9 K tiles (odd: the last iteration's second half is skipped)
    (3, 25, 42, 192, 320, 1, 1, 0, False, False),      # 3 K tiles, ragged row / channel tiles
    (2, 50, 84, 128, 256, 1, 1, 0, True, True),        # 2 K tiles, fp32 result
    (2, 51, 85, 256, 384, 3, 2, 1, False, False),      # stride 2
    (1, 13, 21, 256, 54, 3, 1, 1, False, True),        # ragged channel count: routed to the small tiles
])
def test_bf16_eight_phase_kernel_is_bit_identical(cfg):
    """the 256 x 256 eight-phase kernel (conv_pp_bf16.hip) accumulates over K in the order of the two-buffer kernel:
    identical results, for every K-tile count parity, ragged tiles, strides, residual / fp32 epilogues, several
    launches in a row (LDS slots and the DMA pipeline start clean each time)"""
    from brcnn import lib as _lib
    L = _lib.load()
    n, h, w_, ci, co, k, stride, pad, res, of32 = cfg
    g = torch.Generator().manual_seed(33)
    x = torch.randn(n, h, w_, ci, generator=g).bfloat16().to(DEV)
    w = (torch.randn(co, k, k, ci, generator=g) * 0.05).bfloat16().to(DEV)
    sc = (torch.rand(co, generator=g) + 0.5).to(DEV)
    sh = torch.randn(co, generator=g).to(DEV)
    ho, wo = ops.conv_out_size(h, w_, k, k, stride, pad)
    r = torch.randn(n, ho, wo, co, generator=g).bfloat16().to(DEV) if res else None
    try:
        assert L.brcnn_conv_set_tile_bf16(11) == 0
        ref = ops.conv2d_nhwc(x, w, sc, sh, r, True, stride, pad, out_f32=of32)
        for tile in (8844, 8842):       # 256 x 256 eight-phase; 256 x 128 two-group (conv_pp128_bf16.hip, where its shape rules allow)
            assert L.brcnn_conv_set_tile_bf16(tile) == 0
            for rep in range(3):
                out = ops.conv2d_nhwc(x, w, sc, sh, r, True, stride, pad, out_f32=of32)
                assert torch.equal(out, ref), (tile, rep, (out.float() - ref.float()).abs().max().item())
    finally:
        L.brcnn_conv_set_tile_bf16(0)


@pytest.mark.parametrize('et', [BF, torch.float16])
def test_bf16_256x128_kernel_every_k_tile_count(et):
    """conv_pp128_bf16.hip keeps three K tiles in nine LDS slots and unrolls six K tiles per loop iteration: every K-tile
    count 3 .. 14 (all tails of the unrolled loop), 1x1 and 3x3 taps, ragged rows, one and three column tiles, residual
    and ReLU -- bit-identical to the two-buffer kernel, launch after launch"""
    from brcnn import lib as _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(61)
    try:
        for nk in range(3, 15):
            for (k, co, res) in ((1, 128, False), (1, 384, True)) + (((3, 128, True),) if nk % 9 == 0 else ()):
                ci = 64 * nk // (k * k)
                x = torch.randn(2, 37, 53, ci, generator=g).to(DEV, et)
                w = (torch.randn(co, k, k, ci, generator=g) * 0.05).to(DEV, et)
                sc = (torch.rand(co, generator=g) + 0.5).to(DEV)
                sh = torch.randn(co, generator=g).to(DEV)
                r = torch.randn(2, 37, 53, co, generator=g).to(DEV, et) if res else None
                assert L.brcnn_conv_set_tile_bf16(11) == 0
                ref = ops.conv2d_nhwc(x, w, sc, sh, r, True, 1, k // 2)
                assert L.brcnn_conv_set_tile_bf16(8842) == 0
                for rep in range(2):
                    out = ops.conv2d_nhwc(x, w, sc, sh, r, True, 1, k // 2)
                    assert torch.equal(out, ref), (nk, k, co, rep, (out.float() - ref.float()).abs().max().item())
    finally:
        L.brcnn_conv_set_tile_bf16(0)


def test_bf16_wgrad_slab_reduction_is_reproducible_and_equals_the_atomics_form():
    """the weight gradient's M slices reduced through per-workgroup slabs + a fixed-order second stage: two launches
    give the same bits (the fp32-atomics form does not promise that), the values agree with the atomics form to the
    order of the additions, accumulation into an existing dW is kept, for the three tile shapes and a five-level launch"""
    import ctypes
    from brcnn import lib as _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(17)
    LV = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    cases = [(2, LV, 256, 256, 3, 1, 1), (8, LV[1:2], 256, 256, 3, 1, 1), (8, LV[1:2], 1024, 256, 1, 1, 0), (3, LV[:1], 128, 128, 3, 2, 1)]
    try:
        for (N, lv, ci, co, k, st, pd) in cases:
            outs = [ops.conv_out_size(H, W, k, k, st, pd) for H, W in lv]
            M = sum(N * ho * wo for ho, wo in outs)
            x = torch.randn(sum(N * H * W for H, W in lv), ci, generator=g).to(DEV, BF)
            dy = torch.randn(M, co, generator=g).to(DEV, BF)
            hs = (ctypes.c_int * len(lv))(*[h for h, _ in lv]); ws = (ctypes.c_int * len(lv))(*[w for _, w in lv])
            base = torch.randn(co, k, k, ci, generator=g).to(DEV)
            for tile in (1, 2, 4):
                assert L.brcnn_conv_set_tile_wgrad_bf16(tile) == 0
                res = {}
                for mode in (11, 11, 10):
                    assert L.brcnn_conv_set_tile_wgrad_bf16(mode) == 0
                    dw = base.clone()
                    assert L.brcnn_conv2d_wgrad_nhwc_multi(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, len(lv), hs, ws, ci, co, k, k,
                                                           st, pd, 1, None) == 0
                    torch.cuda.synchronize()
                    res.setdefault(mode, []).append(dw)
                assert torch.equal(res[11][0], res[11][1]), (tile, 'slab form not reproducible')
                scale = (res[10][0] - base).abs().max().item()
                assert (res[11][0] - res[10][0]).abs().max().item() <= 2e-5 * scale, tile
    finally:
        L.brcnn_conv_set_tile_wgrad_bf16(0)
        L.brcnn_conv_set_tile_wgrad_bf16(11)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_bf16_eight_phase_kernel_training_epilogues(dtype):
    """the dual-store forward (MODE 1) and the data gradient + BatchNorm backward (MODE 2, incl. the zero-stuffed
    stride-2 form) of the eight-phase kernel: the same bit-for-bit checks against the separate bn_act launches as the
    two-buffer kernel's, with the 256 x 256 tile forced"""
    from brcnn import lib as _lib
    L = _lib.load()
    try:
        for tile in (8844, 8842):       # (8842: the 256 x 128 two-group kernel, same epilogues through conv_pp_epilogue.h)
            assert L.brcnn_conv_set_tile_bf16(tile) == 0
            for cfg in [(8, 256, 50, 84, 1024, 1, 1, True, True, False), (2, 256, 40, 56, 256, 3, 1, False, True, True),
                        (2, 128, 30, 44, 512, 3, 2, False, True, False)] + \
                    ([(2, 128, 40, 56, 128, 3, 1, False, True, True)] if tile == 8842 else []):
                test_conv_bn_act_one_launch_training_forward(cfg, dtype)
            for cfg in [(2, 1024, 256, 1, 26, 40, False), (1, 1024, 512, 2, 13, 17, True)] + \
                    ([(2, 512, 128, 1, 26, 40, False)] if tile == 8842 else []):
                test_bottleneck_bn_backward_inside_data_gradient_launch(cfg, dtype)
    finally:
        L.brcnn_conv_set_tile_bf16(0)


def test_bf16_eight_phase_kernel_multi_level_and_data_gradient():
    """five pyramid levels in one launch and the zero-stuffed data gradient of a stride-2 conv on the eight-phase kernel"""
    from brcnn import lib as _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(34)
    sizes = [(40, 64), (20, 32), (10, 16), (5, 8), (3, 4)]
    B, C = 2, 256
    xc = torch.cat([torch.randn(B, h, w, C, generator=g).reshape(-1, C) for h, w in sizes], 0).to(DEV, BF)
    wt = (torch.randn(C, 3, 3, C, generator=g) / 48).to(DEV, BF)
    dy = torch.randn(2, 25, 42, 256, generator=g).to(DEV, BF)
    w2 = (torch.randn(256, 256, 3, 3, generator=g) / 48).to(DEV)
    try:
        outs = {}
        for t in (11, 8844, 8842):
            assert L.brcnn_conv_set_tile_bf16(t) == 0
            y, _ = ops.conv2d_nhwc_multi(xc, wt, B, sizes, None, None, None, False, 1, 1)
            x = torch.randn(2, 50, 84, 256, generator=torch.Generator().manual_seed(35)).to(DEV, BF).requires_grad_(True)
            from brcnn.autograd import conv2d_nhwc_autograd
            z = conv2d_nhwc_autograd(x, w2.clone().requires_grad_(True), None, 2, 1)
            z.backward(dy)
            outs[t] = (y, x.grad.clone())
        for t in (8844, 8842):
            assert torch.equal(outs[11][0], outs[t][0]) and torch.equal(outs[11][1], outs[t][1]), t
    finally:
        L.brcnn_conv_set_tile_bf16(0)


@pytest.mark.parametrize('shape', [
    # batch, H, W, Cin, Cout, k, residual      (8 x 50 x 84 = 33 600 rows: 263 row tiles of 128 -- the stage-3 maps)
    (8, 50, 84, 256, 256, 3, False),
    (8, 50, 84, 1024, 256, 1, False),
    (8, 50, 84, 256, 1024, 1, True),
    (8, 25, 42, 512, 512, 3, False),
    (3, 50, 84, 256, 256, 3, True),          # fewer tiles than resident workgroups on most tile shapes: plain launch
    (8, 100, 168, 256, 256, 3, False),       # 525 tiles of 256 x 256: the eight-phase kernel's stream-K case
    (8, 100, 168, 128, 128, 3, True),        # 525 tiles of 256 x 128: the two-group 128-column kernel's (stage 2)
])
def test_bf16_stream_k_schedule_is_bit_identical(shape):
    """the chained stream-K schedule (a tile that straddles two workgroup ranges is started by one workgroup and
    finished by another from the stored fp32 accumulators) keeps the MFMA chain over K: forced on (-5) it equals the
    one-tile-per-workgroup launch (-3) bit for bit on every production tile shape, with and without residual / ReLU,
    repeated (the hand-over slots and epoch flags are reused launch after launch)"""
    from brcnn import lib as _lib
    L = _lib.load()
    n, h, w_, ci, co, k, res = shape
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, h, w_, ci, generator=g).bfloat16().to(DEV)
    w = (torch.randn(co, k, k, ci, generator=g) * 0.05).bfloat16().to(DEV)
    sc = (torch.rand(co, generator=g) + 0.5).to(DEV)
    sh = torch.randn(co, generator=g).to(DEV)
    r = torch.randn(n, h, w_, co, generator=g).bfloat16().to(DEV) if res else None
    try:
        for t in (0, 11, 21, 81, 82, 8844, 8842):
            assert L.brcnn_conv_set_tile_bf16(t) == 0
            assert L.brcnn_conv_set_tile_bf16(-3) == 0
            ref = ops.conv2d_nhwc(x, w, sc, sh, r, True, 1, k // 2)
            assert L.brcnn_conv_set_tile_bf16(-5) == 0
            for rep in range(3):
                out = ops.conv2d_nhwc(x, w, sc, sh, r, True, 1, k // 2)
                assert torch.equal(out, ref), (t, rep, (out.float() - ref.float()).abs().max().item())
    finally:
        L.brcnn_conv_set_tile_bf16(0)
        L.brcnn_conv_set_tile_bf16(-4)


@pytest.mark.parametrize('et', [BF, torch.float16])
@pytest.mark.parametrize('cfg', [
    # batch, [(H, W), ...], Cin, Cout, k, stride, pad
    (2, [(40, 64)], 256, 256, 3, 1, 1),            # M = 5120: several slices, whole tiles
    (2, [(37, 53)], 128, 256, 3, 1, 1),            # ragged map (M = 3922, not a multiple of 64), K = 1152: half-empty last k tile
    (3, [(25, 42)], 512, 256, 1, 1, 0),            # plain 1x1 form, map narrower than 64 (full decode every step)
    (2, [(41, 67)], 256, 256, 3, 2, 1),            # stride 2
    (2, [(24, 40), (12, 20), (6, 10), (3, 5)], 256, 256, 3, 1, 1),     # pyramid levels sharing the weights (per-lane geometry)
    (1, [(33, 70)], 64, 320, 3, 1, 1),             # Cout not a multiple of 256 (forced): out-of-range co rows are zeros
    (4096, [(1, 1)], 1024, 1024, 1, 1, 0),         # FC shape
])
def test_wgrad_eight_phase_kernel_against_fp64(cfg, et):
    """conv_wgrad_pp_bf16.hip (256 x 256 tile, eight-phase schedule, transposing LDS reads, slab reduction) against the
    fp64 weight gradient of the same 16-bit operands: fp32 accumulation error only; bit-reproducible; the two-buffer
    kernel of conv_wgrad_bf16.hip agrees to the same bound"""
    import ctypes
    from brcnn import lib
    L = lib.load()
    n, lv, ci, co, k, st, pd = cfg
    g = torch.Generator().manual_seed(31)
    xs = [torch.randn(n, h, w, ci, generator=g).to(et) for h, w in lv]
    outs = [ops.conv_out_size(h, w, k, k, st, pd) for h, w in lv]
    dys = [torch.randn(n, ho, wo, co, generator=g).to(et) for ho, wo in outs]
    ref = torch.zeros(co, k, k, ci, dtype=torch.float64, device=DEV)
    for x_, dy_ in zip(xs, dys):        # fp64 reference on the device: dW[co, kh, kw, ci] = sum dy * window(x)
        xd = x_.to(DEV).double().permute(0, 3, 1, 2)
        dyd = dy_.to(DEV).double().permute(0, 3, 1, 2)
        gw = torch.nn.grad.conv2d_weight(xd, (co, ci, k, k), dyd, stride=st, padding=pd)
        ref += gw.permute(0, 2, 3, 1)
    x = torch.cat([t.reshape(-1, ci) for t in xs]).to(DEV)
    dy = torch.cat([t.reshape(-1, co) for t in dys]).to(DEV)
    hs = (ctypes.c_int * len(lv))(*[h for h, _ in lv])
    ws = (ctypes.c_int * len(lv))(*[w for _, w in lv])
    dt = 1 if et == BF else 3

    def run(mode):
        assert L.brcnn_conv_set_tile_wgrad_bf16(mode) == 0
        dw = torch.zeros(co, k, k, ci, device=DEV)
        st_ = L.brcnn_conv2d_wgrad_nhwc_multi(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, len(lv), hs, ws, ci, co, k, k,
                                              st, pd, dt, lib.stream_handle())
        assert st_ == 0
        return dw
    try:
        L.brcnn_conv_set_tile_wgrad_bf16(29)
        L.brcnn_conv_set_tile_wgrad_bf16(31)    # slabs added inside the producing launch (last arriver of each slice group)
        new = run(22)                   # eight-phase kernel wherever the shape allows
        again = run(22)
        assert L.brcnn_conv_set_tile_wgrad_bf16(29) == 2        # ... and it was taken both times
        for _ in range(30):             # whoever arrives last adds in the same order: same bits; counters back at zero
            assert torch.equal(run(22), new)
        L.brcnn_conv_set_tile_wgrad_bf16(29)
        L.brcnn_conv_set_tile_wgrad_bf16(30)    # slabs added by separate launches
        sep = run(22)
        assert torch.equal(run(22), sep)
        assert L.brcnn_conv_set_tile_wgrad_bf16(29) == 2
        old = run(20)                   # two-buffer kernel
        assert L.brcnn_conv_set_tile_wgrad_bf16(29) == 0
    finally:
        L.brcnn_conv_set_tile_wgrad_bf16(21)
        L.brcnn_conv_set_tile_wgrad_bf16(30)
    m_total = sum(n * ho * wo for ho, wo in outs)
    tol = 3e-6 * (m_total ** 0.5) * float(ref.abs().max()) + 1e-6      # fp32 accumulation of M products, random signs
    assert float((new.double() - ref).abs().max()) <= tol, (float((new.double() - ref).abs().max()), tol)
    assert float((old.double() - ref).abs().max()) <= tol
    assert float((sep.double() - ref).abs().max()) <= tol
    assert torch.equal(new, again)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_straight_line_readout_and_plain_setup_keep_the_bits(dtype):
    """round 4: tiles inside the output take a read-out without guards (vector scale / shift loads, packed fp32 math,
    hardware conversions, packed-int16 ReLU), 1x1 stride-1 layers a set-up without the map decode.  With the test hook
    brcnn_conv_set_tile(-4, 1) every launch takes the general forms instead: same bits -- inference layers (scale /
    shift / residual / ReLU, 1x1 and 3x3, full and ragged tiles, stride 2), and a whole training stage (dual store,
    the BatchNorm backward inside the data-gradient launches with and without the residual producer, stride-2 data
    gradients)."""
    from brcnn import lib, autograd as A
    from brcnn.backbones import Bottleneck, ResLayer
    L = lib.load()
    g = torch.Generator().manual_seed(5)

    def layers():
        outs = []
        for (n, h, w, ci, co, k, st, res, relu) in [(4, 32, 64, 64, 256, 1, 1, True, True), (2, 50, 84, 256, 64, 1, 1, False, True),
                                                    (2, 33, 41, 128, 512, 1, 1, True, False), (2, 32, 32, 128, 128, 3, 1, False, True),
                                                    (2, 32, 32, 256, 256, 3, 2, False, False), (1, 64, 64, 512, 256, 1, 1, False, False)]:
            x = torch.randn(n, h, w, ci, generator=g).to(dtype).to(DEV)
            wt = (torch.randn(co, k, k, ci, generator=g) * 0.05).to(dtype).to(DEV)
            sc = (torch.rand(co, generator=g) + 0.5).to(DEV); sh = torch.randn(co, generator=g).to(DEV)
            ho, wo = ops.conv_out_size(h, w, k, k, st, k // 2)
            r = torch.randn(n, ho, wo, co, generator=g).to(dtype).to(DEV) if res else None
            outs.append(ops.conv2d_nhwc(x, wt, scale=sc, shift=sh, residual=r, relu=relu, stride=st, pad=k // 2))
        return outs

    def stage():
        if dtype == torch.float32:
            return []
        torch.manual_seed(11)
        layer = ResLayer(Bottleneck, 256, 128, 3, 2).to(DEV)
        for m in layer.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                with torch.no_grad():
                    m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
        layer.eval()
        x = torch.randn(4, 64, 64, 256, device=DEV, generator=torch.Generator(DEV).manual_seed(2)).to(dtype).requires_grad_()
        out = layer.forward_nhwc(x)
        out.backward(torch.randn(out.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(3)).to(dtype))
        return [out.detach(), x.grad] + [p.grad for _, p in sorted(layer.named_parameters()) if p.dim() == 1]

    saved = A.WGRAD_SIDE_STREAM
    try:
        A.WGRAD_SIDE_STREAM = False
        g.manual_seed(5)
        fast = layers() + stage()
        assert L.brcnn_conv_set_tile(-4, 1) == 0
        g.manual_seed(5)
        general = layers() + stage()
    finally:
        L.brcnn_conv_set_tile(-4, 0)
        A.WGRAD_SIDE_STREAM = saved
    assert len(fast) == len(general)
    for i, (a, b) in enumerate(zip(fast, general)):
        assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize('et', [BF, torch.float16])
def test_deferred_slab_reduction_equals_the_per_layer_second_stage(et):
    """csrc/wgrad_defer.hip: with deferral on for a stream the sliced weight-gradient launches only record an item and ONE
    table-driven launch (brcnn_wgrad_defer_flush) adds the slabs -- bit for bit the dW of the per-layer second stage, for
    the three two-buffer tiles, the eight-phase kernel, one- and two-level slice grouping, accumulation into an existing dW,
    an arena too small for all layers (it flushes by itself) and a layer too large for the arena (immediate form)"""
    import ctypes
    from brcnn import lib as _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(23)
    LV = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    # batch, levels, Cin, Cout, k, stride, pad, tile hook (0: heuristic -> eight-phase where the shape allows)
    cases = [(2, LV, 256, 256, 3, 1, 1, 0), (8, LV[1:2], 256, 256, 3, 1, 1, 0), (8, LV[1:2], 1024, 256, 1, 1, 0, 0),
             (3, LV[:1], 128, 128, 3, 2, 1, 2), (8, LV[:1], 64, 64, 1, 1, 0, 1), (2, LV[:1], 128, 64, 3, 1, 1, 1),
             (8, LV[1:2], 512, 256, 1, 1, 0, 4), (4096, [(1, 1)], 1024, 1024, 1, 1, 0, 0), (8, LV[2:3], 512, 2048, 1, 1, 0, 0)]
    dt = 1 if et == BF else 3
    stream = torch.cuda.Stream(DEV)
    h = stream.cuda_stream
    layers = []
    for (N, lv, ci, co, k, st, pd, tile) in cases:
        outs = [ops.conv_out_size(H, W, k, k, st, pd) for H, W in lv]
        M = sum(N * ho * wo for ho, wo in outs)
        x = torch.randn(sum(N * H * W for H, W in lv), ci, generator=g).to(DEV, et)
        dy = torch.randn(M, co, generator=g).to(DEV, et)
        base = torch.randn(co, k, k, ci, generator=g).to(DEV)
        hs = (ctypes.c_int * len(lv))(*[h_ for h_, _ in lv]); ws = (ctypes.c_int * len(lv))(*[w for _, w in lv])
        layers.append((x, dy, base, (N, len(lv), hs, ws, ci, co, k, k, st, pd, dt), tile))

    def run_all():
        outs = []
        for x, dy, base, args, tile in layers:
            assert L.brcnn_conv_set_tile_wgrad_bf16(tile) == 0
            dw = base.clone()
            assert L.brcnn_conv2d_wgrad_nhwc_multi(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), *args, h) == 0
            outs.append(dw)
        return outs
    flushes, items = ctypes.c_longlong(0), ctypes.c_longlong(0)
    try:
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            ref = run_all()
            stream.synchronize()
            for mb, max_items, want_flushes in ((1024, 64, 1), (1024, 3, None), (96, 64, None), (8, 64, None)):
                arena = torch.empty(mb << 20, dtype=torch.uint8, device=DEV)
                L.brcnn_wgrad_defer_stats(h, ctypes.byref(flushes), ctypes.byref(items))
                f0, i0 = flushes.value, items.value
                assert L.brcnn_wgrad_defer_begin(h, arena.data_ptr(), arena.numel(), max_items) == 0
                got = run_all()
                pending = L.brcnn_wgrad_defer_pending(h)
                if mb == 1024 and max_items == 64:
                    assert pending >= 6, pending          # (the sliced layers; a one-slice launch has no second stage)
                    stream.synchronize()
                    assert not torch.equal(got[0], ref[0])      # ... and their dW is not complete before the flush
                assert L.brcnn_wgrad_defer_flush(h) == pending
                assert L.brcnn_wgrad_defer_pending(h) == 0 and L.brcnn_wgrad_defer_flush(h) == 0
                stream.synchronize()
                L.brcnn_wgrad_defer_stats(h, ctypes.byref(flushes), ctypes.byref(items))
                if want_flushes is not None:
                    assert flushes.value - f0 == want_flushes
                if mb == 8:
                    assert items.value - i0 < 6         # (most layers' slabs do not fit 8 MiB: immediate form)
                for i, (a, b) in enumerate(zip(got, ref)):
                    assert torch.equal(a, b), (mb, max_items, i, float((a - b).abs().max()))
                assert L.brcnn_wgrad_defer_begin(h, None, 0, 0) == 0
            again = run_all()           # deferral off again: the immediate form, complete without a flush
            stream.synchronize()
            for a, b in zip(again, ref):
                assert torch.equal(a, b)
    finally:
        L.brcnn_wgrad_defer_begin(h, None, 0, 0)
        L.brcnn_conv_set_tile_wgrad_bf16(0)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_deferred_bn_second_stages_give_the_same_bits(dtype):
    """round 6: the dgamma / dbeta reductions of a backward pass are recorded and run in ONE launch when the pass ends
    (`brcnn_bn_reduce_flush`) instead of ~40 small serial launches.  A ResNet stage (fused conv+BN forward, BatchNorm
    backward inside the data-gradient launches and as its own kernel) gives the same bits either way; the values are
    there when backward() returns; a parameter that already holds a gradient (accumulation) takes the immediate form."""
    from brcnn import autograd as A
    from brcnn import lib as _lib
    from brcnn.backbones import Bottleneck, ResLayer
    L = _lib.load()
    torch.manual_seed(5)
    layer = ResLayer(Bottleneck, 256, 128, 3, stride=2).to(DEV)
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    layer.eval()
    x = torch.randn(2, 40, 56, 256, device=DEV).to(dtype)
    go = None
    got = {}
    saved = A.BN_REDUCE_DEFER
    try:
        for defer in (False, True):
            A.BN_REDUCE_DEFER = defer
            layer.zero_grad(set_to_none=True)
            xd = x.clone().requires_grad_()
            out = layer.forward_nhwc(xd)
            if go is None:
                go = torch.randn(out.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(3)).to(dtype)
            out.backward(go)
            assert L.brcnn_bn_reduce_pending() == 0 and not A._bn_keep        # flushed by the end-of-pass callback
            got[defer] = {k: p.grad.clone() for k, p in layer.named_parameters() if 'bn' in k or 'downsample.1' in k}
            assert len(got[defer]) == 2 * (3 * 3 + 1)
        for k, g in got[True].items():
            assert torch.equal(g, got[False][k]), k
        # accumulation: .grad exists, autograd ADDS the new gradient right away -> the immediate form, twice the values
        A.BN_REDUCE_DEFER = True
        xd = x.clone().requires_grad_()
        layer.forward_nhwc(xd).backward(go)
        torch.cuda.synchronize()
        for k, p in layer.named_parameters():
            if k in got[True]:
                assert torch.allclose(p.grad, 2 * got[True][k], rtol=1e-6, atol=1e-6), k
    finally:
        A.BN_REDUCE_DEFER = saved


def test_fused_bottleneck_tail_16bit_equals_the_two_launches():
    """csrc/bottleneck_tail_bf16.hip: conv2 3x3 + bn2 + relu + conv3 1x1 + bn3 + identity + relu of a frozen stage-1 block in
    one launch -- the intermediate rounded to the 16-bit type as the two-launch form stores it, same K order and epilogue
    arithmetic: equal outputs"""
    g = torch.Generator().manual_seed(78)
    for (n, h, w) in [(1, 8, 16), (2, 16, 24), (1, 32, 40), (2, 50, 64)]:
        x = torch.randn(n, h, w, 64, generator=g).to(DEV, BF)
        idn = torch.randn(n, h, w, 256, generator=g).to(DEV, BF)
        w2 = (torch.randn(64, 3, 3, 64, generator=g) / 24).to(DEV, BF)
        w3 = (torch.randn(256, 1, 1, 64, generator=g) / 8).to(DEV, BF)
        s2, b2 = (torch.rand(64, generator=g) + 0.5).to(DEV), torch.randn(64, generator=g).to(DEV)
        s3, b3 = (torch.rand(256, generator=g) + 0.5).to(DEV), torch.randn(256, generator=g).to(DEV)
        t = ops.conv2d_nhwc(x, w2, s2, b2, None, True, 1, 1)
        ref = ops.conv2d_nhwc(t, w3, s3, b3, idn, True, 1, 0)
        assert ops.bottleneck_tail_supported(x, w2, w3, idn)
        y = ops.bottleneck_tail_nhwc(x, w2, s2, b2, w3, s3, b3, idn)
        assert y.dtype == ref.dtype and torch.equal(y, ref), (n, h, w, (y.float() - ref.float()).abs().max().item())
