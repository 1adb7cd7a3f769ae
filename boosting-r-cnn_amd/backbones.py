"""ResNet-50/101 backbone, executed as NHWC implicit-GEMM convolutions with eval-mode BN,
ReLU and the residual add folded into the conv epilogues.

Mirrors `mmdet/models/backbones/resnet.py` (`Bottleneck` :97-302, `ResNet` :306-657) and
`mmdet/models/utils/res_layer.py`: same constructor arguments as the configs pass
(`depth, num_stages, out_indices, frozen_stages, norm_cfg, norm_eval, style, init_cfg`), the
same state-dict key layout (`conv1/bn1/layer{1-4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,
downsample.{0,1}}`), the same freeze / norm_eval behaviour in `train()`.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .blocks import PackedCache, build_norm_layer, conv_bn_act_nhwc, folded_conv_operands, to_nchw_view, to_nhwc
from .registry import BACKBONES


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, style='pytorch',
                 norm_cfg=dict(type='BN')):
        super().__init__()
        assert style in ('pytorch', 'caffe')
        self.inplanes, self.planes, self.stride, self.style = inplanes, planes, stride, style
        if style == 'pytorch':      # stride on the 3x3 (resnet.py:150-155)
            self.conv1_stride, self.conv2_stride = 1, stride
        else:
            self.conv1_stride, self.conv2_stride = stride, 1
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=self.conv1_stride, bias=False)
        self.add_module('bn1', build_norm_layer(norm_cfg, planes, 1)[1])
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=self.conv2_stride, padding=1, bias=False)
        self.add_module('bn2', build_norm_layer(norm_cfg, planes, 2)[1])
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.add_module('bn3', build_norm_layer(norm_cfg, planes * self.expansion, 3)[1])
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self._c = [PackedCache() for _ in range(4)]

    chain_fusion = True        # ResLayer may pass `out_single_use` / `in_from_prev`

    FUSE_TAIL = os.environ.get('BRCNN_FUSE_BLOCK_TAIL', '1') != '0'

    def _tail_fusable(self, out, identity):
        """conv2 + bn2 + relu + conv3 + bn3 + identity + relu as one launch: fp32, nothing to differentiate, the 64 -> 64 (3x3,
        stride 1) -> 256 shape of stage 1"""
        from .autograd import wants_grad
        c2, c3 = self.conv2, self.conv3
        if not (self.FUSE_TAIL and type(self) is Bottleneck and out.is_cuda and identity.dtype == out.dtype and
                out.dtype in (torch.float32, torch.bfloat16, torch.float16) and not self.bn2.training and not self.bn3.training):
            return False
        if not (c2.kernel_size == (3, 3) and c2.stride == (1, 1) and c2.padding == (1, 1) and c2.dilation == (1, 1) and
                c2.groups == 1 and c2.in_channels == 64 and c2.out_channels == 64 and c2.bias is None and
                c3.kernel_size == (1, 1) and c3.stride == (1, 1) and c3.groups == 1 and c3.in_channels == 64 and
                c3.out_channels == 256 and c3.bias is None):
            return False
        if out.dim() != 4 or (out.shape[0] * out.shape[1] * out.shape[2]) % (64 if out.dtype == torch.float32 else 128) or \
                tuple(identity.shape) != tuple(out.shape[:3]) + (256,):
            return False
        return not wants_grad(out, identity, c2.weight, c3.weight, self.bn2.weight, self.bn2.bias, self.bn3.weight, self.bn3.bias)

    def forward_nhwc(self, x, out_single_use=False, in_from_prev=False):
        """`out_single_use`: the caller promises that the result feeds only the next block of the stage (whose
        conv1 then runs bn3's backward inside its data-gradient launch); `in_from_prev`: x is such a result"""
        if self.downsample is None:
            # the identity branch leaves through conv1's autograd node: its gradient is added in
            # conv1's data-gradient kernel instead of by a separate add over the whole tensor
            out, identity = conv_bn_act_nhwc(x, self.conv1, self.bn1, self._c[0], True, with_skip=True,
                                             sole_consumer=in_from_prev, single_use_output=True)
        else:
            # the downsample branch reads conv1's input through the same alias: its data gradient is added in
            # conv1's data-gradient epilogue too (stride-1 conv1, the 'pytorch' style) instead of by an autograd add
            out, xs = conv_bn_act_nhwc(x, self.conv1, self.bn1, self._c[0], True, with_skip=True, single_use_output=True)
            identity = conv_bn_act_nhwc(xs, self.downsample[0], self.downsample[1], self._c[3], False)
        if self._tail_fusable(out, identity):
            # frozen / inference block with 64 -> 64 -> 256 channels (stage 1): conv2 .. the block's ReLU in ONE launch,
            # the 64-channel intermediate stays in LDS (csrc/bottleneck_tail_f32.hip / _bf16.hip; equal to the two launches)
            w2, s2, b2 = folded_conv_operands(self.conv2, self.bn2, self._c[1], out.dtype)
            w3, s3, b3 = folded_conv_operands(self.conv3, self.bn3, self._c[2], out.dtype)
            return ops.bottleneck_tail_nhwc(out, w2, s2, b2, w3, s3, b3, identity)
        # out of conv1 / conv2 feeds the next conv only: that conv's data-gradient launch runs bn1's / bn2's backward
        out = conv_bn_act_nhwc(out, self.conv2, self.bn2, self._c[1], True, sole_consumer=True, single_use_output=True)
        # relu(bn3(conv3(out)) + identity) in one epilogue (resnet.py:288-300)
        return conv_bn_act_nhwc(out, self.conv3, self.bn3, self._c[2], True, residual=identity, sole_consumer=True,
                                single_use_output=out_single_use)

    def forward(self, x):
        return to_nchw_view(self.forward_nhwc(to_nhwc(x)))


class ResLayer(nn.Sequential):
    """mmdet/models/utils/res_layer.py: first block carries the stride and the 1x1 downsample."""

    def __init__(self, block, inplanes, planes, num_blocks, stride=1, style='pytorch',
                 norm_cfg=dict(type='BN'), **block_kwargs):
        downsample = None
        if stride != 1 or inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                build_norm_layer(norm_cfg, planes * block.expansion)[1])
        layers = [block(inplanes, planes, stride, downsample, style, norm_cfg, **block_kwargs)]
        inplanes = planes * block.expansion
        for _ in range(1, num_blocks):
            layers.append(block(inplanes, planes, 1, None, style, norm_cfg, **block_kwargs))
        super().__init__(*layers)

    def forward_nhwc(self, x):
        n = len(self)
        for i, blk in enumerate(self):
            if getattr(type(blk), 'chain_fusion', False):
                # the output of every block but the last feeds only the next block of the stage
                x = blk.forward_nhwc(x, out_single_use=i + 1 < n, in_from_prev=i > 0)
            else:
                x = blk.forward_nhwc(x)
        return x


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)),
                     152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3),
                 style='pytorch', deep_stem=False, avg_down=False, frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False,
                 zero_init_residual=True, pretrained=None, init_cfg=None):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet (hot path: 50/101/152)')
        assert not deep_stem and not avg_down and dcn is None and plugins is None and \
            conv_cfg is None, 'ResNetV1d / DCN / plugin variants are outside the hot path'
        assert all(d == 1 for d in dilations)
        assert 1 <= num_stages <= 4 and max(out_indices) < num_stages
        self.depth, self.num_stages, self.out_indices = depth, num_stages, out_indices
        self.style, self.frozen_stages, self.norm_cfg, self.norm_eval = \
            style, frozen_stages, norm_cfg, norm_eval
        self.zero_init_residual = zero_init_residual
        self.init_cfg = init_cfg
        stem_channels = stem_channels or base_channels
        self.block, stage_blocks = self.arch_settings[depth]
        self.stage_blocks = stage_blocks[:num_stages]
        self.inplanes = stem_channels
        self.conv1 = nn.Conv2d(in_channels, stem_channels, 7, stride=2, padding=3, bias=False)
        self.add_module('bn1', build_norm_layer(norm_cfg, stem_channels, 1)[1])
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.res_layers = []
        for i, num_blocks in enumerate(self.stage_blocks):
            planes = base_channels * 2 ** i
            layer = ResLayer(self.block, self.inplanes, planes, num_blocks, strides[i], style,
                             norm_cfg, **self.block_kwargs())
            self.inplanes = planes * self.block.expansion
            name = f'layer{i + 1}'
            self.add_module(name, layer)
            self.res_layers.append(name)
        self.feat_dim = self.block.expansion * base_channels * 2 ** (len(self.stage_blocks) - 1)
        self._stem_cache = PackedCache()
        self._stem_cache2 = PackedCache()
        self._freeze_stages()
        self.init_weights()

    def block_kwargs(self):
        """extra constructor arguments of the residual block (ResNeXt: groups / base_width)"""
        return {}

    def init_weights(self, pretrained=False):
        """Kaiming (fan_out, relu) for convs, constant 1 for norms, zero for the last BN of each
        block (resnet.py:392-412).  `pretrained=True` (the detector's `init_weights()`, i.e. the train
        driver) then loads `init_cfg=dict(type='Pretrained', checkpoint=...)` from the local model
        directory (`blocks.load_pretrained`); construction alone never touches the file system."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, a=0, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual and not (isinstance(self.init_cfg, dict) and
                                            self.init_cfg.get('type') == 'Pretrained'):
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
        if pretrained:
            from .blocks import load_pretrained
            load_pretrained(self, self.init_cfg)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def _stem_fast_ok(self):
        c = self.conv1
        return c.kernel_size == (7, 7) and c.stride == (2, 2) and c.padding == (3, 3) and \
            c.in_channels == 3 and c.bias is None and not c.weight.requires_grad

    # the frozen 64-channel stem and its max-pool as one launch (csrc/stem_pool.hip); BRCNN_STEM_POOL_FUSED=0: two launches
    FUSED_STEM_POOL = os.environ.get('BRCNN_STEM_POOL_FUSED', '1') != '0'

    def stem_from_nchw(self, img):
        """frozen stem straight from the NCHW image (no NHWC copy of the input): 7x7 conv + folded BN + ReLU +
        3x3/s2 max-pool in one launch (64 stem channels), else repack + vector-path conv, then the max-pool"""
        fused = self.FUSED_STEM_POOL and self.conv1.out_channels == 64

        def builder():
            from .blocks import fold_bn, compute_dtype
            scale, shift = fold_bn(self.bn1)
            pack = ops.pack_stem_pool_weight if fused else ops.pack_stem_weight
            return pack(self.conv1.weight, compute_dtype()), scale, shift
        w, scale, shift = self._stem_cache2.get(
            [self.conv1.weight, self.bn1.weight, self.bn1.bias, self.bn1.running_mean, self.bn1.running_var],
            builder)
        if fused:
            return ops.stem7x7s2_pool_nchw(img, w, scale, shift)
        return ops.maxpool3x3s2_nhwc(ops.stem7x7s2_nchw(img, w, scale, shift, True))

    supports_tap = True          # forward_nhwc / forward_from_nchw take `tap` (see _stages)

    def forward_from_nchw(self, img, tap=None):
        """(N,3,H,W) NCHW image -> tuple of NHWC stage outputs; `tap`: see _stages"""
        if self._stem_fast_ok() and not self.bn1.training and img.is_contiguous() and \
                img.shape[2] >= 7 and img.shape[3] >= 7:
            return self._stages(self.stem_from_nchw(img), tap)
        return self.forward_nhwc(to_nhwc(img), tap)

    def _stages(self, x, tap=None):
        """`tap(k, x)` (training, optional): called with the k-th OUTPUT stage's result as soon as it exists; what it
        returns replaces x both as that output and as the next stage's input.  The detector passes the neck's
        `FPN.lateral_tap`: the lateral conv of level k runs right here and hands back an alias of its input, so the
        gradient of everything downstream of the alias (the next stage) is added inside the lateral conv's
        data-gradient launch instead of by an aten add over the whole stage output (autograd.ConvNHWCFunction,
        `with_skip`; C3 / C4 at batch 8: 137 / 69 MB tensors, 0.13 ms per bf16 train step)."""
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name).forward_nhwc(x)
            if i in self.out_indices:
                if tap is not None:
                    x = tap(len(outs), x)
                outs.append(x)
        return tuple(outs)

    def forward_nhwc(self, x, tap=None):
        """x (N,H,W,3) -> tuple of (N,h,w,C) for out_indices"""
        x = conv_bn_act_nhwc(x, self.conv1, self.bn1, self._stem_cache, True)
        x = ops.maxpool3x3s2_nhwc(x)
        return self._stages(x, tap)

    def forward(self, x):
        return tuple(to_nchw_view(o) for o in self.forward_nhwc(to_nhwc(x)))


# --------------------------------------------------------------------------- ResNeXt
class BottleneckX(Bottleneck):
    """resnext.py:10-84: conv1 / conv2 / conv3 run at `width = floor(planes * base_width /
    base_channels) * groups`, conv2 is a grouped 3x3"""

    def __init__(self, inplanes, planes, stride=1, downsample=None, style='pytorch', norm_cfg=dict(type='BN'),
                 groups=1, base_width=4, base_channels=64):
        super().__init__(inplanes, planes, stride, downsample, style, norm_cfg)
        import math
        width = planes if groups == 1 else math.floor(planes * (base_width / base_channels)) * groups
        self.groups, self.width = groups, width
        self.conv1 = nn.Conv2d(inplanes, width, 1, stride=self.conv1_stride, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, width, 1)[1]
        self.conv2 = nn.Conv2d(width, width, 3, stride=self.conv2_stride, padding=1, groups=groups, bias=False)
        self.bn2 = build_norm_layer(norm_cfg, width, 2)[1]
        self.conv3 = nn.Conv2d(width, planes * self.expansion, 1, bias=False)


@BACKBONES.register_module()
class ResNeXt(ResNet):
    """mmdet/models/backbones/resnext.py:87-153 (the x101 64x4d recipe); the grouped 3x3 convs run
    as block-diagonal 64-channel tiles on the MFMA kernel (`brcnn_conv2d_nhwc_grouped`)"""
    arch_settings = {50: (BottleneckX, (3, 4, 6, 3)), 101: (BottleneckX, (3, 4, 23, 3)),
                     152: (BottleneckX, (3, 8, 36, 3))}

    def __init__(self, groups=1, base_width=4, **kwargs):
        self.groups, self.base_width = groups, base_width
        self._base_channels = kwargs.get('base_channels', 64)
        super().__init__(**kwargs)

    def block_kwargs(self):
        return dict(groups=self.groups, base_width=self.base_width, base_channels=self._base_channels)


# --------------------------------------------------------------------------- Res2Net (+ DCNv2)
class DCNv2(nn.Module):
    """mmcv.ops.ModulatedDeformConv2dPack (the `dcn=dict(type='DCNv2', deform_groups=1)` of the
    r2_101 recipes): a 3x3 conv whose taps are displaced by learned offsets and scaled by a
    learned mask, both predicted by `conv_offset` (zero-initialised).  Parameter names as mmcv's:
    `weight`, `conv_offset.{weight,bias}`."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1,
                 deform_groups=1, bias=False):
        super().__init__()
        assert groups == 1 and deform_groups == 1 and kernel_size == 3 and not bias
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = (3, 3), stride, padding, dilation
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.conv_offset = nn.Conv2d(in_channels, 27, 3, stride=stride, padding=padding, dilation=dilation, bias=True)
        nn.init.kaiming_uniform_(self.weight, nonlinearity='relu')
        nn.init.constant_(self.conv_offset.weight, 0)
        nn.init.constant_(self.conv_offset.bias, 0)


def _pad_to(n, mult=32):
    return (n + mult - 1) // mult * mult


class Bottle2neck(nn.Module):
    """res2net.py:13-162.  The `scales` splits of conv1's output are `width` = 26/52/104/208 channels
    wide; here every split is laid out padded to a multiple of 32 channels (zero filters / zero
    BN affine in the pad slots, so the pad channels stay exactly 0 through ReLU, the hierarchical
    adds and the concat), which keeps all convs on the vector (LDS-DMA) MFMA path; the packed,
    padded weights are functions of the reference-layout parameters."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, style='pytorch', norm_cfg=dict(type='BN'),
                 scales=4, base_width=26, base_channels=64, stage_type='normal', dcn=None):
        super().__init__()
        import math
        assert scales > 1 and style == 'pytorch'
        self.inplanes, self.planes, self.stride, self.style = inplanes, planes, stride, style
        self.conv1_stride, self.conv2_stride = 1, stride
        width = int(math.floor(planes * (base_width / base_channels)))
        self.width, self.scales, self.stage_type = width, scales, stage_type
        self.with_dcn = dcn is not None
        self.conv1 = nn.Conv2d(inplanes, width * scales, 1, stride=1, bias=False)
        self.add_module('bn1', build_norm_layer(norm_cfg, width * scales, 1)[1])
        if stage_type == 'stage' and stride != 1:
            self.pool = nn.AvgPool2d(kernel_size=3, stride=stride, padding=1)
        convs, bns = [], []
        for i in range(scales - 1):
            if self.with_dcn:
                convs.append(DCNv2(width, width, 3, stride=stride, padding=1, deform_groups=dcn.get('deform_groups', 1)))
            else:
                convs.append(nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False))
            bns.append(build_norm_layer(norm_cfg, width, i + 1)[1])
        self.convs, self.bns = nn.ModuleList(convs), nn.ModuleList(bns)
        self.conv3 = nn.Conv2d(width * scales, planes * self.expansion, 1, bias=False)
        self.add_module('bn3', build_norm_layer(norm_cfg, planes * self.expansion, 3)[1])
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self._cache = PackedCache()
        self._c_down = PackedCache()

    # ---- padded, packed tensors (inference: cached; rebuilt when a parameter changes) --------
    def _packed(self):
        from .blocks import fold_bn, pack_weight
        w, s, wp = self.width, self.scales, _pad_to(self.width)

        def builder():
            d = {}
            s1, b1 = fold_bn(self.bn1)
            w1 = self.conv1.weight.detach().float().view(s, w, self.inplanes)
            d['w1'] = torch.nn.functional.pad(w1, (0, 0, 0, wp - w)).reshape(s * wp, 1, 1, self.inplanes).contiguous()
            d['s1'] = torch.nn.functional.pad(s1.view(s, w), (0, wp - w)).reshape(-1).contiguous()
            d['b1'] = torch.nn.functional.pad(b1.view(s, w), (0, wp - w)).reshape(-1).contiguous()
            for i, (conv, bn) in enumerate(zip(self.convs, self.bns)):
                si, bi = fold_bn(bn)
                wi = pack_weight(conv.weight)                                   # (w, 3, 3, w)
                d[f'cw{i}'] = torch.nn.functional.pad(wi, (0, wp - w, 0, 0, 0, 0, 0, wp - w)).contiguous()
                d[f'cs{i}'] = torch.nn.functional.pad(si, (0, wp - w)).contiguous()
                d[f'cb{i}'] = torch.nn.functional.pad(bi, (0, wp - w)).contiguous()
                if self.with_dcn:
                    wo = pack_weight(conv.conv_offset.weight)                   # (27, 3, 3, w)
                    d[f'wo{i}'] = torch.nn.functional.pad(wo, (0, wp - w)).contiguous()
                    d[f'bo{i}'] = conv.conv_offset.bias.detach().float().contiguous()
            s3, b3 = fold_bn(self.bn3)
            w3 = self.conv3.weight.detach().float().view(-1, s, w)
            d['w3'] = torch.nn.functional.pad(w3, (0, wp - w)).reshape(-1, 1, 1, s * wp).contiguous()
            d['s3'], d['b3'] = s3, b3
            return d
        srcs = [p for p in self.parameters()] + [b for b in self.buffers() if b.dtype.is_floating_point]
        return self._cache.get(srcs, builder)

    def _conv_i(self, d, i, sp):
        """convs[i] + bns[i] + ReLU on a (N,h,w,wp) split"""
        conv = self.convs[i]
        if not self.with_dcn:
            return ops.conv2d_nhwc(sp, d[f'cw{i}'], d[f'cs{i}'], d[f'cb{i}'], None, True, self.conv2_stride, 1)
        om = ops.conv2d_nhwc(sp, d[f'wo{i}'], None, d[f'bo{i}'], None, False, self.conv2_stride, 1)
        col, (ho, wo) = ops.deform_im2col_nhwc(sp, om, 3, self.conv2_stride, 1, 1)
        n, wp = sp.shape[0], sp.shape[3]
        y = ops.conv2d_nhwc(col.view(n * ho * wo, 1, 1, 9 * wp), d[f'cw{i}'].reshape(wp, 1, 1, 9 * wp),
                            d[f'cs{i}'], d[f'cb{i}'], None, True, 1, 0)
        return y.view(n, ho, wo, wp)

    # ---- training: the same padded layout built differentiably from the parameters ----------
    def _affine(self, bn, pad_to=None):
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
        return scale, shift

    def _bn_act(self, y, scale, shift, residual, relu):
        from .autograd import bn_act_autograd, bn_act_supported
        if bn_act_supported(y) and (residual is None or residual.dtype == y.dtype):
            return bn_act_autograd(y, scale, shift, residual, relu)
        y = y * scale + shift
        if residual is not None:
            y = y + residual
        return y.relu() if relu else y

    def _forward_train(self, x):
        import torch.nn.functional as F
        from .autograd import conv2d_nhwc_autograd, deform_im2col_autograd, linear_autograd
        w, s, wp = self.width, self.scales, _pad_to(self.width)
        pw = wp - w
        s1, b1 = self._affine(self.bn1)
        w1 = F.pad(self.conv1.weight.view(s, w, self.inplanes), (0, 0, 0, pw)).reshape(s * wp, self.inplanes, 1, 1)
        out = conv2d_nhwc_autograd(x, w1, None, 1, 0)
        out = self._bn_act(out, F.pad(s1.view(s, w), (0, pw)).reshape(-1), F.pad(b1.view(s, w), (0, pw)).reshape(-1),
                           None, True)
        spx = [out[..., i * wp:(i + 1) * wp] for i in range(s)]
        outs, sp = [], None
        for i in range(s - 1):
            sp = spx[i].contiguous() if (i == 0 or self.stage_type == 'stage') else sp + spx[i]
            conv, bn = self.convs[i], self.bns[i]
            si, bi = self._affine(bn)
            wi = F.pad(conv.weight, (0, 0, 0, 0, 0, pw, 0, pw))                      # (wp, wp, 3, 3)
            if self.with_dcn:
                wo = F.pad(conv.conv_offset.weight, (0, 0, 0, 0, 0, pw))             # (27, wp, 3, 3)
                om = conv2d_nhwc_autograd(sp, wo, conv.conv_offset.bias, self.conv2_stride, 1)
                col = deform_im2col_autograd(sp, om, self.conv2_stride, 1)
                n, ho, wo_ = om.shape[0], om.shape[1], om.shape[2]
                y = linear_autograd(col, wi.permute(0, 2, 3, 1).reshape(wp, 9 * wp), None).view(n, ho, wo_, wp)
            else:
                y = conv2d_nhwc_autograd(sp, wi, None, self.conv2_stride, 1)
            sp = self._bn_act(y, F.pad(si, (0, pw)), F.pad(bi, (0, pw)), None, True)
            outs.append(sp)
        if self.stage_type == 'normal' or self.conv2_stride == 1:
            outs.append(spx[s - 1])
        else:
            outs.append(F.avg_pool2d(spx[s - 1].permute(0, 3, 1, 2), 3, self.conv2_stride, 1).permute(0, 2, 3, 1))
        cat = torch.cat(outs, 3).contiguous()
        identity = x
        if self.downsample is not None:
            pool, conv, bn = self.downsample[0], self.downsample[1], self.downsample[2]
            k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
            xi = x if k == 1 else F.avg_pool2d(x.permute(0, 3, 1, 2), k, k, 0, ceil_mode=True,
                                               count_include_pad=False).permute(0, 2, 3, 1).contiguous()
            identity = conv_bn_act_nhwc(xi, conv, bn, self._c_down, False)
        s3, b3 = self._affine(self.bn3)
        w3 = F.pad(self.conv3.weight.view(-1, s, w), (0, pw)).reshape(-1, s * wp, 1, 1)
        y = conv2d_nhwc_autograd(cat, w3, None, 1, 0)
        return self._bn_act(y, s3, b3, identity.contiguous(), True)

    def forward_nhwc(self, x):
        if x.dtype != torch.float32:
            raise NotImplementedError('Res2Net runs in fp32 only this round')
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self._forward_train(x)
        d = self._packed()
        wp, s = _pad_to(self.width), self.scales
        out = ops.conv2d_nhwc(x, d['w1'], d['s1'], d['b1'], None, True, 1, 0)     # (N,h,w,s*wp)
        spx = [out[..., i * wp:(i + 1) * wp] for i in range(s)]
        sp = self._conv_i(d, 0, spx[0].contiguous())
        outs = [sp]
        for i in range(1, s - 1):
            sp = spx[i].contiguous() if self.stage_type == 'stage' else sp + spx[i]
            sp = self._conv_i(d, i, sp)
            outs.append(sp)
        if self.stage_type == 'normal' or self.conv2_stride == 1:
            outs.append(spx[s - 1])
        else:
            outs.append(ops.avgpool_nhwc(spx[s - 1].contiguous(), 3, self.conv2_stride, 1, False, True))
        cat = torch.cat(outs, 3)
        identity = x
        if self.downsample is not None:
            pool, conv, bn = self.downsample[0], self.downsample[1], self.downsample[2]
            k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
            xi = x if k == 1 else ops.avgpool_nhwc(x, k, k, 0, True, False)
            identity = conv_bn_act_nhwc(xi, conv, bn, self._c_down, False)
        return ops.conv2d_nhwc(cat, d['w3'], d['s3'], d['b3'], identity, True, 1, 0)

    def forward(self, x):
        return to_nchw_view(self.forward_nhwc(to_nhwc(x)))


class Res2Layer(nn.Sequential):
    """res2net.py:165-233: avg_down shortcut (AvgPool(stride, ceil_mode, no pad count) + 1x1 conv + BN),
    first block of a stage is of type 'stage'"""

    def __init__(self, block, inplanes, planes, num_blocks, stride=1, style='pytorch', norm_cfg=dict(type='BN'),
                 scales=4, base_width=26, base_channels=64, dcn=None):
        downsample = None
        if stride != 1 or inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.AvgPool2d(kernel_size=stride, stride=stride, ceil_mode=True, count_include_pad=False),
                nn.Conv2d(inplanes, planes * block.expansion, 1, stride=1, bias=False),
                build_norm_layer(norm_cfg, planes * block.expansion)[1])
        kw = dict(scales=scales, base_width=base_width, base_channels=base_channels, dcn=dcn)
        layers = [block(inplanes, planes, stride, downsample, style, norm_cfg, stage_type='stage', **kw)]
        inplanes = planes * block.expansion
        for _ in range(1, num_blocks):
            layers.append(block(inplanes, planes, 1, None, style, norm_cfg, **kw))
        super().__init__(*layers)

    def forward_nhwc(self, x):
        n = len(self)
        for i, blk in enumerate(self):
            if getattr(type(blk), 'chain_fusion', False):
                # the output of every block but the last feeds only the next block of the stage
                x = blk.forward_nhwc(x, out_single_use=i + 1 < n, in_from_prev=i > 0)
            else:
                x = blk.forward_nhwc(x)
        return x


@BACKBONES.register_module()
class Res2Net(nn.Module):
    """mmdet/models/backbones/res2net.py:236-327 (always the v1d form: deep 3x3 stem, avg_down),
    with optional DCNv2 in the stages of `stage_with_dcn` (resnet.py:500-520)"""
    arch_settings = {50: (Bottle2neck, (3, 4, 6, 3)), 101: (Bottle2neck, (3, 4, 23, 3)),
                     152: (Bottle2neck, (3, 8, 36, 3))}

    def __init__(self, depth, scales=4, base_width=26, in_channels=3, stem_channels=None, base_channels=64,
                 num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3),
                 style='pytorch', deep_stem=True, avg_down=True, frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False,
                 zero_init_residual=True, pretrained=None, init_cfg=None):
        super().__init__()
        assert depth in self.arch_settings and conv_cfg is None and plugins is None
        assert all(d == 1 for d in dilations) and 1 <= num_stages <= 4 and max(out_indices) < num_stages
        self.depth, self.scales, self.base_width = depth, scales, base_width
        self.num_stages, self.out_indices, self.frozen_stages = num_stages, out_indices, frozen_stages
        self.norm_cfg, self.norm_eval, self.init_cfg = norm_cfg, norm_eval, init_cfg
        self.zero_init_residual = zero_init_residual
        stem_channels = stem_channels or base_channels
        block, stage_blocks = self.arch_settings[depth]
        h = stem_channels // 2
        self.stem = nn.Sequential(
            nn.Conv2d(in_channels, h, 3, stride=2, padding=1, bias=False), build_norm_layer(norm_cfg, h)[1],
            nn.ReLU(inplace=True),
            nn.Conv2d(h, h, 3, stride=1, padding=1, bias=False), build_norm_layer(norm_cfg, h)[1],
            nn.ReLU(inplace=True),
            nn.Conv2d(h, stem_channels, 3, stride=1, padding=1, bias=False),
            build_norm_layer(norm_cfg, stem_channels)[1], nn.ReLU(inplace=True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.res_layers, inplanes = [], stem_channels
        for i, nb in enumerate(stage_blocks[:num_stages]):
            planes = base_channels * 2 ** i
            stage_dcn = None
            if dcn is not None and stage_with_dcn[i]:
                stage_dcn = dict(dcn)
                stage_dcn.pop('fallback_on_stride', None)
                assert stage_dcn.pop('type') == 'DCNv2'
            layer = Res2Layer(block, inplanes, planes, nb, strides[i], style, norm_cfg, scales, base_width,
                              base_channels, stage_dcn)
            inplanes = planes * block.expansion
            self.add_module(f'layer{i + 1}', layer)
            self.res_layers.append(f'layer{i + 1}')
        self.feat_dim = inplanes
        self._stem_caches = [PackedCache() for _ in range(3)]
        self._freeze_stages()
        self.init_weights()

    def init_weights(self, pretrained=False):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, a=0, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        for m in self.modules():
            if isinstance(m, DCNv2):
                nn.init.constant_(m.conv_offset.weight, 0)
                nn.init.constant_(m.conv_offset.bias, 0)
        if self.zero_init_residual and not (isinstance(self.init_cfg, dict) and
                                            self.init_cfg.get('type') == 'Pretrained'):
            for m in self.modules():
                if isinstance(m, Bottle2neck):
                    nn.init.constant_(m.bn3.weight, 0)
        if pretrained:
            from .blocks import load_pretrained
            load_pretrained(self, self.init_cfg)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.stem.eval()
            for p in self.stem.parameters():
                p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def forward_nhwc(self, x):
        for j in range(3):
            x = conv_bn_act_nhwc(x, self.stem[3 * j], self.stem[3 * j + 1], self._stem_caches[j], True)
        x = ops.maxpool3x3s2_nhwc(x)
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name).forward_nhwc(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def forward(self, x):
        return tuple(to_nchw_view(o) for o in self.forward_nhwc(to_nhwc(x)))
