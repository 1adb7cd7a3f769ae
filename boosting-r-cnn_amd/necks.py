"""FPN / PAFPN necks on NHWC activations.

Mirrors `mmdet/models/necks/fpn.py:63-204` and `mmdet/models/necks/pafpn.py:12-158`
(constructor arguments, `lateral_convs / fpn_convs / downsample_convs / pafpn_convs`
ModuleLists of ConvModules -> the reference's state-dict keys, forward data flow).  The
top-down `laterals[i-1] += interpolate(laterals[i])` is one fused nearest-upsample-add
kernel; the bottom-up `inter_outs[i+1] += downsample_conv(inter_outs[i])` is the residual
operand of the stride-2 conv's epilogue.
"""
import torch.nn as nn

from . import ops
from .blocks import ConvModule, to_nchw_view, to_nhwc
from .registry import NECKS


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1,
                 add_extra_convs=False, relu_before_extra_convs=False, no_norm_on_lateral=False,
                 conv_cfg=None, norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest'),
                 init_cfg=dict(type='Xavier', layer='Conv2d', distribution='uniform')):
        super().__init__()
        assert isinstance(in_channels, list)
        assert upsample_cfg.get('mode', 'nearest') == 'nearest' and 'scale_factor' not in upsample_cfg
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.relu_before_extra_convs = relu_before_extra_convs
        self.no_norm_on_lateral = no_norm_on_lateral
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels)
            assert num_outs == end_level - start_level
        self.start_level, self.end_level = start_level, end_level
        self.add_extra_convs = add_extra_convs
        assert isinstance(add_extra_convs, (str, bool))
        if isinstance(add_extra_convs, str):
            assert add_extra_convs in ('on_input', 'on_lateral', 'on_output')
        elif add_extra_convs:
            self.add_extra_convs = 'on_input'
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(
                in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                norm_cfg=norm_cfg if not no_norm_on_lateral else None, act_cfg=act_cfg,
                inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1,
                                             conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg,
                                             inplace=False))
        extra_levels = num_outs - self.backbone_end_level + self.start_level
        if self.add_extra_convs and extra_levels >= 1:
            for i in range(extra_levels):
                if i == 0 and self.add_extra_convs == 'on_input':
                    cin = self.in_channels[self.backbone_end_level - 1]
                else:
                    cin = out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1,
                                                 conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                                 act_cfg=act_cfg, inplace=False))
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight, gain=1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    # ---- gradient fan-in without aten adds (training) ------------------------------------------------
    # A tensor with two consumers gets its gradient as an autograd add over the whole tensor.  Where one of the
    # consumers is a stride-1 conv, that conv can hand out an ALIAS of its input (`with_skip`): whatever the other
    # consumer sends back through the alias is added in the conv's data-gradient epilogue (the residual operand of the
    # MFMA kernel) -- no extra launch, one stream of bytes less than the add.  Used for
    #   * the backbone's stage outputs: lateral conv (alias -> the next stage), `lateral_tap`;
    #   * the top-down path: laterals[i] feeds fpn_convs[i] (alias -> the upsample-add into level i-1);
    #   * PAFPN's bottom-up path: inter[i] feeds pafpn_convs[i-1] (alias -> downsample_convs[i]).
    # Same values forward; the gradient sums have the same terms in another association.
    FUSE_FAN_IN = __import__('os').environ.get('BRCNN_FUSE_FAN_IN', '1') != '0'      # (A/B switch)

    @staticmethod
    def _skip_ok(conv_module, x):
        import torch
        c = conv_module.conv
        return (FPN.FUSE_FAN_IN and torch.is_grad_enabled() and x.is_cuda and x.requires_grad and c.groups == 1 and
                c.stride == (1, 1) and c.kernel_size[0] == c.kernel_size[1] and
                2 * c.padding[0] == c.kernel_size[0] - 1 and c.dilation == (1, 1) and
                (conv_module.norm is None or isinstance(conv_module.norm, nn.BatchNorm2d)))

    # ---- the output levels back to back in one buffer (what the RPN's multi-level launches read) --------------------
    SHARED_OUTPUT_BUFFER = __import__('os').environ.get('BRCNN_PYRAMID_BUFFER', '1') != '0'       # (A/B switch)

    def _out_views(self, laterals):
        """per output level a (rows, C) slice of ONE buffer, or None: the convs that produce the levels write there
        (ops.output_into), so `autograd.cat_rows` needs no copy"""
        import torch
        if not (self.SHARED_OUTPUT_BUFFER and laterals[0].is_cuda) or self.num_outs < len(laterals) or \
                (self.num_outs > len(laterals) and not self.add_extra_convs):
            return [None] * self.num_outs
        b = laterals[0].shape[0]
        hw = [tuple(l.shape[1:3]) for l in laterals]
        for i in range(len(laterals), self.num_outs):          # the extra levels: 3x3 stride-2 convs
            h, w = hw[-1]
            hw.append(ops.conv_out_size(h, w, 3, 3, 2, 1))
        rows = [b * h * w for h, w in hw]
        buf = torch.empty((sum(rows), self.out_channels), dtype=laterals[0].dtype, device=laterals[0].device)
        views, r0 = [], 0
        for n in rows:
            views.append(buf[r0:r0 + n])
            r0 += n
        return views

    @staticmethod
    def _into(view, fn):
        if view is None:
            return fn()
        with ops.output_into(view):
            return fn()

    def lateral_tap(self, k, x):
        """the backbone's `tap` (ResNet._stages): output stage k's result x -> what the backbone continues with.  Runs
        the lateral conv of that level now, keeps its result for `_laterals`, returns the alias of x."""
        i = k - self.start_level
        if not (0 <= i < len(self.lateral_convs)) or not self._skip_ok(self.lateral_convs[i], x):
            return x
        pre = self.__dict__.setdefault('_pre_laterals', {})
        pre[i], alias = self.lateral_convs[i].forward_nhwc(x, with_skip=True)
        return alias

    def _laterals(self, inputs):
        assert len(inputs) == len(self.in_channels)
        pre = self.__dict__.pop('_pre_laterals', None) or {}
        laterals = [pre[i] if i in pre else conv.forward_nhwc(inputs[i + self.start_level])
                    for i, conv in enumerate(self.lateral_convs)]
        return laterals

    def _top_down(self, laterals, convs, views=None):
        """laterals[i-1] += upsample(laterals[i]) from the top, and outs[i] = convs[i](laterals[i]) as soon as level i
        is final (the reference runs all the adds first and the convs afterwards: same values)"""
        outs = [None] * len(laterals)
        for i in range(len(laterals) - 1, -1, -1):
            src = laterals[i]
            v = views[i] if views is not None else None
            if i > 0 and self._skip_ok(convs[i], laterals[i]):
                outs[i], src = self._into(v, lambda: convs[i].forward_nhwc(laterals[i], with_skip=True))
            else:
                outs[i] = self._into(v, lambda: convs[i].forward_nhwc(laterals[i]))
            if i == 0:
                break
            if src.requires_grad or laterals[i - 1].requires_grad:
                if src.is_cuda and src.shape[3] % 4 == 0 and src.dtype == laterals[i - 1].dtype:
                    laterals[i - 1] = ops.upsample_nearest_add_nhwc(laterals[i - 1], src)
                    continue
                import torch.nn.functional as F
                up = F.interpolate(src.permute(0, 3, 1, 2), size=laterals[i - 1].shape[1:3],
                                   mode='nearest').permute(0, 2, 3, 1)
                laterals[i - 1] = laterals[i - 1] + up
            else:
                ops.upsample_nearest_add_nhwc_(laterals[i - 1], src)
        return outs

    def _extra(self, inputs, laterals, outs, views=None):
        used = len(laterals)
        if self.num_outs > len(outs):
            if not self.add_extra_convs:
                for _ in range(self.num_outs - used):
                    outs.append(outs[-1][:, ::2, ::2, :].contiguous())   # max_pool2d(k=1, s=2)
            else:
                if self.add_extra_convs == 'on_input':
                    src = inputs[self.backbone_end_level - 1]
                elif self.add_extra_convs == 'on_lateral':
                    src = laterals[-1]
                else:
                    src = outs[-1]
                vw = views if views is not None else [None] * self.num_outs
                outs.append(self._into(vw[used], lambda: self.fpn_convs[used].forward_nhwc(src)))
                for i in range(used + 1, self.num_outs):
                    x = outs[-1].relu() if self.relu_before_extra_convs else outs[-1]
                    outs.append(self._into(vw[i], lambda: self.fpn_convs[i].forward_nhwc(x)))
        return outs

    def forward_nhwc(self, inputs):
        laterals = self._laterals(inputs)
        views = self._out_views(laterals)
        outs = self._top_down(laterals, self.fpn_convs, views)
        return tuple(self._extra(inputs, laterals, outs, views))

    def forward(self, inputs):
        return tuple(to_nchw_view(o) for o in self.forward_nhwc([to_nhwc(x) for x in inputs]))


@NECKS.register_module()
class PAFPN(FPN):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1,
                 add_extra_convs=False, relu_before_extra_convs=False, no_norm_on_lateral=False,
                 conv_cfg=None, norm_cfg=None, act_cfg=None,
                 init_cfg=dict(type='Xavier', layer='Conv2d', distribution='uniform')):
        super().__init__(in_channels, out_channels, num_outs, start_level, end_level,
                         add_extra_convs, relu_before_extra_convs, no_norm_on_lateral, conv_cfg,
                         norm_cfg, act_cfg, init_cfg=init_cfg)
        self.downsample_convs = nn.ModuleList()
        self.pafpn_convs = nn.ModuleList()
        for i in range(self.start_level + 1, self.backbone_end_level):
            self.downsample_convs.append(ConvModule(out_channels, out_channels, 3, stride=2,
                                                    padding=1, conv_cfg=conv_cfg,
                                                    norm_cfg=norm_cfg, act_cfg=act_cfg,
                                                    inplace=False))
            self.pafpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1,
                                               conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                               act_cfg=act_cfg, inplace=False))
        self.init_weights()

    def forward_nhwc(self, inputs):
        laterals = self._laterals(inputs)
        used = len(laterals)
        views = self._out_views(laterals)
        # (level 0 leaves the top-down path as it is: its conv writes into the shared buffer; the others go there from
        # their pafpn conv)
        inter = self._top_down(laterals, self.fpn_convs, [views[0]] + [None] * (used - 1))
        outs = [inter[0]] + [None] * (used - 1)
        for i in range(used):
            src = inter[i]
            if i >= 1:      # level i is final: its output conv runs now, handing out the alias the next downsample conv reads
                if i < used - 1 and self._skip_ok(self.pafpn_convs[i - 1], inter[i]):
                    outs[i], src = self._into(views[i], lambda: self.pafpn_convs[i - 1].forward_nhwc(inter[i], with_skip=True))
                else:
                    outs[i] = self._into(views[i], lambda: self.pafpn_convs[i - 1].forward_nhwc(inter[i]))
            if i == used - 1:
                break
            d = self.downsample_convs[i]
            if d.with_norm or d.with_activation:
                inter[i + 1] = inter[i + 1] + d.forward_nhwc(src)
            else:   # inter[i+1] += conv(inter[i]) as the conv epilogue's residual operand
                inter[i + 1] = d.forward_nhwc(src, residual=inter[i + 1])
        return tuple(self._extra(inputs, laterals, outs, views))
