"""Parameter containers + NHWC execution of conv / norm / activation bricks.

The reference builds its layers from mmcv's `ConvModule` / `build_conv_layer` /
`build_norm_layer` / `Scale` (external).  Here a brick keeps its parameters in plain
`nn.Conv2d` / `nn.BatchNorm2d` / `nn.GroupNorm` modules -- so the state-dict key layout is the
reference's (`...conv.weight`, `...bn.weight`, `...gn.bias`, ...) and
published checkpoints load -- but those modules' own forward is never used: execution goes
through the HIP implicit-GEMM kernel on NHWC activations with BN (eval) / bias / residual /
ReLU folded into its epilogue.

Activations between bricks are (N,H,W,C) contiguous fp32 tensors ("nhwc").  `to_nchw_view`
gives the zero-copy logical (N,C,H,W) view (== torch.channels_last) used at the public
module boundaries.
"""
import os

import torch
import torch.nn as nn

from . import ops


# Arithmetic type of the conv stack: torch.float32 (exact-fp32 MFMA, the parity path and the default),
# torch.bfloat16 or torch.float16 (16-bit MFMA operands, fp32 accumulation; in training with fp32 master
# weights, fp32 heads / losses and, for fp16, static loss scaling).  Process-wide; set through
# `TwoStageDetector.set_compute_dtype`.
_COMPUTE_DTYPE = torch.float32


def set_compute_dtype(dtype):
    global _COMPUTE_DTYPE
    dtype = {'f32': torch.float32, 'fp32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16,
             'fp16': torch.float16}.get(dtype, dtype)
    assert dtype in (torch.float32, torch.bfloat16, torch.float16)
    _COMPUTE_DTYPE = dtype


def compute_dtype():
    return _COMPUTE_DTYPE


def to_nchw_view(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2)


def to_nhwc(x):
    """logical (N,C,H,W) -> (N,H,W,C) contiguous; free when x is channels_last already."""
    if x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
        return x.permute(0, 2, 3, 1)
    if x.shape[1] == 1 or (x.shape[2] == 1 and x.shape[3] == 1):
        return x.permute(0, 2, 3, 1).contiguous()
    return ops.nchw_to_nhwc(x)


class PackedCache:
    """Device-side packed operands of one conv (weights in (Cout,KH,KW,Cin) order, folded
    per-channel scale/shift), rebuilt when any source parameter changes version."""

    def __init__(self):
        self.key = None
        self.val = None

    def get(self, sources, builder):
        key = tuple((s.data_ptr(), s._version, s.device) for s in sources if s is not None) + \
            (_COMPUTE_DTYPE,)
        if key != self.key:
            with torch.no_grad():
                self.val = builder()
            self.key = key
        return self.val


def pack_weight(w):
    """(Cout,Cin,KH,KW) -> (Cout,KH,KW,Cin) contiguous fp32"""
    return w.detach().float().permute(0, 2, 3, 1).contiguous()


def fold_bn(bn):
    """eval-mode BatchNorm as y = x*scale + shift (resnet.py:648-657: norm_eval=True keeps BN
    in eval mode during training too)."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.float() * scale
    return scale.contiguous(), shift.contiguous()


def resolve_pretrained(checkpoint):
    """local file of an `init_cfg=dict(type='Pretrained', checkpoint=...)` entry.  The reference hands
    `torchvision://resnet50` / `open-mmlab://resnext101_64x4d` ... to mmcv's load_checkpoint, which downloads
    them; here they resolve against a local model directory: `$BRCNN_PRETRAINED_DIR/<name>.pth` (also
    `<scheme>/<name>.pth`, `.pt`, or the bare name), or `~/.cache/brcnn/pretrained`.  A plain path is used
    as it is.  Returns None when nothing is found."""
    import os
    if '://' not in checkpoint:
        return checkpoint if os.path.isfile(checkpoint) else None
    scheme, name = checkpoint.split('://', 1)
    roots = [os.environ.get('BRCNN_PRETRAINED_DIR'), os.path.expanduser('~/.cache/brcnn/pretrained')]
    for root in roots:
        if not root:
            continue
        for rel in (name, os.path.join(scheme, name), name.replace('/', '_')):
            for ext in ('.pth', '.pt', ''):
                cand = os.path.join(root, rel + ext)
                if os.path.isfile(cand):
                    return cand
    return None


def load_pretrained(module, init_cfg, prefixes=('backbone.', 'module.')):
    """mmcv PretrainedInit for a backbone: load `init_cfg.checkpoint` (strict=False) into `module`.
    Returns True when weights were loaded, False when `init_cfg` is not a Pretrained entry.  A Pretrained
    entry whose file cannot be found raises, unless BRCNN_ALLOW_RANDOM_INIT=1 (tests / benches with
    synthetic weights): training a recipe from random weights with a frozen random stem silently misses the
    reference's accuracy, so that must be an explicit choice."""
    import logging
    import os
    if not (isinstance(init_cfg, dict) and init_cfg.get('type') == 'Pretrained'):
        return False
    ckpt = init_cfg.get('checkpoint')
    path = resolve_pretrained(ckpt) if ckpt else None
    log = logging.getLogger('brcnn')
    if path is None:
        msg = (f'pretrained weights {ckpt!r} not found: put the file under $BRCNN_PRETRAINED_DIR (e.g. '
               f'<dir>/{str(ckpt).split("://")[-1]}.pth) or give a path; set BRCNN_ALLOW_RANDOM_INIT=1 (tools/train.py '
               f'--allow-random-init) to train from random weights instead')
        if os.environ.get('BRCNN_ALLOW_RANDOM_INIT') == '1':
            log.warning('RANDOM INIT: ' + msg)
            return False
        raise FileNotFoundError(msg)
    sd = torch.load(path, map_location='cpu', weights_only=False)
    sd = sd.get('state_dict', sd) if isinstance(sd, dict) else sd
    out = {}
    for k, v in sd.items():
        for pre in prefixes:
            if k.startswith(pre):
                k = k[len(pre):]
        out[k] = v
    missing, unexpected = module.load_state_dict(out, strict=False)
    missing = [k for k in missing if not k.endswith('num_batches_tracked')]
    log.info(f'loaded pretrained {ckpt} from {path}: {len(out) - len(unexpected)} tensors, missing {missing[:6]}'
             f'{"..." if len(missing) > 6 else ""}, unexpected {list(unexpected)[:6]}{"..." if len(unexpected) > 6 else ""}')
    return True


def build_norm_layer(cfg, num_features, postfix=''):
    """(name, module) like mmcv.cnn.build_norm_layer: BN -> 'bn', GN -> 'gn'."""
    cfg = dict(cfg)
    layer_type = cfg.pop('type')
    requires_grad = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    if layer_type in ('BN', 'BN2d', 'SyncBN'):
        name, layer = 'bn', nn.BatchNorm2d(num_features, **cfg)
    elif layer_type == 'GN':
        assert 'num_groups' in cfg
        name, layer = 'gn', nn.GroupNorm(num_channels=num_features, **cfg)
    else:
        raise KeyError(f'Unrecognized norm type {layer_type}')
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return name + str(postfix), layer


def conv_weights_channels_last(module):
    """store the spatial (KH*KW > 1) weights of the trainable dense nn.Conv2d layers of `module` with
    torch.channels_last strides -- (Cout, KH, KW, Cin) in memory, the layout the weight-gradient kernel writes --
    so that its result becomes `weight.grad` as it is (autograd's layout contract) instead of through one copy
    per layer and step.  Values, shapes and state_dict keys are unchanged; call before wrapping the model in
    DistributedDataParallel (its bucket views take the parameters' strides).  Returns the number converted."""
    n = 0
    if os.environ.get('BRCNN_CL_WEIGHTS', '1') == '0':
        return n
    for m in module.modules():
        if isinstance(m, nn.Conv2d) and m.groups == 1 and m.weight.requires_grad and m.weight.shape[2] * m.weight.shape[3] > 1 \
                and not m.weight.is_contiguous(memory_format=torch.channels_last):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
            n += 1
    return n


# training forward of conv -> eval-BN -> act as one launch where `autograd.conv_bn_eval_act_fusable` allows
# (False: conv kernel + bn_act kernel, the fp32 structure)
FUSE_CONV_BN_TRAIN = True


def conv_bn_act_tail(y, bn, relu, residual):
    """eval-BN affine (+residual, +ReLU) behind a differentiable conv (fused kernel when it applies)"""
    if bn is not None:
        from .autograd import bn_act_supported, bn_eval_act_autograd
        if y.is_cuda and bn_act_supported(y) and (residual is None or residual.dtype == y.dtype):
            return bn_eval_act_autograd(y, bn, residual, relu)
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
        y = y * scale.to(y.dtype) + shift.to(y.dtype)
    if residual is not None:
        y = y + residual
    return y.relu() if relu else y


def conv_bn_act_nhwc(x, conv, bn, cache, relu, residual=None, with_skip=False, sole_consumer=False,
                     single_use_output=False):
    """act(bn(conv(x)) + residual).  Inference / frozen layers: everything folded into one
    kernel launch.  Trainable layers under grad mode: the differentiable conv kernel followed
    by the (cheap, element-wise) eval-BN affine / add / ReLU as torch ops."""
    if bn is not None and bn.training:
        raise NotImplementedError('training-mode BatchNorm is not on the HIP path '
                                  '(the reference runs BN in eval mode: norm_eval=True)')
    from .autograd import conv2d_nhwc_autograd, wants_grad
    if conv.groups > 1:
        out = _grouped_conv_bn_act_nhwc(x, conv, bn, cache, relu, residual)
        return (out, x) if with_skip else out
    if wants_grad(x, conv.weight, conv.bias, bn.weight if bn is not None else None,
                  residual if residual is not None and residual.requires_grad else None):
        skip = None
        if bn is not None and FUSE_CONV_BN_TRAIN:
            from .autograd import conv_bn_eval_act_autograd, conv_bn_eval_act_fusable
            if conv_bn_eval_act_fusable(x, conv, bn, residual):
                # conv + eval-BN (+ residual) (+ ReLU) in one forward launch (16-bit compute dtypes)
                # `single_use_output` / `sole_consumer` (see conv_bn_eval_act_autograd): a Bottleneck's conv2 / conv3
                # run the BatchNorm backward of the layer above inside their data-gradient launch
                if with_skip and x.requires_grad:
                    return conv_bn_eval_act_autograd(x, conv, bn, residual, relu, True, sole_consumer, single_use_output)
                out = conv_bn_eval_act_autograd(x, conv, bn, residual, relu, False, sole_consumer, single_use_output)
                return (out, x) if with_skip else out
        if with_skip and x.requires_grad:
            y, skip = conv2d_nhwc_autograd(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], True)
        else:
            y = conv2d_nhwc_autograd(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0])
            skip = x if with_skip else None
        if with_skip:
            return conv_bn_act_tail(y, bn, relu, residual), skip
        if bn is not None:
            from .autograd import bn_act_supported, bn_eval_act_autograd
            if y.is_cuda and bn_act_supported(y) and (residual is None or residual.dtype == y.dtype):
                return bn_eval_act_autograd(y, bn, residual, relu)     # one kernel each way, BN fold included
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            shift = bn.bias - bn.running_mean * scale
            # bf16 activations stay bf16 (the affine parameters are rounded once per step)
            y = y * scale.to(y.dtype) + shift.to(y.dtype)
        if residual is not None:
            y = y + residual
        return y.relu() if relu else y
    w, scale, shift = folded_conv_operands(conv, bn, cache, x.dtype)
    out = ops.conv2d_nhwc(x, w, scale, shift, residual, relu, conv.stride[0], conv.padding[0])
    return (out, x) if with_skip else out


def folded_conv_operands(conv, bn, cache, dtype):
    """(packed weight (Cout,KH,KW,Cin) in `dtype`, scale, shift) of a frozen / inference conv + eval-BN pair, from the
    layer's PackedCache (rebuilt when a source parameter changes version)"""
    srcs = [conv.weight, conv.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var]
                                       if bn is not None else [])

    def builder():
        w = pack_weight(conv.weight).to(dtype)
        if bn is not None:
            scale, shift = fold_bn(bn)
            if conv.bias is not None:
                shift = shift + conv.bias.detach().float() * scale
            return w, scale, shift
        return w, None, (conv.bias.detach().float().contiguous() if conv.bias is not None else None)

    return cache.get(srcs, builder)


def _grouped_conv_bn_act_nhwc(x, conv, bn, cache, relu, residual):
    """grouped conv (ResNeXt conv2) + folded eval-BN + ReLU in one launch; inference / frozen only"""
    from .autograd import bn_act_autograd, bn_act_supported, grouped_conv_autograd, wants_grad
    cg_in, cg_out = conv.in_channels // conv.groups, conv.out_channels // conv.groups
    if x.dtype != torch.float32 and ((64 // cg_out) * cg_in) % 64:
        # bf16 tiles need a 64-channel input window per 64-channel output tile; other group shapes
        # widen to the fp32 tiles
        y = _grouped_conv_bn_act_nhwc(x.float(), conv, bn, cache, relu,
                                      residual.float() if residual is not None else None)
        return y.to(x.dtype)
    if wants_grad(x, conv.weight, conv.bias, bn.weight if bn is not None else None):
        assert conv.bias is None
        y = grouped_conv_autograd(x, conv.weight, conv.groups, conv.stride[0], conv.padding[0])
        if bn is not None:
            if y.is_cuda and bn_act_supported(y) and (residual is None or residual.dtype == y.dtype):
                from .autograd import bn_eval_act_autograd
                return bn_eval_act_autograd(y, bn, residual, relu)
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            shift = bn.bias - bn.running_mean * scale
            y = y * scale + shift
        if residual is not None:
            y = y + residual
        return y.relu() if relu else y
    srcs = [conv.weight, conv.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var]
                                       if bn is not None else [])

    def builder():
        w, window = ops.pack_grouped_weight(conv.weight, conv.groups)
        w = w.to(x.dtype)
        scale = shift = None
        if bn is not None:
            scale, shift = fold_bn(bn)
            if conv.bias is not None:
                shift = shift + conv.bias.detach().float() * scale
        elif conv.bias is not None:
            shift = conv.bias.detach().float().contiguous()
        return w, window, scale, shift
    w, window, scale, shift = cache.get(srcs, builder)
    return ops.conv2d_nhwc_grouped(x, w, window, scale, shift, residual, relu, conv.stride[0], conv.padding[0])


class ConvModule(nn.Module):
    """conv -> norm -> activation brick with mmcv.cnn.ConvModule's constructor and attribute
    names (`conv`, `bn`/`gn`, `activate`); bias='auto' means "no bias when a norm follows"."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'),
                 inplace=True, with_spectral_norm=False, padding_mode='zeros',
                 order=('conv', 'norm', 'act')):
        super().__init__()
        assert conv_cfg is None or conv_cfg.get('type', 'Conv2d') in ('Conv2d', 'Conv'), \
            'only plain Conv2d is on the hot path'
        assert dilation == 1 and groups == 1 and padding_mode == 'zeros' and not with_spectral_norm
        assert tuple(order) == ('conv', 'norm', 'act')
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if self.with_activation:
            assert act_cfg.get('type') == 'ReLU', 'only ReLU activations are on the hot path'
        if bias == 'auto':
            bias = not self.with_norm
        self.with_bias = bias
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm_name = None
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            self.activate = nn.ReLU(inplace=inplace)
        self._cache = PackedCache()
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.norm_name else None

    def init_weights(self):
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)
        if self.with_norm:
            nn.init.constant_(self.norm.weight, 1)
            nn.init.constant_(self.norm.bias, 0)

    def forward_nhwc(self, x, residual=None, with_skip=False):
        """`with_skip`: returns (out, alias of x) -- the alias's gradient is added in this conv's data-gradient launch
        (conv_bn_act_nhwc); plain / BatchNorm modules only"""
        norm = self.norm
        if norm is None or isinstance(norm, nn.BatchNorm2d):
            return conv_bn_act_nhwc(x, self.conv, norm, self._cache, self.with_activation, residual, with_skip)
        # GroupNorm needs the statistics of the whole conv output: conv, then fused GN(+ReLU)
        assert residual is None and not with_skip
        y = conv_bn_act_nhwc(x, self.conv, None, self._cache, False)
        if y.requires_grad:
            from . import autograd as ag
            if y.shape[3] <= 256 and y.shape[3] % 4 == 0:
                return ag.groupnorm_nhwc_autograd(y, norm.weight, norm.bias, norm.num_groups, norm.eps,
                                                  self.with_activation)
            import torch.nn.functional as F
            z = F.group_norm(y.permute(0, 3, 1, 2).float(), norm.num_groups, norm.weight, norm.bias, norm.eps)
            z = z.permute(0, 2, 3, 1).to(y.dtype)
            return (z.relu() if self.with_activation else z).contiguous()
        return ops.groupnorm_nhwc(y, norm.weight.detach(), norm.bias.detach(), norm.num_groups,
                                  norm.eps, self.with_activation)

    def forward(self, x):
        return to_nchw_view(self.forward_nhwc(to_nhwc(x)))


class Scale(nn.Module):
    """learnable scalar (mmcv.cnn.Scale); state-dict key `scale`."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


def bias_init_with_prob(prior_prob):
    import math
    return float(-math.log((1 - prior_prob) / prior_prob))
