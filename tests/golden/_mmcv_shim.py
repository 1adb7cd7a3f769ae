"""Authoring-container-only helper: lets the read-only reference (/root/reference, an
mmdetection-2.17 fork) be imported on CPU although its un-vendored dependency mmcv-full
(and torchvision / cv2 / pycocotools / terminaltables) is not installed.

It is used by `tests/golden/make_golden.py` to GENERATE the committed golden fixtures; it
never travels with a test run (the `-m gpu` tests and the CPU tests read the fixtures, not
the reference).  Everything here is trivial glue written for this repo: a registry, an
attribute dict, conv/norm/activation wrappers around torch.nn, identity decorators.  The
`mmcv.ops` names the reference reaches (RoIAlign, nms, batched_nms, soft_nms,
sigmoid_focal_loss) are bound to this repo's CPU oracle (oracle/orc.py).
"""
import importlib.abc
import importlib.machinery
import sys
import types
from unittest import mock

import torch
import torch.nn as nn

REFERENCE_ROOT = '/root/reference'
_FAKE_ROOTS = ('mmcv', 'torchvision', 'cv2', 'pycocotools', 'terminaltables')


# ----------------------------------------------------------------------------- registry
class Registry:
    def __init__(self, name, build_func=None, parent=None, scope=None):
        self.name = name
        self.module_dict = {}
        self.parent = parent
        self.build_func = build_func or (parent.build_func if parent is not None else build_from_cfg)

    def __contains__(self, key):
        return self.get(key) is not None

    def __len__(self):
        return len(self.module_dict)

    def get(self, key):
        if key in self.module_dict:
            return self.module_dict[key]
        if self.parent is not None:
            return self.parent.get(key)
        return None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register(self, cls, name=None, force=False):
        names = [cls.__name__] if name is None else ([name] if isinstance(name, str) else name)
        for n in names:
            self.module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if isinstance(name, type):  # deprecated @REG.register_module without ()
            self._register(name)
            return name
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise KeyError(f'cfg must be a dict with "type", got {cfg}')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    t = args.pop('type')
    if isinstance(t, str):
        cls = registry.get(t)
        if cls is None:
            raise KeyError(f'{t} is not in the {registry.name} registry')
    else:
        cls = t
    return cls(**args)


# ----------------------------------------------------------------------------- config dict
class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(x):
        if isinstance(x, dict):
            return ConfigDict({k: ConfigDict.wrap(v) for k, v in x.items()})
        if isinstance(x, (list, tuple)):
            return type(x)(ConfigDict.wrap(v) for v in x)
        return x

    def copy(self):
        return ConfigDict(dict.copy(self))

    def __deepcopy__(self, memo):
        import copy
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


# ----------------------------------------------------------------------------- nn glue
class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = init_cfg

    @property
    def is_init(self):
        return self._is_init

    def init_weights(self):
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()
        self._is_init = True


class ModuleList(BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


class Sequential(BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


def _identity_decorator_factory(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def deco(fn):
        return fn
    return deco


def build_conv_layer(cfg, *args, **kwargs):
    if cfg is None or cfg.get('type', 'Conv2d') in ('Conv2d', 'Conv'):
        return nn.Conv2d(*args, **kwargs)
    raise NotImplementedError(cfg)


def build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    t = cfg.pop('type')
    requires_grad = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    if t in ('BN', 'BN2d', 'SyncBN'):
        name, layer = 'bn', nn.BatchNorm2d(num_features, **cfg)
    elif t == 'GN':
        name, layer = 'gn', nn.GroupNorm(num_channels=num_features, **cfg)
    else:
        raise NotImplementedError(t)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return name + str(postfix), layer


def build_activation_layer(cfg):
    cfg = dict(cfg)
    t = cfg.pop('type')
    return {'ReLU': nn.ReLU, 'Sigmoid': nn.Sigmoid}[t](**cfg)


def build_plugin_layer(cfg, postfix='', **kwargs):
    raise NotImplementedError('plugins are not on the hot path')


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'),
                 inplace=True, with_spectral_norm=False, padding_mode='zeros',
                 order=('conv', 'norm', 'act')):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        self.order = order
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size,
                                     stride=stride, padding=padding, dilation=dilation,
                                     groups=groups, bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            a = dict(act_cfg)
            a.setdefault('inplace', inplace)
            self.activate = build_activation_layer(a)
        nn.init.kaiming_normal_(self.conv.weight, a=0, nonlinearity='relu')
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.with_norm else None

    def forward(self, x, activate=True, norm=True):
        for layer in self.order:
            if layer == 'conv':
                x = self.conv(x)
            elif layer == 'norm' and norm and self.with_norm:
                x = self.norm(x)
            elif layer == 'act' and activate and self.with_activation:
                x = self.activate(x)
        return x


class Scale(nn.Module):
    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


def bias_init_with_prob(prior_prob):
    import numpy as np
    return float(-np.log((1 - prior_prob) / prior_prob))


def _init(fn):
    def wrapper(module, *a, bias=0, **k):
        if hasattr(module, 'weight') and module.weight is not None:
            fn(module.weight, *a, **k)
        if hasattr(module, 'bias') and module.bias is not None:
            nn.init.constant_(module.bias, bias)
    return wrapper


def normal_init(module, mean=0, std=1, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.normal_(module.weight, mean, std)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if hasattr(module, 'weight') and module.weight is not None:
        (nn.init.xavier_uniform_ if distribution == 'uniform' else nn.init.xavier_normal_)(
            module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    if hasattr(module, 'weight') and module.weight is not None:
        (nn.init.kaiming_uniform_ if distribution == 'uniform' else nn.init.kaiming_normal_)(
            module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def to_2tuple(x):
    return tuple(x) if isinstance(x, (list, tuple)) else (x, x)


def get_dist_info():
    return 0, 1


# ----------------------------------------------------------------------------- fake modules
class _FakeModule(types.ModuleType):
    """Any attribute exists: ALL-CAPS -> Registry, Capitalised -> empty nn.Module subclass
    (usable as a base class), anything else -> MagicMock."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        if name.isupper() and name != 'VGG':
            v = Registry(name.lower())
        elif name[0].isupper():
            v = type(name, (nn.Module,), {'__module__': self.__name__})
        else:
            v = mock.MagicMock(name=f'{self.__name__}.{name}')
        setattr(self, name, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in _FAKE_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _FakeModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        _populate(module)


def _populate(m):
    name = m.__name__
    if name == 'mmcv':
        m.__version__ = '1.4.0'
        m.ConfigDict = ConfigDict
        m.jit = _identity_decorator_factory
        m.is_tuple_of = lambda seq, t: isinstance(seq, tuple) and all(isinstance(s, t) for s in seq)
        m.is_list_of = lambda seq, t: isinstance(seq, list) and all(isinstance(s, t) for s in seq)
        m.is_seq_of = lambda seq, t, seq_type=None: all(isinstance(s, t) for s in seq)
        m.is_str = lambda x: isinstance(x, str)
        m.list_from_file = lambda f, **k: [ln.rstrip('\n\r') for ln in open(f)]
        # image arithmetic (mmcv -> OpenCV, both absent): this repo's restatement, so that the
        # reference's TRANSFORM LOGIC (scales, boxes, flips, metas) can be pinned around it
        from brcnn import pipelines as _P
        m.imrescale = lambda img, scale, return_scale=False, interpolation='bilinear', backend=None: \
            _P.imrescale(img, scale, return_scale)
        m.imresize = lambda img, size, return_scale=False, interpolation='bilinear', out=None, backend=None: \
            _P.imresize(img, size, return_scale)
        m.imflip = _P.imflip
        m.imnormalize = _P.imnormalize
        m.impad = lambda img, shape=None, padding=None, pad_val=0, padding_mode='constant': \
            _P.impad(img, shape, pad_val)
        m.impad_to_multiple = _P.impad_to_multiple
    elif name == 'terminaltables':
        class AsciiTable:
            def __init__(self, data, title=None):
                self.table_data, self.inner_footing_row_border = data, False

            @property
            def table(self):
                return '\n'.join(' | '.join(str(c) for c in row) for row in self.table_data)
        m.AsciiTable = AsciiTable
    elif name == 'mmcv.parallel':
        from brcnn import pipelines as _P
        m.DataContainer = _P.DataContainer
    elif name in ('pycocotools.coco', 'pycocotools'):
        from brcnn import datasets as _D

        class COCO(_D.COCO):      # camelCase surface of pycocotools.coco.COCO over this repo's index
            def getCatIds(self, catNms=(), supNms=(), catIds=()):
                return _D.COCO.get_cat_ids(self, catNms)

            def getImgIds(self, imgIds=(), catIds=()):
                return _D.COCO.get_img_ids(self)

            def getAnnIds(self, imgIds=(), catIds=(), areaRng=(), iscrowd=None):
                return _D.COCO.get_ann_ids(self, imgIds)

            loadAnns = _D.COCO.load_anns
            loadImgs = _D.COCO.load_imgs
            loadCats = _D.COCO.load_cats
        m.COCO = COCO
    elif name == 'mmcv.utils':
        m.Registry = Registry
        m.build_from_cfg = build_from_cfg
        m.ConfigDict = ConfigDict
        m.to_2tuple = to_2tuple
        m.print_log = lambda *a, **k: None
        m.deprecated_api_warning = _identity_decorator_factory
    elif name in ('mmcv.runner', 'mmcv.runner.base_module'):
        m.BaseModule = BaseModule
        m.ModuleList = ModuleList
        m.Sequential = Sequential
        m.force_fp32 = _identity_decorator_factory
        m.auto_fp16 = _identity_decorator_factory
        m.get_dist_info = get_dist_info
    elif name in ('mmcv.cnn', 'mmcv.cnn.bricks', 'mmcv.cnn.bricks.norm'):
        m.ConvModule = ConvModule
        m.Scale = Scale
        m.build_conv_layer = build_conv_layer
        m.build_norm_layer = build_norm_layer
        m.build_activation_layer = build_activation_layer
        m.build_plugin_layer = build_plugin_layer
        m.bias_init_with_prob = bias_init_with_prob
        m.normal_init = normal_init
        m.constant_init = constant_init
        m.xavier_init = xavier_init
        m.kaiming_init = kaiming_init
        if name == 'mmcv.cnn':
            m.MODELS = Registry('model')
    elif name == 'mmcv.utils.parrots_wrapper':
        m._BatchNorm = nn.modules.batchnorm._BatchNorm
        m._InstanceNorm = nn.modules.instancenorm._InstanceNorm
    elif name in ('mmcv.ops', 'mmcv.ops.nms', 'mmcv.ops.roi_align'):
        from oracle import orc
        m.RoIAlign = orc.RoIAlign
        m.roi_align = orc.roi_align
        m.nms = orc.nms
        m.batched_nms = orc.batched_nms
        m.soft_nms = orc.soft_nms
        m.sigmoid_focal_loss = orc.sigmoid_focal_loss


def install():
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import os
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
