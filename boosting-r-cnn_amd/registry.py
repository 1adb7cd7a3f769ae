"""Name -> class registries with the reference's surface.

The reference resolves `dict(type='FasterRCNN', ...)` through one `MODELS` registry aliased
as BACKBONES/NECKS/ROI_EXTRACTORS/SHARED_HEADS/HEADS/LOSSES/DETECTORS
(mmdet/models/builder.py:7-59) and assigners/samplers/coders/prior generators/IoU
calculators through their own (mmdet/core/bbox/builder.py:4-21,
mmdet/core/anchor/builder.py, mmdet/core/bbox/iou_calculators/builder.py).  The registry
class itself is mmcv's (external); this is this repo's own implementation of that contract:
`register_module(name=None, force=False, module=None)`, `get`, `build`, `in`.
"""
import inspect


def build_from_cfg(cfg, registry, default_args=None):
    """Instantiate `cfg['type']` from `registry` with the remaining keys as kwargs."""
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg:
        if default_args is None or 'type' not in default_args:
            raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}')
    if not isinstance(registry, Registry):
        raise TypeError(f'registry must be a Registry, but got {type(registry)}')
    if not (isinstance(default_args, dict) or default_args is None):
        raise TypeError(f'default_args must be a dict or None, but got {type(default_args)}')
    args = dict(cfg)
    if default_args is not None:
        for name, value in default_args.items():
            args.setdefault(name, value)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    try:
        return obj_cls(**args)
    except Exception as e:
        raise type(e)(f'{obj_cls.__name__}: {e}')


class Registry:
    def __init__(self, name, build_func=None, parent=None):
        self._name = name
        self._module_dict = {}
        self.parent = parent
        if build_func is None:
            build_func = parent.build_func if parent is not None else build_from_cfg
        self.build_func = build_func

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f'{self.__class__.__name__}(name={self._name}, items={sorted(self._module_dict)})'

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        if self.parent is not None:
            return self.parent.get(key)
        return None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register_module(self, module_class, module_name=None, force=False):
        if not inspect.isclass(module_class):
            raise TypeError(f'module must be a class, but got {type(module_class)}')
        if module_name is None:
            module_name = module_class.__name__
        if isinstance(module_name, str):
            module_name = [module_name]
        for name in module_name:
            if not force and name in self._module_dict:
                raise KeyError(f'{name} is already registered in {self.name}')
            self._module_dict[name] = module_class

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f'force must be a boolean, but got {type(force)}')
        if inspect.isclass(name):  # bare @REG.register_module
            self._register_module(name)
            return name
        if not (name is None or isinstance(name, str) or
                (isinstance(name, (list, tuple)) and all(isinstance(n, str) for n in name))):
            raise TypeError(f'name must be None, a str or a sequence of str, got {type(name)}')
        if module is not None:
            self._register_module(module_class=module, module_name=name, force=force)
            return module

        def _register(cls):
            self._register_module(module_class=cls, module_name=name, force=force)
            return cls
        return _register


# ---- the reference's registries (same names) --------------------------------------------
MODELS = Registry('models')
BACKBONES = NECKS = ROI_EXTRACTORS = SHARED_HEADS = HEADS = LOSSES = DETECTORS = MODELS
BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
BBOX_CODERS = Registry('bbox_coder')
PRIOR_GENERATORS = Registry('Generator for anchors and points')
ANCHOR_GENERATORS = PRIOR_GENERATORS
IOU_CALCULATORS = Registry('IoU calculator')


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_roi_extractor(cfg):
    return ROI_EXTRACTORS.build(cfg)


def build_shared_head(cfg):
    return SHARED_HEADS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet/models/builder.py:48-59."""
    assert cfg.get('train_cfg') is None or train_cfg is None, \
        'train_cfg specified in both outer field and model field '
    assert cfg.get('test_cfg') is None or test_cfg is None, \
        'test_cfg specified in both outer field and model field '
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


def build_prior_generator(cfg, default_args=None):
    return build_from_cfg(cfg, PRIOR_GENERATORS, default_args)


build_anchor_generator = build_prior_generator


def build_iou_calculator(cfg, default_args=None):
    return build_from_cfg(cfg, IOU_CALCULATORS, default_args)
