"""Python-dict config files with `_base_` inheritance -- the config language of the
reference (mmcv.Config, external to the reference tree; behaviour restated from its use in
tools/train.py:90-92 and configs/boosting_rcnn/*.py).

Rules reproduced:
  * a config file is a Python file; every non-dunder, non-module top-level name is a key;
  * `_base_ = 'x.py'` or a list of paths (relative to the file) are loaded first; keys of
    several bases must not collide; the child is merged INTO the bases;
  * dict values merge recursively; `_delete_=True` inside a child dict replaces the base
    dict instead of merging.  `_delete_` is only consumed when the key exists in the base
    (so `sampler=dict(_delete_=True, type='PseudoSampler')` in
    boosting_rcnn_r50_pafpn_1x_utdac.py:90, which has no base to delete from, keeps a
    `_delete_` entry that PseudoSampler(**kwargs) swallows);
  * `--cfg-options a.b=1` style overrides via `merge_from_dict`.
"""
import argparse
import ast
import copy
import os
import types

BASE_KEY = '_base_'
DELETE_KEY = '_delete_'


class ConfigDict(dict):
    """dict with attribute access (missing attribute -> AttributeError)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _hook(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._hook(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._hook(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'{self.__class__.__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name)

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]

    def copy(self):
        return ConfigDict(dict.copy(self))

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self):
        def plain(v):
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(plain(x) for x in v)
            return v
        return plain(self)


def _merge_a_into_b(a, b):
    b = dict(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and not v.pop(DELETE_KEY, False):
            if not isinstance(b[k], dict):
                raise TypeError(
                    f'{k}={v} in child config cannot inherit from base because {k} is a dict in '
                    f'the child config but is of type {type(b[k])} in base config. You may set '
                    f'`{DELETE_KEY}=True` to ignore the base config')
            b[k] = _merge_a_into_b(v, b[k])
        else:
            b[k] = v
    return b


def _file2dict(filename):
    filename = os.path.abspath(os.path.expanduser(filename))
    if not os.path.isfile(filename):
        raise FileNotFoundError(f'file "{filename}" does not exist')
    if not filename.endswith('.py'):
        raise IOError('Only py type are supported now!')
    with open(filename, 'r', encoding='utf-8') as f:
        text = f.read()
    try:
        ast.parse(text)
    except SyntaxError as e:
        raise SyntaxError(f'There are syntax errors in config file {filename}: {e}')
    ns = {'__file__': filename, '__name__': '_brcnn_cfg_'}
    exec(compile(text, filename, 'exec'), ns)
    cfg_dict = {k: v for k, v in ns.items()
                if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType))}
    if BASE_KEY in cfg_dict:
        base = cfg_dict.pop(BASE_KEY)
        base = base if isinstance(base, list) else [base]
        base_cfg = {}
        for b in base:
            d = _file2dict(os.path.join(os.path.dirname(filename), b))
            dup = base_cfg.keys() & d.keys()
            if dup:
                raise KeyError(f'Duplicate key is not allowed among bases. Duplicate keys: {dup}')
            base_cfg.update(d)
        cfg_dict = _merge_a_into_b(cfg_dict, base_cfg)
    return cfg_dict


class Config:
    """`Config.fromfile(path)`; attribute and item access; `merge_from_dict`."""

    def __init__(self, cfg_dict=None, filename=None):
        if cfg_dict is None:
            cfg_dict = {}
        if not isinstance(cfg_dict, dict):
            raise TypeError(f'cfg_dict must be a dict, but got {type(cfg_dict)}')
        object.__setattr__(self, '_cfg_dict', ConfigDict(cfg_dict))
        object.__setattr__(self, '_filename', filename)

    @staticmethod
    def fromfile(filename):
        return Config(_file2dict(filename), filename=filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def to_dict(self):
        return self._cfg_dict.to_dict()

    def merge_from_dict(self, options):
        """`{'model.backbone.depth': 101}` style overrides (tools/train.py:91-92)."""
        option_cfg = {}
        for full_key, v in options.items():
            d = option_cfg
            keys = full_key.split('.')
            for sub in keys[:-1]:
                d = d.setdefault(sub, {})
            d[keys[-1]] = v
        merged = _merge_a_into_b(option_cfg, self._cfg_dict.to_dict())
        object.__setattr__(self, '_cfg_dict', ConfigDict(merged))

    # ---- text form (mmcv Config.pretty_text / dump): a python file that round-trips ------
    @property
    def pretty_text(self):
        import pprint
        lines = []
        for k, v in self._cfg_dict.to_dict().items():
            lines.append(f'{k} = {pprint.pformat(v, width=100, sort_dicts=False)}')
        return '\n'.join(lines) + '\n'

    def dump(self, file=None):
        text = self.pretty_text
        if file is None:
            return text
        with open(file, 'w', encoding='utf-8') as f:
            f.write(text)

    def __repr__(self):
        return f'Config (path: {self.filename}): {self._cfg_dict.to_dict()!r}'


class DictAction(argparse.Action):
    """`--cfg-options a.b=1 c=x,y d="[1,2]"` -> {'a.b': 1, 'c': ['x','y'], 'd': [1,2]}
    (the argparse action the reference's tools take from mmcv)"""

    @staticmethod
    def _parse_scalar(val):
        for cast in (int, float):
            try:
                return cast(val)
            except ValueError:
                pass
        if val.lower() in ('true', 'false'):
            return val.lower() == 'true'
        if val == 'None':
            return None
        return val

    @staticmethod
    def _split_top(s):
        """split on commas that are not nested in () / []"""
        out, depth, cur = [], 0, ''
        for ch in s:
            if ch in '([':
                depth += 1
            elif ch in ')]':
                depth -= 1
            if ch == ',' and depth == 0:
                out.append(cur)
                cur = ''
            else:
                cur += ch
        out.append(cur)
        return out

    @classmethod
    def _parse_value(cls, val):
        val = val.strip().strip('\'"').replace(' ', '')
        is_tuple = False
        if val.startswith('(') and val.endswith(')'):
            is_tuple, val = True, val[1:-1]
        elif val.startswith('[') and val.endswith(']'):
            val = val[1:-1]
        elif ',' not in val:
            return cls._parse_scalar(val)
        items = [cls._parse_value(v) for v in cls._split_top(val) if v != '']
        return tuple(items) if is_tuple else items

    def __call__(self, parser, namespace, values, option_string=None):
        options = {}
        for kv in values:
            key, val = kv.split('=', maxsplit=1)
            options[key] = self._parse_value(val)
        setattr(namespace, self.dest, options)
