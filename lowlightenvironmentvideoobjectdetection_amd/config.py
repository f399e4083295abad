"""Python-file configs with `_base_` inheritance, `_delete_`, attribute access and dotted-key overrides: the part of
mmcv.Config the reference's tools use (mmtracking/tools/train.py:67-90: Config.fromfile, cfg.merge_from_dict,
cfg.get; `_base_` lists such as mmtracking/configs/vid/fgfa/*.py:1-5). The reference's config files load unchanged."""
import copy
import os
import runpy

BASE_KEY = '_base_'
DELETE_KEY = '_delete_'
RESERVED_KEYS = ['filename', 'text', 'pretty_text']


class ConfigDict(dict):
    def __missing__(self, name):
        raise KeyError(name)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'{self.__class__.__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _to_configdict(v):
    if isinstance(v, dict):
        return ConfigDict({k: _to_configdict(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_to_configdict(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_to_configdict(x) for x in v)
    return v


def _merge_a_into_b(a, b, allow_list_keys=False):
    """mmcv Config._merge_a_into_b: dicts merge recursively; `_delete_=True` replaces instead of merging."""
    b = b.copy()
    for k, v in a.items():
        if allow_list_keys and k.isdigit() and isinstance(b, list):
            k = int(k)
            if len(b) <= k:
                raise KeyError(f'Index {k} exceeds the length of list {b}')
            b[k] = _merge_a_into_b(v, b[k], allow_list_keys)
        elif isinstance(v, dict) and k in b and not v.pop(DELETE_KEY, False):
            allowed = (dict, list) if allow_list_keys else dict
            if not isinstance(b[k], allowed):
                raise TypeError(f'{k}={v} in child config cannot inherit from base because {k} is a dict in the child '
                                f'config but is of type {type(b[k])} in base config. You may set `{DELETE_KEY}=True` to '
                                f'ignore the base config')
            b[k] = _merge_a_into_b(v, b[k], allow_list_keys)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            b[k] = v
    return b


def _file2dict(filename):
    filename = os.path.abspath(os.path.expanduser(filename))
    if not os.path.isfile(filename):
        raise FileNotFoundError(f'file "{filename}" does not exist')
    if not filename.endswith('.py'):
        raise IOError('Only py type are supported now!')
    ns = runpy.run_path(filename)
    cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v) and not isinstance(v, type(os))}
    if BASE_KEY in cfg:
        base = cfg.pop(BASE_KEY)
        base = base if isinstance(base, list) else [base]
        merged = dict()
        for f in base:
            c = _file2dict(os.path.join(os.path.dirname(filename), f))
            dup = merged.keys() & c.keys()
            if dup:
                raise KeyError(f'Duplicate key is not allowed among bases: {dup}')
            merged.update(c)
        cfg = _merge_a_into_b(cfg, merged)
    return cfg


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        cfg_dict = dict() if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError(f'cfg_dict must be a dict, but got {type(cfg_dict)}')
        for key in cfg_dict:
            if key in RESERVED_KEYS:
                raise KeyError(f'{key} is reserved for config file')
        super().__setattr__('_cfg_dict', _to_configdict(cfg_dict))
        super().__setattr__('_filename', filename)

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
        self._cfg_dict[name] = _to_configdict(value)

    def __setitem__(self, name, value):
        self._cfg_dict[name] = _to_configdict(value)

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))

    def merge_from_dict(self, options, allow_list_keys=True):
        """--cfg-options a.b.c=v (mmtracking/tools/train.py:77-78)."""
        option_cfg_dict = {}
        for full_key, v in options.items():
            d = option_cfg_dict
            keys = full_key.split('.')
            for subkey in keys[:-1]:
                d.setdefault(subkey, ConfigDict())
                d = d[subkey]
            d[keys[-1]] = v
        merged = _merge_a_into_b(option_cfg_dict, dict(self._cfg_dict), allow_list_keys=allow_list_keys)
        super().__setattr__('_cfg_dict', _to_configdict(merged))
