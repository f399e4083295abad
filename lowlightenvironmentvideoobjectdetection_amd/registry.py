"""Registry + build_from_cfg with mmcv.utils.Registry's behaviour, for the reference's registries
(mmdetection/mmdet/models/builder.py:6-77, mmtracking/mmtrack/models/builder.py:5-80,
mmdetection/mmdet/core/anchor/builder.py, mmdetection/mmdet/core/bbox/builder.py). mmcv itself is not
available; this is the part of its surface the reference's configs exercise."""
import inspect

from torch import nn


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = dict()

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f'{self.__class__.__name__}(name={self._name}, items={list(self._module_dict)})'

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def _register_module(self, module_class, module_name=None, force=False):
        if not inspect.isclass(module_class):
            raise TypeError(f'module must be a class, but got {type(module_class)}')
        if module_name is None:
            module_name = module_class.__name__
        names = [module_name] if isinstance(module_name, str) else module_name
        for name in names:
            if not force and name in self._module_dict:
                raise KeyError(f'{name} is already registered in {self.name}')
            self._module_dict[name] = module_class

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f'force must be a boolean, but got {type(force)}')
        if not (name is None or isinstance(name, str) or (isinstance(name, (list, tuple)) and all(isinstance(n, str) for n in name))):
            raise TypeError(f'name must be either of None, an instance of str or a sequence of str, but got {type(name)}')
        if module is not None:
            self._register_module(module_class=module, module_name=name, force=force)
            return module

        def _register(cls):
            self._register_module(module_class=cls, module_name=name, force=force)
            return cls

        return _register


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg:
        if default_args is None or 'type' not in default_args:
            raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    if not isinstance(registry, Registry):
        raise TypeError(f'registry must be an mmcv.Registry object, but got {type(registry)}')
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
    return obj_cls(**args)


# mmdet registries (mmdetection/mmdet/models/builder.py:6-12)
BACKBONES = Registry('backbone')
NECKS = Registry('neck')
ROI_EXTRACTORS = Registry('roi_extractor')
SHARED_HEADS = Registry('shared_head')
HEADS = Registry('head')
LOSSES = Registry('loss')
DETECTORS = Registry('detector')
# mmdet core registries
ANCHOR_GENERATORS = Registry('Anchor generator')
BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
BBOX_CODERS = Registry('bbox_coder')
IOU_CALCULATORS = Registry('IoU calculator')
# mmtrack registries (mmtracking/mmtrack/models/builder.py:5-10)
MODELS = Registry('model')
TRACKERS = Registry('tracker')
MOTION = Registry('motion')
REID = Registry('reid')
AGGREGATORS = Registry('aggregator')
CLEANER = Registry('cleaner')


def build(cfg, registry, default_args=None):
    """mmdetection/mmdet/models/builder.py:15-34: a list of cfgs becomes nn.Sequential."""
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdetection/mmdet/models/builder.py:67-77"""
    if train_cfg is not None or test_cfg is not None:
        import warnings
        warnings.warn('train_cfg and test_cfg is deprecated, please specify them in model', UserWarning)
    assert cfg.get('train_cfg') is None or train_cfg is None, 'train_cfg specified in both outer field and model field '
    assert cfg.get('test_cfg') is None or test_cfg is None, 'test_cfg specified in both outer field and model field '
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_anchor_generator(cfg, default_args=None):
    return build_from_cfg(cfg, ANCHOR_GENERATORS, default_args)


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_cleaner(cfg):
    return build(cfg, CLEANER)


def build_aggregator(cfg):
    return build(cfg, AGGREGATORS)


def build_motion(cfg):
    return build(cfg, MOTION)


def build_model(cfg, train_cfg=None, test_cfg=None):
    """mmtracking/mmtrack/models/builder.py:58-72"""
    if train_cfg is None and test_cfg is None:
        return build(cfg, MODELS)
    return build(cfg, MODELS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
