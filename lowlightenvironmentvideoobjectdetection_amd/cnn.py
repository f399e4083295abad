"""The small part of mmcv.cnn the reference's hot path touches (ConvModule, init helpers), as parameter
containers: these modules own fp32 parameters under the reference's state_dict names and initialise them
with the reference's schemes; they do not compute -- forward runs in the HIP plans (engine.py).

mmcv is un-vendored (pinned >=1.2.4,<=1.4.0 by mmdetection/mmdet/__init__.py:18-26). Call sites:
mmdetection/mmdet/models/necks/fpn.py:121-160, dense_heads/retina_head.py:57-74, backbones/resnet.py:3-4,
mmtracking/mmtrack/models/aggregators/embed_aggregator.py:41-48."""
import numpy as np
import torch.nn as nn


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    assert distribution in ['uniform', 'normal']
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.normal_(module.weight, mean, std)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    assert distribution in ['uniform', 'normal']
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
        else:
            nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(prior_prob):
    return float(-np.log((1 - prior_prob) / prior_prob))


class ConvModule(nn.Module):
    """conv (bias iff no norm) -> norm -> activation (default ReLU), Kaiming init. Parameter container."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias='auto',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True, **kwargs):
        super().__init__()
        if conv_cfg is not None and conv_cfg.get('type', 'Conv2d') not in ('Conv2d', 'Conv'):
            raise NotImplementedError(f'conv_cfg {conv_cfg} is outside the hot path')
        if dilation != 1 or groups != 1:
            raise NotImplementedError('dilated / grouped ConvModule is outside the hot path')
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if self.with_activation and act_cfg.get('type') not in ('ReLU', 'LeakyReLU'):
            raise NotImplementedError(f'activation {act_cfg} is outside the hot path')
        self.act_type = act_cfg.get('type') if self.with_activation else None
        self.negative_slope = act_cfg.get('negative_slope', 0.01) if self.act_type == 'LeakyReLU' else 0.0
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        if self.with_norm:
            if norm_cfg.get('type') != 'BN':
                raise NotImplementedError(f'norm {norm_cfg} is outside the hot path')
            self.bn = nn.BatchNorm2d(out_channels)
            for prm in self.bn.parameters():
                prm.requires_grad = norm_cfg.get('requires_grad', True)
        self.init_weights()

    def init_weights(self):
        if getattr(self, 'act_type', None) == 'LeakyReLU':      # mmcv ConvModule.init_weights: a = negative_slope
            kaiming_init(self.conv, a=self.negative_slope, nonlinearity='leaky_relu')
        else:
            kaiming_init(self.conv, a=0, nonlinearity='relu')
        if self.with_norm:
            constant_init(self.bn, 1, bias=0)

    def forward(self, x):
        raise RuntimeError('ConvModule is a parameter container; the arithmetic runs in the HIP plans of the owning model')
