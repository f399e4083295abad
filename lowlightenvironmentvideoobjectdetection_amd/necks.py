"""FPN with the reference's constructor, parameter names and init (mmdetection/mmdet/models/necks/fpn.py:11-221)."""
import warnings

import torch
import torch.nn as nn

from . import engine, ops
from .cnn import ConvModule, xavier_init
from .planned import PlannedModule
from .registry import NECKS


@NECKS.register_module()
class FPN(PlannedModule):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest')):
        super().__init__()
        assert isinstance(in_channels, list)
        if norm_cfg is not None or act_cfg is not None or upsample_cfg.get('mode', 'nearest') != 'nearest' or \
                'scale_factor' in upsample_cfg:
            raise NotImplementedError('FPN with norm/act or non-nearest upsampling is outside the hot path')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.relu_before_extra_convs, self.no_norm_on_lateral = relu_before_extra_convs, no_norm_on_lateral
        self.fp16_enabled = False
        self.upsample_cfg = upsample_cfg.copy()
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
            if extra_convs_on_inputs:
                warnings.simplefilter('once')
                warnings.warn('"extra_convs_on_inputs" will be deprecated in v2.9.0, Please use "add_extra_convs"',
                              DeprecationWarning)
                self.add_extra_convs = 'on_input'
            else:
                self.add_extra_convs = 'on_output'
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, act_cfg=None, inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, act_cfg=None, inplace=False))
        extra_levels = num_outs - self.backbone_end_level + self.start_level
        if self.add_extra_convs and extra_levels >= 1:
            for i in range(extra_levels):
                if i == 0 and self.add_extra_convs == 'on_input':
                    ic = self.in_channels[self.backbone_end_level - 1]
                else:
                    ic = out_channels
                self.fpn_convs.append(ConvModule(ic, out_channels, 3, stride=2, padding=1, act_cfg=None, inplace=False))

    def init_weights(self):             # fpn.py:163-167
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')
        self.invalidate_plans()

    def plan_kwargs(self):
        return dict(start_level=self.start_level, end_level=self.end_level, num_outs=self.num_outs,
                    add_extra_convs=self.add_extra_convs, relu_before_extra_convs=self.relu_before_extra_convs)

    def forward(self, inputs):
        """tuple of NCHW fp32 GPU tensors -> tuple of num_outs NCHW fp32 tensors."""
        assert len(inputs) == len(self.in_channels)
        for t in inputs:
            self._check_input(t)
        key = tuple(tuple(t.shape) for t in inputs) + (str(inputs[0].device),)

        def make():
            b = engine.Builder(inputs[0].device)
            ins = [b.buf(t.shape[0], t.shape[2], t.shape[3], ops.round_up(t.shape[1], 8)) for t in inputs]
            pyr = engine.build_fpn(b, engine._cpu_sd(self.state_dict()), ins, **self.plan_kwargs())
            return b.plan, ins, pyr
        plan, ins, pyr = self._plan(key, make)
        for t, buf in zip(inputs, ins):
            ops.pack_input(t.float().contiguous(), buf.shape[-1], out=buf)
        plan.run()
        return tuple(ops.unpack_output(pyr.level(i).contiguous(), self.out_channels, owner=plan) for i in range(len(pyr.level_hw)))
