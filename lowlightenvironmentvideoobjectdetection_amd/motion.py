"""FlowNetSimple with the reference's constructor and parameter names
(mmtracking/mmtrack/models/motion/flownet_simple.py:8-251). Parameter container; forward = engine.FlowNetPlan."""
import torch
import torch.nn as nn

from . import engine, ops
from .planned import PlannedModule
from .registry import MOTION


class _Conv(nn.Module):
    """ConvModule stand-in holding `.conv` (Conv2d or ConvTranspose2d) so state_dict keys read `<name>.conv.weight`."""

    def __init__(self, conv):
        super().__init__()
        self.conv = conv


@MOTION.register_module()
class FlowNetSimple(PlannedModule):
    arch_setting = {'conv_layers': {'inplanes': (6, 64, 128, 256, 512, 512), 'kernel_size': (7, 5, 5, 3, 3, 3),
                                    'num_convs': (1, 1, 2, 2, 2, 2)},
                    'deconv_layers': {'inplanes': (386, 770, 1026, 1024)}}

    def __init__(self, img_scale_factor, out_indices=[2, 3, 4, 5, 6], flow_scale_factor=5.0,
                 flow_img_norm_std=[255.0, 255.0, 255.0], flow_img_norm_mean=[0.411, 0.432, 0.450]):
        super().__init__()
        if list(out_indices) != [2, 3, 4, 5, 6]:
            raise NotImplementedError('only the default out_indices are on the hot path')
        self.img_scale_factor, self.out_indices, self.flow_scale_factor = img_scale_factor, out_indices, flow_scale_factor
        self.flow_img_norm_mean, self.flow_img_norm_std = flow_img_norm_mean, flow_img_norm_std
        cs = self.arch_setting['conv_layers']
        self.conv_layers = []
        for i in range(6):
            inplanes = cs['inplanes'][i]
            planes = 2 * inplanes if i == 5 else cs['inplanes'][i + 1]
            k = cs['kernel_size'][i]
            layer = nn.ModuleList([_Conv(nn.Conv2d(inplanes, planes, k, 2, (k - 1) // 2, bias=True))])
            for _ in range(1, cs['num_convs'][i]):
                k2 = 3 if i == 2 else k
                layer.append(_Conv(nn.Conv2d(planes, planes, k2, 1, (k2 - 1) // 2, bias=True)))
            self.add_module(f'conv{i + 1}', layer)
            self.conv_layers.append(f'conv{i + 1}')
        ds = self.arch_setting['deconv_layers']['inplanes']
        planes = ds[-1] // 2
        for i in range(len(ds) - 1, -1, -1):
            self.add_module(f'deconv{i + 2}', _Conv(nn.ConvTranspose2d(ds[i], planes, 4, 2, 1, bias=False)))
            self.add_module(f'predict_flow{i + 3}', _Conv(nn.Conv2d(ds[i], 2, 3, 1, 1, bias=False)))
            self.add_module(f'upsample_flow{i + 2}', _Conv(nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=False)))
            planes = planes // 2
        self.predict_flow = _Conv(nn.Conv2d(planes * (2 + 4) + 2, 2, 3, 1, 1, bias=False))

    def init_weights(self):
        """flownet_simple.py:145-147 (no-op in the reference)."""
        self.invalidate_plans()

    def make_plan(self, batch, height, width, device, img_norm_cfg=None):
        nc = img_norm_cfg or dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375])
        return engine.FlowNetPlan(self.state_dict(), batch, height, width, device, self.img_scale_factor, self.flow_scale_factor,
                                  self.flow_img_norm_std, self.flow_img_norm_mean, nc['mean'], nc['std'])

    def forward(self, imgs, img_metas):
        """imgs (N,6,H,W) normalised image pairs on the GPU -> flow (N,2,H',W') (flownet_simple.py:192-244)."""
        self._check_input(imgs)
        n, c, h, w = imgs.shape
        assert c == 6
        key = (n, h, w, str(imgs.device), tuple(img_metas[0].get('img_norm_cfg', {}).get('mean', ())))
        plan = self._plan(key, lambda: self.make_plan(n, h, w, imgs.device, img_metas[0].get('img_norm_cfg')))
        plan.set_img_shape(img_metas[0]['img_shape'])
        plan.pairs.copy_(imgs.float())
        plan.plan.run()
        return plan.flow.clone()
