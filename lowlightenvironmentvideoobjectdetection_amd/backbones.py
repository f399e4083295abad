"""ResNet backbone with the reference's constructor, parameter names and init
(mmdetection/mmdet/models/backbones/resnet.py:12-666, utils/res_layer.py:5-103).
The modules are parameter containers; forward builds/replays a HIP plan (engine.build_resnet)."""
import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from . import engine, ops
from .cnn import constant_init, kaiming_init
from .planned import PlannedModule
from .registry import BACKBONES


def _norm(norm_cfg, ch):
    if norm_cfg.get('type', 'BN') != 'BN':
        raise NotImplementedError(f'norm {norm_cfg} is outside the hot path (BN with running stats is folded into the conv)')
    bn = nn.BatchNorm2d(ch)
    for p in bn.parameters():
        p.requires_grad = norm_cfg.get('requires_grad', True)
    return bn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', with_cp=False,
                 conv_cfg=None, norm_cfg=dict(type='BN'), dcn=None, plugins=None):
        super().__init__()
        assert dcn is None, 'Not implemented yet.'
        assert plugins is None, 'Not implemented yet.'
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)   # resnet.py:38-46
        self.bn1 = _norm(norm_cfg, planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = _norm(norm_cfg, planes)
        self.downsample = downsample
        self.stride, self.dilation = stride, dilation

    norm1 = property(lambda self: self.bn1)
    norm2 = property(lambda self: self.bn2)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', with_cp=False,
                 conv_cfg=None, norm_cfg=dict(type='BN'), dcn=None, plugins=None):
        super().__init__()
        assert style in ['pytorch', 'caffe']
        if dcn is not None or plugins is not None:
            raise NotImplementedError('dcn / plugins are outside the hot path')
        self.style = style
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=s1, bias=False)
        self.bn1 = _norm(norm_cfg, planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=s2, padding=dilation, dilation=dilation, bias=False)   # resnet.py:163-171
        self.bn2 = _norm(norm_cfg, planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = _norm(norm_cfg, planes * self.expansion)
        self.downsample = downsample
        self.stride = stride

    norm1 = property(lambda self: self.bn1)
    norm2 = property(lambda self: self.bn2)
    norm3 = property(lambda self: self.bn3)


class DenoisingBlock(nn.Module):
    """mmtracking/mmtrack/models/backbones/dark_resnet.py:15-285: a Bottleneck whose 3x3 conv is the gate convolution of a ConvLSTM
    that runs along the batch (the frames of a clip): conv2 maps [x_t | h_{t-1}] (2 * planes) to the four gates (4 * planes, weights
    shared by all frames, :119-130), norm2 / conv3 see the hidden state (planes channels). Parameter container; the launches are built
    by engine._denoising_block."""
    expansion = 4

    def __init__(self, inplanes, planes, seqs=4, stride=1, dilation=1, downsample=None, style='pytorch', with_cp=False, conv_cfg=None,
                 norm_cfg=dict(type='BN'), dcn=None, plugins=None):
        super().__init__()
        assert style in ['pytorch', 'caffe']
        if dcn is not None or plugins is not None:
            raise NotImplementedError('dcn / plugins are outside the hot path')
        self.inplanes, self.planes, self.seqs, self.stride, self.dilation, self.style = inplanes, planes, seqs, stride, dilation, style
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=s1, bias=False)
        self.bn1 = _norm(norm_cfg, planes)
        self.conv2 = nn.Conv2d(planes * 2, planes * 4, 3, stride=s2, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = _norm(norm_cfg, planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = _norm(norm_cfg, planes * self.expansion)
        self.downsample = downsample

    norm1 = property(lambda self: self.bn1)
    norm2 = property(lambda self: self.bn2)
    norm3 = property(lambda self: self.bn3)


class ResLayer(nn.Sequential):
    """res_layer.py:5-103 (downsample_first=True, no avg_down)."""

    def __init__(self, block, inplanes, planes, num_blocks, stride=1, avg_down=False, conv_cfg=None,
                 norm_cfg=dict(type='BN'), downsample_first=True, insert_plugins=False, **kwargs):
        if avg_down or not downsample_first:
            raise NotImplementedError('avg_down / Hourglass ordering is outside the hot path')
        self.block = block
        downsample = None
        if stride != 1 or inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                       _norm(norm_cfg, planes * block.expansion))
        layers = [block(inplanes=inplanes, planes=planes, stride=stride, downsample=downsample, conv_cfg=conv_cfg,
                        norm_cfg=norm_cfg, **kwargs)]
        inplanes = planes * block.expansion
        for _ in range(1, num_blocks):
            layers.append(block(inplanes=inplanes, planes=planes, stride=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, **kwargs))
        super().__init__(*layers)


@BACKBONES.register_module()
class ResNet(PlannedModule):
    """Same arguments as mmdet's ResNet (resnet.py:302-447).

    Example (mirrors resnet.py:340-352, on a GPU):
        >>> self = ResNet(depth=18).cuda().eval()
        >>> outs = self.forward(torch.rand(1, 3, 32, 32).cuda())
        >>> [tuple(o.shape) for o in outs]
        [(1, 64, 8, 8), (1, 128, 4, 4), (1, 256, 2, 2), (1, 512, 1, 1)]
    """
    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)), 50: (Bottleneck, (3, 4, 6, 3)),
                     101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch', deep_stem=False, avg_down=False,
                 frozen_stages=-1, conv_cfg=None, norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False, zero_init_residual=True):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        if deep_stem or avg_down or dcn is not None or plugins is not None or in_channels != 3:
            raise NotImplementedError('deep_stem / avg_down / dcn / plugins are outside the hot path')
        self.depth = depth
        stem_channels = stem_channels or base_channels
        self.stem_channels, self.base_channels, self.num_stages = stem_channels, base_channels, num_stages
        assert 1 <= num_stages <= 4
        self.strides, self.dilations = strides, dilations
        assert len(strides) == len(dilations) == num_stages
        self.out_indices = out_indices
        assert max(out_indices) < num_stages
        self.style, self.frozen_stages, self.norm_cfg, self.norm_eval = style, frozen_stages, norm_cfg, norm_eval
        self.zero_init_residual = zero_init_residual
        self.block, stage_blocks = self.arch_settings[depth]
        self.stage_blocks = stage_blocks[:num_stages]
        self.inplanes = stem_channels
        self.conv1 = nn.Conv2d(in_channels, stem_channels, 7, stride=2, padding=3, bias=False)
        self.bn1 = _norm(norm_cfg, stem_channels)
        self.res_layers = []
        for i, num_blocks in enumerate(self.stage_blocks):
            planes = base_channels * 2 ** i
            layer = ResLayer(block=self.block, inplanes=self.inplanes, planes=planes, num_blocks=num_blocks,
                             stride=strides[i], dilation=dilations[i], style=style, norm_cfg=norm_cfg)
            self.inplanes = planes * self.block.expansion
            name = f'layer{i + 1}'
            self.add_module(name, layer)
            self.res_layers.append(name)
        self._freeze_stages()
        self.feat_dim = self.block.expansion * base_channels * 2 ** (len(self.stage_blocks) - 1)

    norm1 = property(lambda self: self.bn1)

    def _freeze_stages(self):           # resnet.py:572-588
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in [self.conv1, self.bn1]:
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):   # resnet.py:590-620
        if isinstance(pretrained, str):
            from .checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
        elif pretrained is None:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    kaiming_init(m)
                elif isinstance(m, (_BatchNorm, nn.GroupNorm)):
                    constant_init(m, 1)
            if self.zero_init_residual:
                for m in self.modules():
                    if isinstance(m, Bottleneck):
                        constant_init(m.bn3, 0)
                    elif isinstance(m, BasicBlock):
                        constant_init(m.bn2, 0)
        else:
            raise TypeError('pretrained must be a str or None')
        self.invalidate_plans()

    def train(self, mode=True):         # resnet.py:642-651
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self

    dark_stages = ()

    def forward_train(self, x):
        """The stages under autograd (dev_autograd.resnet_forward: every convolution a differentiable device op; the denoising plugins of
        InsertResNet and the ConvLSTM blocks of DarkResNet included): (N, 3, H, W) fp32 -> tuple of NCHW fp32 stage outputs carrying the
        graph to every trainable parameter. The dense detectors do not come through here (their whole step is one launch tape,
        train_engine.py); this is the training path of the backbones whose blocks fuse the frames of the batch."""
        from . import dev_autograd as D
        return D.resnet_forward(self, x)

    def plan_kwargs(self):
        kw = dict(depth=self.depth, strides=self.strides, style=self.style, out_indices=self.out_indices,
                  num_stages=self.num_stages, dilations=tuple(self.dilations))
        if self.dark_stages:
            kw['dark_stages'] = tuple(self.dark_stages)
        return kw

    def forward(self, x):
        """NCHW fp32 image on the GPU -> tuple of NCHW fp32 stage outputs (BN uses running statistics)."""
        self._check_input(x)
        n, c, h, w = x.shape

        def make():
            b = engine.Builder(x.device)
            x_in = b.buf(n, h, w, 8)
            # inference plan: the stride-1 Bottlenecks of layer1 / layer2 as one launch each (bit-identical to the three launches)
            feats = engine.build_resnet(b, engine._cpu_sd(self.state_dict()), x_in, fuse_blocks=True, **self.plan_kwargs())
            return b.plan, x_in, feats
        plan, x_in, feats = self._plan((n, h, w, str(x.device)), make)
        ops.pack_input(x.float().contiguous(), 8, out=x_in)
        plan.run()
        return tuple(ops.unpack_output(f, f.shape[-1], owner=plan) for f in feats)


@BACKBONES.register_module()
class InsertResNet(ResNet):
    """ResNet with the authors' denoising plugin appended to chosen stages
    (mmtracking/mmtrack/models/backbones/insert_resnet.py:93-97,134-230): stage i with insert_tag[i] ends in
    build_aggregator(insert_cfg) with in_channels = the stage's output channels, mid_channels = its planes and
    channel_growth = 64; the plugin is module `layer{i+1}.{num_blocks}` of the stage (state_dict keys
    `layer3.6.conv1.weight`, `layer3.6.taf.dcn_pack.weight`, ...). The batch dimension is the clip: the plugin's temporal
    attention fuses the frames of the batch. One launch plan: stages and plugins interleaved. Dilated stages are not built."""

    def __init__(self, depth, insert_tag=(False, False, False, False), insert_cfg=None, **kwargs):
        if depth not in (50, 101):
            raise KeyError(f'invalid depth {depth} for InsertResNet')          # insert_resnet.py:139-142
        super().__init__(depth, **kwargs)
        from .registry import build_aggregator
        self.insert_tag = tuple(insert_tag)
        assert len(self.insert_tag) == self.num_stages
        self.insert_cfg = dict(insert_cfg) if insert_cfg is not None else None
        self._plugins = {}
        for i, name in enumerate(self.res_layers):
            if not self.insert_tag[i]:
                continue
            planes = self.base_channels * 2 ** i
            cfg = dict(self.insert_cfg, in_channels=planes * self.block.expansion, mid_channels=planes, channel_growth=64)
            layer = getattr(self, name)
            layer.add_module(str(len(layer)), build_aggregator(cfg))
            self._plugins[i] = layer[len(layer) - 1]

    def forward(self, x):
        """x (N, 3, H, W): the N frames of a clip -> tuple of stage outputs, NCHW fp32."""
        self._check_input(x)
        n, c, h, w = x.shape

        def make():
            b = engine.Builder(x.device)
            x_in = b.buf(n, h, w, 8)
            sd = engine._cpu_sd(self.state_dict())

            def after_stage(i, nb, t):
                if i not in self._plugins:
                    return t
                return engine.build_denoising_aggregator(b, sd, t, n, p=f'layer{i + 1}.{nb}.', tag=f'layer{i + 1}.plugin',
                                                         **self._plugins[i].plan_kwargs())
            feats = engine.build_resnet(b, sd, x_in, after_stage=after_stage, **self.plan_kwargs())
            return b.plan, x_in, feats
        plan, x_in, feats = self._plan((n, h, w, str(x.device)), make)
        ops.pack_input(x.float().contiguous(), 8, out=x_in)
        plan.run()
        return tuple(ops.unpack_output(f, f.shape[-1], owner=plan) for f in feats)


@BACKBONES.register_module()
class ResNetH(ResNet):
    """mmtracking/mmtrack/models/backbones/resnet_H.py:123-426 ("ResNet backbone for Dark Denoising Network"), the one A..H experiment
    fork a config names (configs/vid/llvod/done/llvod_selsa_tra_rh50_l34*.py). It is InsertResNet's ResLayer with the plugin switch
    renamed: a stage with insert_plugins[i] ends in `build_aggregator(plugins_cfg['type'], **plugins_cfg['cfg'])` (resnet_H.py:94-95) --
    but `plugins_cfg` is a ResLayer argument that ResNetH never passes (it hands `plugins=` to the blocks instead, :204-217), so with any
    insert_plugins entry true (the default, and what both configs set) the reference's constructor fails with
    `TypeError: 'NoneType' object is not subscriptable`. Kept constructor-compatible: that case raises the same TypeError, saying why;
    with every entry false it is mmdet's ResNet on the conv engine (depth 50 / 101 only, :129-132), parameter names unchanged. The working
    form of the same idea is InsertResNet (insert_tag / insert_cfg)."""

    def __init__(self, depth, insert_plugins=(True, True, True, True), plugins=None, num_stages=4, **kwargs):
        if depth not in (50, 101):
            raise KeyError(f'invalid depth {depth} for resnet')
        assert len(insert_plugins) == num_stages
        if any(insert_plugins):
            raise TypeError("'NoneType' object is not subscriptable (ResNetH: resnet_H.py:94-95 builds the stage plugin from `plugins_cfg`, "
                            'which ResNetH never passes to its ResLayer -- the reference cannot construct this backbone with insert_plugins '
                            'set either; use InsertResNet(insert_tag=..., insert_cfg=...) for a ResNet with denoising plugins)')
        if plugins is not None:
            raise NotImplementedError('mmdet block plugins (context / non-local blocks) are outside the hot path')
        super().__init__(depth, num_stages=num_stages, **kwargs)
        self.insert_plugins = tuple(insert_plugins)
        self.feat_dim = 4 * self.base_channels * 2 ** (self.num_stages - 1)          # resnet_H.py:226-227


@BACKBONES.register_module()
class DarkResNet(ResNet):
    """mmtracking/mmtrack/models/backbones/dark_resnet.py:386-709 ("ResNet backbone for Dark Denoising Network"): ResNet-50 whose
    SECOND stage, or ResNet-101 whose FIRST stage, is made of DenoisingBlocks (arch_settings :392-395) -- Bottlenecks whose 3x3 conv is
    a ConvLSTM along the frames of the batch. Same constructor; `seqs` is accepted and, as in the reference, never reaches the blocks
    (:455 tests `isinstance(block[0], DenoisingBlock)` on the CLASS, which is false, so the stage is built by the plain ResLayer with
    the block's default seqs=4, which the block's forward does not read either). Parameter names as the reference's
    (`layer2.0.conv2.weight` of shape (4P, 2P, 3, 3), ...). zero_init_residual does not touch the DenoisingBlocks (:676-681 list
    Bottleneck and BasicBlock only)."""
    dark_arch = {50: (1,), 101: (0,)}

    def __init__(self, depth, seqs=4, **kwargs):
        if depth not in self.dark_arch:
            raise KeyError(f'invalid depth {depth} for resnet')
        super().__init__(depth, **kwargs)
        self.seqs = seqs
        self.dark_stages = tuple(i for i in self.dark_arch[depth] if i < self.num_stages)
        inplanes = self.stem_channels
        for i, num_blocks in enumerate(self.stage_blocks):
            planes = self.base_channels * 2 ** i
            if i in self.dark_stages:
                layer = ResLayer(block=DenoisingBlock, inplanes=inplanes, planes=planes, num_blocks=num_blocks, stride=self.strides[i],
                                 dilation=self.dilations[i], style=self.style, norm_cfg=self.norm_cfg)
                setattr(self, f'layer{i + 1}', layer)
            inplanes = planes * 4
        self._freeze_stages()
        self.block_expansion = 4
