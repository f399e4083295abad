"""Aggregators with the reference's constructors and parameter names: EmbedAggregator and the authors' DenoisingAggregator.

EmbedAggregator
(mmtracking/mmtrack/models/aggregators/embed_aggregator.py:8-81). The embedding convs run on the conv kernel over
key + references in one launch; the tail (L2-normalise over C, cosine, softmax over frames, weighted sum) is
llv_embed_aggregate, one read of the reference stack."""
import torch
import torch.nn as nn

from . import engine, ops
import math

from ._lib import CONV_RELU
from .cnn import ConvModule
from .planned import PlannedModule
from .registry import AGGREGATORS


def embed_packed_convs(sd, num_convs, prefix=''):
    """ConvModule stack of the aggregator: ReLU on all but the last conv (embed_aggregator.py:33-48)."""
    pcs = []
    for i in range(num_convs):
        w = sd[f'{prefix}embed_convs.{i}.conv.weight']
        pcs.append((ops.pack_conv(w, sd.get(f'{prefix}embed_convs.{i}.conv.bias'), 1, (w.shape[-1] - 1) // 2),
                    CONV_RELU if i != num_convs - 1 else 0))
    return pcs


@AGGREGATORS.register_module()
class EmbedAggregator(PlannedModule):
    def __init__(self, num_convs=1, channels=256, kernel_size=3, norm_cfg=None, act_cfg=dict(type='ReLU')):
        super().__init__()
        assert num_convs > 0, 'The number of convs must be bigger than 1.'
        if norm_cfg is not None:
            raise NotImplementedError('EmbedAggregator with norm_cfg is outside the hot path')
        self.num_convs, self.channels, self.kernel_size = num_convs, channels, kernel_size
        self.embed_convs = nn.ModuleList()
        for i in range(num_convs):
            last = i == num_convs - 1
            self.embed_convs.append(ConvModule(channels, channels, kernel_size, padding=(kernel_size - 1) // 2,
                                               norm_cfg=None, act_cfg=None if last else act_cfg))

    def init_weights(self):
        for m in self.embed_convs:
            m.init_weights()
        self.invalidate_plans()

    def forward(self, x, ref_x):
        """x (1,C,H,W), ref_x (N,C,H,W) NCHW fp32 on the GPU -> aggregated (1,C,H,W) (embed_aggregator.py:50-81)."""
        self._check_input(x)
        self._check_input(ref_x)
        assert len(x.shape) == 4 and len(x) == 1, "Only support 'batch_size == 1' for x"
        n, c, h, w = ref_x.shape
        cp = ops.round_up(c, 8)

        def make():
            b = engine.Builder(x.device)
            feat = b.buf(n + 1, h, w, cp)          # slot 0 = key frame, 1.. = references
            t = feat
            for i, (pc, flags) in enumerate(embed_packed_convs(engine._cpu_sd(self.state_dict()), self.num_convs)):
                t = b.conv(f'embed{i}', t, pc, flags)
            out = b.buf(1, h, w, cp)
            b.plan.add('aggregate', ops.lib().llv_embed_aggregate,
                       (ops._ptr(t[0:1]), ops._ptr(t[1:]), ops._ptr(feat[1:]), ops._ptr(out), n, h * w, cp, t.shape[-1],
                        cp, cp))
            b.plan.keep += [t, out]
            return b.plan, feat, out
        plan, feat, out = self._plan((n, c, h, w, str(x.device)), make)
        ops.pack_input(torch.cat([x, ref_x], 0).float().contiguous(), cp, out=feat)
        plan.run()
        return ops.unpack_output(out, c)


# ------------------------------------------------------------------------------------------------
# DenoisingAggregator: residual dense blocks + DCN-based temporal attention fusion
# mmtracking/mmtrack/models/aggregators/denoising_aggregator.py
# ------------------------------------------------------------------------------------------------
def _conv3(cin, cout, k=3):
    return nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2)


class DenseLayer(nn.Module):
    """denoising_aggregator.py:10-34 (parameter container: `conv`)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = _conv3(in_channels, out_channels)


class RDB(nn.Module):
    """denoising_aggregator.py:37-66 (`layers.{i}.conv`, `lff`)."""

    def __init__(self, in_channels, channel_growth, num_layers):
        super().__init__()
        self.layers = nn.Sequential(*[DenseLayer(in_channels + channel_growth * i, channel_growth) for i in range(num_layers)])
        self.lff = nn.Conv2d(in_channels + channel_growth * num_layers, in_channels, kernel_size=1)


class ModulatedDCNPack(nn.Module):
    """denoising_aggregator.py:69-111 over mmcv's ModulatedDeformConv2d: `weight`, `bias`, `conv_offset`. Initialisation
    as mmcv 1.3: weight uniform(+-1/sqrt(Cin*kh*kw)), bias zero, conv_offset zero (offsets 0, mask 0.5 at the start)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deform_groups=1,
                 bias=True):
        super().__init__()
        if groups != 1 or stride != 1 or dilation != 1:
            raise NotImplementedError('ModulatedDCNPack: the temporal fusion uses groups = stride = dilation = 1')
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, (k, k)
        self.stride, self.padding, self.dilation, self.groups, self.deform_groups = stride, padding, dilation, groups, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, k, k))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.conv_offset = nn.Conv2d(in_channels, deform_groups * 3 * k * k, kernel_size=k, stride=stride, padding=padding, bias=True)
        stdv = 1.0 / math.sqrt(in_channels * k * k)
        nn.init.uniform_(self.weight, -stdv, stdv)
        nn.init.constant_(self.conv_offset.weight, 0)
        nn.init.constant_(self.conv_offset.bias, 0)


class TemporalAttentionFusion(nn.Module):
    """denoising_aggregator.py:114-152 (`conv1`, `offset_conv`, `dcn_pack`, `emb_conv.{i}`, `conv2`)."""

    def __init__(self, channels, mid_channels, emb_nums=3):
        super().__init__()
        self.channels, self.mid_channels, self.emb_nums = channels, mid_channels, emb_nums
        self.conv1 = _conv3(channels, mid_channels)
        self.offset_conv = _conv3(mid_channels * 2, mid_channels)
        self.dcn_pack = ModulatedDCNPack(mid_channels, mid_channels, 3, padding=1, deform_groups=8)
        self.emb_conv = nn.Sequential(*[_conv3(mid_channels, mid_channels) for _ in range(emb_nums)])
        self.conv2 = _conv3(mid_channels, channels)


@AGGREGATORS.register_module()
class DenoisingAggregator(PlannedModule):
    """The authors' denoising aggregator (denoising_aggregator.py:155-200): conv1 -> residual dense blocks -> conv2 ->
    TemporalAttentionFusion over the N frames of the batch -> conv3, with the two inner skips and the outer identity.

    forward(x): x (N, C, H, W) fp32 NCHW on the GPU = the feature maps of N frames -> same shape. The whole block is one
    launch plan (engine.build_denoising_aggregator): every step a libllv.so launch on NHWC fp16 buffers, replayable from a
    HIP graph, and embeddable between the stages of a backbone plan (InsertResNet)."""

    def __init__(self, in_channels=64, mid_channels=64, num_blocks=1, num_layers=6, channel_growth=64, emb_nums=3,
                 with_rdb=True, with_taf=True):
        super().__init__()
        if in_channels % 8 or mid_channels % 64 or channel_growth % 8:
            raise NotImplementedError('DenoisingAggregator: channel counts must keep 8-channel vectors whole and each of '
                                      'the 8 deformable groups a multiple of 8 channels')
        self.in_channels, self.mid_channels, self.num_blocks, self.num_layers = in_channels, mid_channels, num_blocks, num_layers
        self.channel_growth, self.emb_nums, self.with_rdb, self.with_taf = channel_growth, emb_nums, with_rdb, with_taf
        self.conv1 = _conv3(in_channels, mid_channels)
        if with_rdb:
            self.rdbs = nn.Sequential(*[RDB(mid_channels, channel_growth, num_layers) for _ in range(num_blocks)])
        self.conv2 = _conv3(mid_channels, mid_channels)
        if with_taf:
            self.taf = TemporalAttentionFusion(mid_channels, mid_channels=mid_channels, emb_nums=emb_nums)
        self.conv3 = _conv3(mid_channels, in_channels)

    def plan_kwargs(self):
        return dict(num_blocks=self.num_blocks, num_layers=self.num_layers, emb_nums=self.emb_nums, with_rdb=self.with_rdb,
                    with_taf=self.with_taf, deform_groups=self.taf.dcn_pack.deform_groups if self.with_taf else 8)

    def make_plan(self, n, h, w, device):
        b = engine.Builder(device)
        x_in = b.buf(n, h, w, self.in_channels)
        out = engine.build_denoising_aggregator(b, engine._cpu_sd(self.state_dict()), x_in, n, **self.plan_kwargs())
        return b.plan, x_in, out

    def forward_train(self, x):
        """The same block under autograd (dev_autograd.py: every convolution and the deformable convolution a differentiable device
        op -- forward, data and weight gradients on libllv.so; mmcv modulated_deform_conv2d's gradients w.r.t. input, offsets and
        mask = llv_modulated_deform_col2im): x (N, C, H, W) fp32 -> (N, C, H, W) fp32 carrying the graph to x and to every parameter."""
        from . import dev_autograd as D
        assert x.shape[1] == self.in_channels
        return D.from_dev(D.denoising_aggregator_forward(self, D.to_dev(x)), self.in_channels)

    def forward(self, x):
        self._check_input(x)
        if torch.is_grad_enabled() and x.requires_grad:
            return self.forward_train(x)
        n, c, h, w = x.shape
        assert c == self.in_channels
        plan, x_in, out = self._plan((n, h, w, str(x.device)), lambda: self.make_plan(n, h, w, x.device))
        ops.pack_input(x.float().contiguous(), c, out=x_in)
        plan.run()
        return ops.unpack_output(out, c)


@AGGREGATORS.register_module()
class Denoising2Aggergator(PlannedModule):
    """The authors' multi-stage denoising aggregator (mmtracking/mmtrack/models/aggregators/denoising2_aggregator.py:157-244; the class
    name is the reference's spelling, its configs say `type='Denoising2Aggergator'`): for every backbone stage conv1 (from the second
    stage on over the stage's features concatenated with the previous stage's output) -> residual dense blocks of three layers ->
    TemporalAttentionFusion over the N frames -> + the stage's features; conv2 (stride 2 where `downsample`) feeds the next stage.

    forward(x_noise, all_x): x_noise = tuple of (N, C_i, H_i, W_i) fp32 NCHW stage features on the GPU, all_x = tuple whose last
    entry (N, out_channel[-1], H, W) receives the last stage's output -> (tuple of x_noise_out, tuple of len(all_x) copies of the sum),
    as the reference returns them. One launch plan (engine.build_denoising2_aggregator)."""

    def __init__(self, in_channel=(256, 512, 1024, 2048), mid_channel=(64, 128, 256, 512), out_channel=(512, 1024, 2048, 512),
                 layer_name=('layer1', 'layer2', 'layer3', 'layer4'), rdb_blocks=(2, 2, 4, 2), rdb_channel_growth=(64, 64, 64, 64),
                 taf_embs=(3, 3, 3, 3), downsample=(True, True, False, False), with_rdb=(True, True, True, True),
                 with_taf=(True, True, True, True)):
        super().__init__()
        self.num_stage = len(in_channel)
        self.in_channel, self.mid_channel, self.out_channel = list(in_channel), list(mid_channel), list(out_channel)
        self.layer_name, self.rdb_blocks, self.taf_embs = list(layer_name), list(rdb_blocks), list(taf_embs)
        self.downsample, self.with_rdb, self.with_taf = list(downsample), list(with_rdb), list(with_taf)
        for i in range(self.num_stage):
            if in_channel[i] % 8 or out_channel[i] % 8 or rdb_channel_growth[i] % 8 or (with_taf[i] and mid_channel[i] % 64):
                raise NotImplementedError('Denoising2Aggergator: channel counts must keep 8-channel vectors whole and each of the 8 '
                                          'deformable groups a multiple of 8 channels')
        self.layers = nn.ModuleDict()
        for i in range(self.num_stage):
            l = layer_name[i]
            self.layers[l + '_conv1'] = _conv3(in_channel[0] if i == 0 else in_channel[i] + out_channel[i - 1], in_channel[i])
            if with_rdb[i]:
                self.layers[l + '_rdb'] = nn.Sequential(*[RDB(in_channel[i], rdb_channel_growth[i], 3) for _ in range(rdb_blocks[i])])
            if with_taf[i]:
                self.layers[l + '_taf'] = TemporalAttentionFusion(in_channel[i], mid_channel[i], emb_nums=taf_embs[i])
            self.layers[l + '_conv2'] = nn.Conv2d(in_channel[i], out_channel[i], kernel_size=3, padding=1, stride=2 if downsample[i] else 1)

    def make_plan(self, n, shapes, last_shape, device):
        b = engine.Builder(device)
        xs = [b.buf(n, h, w, c) for (c, h, w) in shapes]
        last = b.buf(n, last_shape[1], last_shape[2], last_shape[0])
        outs, total = engine.build_denoising2_aggregator(b, engine._cpu_sd(self.state_dict()), xs, last, n, self.in_channel, self.out_channel,
                                                         self.layer_name, self.rdb_blocks, self.taf_embs, self.downsample, self.with_rdb,
                                                         self.with_taf)
        return b.plan, xs, last, outs, total

    def forward_train(self, x_noise, all_x):
        """forward under autograd (dev_autograd.py), same arguments and return value: gradients reach every x_noise[i], all_x[-1] and
        every parameter."""
        from . import dev_autograd as D
        outs, total = D.denoising2_aggregator_forward(self, [D.to_dev(t) for t in x_noise], D.to_dev(all_x[-1]))
        s = D.from_dev(total, all_x[-1].shape[1])
        return tuple(D.from_dev(o, t.shape[1]) for o, t in zip(outs, x_noise)), tuple(s for _ in all_x)

    def forward(self, x_noise, all_x):
        for t in tuple(x_noise) + (all_x[-1],):
            self._check_input(t)
        if torch.is_grad_enabled() and any(t.requires_grad for t in tuple(x_noise) + (all_x[-1],)):
            return self.forward_train(x_noise, all_x)
        n = x_noise[0].shape[0]
        shapes = tuple(tuple(t.shape[1:]) for t in x_noise)
        last_shape = tuple(all_x[-1].shape[1:])
        dev = x_noise[0].device
        plan, xs, last, outs, total = self._plan((n, shapes, last_shape, str(dev)), lambda: self.make_plan(n, shapes, last_shape, dev))
        for t, buf in zip(x_noise, xs):
            ops.pack_input(t.float().contiguous(), t.shape[1], out=buf)
        ops.pack_input(all_x[-1].float().contiguous(), last_shape[0], out=last)
        plan.run()
        s = ops.unpack_output(total, last_shape[0])
        return tuple(ops.unpack_output(o, o.shape[-1]) for o in outs), tuple(s for _ in all_x)
