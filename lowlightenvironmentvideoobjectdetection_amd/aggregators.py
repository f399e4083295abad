"""EmbedAggregator with the reference's constructor and parameter names
(mmtracking/mmtrack/models/aggregators/embed_aggregator.py:8-81). The embedding convs run on the conv kernel over
key + references in one launch; the tail (L2-normalise over C, cosine, softmax over frames, weighted sum) is
llv_embed_aggregate, one read of the reference stack."""
import torch
import torch.nn as nn

from . import engine, ops
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
