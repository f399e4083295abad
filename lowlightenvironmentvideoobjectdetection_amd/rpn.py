"""Proposal generation of the two-stage path (second piece): `ChannelMapper` neck and `RPNHead` with the
reference's constructors and parameter names (mmdetection/mmdet/models/necks/channel_mapper.py:10-74,
mmdetection/mmdet/models/dense_heads/rpn_head.py:15-234), inference only.

Convolutions run on the conv engine. `RPNHead.get_bboxes` is the dense-head decode kernel with one class (sigmoid -> per-level
top nms_pre -> delta2bbox -> clip, `llv_retina_decode_topk`) followed by `llv_multiclass_nms` with the pyramid LEVEL in the role of
the class, which is exactly mmcv's `batched_nms(proposals, scores, level_ids)`; a level can contribute at most max_per_img
proposals to the final list, so the per-class cap of that kernel is exact here too. The RoI heads that consume the proposals
(`SelsaRoIHead`, `SelsaBBoxHead`, `SelsaAggregator`: selsa.py; `TemporalRoIAlign`: roi_extractors.py) are built for test time."""
import torch
import torch.nn as nn

from . import ops
from ._lib import CONV_OUT_F32, CONV_RELU, LlvError
from .cnn import ConvModule, normal_init, xavier_init
from .dense_heads import AnchorHead, _cfg
from .planned import PlannedModule
from .registry import HEADS, NECKS


class _FreshBuffer:
    """Owner tag of a buffer that is written once and never again (ops._nhwc_source)."""
    epoch = 0


def _conv_nchw(x, pc, flags=0):
    """(N, C, H, W) fp32 on the GPU -> conv engine -> (N, Cout_stored, Ho, Wo) fp32 (API boundary of the module-level calls). The map is
    produced in NHWC fp16, the layout and type every consumer of this package converts it to anyway; the NCHW fp32 tensor handed out
    remembers that buffer, so such a consumer's pack_input is free."""
    c = x.shape[1]
    cp = ops.round_up(c, 8)
    y = ops.conv2d(ops.pack_input(x.float().contiguous(), cp), pc, flags=flags)
    return ops.unpack_output(y, y.shape[-1], owner=_FreshBuffer)


@NECKS.register_module()
class ChannelMapper(PlannedModule):
    def __init__(self, in_channels, out_channels, kernel_size=3, conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU')):
        super().__init__()
        assert isinstance(in_channels, list)
        if norm_cfg is not None or conv_cfg is not None:
            raise NotImplementedError('ChannelMapper with conv_cfg / norm_cfg is not on this path')
        self.act = act_cfg is not None
        self.convs = nn.ModuleList([ConvModule(c, out_channels, kernel_size, padding=(kernel_size - 1) // 2, norm_cfg=None, act_cfg=act_cfg)
                                    for c in in_channels])

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')
        self.invalidate_plans()

    def forward(self, inputs):
        assert len(inputs) == len(self.convs)
        outs = []
        for i, x in enumerate(inputs):
            self._check_input(x)
            conv = self.convs[i].conv
            pc = self._plan(('pc', i, str(x.device)),
                            lambda: ops.pack_conv(conv.weight.detach().cpu(), None if conv.bias is None else conv.bias.detach().cpu(), 1,
                                                  (conv.kernel_size[0] - 1) // 2).to(x.device))
            y = _conv_nchw(x, pc, CONV_RELU if self.act else 0)
            outs.append(y if y.shape[1] == conv.out_channels else y[:, :conv.out_channels])
        return tuple(outs)


@HEADS.register_module()
class RPNHead(AnchorHead):
    def __init__(self, in_channels, **kwargs):
        super().__init__(1, in_channels, **kwargs)

    def _init_layers(self):
        self.rpn_conv = nn.Conv2d(self.in_channels, self.feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(self.feat_channels, self.num_anchors * self.cls_out_channels, 1)
        self.rpn_reg = nn.Conv2d(self.feat_channels, self.num_anchors * 4, 1)

    def init_weights(self):
        for m in (self.rpn_conv, self.rpn_cls, self.rpn_reg):
            normal_init(m, std=0.01)
        self.invalidate_plans()

    def _packed(self, dev):
        def make():
            g = lambda m, pad: ops.pack_conv(m.weight.detach().cpu(), m.bias.detach().cpu(), 1, pad).to(dev)
            return g(self.rpn_conv, 1), g(self.rpn_cls, 0), g(self.rpn_reg, 0)
        return self._plan(('pc', str(dev)), make)

    def forward(self, feats):
        """rpn_head.py:43-50 over the levels: (cls_scores, bbox_preds), lists of NCHW fp32."""
        cls, reg = [], []
        for x in feats:
            self._check_input(x)
            pc_conv, pc_cls, pc_reg = self._packed(x.device)
            cp = ops.round_up(x.shape[1], 8)
            t = ops.conv2d(ops.pack_input(x.float().contiguous(), cp), pc_conv, flags=CONV_RELU)
            ncls, nreg = self.num_anchors * self.cls_out_channels, self.num_anchors * 4       # the packed convs store multiples of 4
            cls.append(ops.conv2d(t, pc_cls, flags=CONV_OUT_F32)[..., :ncls].permute(0, 3, 1, 2).contiguous())
            reg.append(ops.conv2d(t, pc_reg, flags=CONV_OUT_F32)[..., :nreg].permute(0, 3, 1, 2).contiguous())
        return cls, reg

    def loss(self, cls_scores, bbox_preds, gt_bboxes, img_metas, gt_bboxes_ignore=None):
        """rpn_head.py:49-80: AnchorHead.loss without class labels (every matched anchor is class 0 = foreground), keys renamed."""
        from .training import anchor_head_loss
        losses = anchor_head_loss(self, cls_scores, bbox_preds, gt_bboxes, None, img_metas, gt_bboxes_ignore)
        if losses is None:
            raise LlvError('RPNHead.loss: no valid anchors inside the image')
        return dict(loss_rpn_cls=losses['loss_cls'], loss_rpn_bbox=losses['loss_bbox'])

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None, outs=None, **kwargs):
        """base_dense_head.py:23-59: (losses, proposal_list). `outs` = (cls_scores, bbox_preds) as the detector's training graph
        produced them (two_stage_train.TwoStageBody: the RPN's three convolutions are launches of that graph, which is what carries
        the gradient into their weights); without them this module could only offer constants, so it refuses."""
        if outs is None:
            raise LlvError('RPNHead.forward_train: pass outs=(cls_scores, bbox_preds) from the training graph '
                           '(SELSA.forward_train does); a module-level call would not reach the conv weights')
        losses = self.loss(*outs, gt_bboxes, img_metas, gt_bboxes_ignore=gt_bboxes_ignore)
        if proposal_cfg is None:
            return losses
        with torch.no_grad():
            proposal_list = self.get_bboxes([t.detach() for t in outs[0]], [t.detach() for t in outs[1]], img_metas, cfg=proposal_cfg)
        return losses, proposal_list

    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg=None, rescale=False, with_nms=True):
        """rpn_head.py:88-234. Returns list[N] of (k, 5) proposals (x1, y1, x2, y2, score) on the GPU, best first."""
        cfg = dict(_cfg(self.test_cfg if cfg is None else cfg))
        if not getattr(self, 'use_sigmoid_cls', True) or self.cls_out_channels != 1:
            # rpn_head.py:128-137 also has a softmax branch (2 channels per anchor, score = softmax[..., 0]); the decode plan below
            # reads ONE sigmoid logit per anchor, so such a head would be decoded from misaligned logits
            raise LlvError('RPNHead.get_bboxes: only the sigmoid objectness head (loss_cls use_sigmoid=True, 1 channel per anchor) is built')
        if 'nms' not in cfg:
            cfg['nms'] = dict(type='nms', iou_threshold=cfg['nms_thr'])                # deprecated spelling, rpn_head.py:189-213
        max_per_img = cfg.get('max_per_img', cfg.get('max_num'))
        nms_pre = cfg.get('nms_pre', -1)
        dev = cls_scores[0].device
        batch = cls_scores[0].shape[0]
        level_hw = [tuple(c.shape[-2:]) for c in cls_scores]
        A = self.num_anchors
        results = []
        for n in range(batch):
            # everything that depends only on the pyramid geometry and the image shape is built once and kept: the logit buffers the
            # decode descriptor points into, the descriptor and its workspace, the level index of every candidate row
            img_shape = tuple(int(v) for v in img_metas[n]['img_shape'])
            st = self._plan(('decode', tuple(level_hw), img_shape, nms_pre, str(dev)), lambda: self._decode_state(level_hw, img_shape, nms_pre, dev))
            o = 0
            for c, r in zip(cls_scores, bbox_preds):
                k = c.shape[-2] * c.shape[-1]
                st['cls'][o:o + k].copy_(c[n].detach().permute(1, 2, 0).reshape(-1, A))
                st['reg'][o:o + k].copy_(r[n].detach().permute(1, 2, 0).reshape(-1, A * 4))
                o += k
            boxes, scores = ops.retina_decode_topk(st['plan'])
            s = scores[:, 0]
            if cfg.get('min_bbox_size', 0) > 0:
                ok = ((boxes[:, 2] - boxes[:, 0]) >= cfg['min_bbox_size']) & ((boxes[:, 3] - boxes[:, 1]) >= cfg['min_bbox_size'])
                s = s * ok
            multi = st['multi']                                                        # one column per level + the ignored background column
            multi.zero_()
            multi[st['rows'], st['lvl']] = s
            dets, _ = ops.multiclass_nms(boxes.contiguous(), multi, 0.0, cfg['nms'], max_per_img)
            results.append(dets)
        return results

    def _decode_state(self, level_hw, img_shape, nms_pre, dev):
        A = self.num_anchors
        offs, tot = [], 0
        for h, w in level_hw:
            offs.append(tot)
            tot += h * w
        cls_buf = torch.zeros((tot, A), device=dev, dtype=torch.float32)
        reg_buf = torch.zeros((tot, A * 4), device=dev, dtype=torch.float32)
        plan = ops.make_decode_plan(cls_buf, reg_buf, level_hw, self.anchor_generator.strides, offs, offs, A, A * 4, 1, A,
                                    self.anchor_generator.base_anchors, nms_pre, self.bbox_coder.means, self.bbox_coder.stds,
                                    16 / 1000, img_shape, (1., 1., 1., 1.), False)
        counts = [min(nms_pre, h * w * A) if nms_pre > 0 else h * w * A for h, w in level_hw]
        lvl = torch.cat([torch.full((c,), i, dtype=torch.long) for i, c in enumerate(counts)]).to(dev)
        return dict(cls=cls_buf, reg=reg_buf, plan=plan, lvl=lvl, rows=torch.arange(sum(counts), device=dev),
                    multi=torch.zeros(sum(counts), len(level_hw) + 1, device=dev))

    def simple_test_rpn(self, x, img_metas):
        """rpn_test_mixin.py:27-41."""
        return self.get_bboxes(*self(x), img_metas)
