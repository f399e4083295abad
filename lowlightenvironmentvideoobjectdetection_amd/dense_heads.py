"""AnchorHead / RetinaHead with the reference's constructor, parameter names, init and test-time interface
(mmdetection/mmdet/models/dense_heads/anchor_head.py:14-744, retina_head.py:8-114, base_dense_head.py:7-59)."""
import numpy as np
import torch
import torch.nn as nn

from . import engine, ops
from ._lib import LlvError
from .cnn import ConvModule, bias_init_with_prob, normal_init
from .planned import PlannedModule
from .registry import HEADS, build_anchor_generator, build_bbox_coder, build_loss


class ConfigDict(dict):
    """dict with attribute access (the slice of mmcv.ConfigDict / addict the heads rely on: cfg.score_thr)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return ConfigDict(v) if isinstance(v, dict) and not isinstance(v, ConfigDict) else v

    def __setattr__(self, k, v):
        self[k] = v


def _cfg(d):
    return ConfigDict(d) if isinstance(d, dict) and not isinstance(d, ConfigDict) else d


@HEADS.register_module()
class AnchorHead(PlannedModule):
    def __init__(self, num_classes, in_channels, feat_channels=256,
                 anchor_generator=dict(type='AnchorGenerator', scales=[8, 16, 32], ratios=[0.5, 1.0, 2.0],
                                       strides=[4, 8, 16, 32, 64]),
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True, target_means=(.0, .0, .0, .0),
                                 target_stds=(1.0, 1.0, 1.0, 1.0)),
                 reg_decoded_bbox=False,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0), train_cfg=None, test_cfg=None):
        super().__init__()
        self.in_channels, self.num_classes, self.feat_channels = in_channels, num_classes, feat_channels
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        self.sampling = loss_cls['type'] not in ['FocalLoss', 'GHMC', 'QualityFocalLoss']
        self.cls_out_channels = num_classes if self.use_sigmoid_cls else num_classes + 1
        if self.cls_out_channels <= 0:
            raise ValueError(f'num_classes={num_classes} is too small')
        if not self.use_sigmoid_cls:
            raise NotImplementedError('softmax classification is outside the dense-head hot path')
        self.reg_decoded_bbox = reg_decoded_bbox
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.loss_cls_cfg, self.loss_bbox_cfg = dict(loss_cls), dict(loss_bbox)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.train_cfg, self.test_cfg = _cfg(train_cfg), _cfg(test_cfg)
        if self.train_cfg:
            from .registry import build_assigner, build_sampler
            self.assigner = build_assigner(self.train_cfg['assigner'])
            if self.sampling and 'sampler' in self.train_cfg:
                sampler_cfg = self.train_cfg['sampler']
            else:
                sampler_cfg = dict(type='PseudoSampler')
            self.sampler = build_sampler(sampler_cfg, context=self)
        self.fp16_enabled = False
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self._init_layers()

    def _init_layers(self):
        self.conv_cls = nn.Conv2d(self.in_channels, self.num_anchors * self.cls_out_channels, 1)
        self.conv_reg = nn.Conv2d(self.in_channels, self.num_anchors * 4, 1)

    def init_weights(self):
        normal_init(self.conv_cls, std=0.01)
        normal_init(self.conv_reg, std=0.01)
        self.invalidate_plans()

    # -- test-time box decoding on device ----------------------------------------------------------
    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg=None, rescale=False, with_nms=True):
        """anchor_head.py:495-588. cls_scores / bbox_preds: per-level NCHW GPU tensors. Returns
        list[(dets (k,5), labels (k,))] on the GPU. (The fused detector path never materialises NCHW.)"""
        assert len(cls_scores) == len(bbox_preds)
        if not with_nms:
            raise NotImplementedError('with_nms=False is outside the hot path')
        cfg = _cfg(self.test_cfg if cfg is None else cfg)
        for t in list(cls_scores) + list(bbox_preds):
            self._check_input(t)
        dev = cls_scores[0].device
        batch = cls_scores[0].shape[0]
        level_hw = [tuple(c.shape[-2:]) for c in cls_scores]
        # layout plumbing at the API boundary: NCHW lists -> one pixel-major NHWC buffer
        offs, tot = [], 0
        for h, w in level_hw:
            offs.append(tot)
            tot += batch * h * w
        cls_buf = torch.cat([c.detach().float().permute(0, 2, 3, 1).reshape(-1, c.shape[1]) for c in cls_scores]).contiguous()
        reg_buf = torch.cat([r.detach().float().permute(0, 2, 3, 1).reshape(-1, r.shape[1]) for r in bbox_preds]).contiguous()
        b = engine.Builder(dev)
        pyr = engine.Pyramid(cls_buf, level_hw, offs, batch)
        img_shapes = [m['img_shape'] for m in img_metas[:batch]]
        sfs = [m['scale_factor'] for m in img_metas[:batch]]
        post = engine.PostProcess(b, cls_buf, reg_buf, pyr, self.cls_out_channels, self.anchor_generator, cfg,
                                  self.bbox_coder.means, self.bbox_coder.stds, img_shapes, sfs, rescale,
                                  self.bbox_coder.clip_border)
        b.plan.run()
        return [(d.to(dev), l.to(dev)) for d, l in post.results()]

    def loss(self, *args, **kwargs):
        from .training import anchor_head_loss
        return anchor_head_loss(self, *args, **kwargs)

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        """base_dense_head.py:22-59"""
        outs = self(x)
        if gt_labels is None:
            loss_inputs = outs + (gt_bboxes, img_metas)
        else:
            loss_inputs = outs + (gt_bboxes, gt_labels, img_metas)
        losses = self.loss(*loss_inputs, gt_bboxes_ignore=gt_bboxes_ignore)
        if proposal_cfg is None:
            return losses
        return losses, self.get_bboxes(*outs, img_metas, cfg=proposal_cfg)


@HEADS.register_module()
class RetinaHead(AnchorHead):
    """Example (mirrors retina_head.py:16-25, on a GPU):
        >>> self = RetinaHead(11, 7).cuda()
        >>> cls_score, bbox_pred = self.forward_single(torch.rand(1, 7, 32, 32).cuda())
        >>> assert cls_score.shape[1] / self.num_anchors == self.num_classes
        >>> assert bbox_pred.shape[1] / self.num_anchors == 4
    """

    def __init__(self, num_classes, in_channels, stacked_convs=4, conv_cfg=None, norm_cfg=None,
                 anchor_generator=dict(type='AnchorGenerator', octave_base_scale=4, scales_per_octave=3,
                                       ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128]), **kwargs):
        if conv_cfg is not None or norm_cfg is not None:
            raise NotImplementedError('RetinaHead with conv_cfg / norm_cfg is outside the hot path')
        self.stacked_convs, self.conv_cfg, self.norm_cfg = stacked_convs, conv_cfg, norm_cfg
        super().__init__(num_classes, in_channels, anchor_generator=anchor_generator, **kwargs)

    def _init_layers(self):             # retina_head.py:50-81
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            self.cls_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1))
            self.reg_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1))
        self.retina_cls = nn.Conv2d(self.feat_channels, self.num_anchors * self.cls_out_channels, 3, padding=1)
        self.retina_reg = nn.Conv2d(self.feat_channels, self.num_anchors * 4, 3, padding=1)

    def init_weights(self):             # retina_head.py:83-91
        for m in self.cls_convs:
            normal_init(m.conv, std=0.01)
        for m in self.reg_convs:
            normal_init(m.conv, std=0.01)
        normal_init(self.retina_cls, std=0.01, bias=bias_init_with_prob(0.01))
        normal_init(self.retina_reg, std=0.01)
        self.invalidate_plans()

    def forward(self, feats):
        """tuple of per-level NCHW fp32 GPU features -> (cls_scores list, bbox_preds list), NCHW fp32."""
        for t in feats:
            self._check_input(t)
        batch = feats[0].shape[0]
        key = tuple(tuple(t.shape) for t in feats) + (str(feats[0].device),)

        def make():
            b = engine.Builder(feats[0].device)
            level_hw = [(t.shape[2], t.shape[3]) for t in feats]
            offs, tot = [], 0
            for h, w in level_hw:
                offs.append(tot)
                tot += batch * h * w
            pyr = engine.Pyramid(b.buf(tot, ops.round_up(self.in_channels, 8)), level_hw, offs, batch)
            cls_buf, reg_buf = engine.build_retina_head(b, engine._cpu_sd(self.state_dict()), pyr, self.stacked_convs, True)
            return b.plan, pyr, cls_buf, reg_buf
        plan, pyr, cls_buf, reg_buf = self._plan(key, make)
        for i, t in enumerate(feats):
            ops.pack_input(t.float().contiguous(), pyr.buf.shape[-1], out=pyr.level(i))
        plan.run()
        nc, nr = self.num_anchors * self.cls_out_channels, self.num_anchors * 4
        cls = [ops.unpack_output(pyr.level(i, cls_buf).contiguous(), nc) for i in range(len(feats))]
        reg = [ops.unpack_output(pyr.level(i, reg_buf).contiguous(), nr) for i in range(len(feats))]
        return cls, reg

    def forward_single(self, x):
        cls, reg = self.forward((x,))
        return cls[0], reg[0]
