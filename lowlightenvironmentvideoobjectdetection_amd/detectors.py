"""BaseDetector / SingleStageDetector / RetinaNet with the reference's interface
(mmdetection/mmdet/models/detectors/base.py:16-360, single_stage.py:9-154, retinanet.py:5-17)."""
import warnings
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import engine, ops
from ._lib import LlvError
from .core.bbox import bbox2result
from .planned import PlannedModule
from .registry import DETECTORS, build_backbone, build_head, build_neck


class BaseDetector(PlannedModule):
    """base.py:16-253: forward dispatch, loss parsing, train_step."""

    def __init__(self):
        super().__init__()
        self.fp16_enabled = False

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    @property
    def with_bbox(self):
        return hasattr(self, 'bbox_head') and self.bbox_head is not None

    def init_weights(self, pretrained=None):
        if pretrained is not None:
            print(f'load model from: {pretrained}')

    def extract_feats(self, imgs):
        assert isinstance(imgs, list)
        return [self.extract_feat(img) for img in imgs]

    def forward_train(self, imgs, img_metas, **kwargs):
        batch_input_shape = tuple(imgs[0].size()[-2:])
        for img_meta in img_metas:
            img_meta['batch_input_shape'] = batch_input_shape

    def forward_test(self, imgs, img_metas, **kwargs):
        """base.py:125-167"""
        for var, name in [(imgs, 'imgs'), (img_metas, 'img_metas')]:
            if not isinstance(var, list):
                raise TypeError(f'{name} must be a list, but got {type(var)}')
        num_augs = len(imgs)
        if num_augs != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        for img, img_meta in zip(imgs, img_metas):
            batch_size = len(img_meta)
            for img_id in range(batch_size):
                img_meta[img_id]['batch_input_shape'] = tuple(img.size()[-2:])
        if num_augs == 1:
            if 'proposals' in kwargs:
                kwargs['proposals'] = kwargs['proposals'][0]
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        assert imgs[0].size(0) == 1, f'aug test does not support inference with batch size {imgs[0].size(0)}'
        assert 'proposals' not in kwargs
        return self.aug_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        """base.py:169-183"""
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def _parse_losses(self, losses):
        """base.py:185-218: sum the 'loss' terms; all-reduce (mean) every logged scalar over the ranks."""
        log_vars = OrderedDict()
        for loss_name, loss_value in losses.items():
            if isinstance(loss_value, torch.Tensor):
                log_vars[loss_name] = loss_value.mean()
            elif isinstance(loss_value, list):
                log_vars[loss_name] = sum(_loss.mean() for _loss in loss_value)
            else:
                raise TypeError(f'{loss_name} is not a tensor or list of tensors')
        loss = sum(_value for _key, _value in log_vars.items() if 'loss' in _key)
        log_vars['loss'] = loss
        for loss_name, loss_value in log_vars.items():
            if dist.is_available() and dist.is_initialized():
                loss_value = loss_value.data.clone()
                dist.all_reduce(loss_value.div_(dist.get_world_size()))
            log_vars[loss_name] = loss_value.item()
        return loss, log_vars

    def train_step(self, data, optimizer):
        """base.py:220-253"""
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    def val_step(self, data, optimizer):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))


@DETECTORS.register_module()
class SingleStageDetector(BaseDetector):
    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        if neck is not None:
            self.neck = build_neck(neck)
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg)
        bbox_head.update(test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.logits_f32 = True      # head outputs kept in fp32 for get_bboxes (anchor_head.py:495 force_fp32)
        self.init_weights(pretrained=pretrained)

    def init_weights(self, pretrained=None):
        super().init_weights(pretrained)
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            if isinstance(self.neck, nn.Sequential):
                for m in self.neck:
                    m.init_weights()
            else:
                self.neck.init_weights()
        self.bbox_head.init_weights()
        self.invalidate_plans()

    def invalidate_plans(self):
        super().invalidate_plans()
        for m in self.children():
            if isinstance(m, PlannedModule):
                m.invalidate_plans()

    # -- the fused HIP plan: backbone + neck + head + decode + NMS -----------------------------------
    def make_plan(self, batch, height, width, device, x_in=None):
        if not self.with_neck:
            raise LlvError('a neck (FPN) is required on this path')
        head, neck, bb = self.bbox_head, self.neck, self.backbone
        return engine.RetinaNetPlan(self.state_dict(), head.cls_out_channels, batch, height, width, device,
                                    depth=bb.depth, strides=bb.strides, style=bb.style, out_indices=bb.out_indices,
                                    start_level=neck.start_level, num_outs=neck.num_outs,
                                    add_extra_convs=neck.add_extra_convs, relu_before_extra_convs=neck.relu_before_extra_convs,
                                    stacked_convs=head.stacked_convs, anchor_generator=head.anchor_generator,
                                    test_cfg=dict(head.test_cfg) if head.test_cfg else None,
                                    target_means=head.bbox_coder.means, target_stds=head.bbox_coder.stds,
                                    logits_f32=self.logits_f32, x_in=x_in)

    def _get_plan(self, img):
        n, c, h, w = img.shape
        return self._plan((n, h, w, str(img.device)), lambda: self.make_plan(n, h, w, img.device))

    def extract_feat(self, img):
        """single_stage.py:52-57 -> tuple of NCHW fp32 pyramid features (read back from the plan's buffers)."""
        self._check_input(img)
        plan = self._get_plan(img)
        ops.pack_input(img.float().contiguous(), 8, out=plan.x_in)
        plan.plan.run()
        pyr = plan.pyramid
        return tuple(ops.unpack_output(pyr.level(i).contiguous(), pyr.buf.shape[-1]) for i in range(len(pyr.level_hw)))

    def forward_dummy(self, img):
        x = self.extract_feat(img)
        return self.bbox_head(x)

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        """single_stage.py:68-96"""
        from .training import single_stage_forward_train
        super().forward_train(img, img_metas)
        return single_stage_forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore)

    def simple_test(self, img, img_metas, rescale=False):
        """single_stage.py:98-129: one pack, one plan replay, one D2H of <= max_per_img rows per image."""
        self._check_input(img)
        plan = self._get_plan(img)
        plan.post.set_metas(img_metas, rescale)
        ops.pack_input(img.float().contiguous(), 8, out=plan.x_in)
        plan.plan.run()
        return [bbox2result(d, l, self.bbox_head.num_classes) for d, l in plan.results()]

    def aug_test(self, imgs, img_metas, rescale=False):
        raise NotImplementedError('test-time augmentation is outside the hot path')


@DETECTORS.register_module()
class RetinaNet(SingleStageDetector):
    """retinanet.py:5-17"""

    def __init__(self, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__(backbone, neck, bbox_head, train_cfg, test_cfg, pretrained)
