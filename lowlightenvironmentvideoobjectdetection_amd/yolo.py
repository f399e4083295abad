"""YOLOv3 stand-in for BASELINE.json config #5 with the reference's constructors, parameter names and
test-time behaviour:

  Darknet               mmdetection/mmdet/models/backbones/darknet.py:14-199
  DetectionBlock / YOLOV3Neck   mmdetection/mmdet/models/necks/yolo_neck.py:11-136
  YOLOV3Head            mmdetection/mmdet/models/dense_heads/yolo_head.py:20-170 (layers, forward), :172-366 (get_bboxes)
  YOLOV3                mmdetection/mmdet/models/detectors/yolo.py
  YOLOAnchorGenerator   mmdetection/mmdet/core/anchor/anchor_generator.py:590-665
  YOLOBBoxCoder         mmdetection/mmdet/core/bbox/coder/yolo_bbox_coder.py

All convolutions run on the HIP conv engine (conv + folded BN + LeakyReLU(0.1) epilogue; ResBlock adds after the
activation: LLV_CONV_ACT_FIRST; the neck's upsample + concat is llv_upsample2x_nhwc into a channel slice of the buffer the
backbone stage already wrote its half of). Box decode = llv_yolo_decode, NMS = llv_multiclass_nms_batched_sf (objectness as
score factor). The per-level top-k on objectness and the gather of the <= 3000 selected rows are torch ops on the device
(selection plumbing on small tensors). YOLOV3Head.loss (targets + the four loss terms, element-wise torch on the device) is built; the
backward pass through Darknet / the neck is not (the training engine's tape covers the RetinaNet graph).
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import engine, ops
from ._lib import LlvError, lib, check, CONV_LEAKY, CONV_RES_ADD, CONV_ACT_FIRST, CONV_OUT_F32
from .cnn import ConvModule, constant_init, kaiming_init, normal_init
from .core.bbox import bbox2result
from .detectors import SingleStageDetector
from .planned import PlannedModule
from .registry import ANCHOR_GENERATORS, BACKBONES, BBOX_CODERS, DETECTORS, HEADS, NECKS, build_anchor_generator, build_bbox_coder

_CFG = dict(norm_cfg=dict(type='BN', requires_grad=True), act_cfg=dict(type='LeakyReLU', negative_slope=0.1))
ARCH = {53: ((1, 2, 8, 8, 4), ((32, 64), (64, 128), (128, 256), (256, 512), (512, 1024)))}
SLOPE = 0.1


@ANCHOR_GENERATORS.register_module()
class YOLOAnchorGenerator:
    def __init__(self, strides, base_sizes):
        self.strides = [(s, s) if isinstance(s, int) else tuple(s) for s in strides]
        self.centers = [(s[0] / 2., s[1] / 2.) for s in self.strides]
        self.base_sizes = [[(bs, bs) if isinstance(bs, (int, float)) else tuple(bs) for bs in lvl] for lvl in base_sizes]
        assert len(self.base_sizes) == len(self.strides)
        self.base_anchors = self.gen_base_anchors()

    @property
    def num_levels(self):
        return len(self.base_sizes)

    @property
    def num_base_anchors(self):
        return [b.size(0) for b in self.base_anchors]

    def gen_base_anchors(self):
        out = []
        for sizes, (xc, yc) in zip(self.base_sizes, self.centers):
            out.append(torch.stack([torch.Tensor([xc - 0.5 * w, yc - 0.5 * h, xc + 0.5 * w, yc + 0.5 * h]) for w, h in sizes], 0))
        return out

    def grid_anchors(self, featmap_sizes, device='cuda'):
        """anchor_generator.py:207-270: per level (h*w*A, 4), (y, x, anchor) order. Only the training targets need the grid; inference
        adds the shifts inside llv_yolo_decode."""
        out = []
        for (h, w), base, (sw, sh) in zip(featmap_sizes, self.base_anchors, self.strides):
            base = base.to(device)
            sx = torch.arange(0, w, device=device) * sw
            sy = torch.arange(0, h, device=device) * sh
            xx = sx.repeat(len(sy))
            yy = sy.view(-1, 1).repeat(1, len(sx)).view(-1)
            shifts = torch.stack([xx, yy, xx, yy], dim=-1).type_as(base)
            out.append((base[None, :, :] + shifts[:, None, :]).view(-1, 4))
        return out

    def responsible_flags(self, featmap_sizes, gt_bboxes, device='cuda'):
        """anchor_generator.py:667-728: flag = 1 for the anchors of the grid cell a ground-truth centre falls in."""
        out = []
        for (h, w), (sw, sh), nb in zip(featmap_sizes, self.strides, self.num_base_anchors):
            cx = ((gt_bboxes[:, 0] + gt_bboxes[:, 2]) * 0.5).to(device)
            cy = ((gt_bboxes[:, 1] + gt_bboxes[:, 3]) * 0.5).to(device)
            idx = torch.floor(cy / sh).long() * w + torch.floor(cx / sw).long()
            grid = torch.zeros(h * w, dtype=torch.uint8, device=device)
            grid[idx] = 1
            out.append(grid[:, None].expand(grid.size(0), nb).contiguous().view(-1))
        return out


@BBOX_CODERS.register_module()
class YOLOBBoxCoder:
    def __init__(self, eps=1e-6):
        self.eps = eps

    def encode(self, bboxes, gt_bboxes, stride):
        """yolo_bbox_coder.py:26-58: centre offsets inside the cell (clamped to (eps, 1 - eps)) and log size ratios."""
        assert bboxes.size(0) == gt_bboxes.size(0) and bboxes.size(-1) == gt_bboxes.size(-1) == 4
        xg, yg = (gt_bboxes[..., 0] + gt_bboxes[..., 2]) * 0.5, (gt_bboxes[..., 1] + gt_bboxes[..., 3]) * 0.5
        wg, hg = gt_bboxes[..., 2] - gt_bboxes[..., 0], gt_bboxes[..., 3] - gt_bboxes[..., 1]
        xc, yc = (bboxes[..., 0] + bboxes[..., 2]) * 0.5, (bboxes[..., 1] + bboxes[..., 3]) * 0.5
        w, h = bboxes[..., 2] - bboxes[..., 0], bboxes[..., 3] - bboxes[..., 1]
        wt, ht = torch.log((wg / w).clamp(min=self.eps)), torch.log((hg / h).clamp(min=self.eps))
        xt = ((xg - xc) / stride + 0.5).clamp(self.eps, 1 - self.eps)
        yt = ((yg - yc) / stride + 0.5).clamp(self.eps, 1 - self.eps)
        return torch.stack([xt, yt, wt, ht], dim=-1)

    def decode(self, bboxes, pred_bboxes, stride):
        """yolo_bbox_coder.py:61-89 on explicit anchors (the reference's coder API, pinned by its test_coder.py:8-23); the head's
        test path decodes against the grid inside llv_yolo_decode instead."""
        assert pred_bboxes.size(0) == bboxes.size(0) and pred_bboxes.size(-1) == bboxes.size(-1) == 4
        xc, yc = (bboxes[..., 0] + bboxes[..., 2]) * 0.5, (bboxes[..., 1] + bboxes[..., 3]) * 0.5
        w, h = bboxes[..., 2] - bboxes[..., 0], bboxes[..., 3] - bboxes[..., 1]
        xcp, ycp = (pred_bboxes[..., 0] - 0.5) * stride + xc, (pred_bboxes[..., 1] - 0.5) * stride + yc
        wp, hp = torch.exp(pred_bboxes[..., 2]) * w, torch.exp(pred_bboxes[..., 3]) * h
        return torch.stack((xcp - wp / 2, ycp - hp / 2, xcp + wp / 2, ycp + hp / 2), dim=-1)


class ResBlock(nn.Module):
    def __init__(self, in_channels, **cfg):
        super().__init__()
        assert in_channels % 2 == 0
        self.conv1 = ConvModule(in_channels, in_channels // 2, 1, **cfg)
        self.conv2 = ConvModule(in_channels // 2, in_channels, 3, padding=1, **cfg)


@BACKBONES.register_module()
class Darknet(PlannedModule):
    arch_settings = ARCH

    def __init__(self, depth=53, out_indices=(3, 4, 5), frozen_stages=-1, conv_cfg=None, norm_cfg=dict(type='BN', requires_grad=True),
                 act_cfg=dict(type='LeakyReLU', negative_slope=0.1), norm_eval=True):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for darknet')
        if act_cfg.get('type') != 'LeakyReLU' or abs(act_cfg.get('negative_slope', 0.01) - SLOPE) > 1e-12:
            raise NotImplementedError('Darknet on this path uses LeakyReLU(0.1)')
        self.depth, self.out_indices, self.frozen_stages, self.norm_eval = depth, out_indices, frozen_stages, norm_eval
        self.layers, self.channels = self.arch_settings[depth]
        cfg = dict(conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.conv1 = ConvModule(3, 32, 3, padding=1, **cfg)
        self.cr_blocks = ['conv1']
        for i, n in enumerate(self.layers):
            in_c, out_c = self.channels[i]
            m = nn.Sequential()
            m.add_module('conv', ConvModule(in_c, out_c, 3, stride=2, padding=1, **cfg))
            for r in range(n):
                m.add_module(f'res{r}', ResBlock(out_c, **cfg))
            self.add_module(f'conv_res_block{i + 1}', m)
            self.cr_blocks.append(f'conv_res_block{i + 1}')

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            from .checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
        elif pretrained is None:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    kaiming_init(m)
                elif isinstance(m, nn.BatchNorm2d):
                    constant_init(m, 1)
        else:
            raise TypeError('pretrained must be a str or None')
        self.invalidate_plans()


class DetectionBlock(nn.Module):
    def __init__(self, in_channels, out_channels, **cfg):
        super().__init__()
        d = out_channels * 2
        self.conv1 = ConvModule(in_channels, out_channels, 1, **cfg)
        self.conv2 = ConvModule(out_channels, d, 3, padding=1, **cfg)
        self.conv3 = ConvModule(d, out_channels, 1, **cfg)
        self.conv4 = ConvModule(out_channels, d, 3, padding=1, **cfg)
        self.conv5 = ConvModule(d, out_channels, 1, **cfg)


@NECKS.register_module()
class YOLOV3Neck(PlannedModule):
    def __init__(self, num_scales, in_channels, out_channels, conv_cfg=None, norm_cfg=dict(type='BN', requires_grad=True),
                 act_cfg=dict(type='LeakyReLU', negative_slope=0.1)):
        super().__init__()
        assert num_scales == len(in_channels) == len(out_channels)
        self.num_scales, self.in_channels, self.out_channels = num_scales, in_channels, out_channels
        cfg = dict(conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.detect1 = DetectionBlock(in_channels[0], out_channels[0], **cfg)
        for i in range(1, num_scales):
            in_c, out_c = in_channels[i], out_channels[i]
            self.add_module(f'conv{i}', ConvModule(in_c, out_c, 1, **cfg))
            self.add_module(f'detect{i + 1}', DetectionBlock(in_c + out_c, out_c, **cfg))

    def init_weights(self):
        pass


@HEADS.register_module()
class YOLOV3Head(PlannedModule):
    def __init__(self, num_classes, in_channels, out_channels=(1024, 512, 256),
                 anchor_generator=dict(type='YOLOAnchorGenerator',
                                       base_sizes=[[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)],
                                                   [(10, 13), (16, 30), (33, 23)]], strides=[32, 16, 8]),
                 bbox_coder=dict(type='YOLOBBoxCoder'), featmap_strides=[32, 16, 8], one_hot_smoother=0., conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), act_cfg=dict(type='LeakyReLU', negative_slope=0.1),
                 loss_cls=None, loss_conf=None, loss_xy=None, loss_wh=None, train_cfg=None, test_cfg=None):
        super().__init__()
        assert len(in_channels) == len(out_channels) == len(featmap_strides)
        self.num_classes, self.in_channels, self.out_channels = num_classes, in_channels, out_channels
        self.featmap_strides, self.train_cfg, self.test_cfg = featmap_strides, train_cfg, test_cfg
        self.one_hot_smoother = one_hot_smoother
        from .registry import BBOX_ASSIGNERS, BBOX_SAMPLERS, LOSSES, build
        bce = dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0)                  # yolo_head.py:68-82 defaults
        self.loss_cls = build(dict(loss_cls or bce), LOSSES)
        self.loss_conf = build(dict(loss_conf or bce), LOSSES)
        self.loss_xy = build(dict(loss_xy or bce), LOSSES)
        self.loss_wh = build(dict(loss_wh or dict(type='MSELoss', loss_weight=1.0)), LOSSES)
        if train_cfg:                                                                             # yolo_head.py:91-98
            self.assigner = build(dict(train_cfg['assigner']), BBOX_ASSIGNERS)
            self.sampler = build(dict(train_cfg['sampler']) if 'sampler' in train_cfg else dict(type='PseudoSampler'), BBOX_SAMPLERS)
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        assert len(self.anchor_generator.num_base_anchors) == len(featmap_strides)
        cfg = dict(conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.convs_bridge, self.convs_pred = nn.ModuleList(), nn.ModuleList()
        for i in range(self.num_levels):
            self.convs_bridge.append(ConvModule(in_channels[i], out_channels[i], 3, padding=1, **cfg))
            self.convs_pred.append(nn.Conv2d(out_channels[i], self.num_anchors * self.num_attrib, 1))

    num_levels = property(lambda self: len(self.featmap_strides))
    num_attrib = property(lambda self: 5 + self.num_classes)

    def init_weights(self):
        for m in self.convs_pred:
            normal_init(m, std=0.01)

    def loss(self, pred_maps, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        """yolo_head.py:369-417: targets from GridAssigner + YOLOBBoxCoder.encode, then per level the four terms. pred_maps: list over
        levels of (N, A*(5+C), h, w) tensors on the GPU (gradients flow back to them through torch autograd: the terms are
        element-wise torch on the device); returns dict of four per-level lists, as the reference."""
        ops._need_gpu(*pred_maps)
        from .core.anchor import images_to_levels
        num_imgs, device = len(img_metas), pred_maps[0].device
        featmap_sizes = [tuple(pm.shape[-2:]) for pm in pred_maps]
        anchors = self.anchor_generator.grid_anchors(featmap_sizes, device)
        anchor_list = [anchors for _ in range(num_imgs)]
        flag_list = [self.anchor_generator.responsible_flags(featmap_sizes, gt_bboxes[i], device) for i in range(num_imgs)]
        target_maps_list, neg_maps_list = self.get_targets(anchor_list, flag_list, gt_bboxes, gt_labels)
        out = dict(loss_cls=[], loss_conf=[], loss_xy=[], loss_wh=[])
        for pm, tm, nm in zip(pred_maps, target_maps_list, neg_maps_list):
            lc, lo, lxy, lwh = self.loss_single(pm, tm, nm)
            out['loss_cls'].append(lc); out['loss_conf'].append(lo); out['loss_xy'].append(lxy); out['loss_wh'].append(lwh)
        return out

    def loss_single(self, pred_map, target_map, neg_map):
        """yolo_head.py:419-462."""
        num_imgs = len(pred_map)
        pred_map = pred_map.permute(0, 2, 3, 1).reshape(num_imgs, -1, self.num_attrib)
        pos_mask = target_map[..., 4]
        pos_and_neg_mask = (neg_map.float() + pos_mask).clamp(min=0., max=1.)
        pos_mask = pos_mask.unsqueeze(dim=-1)
        loss_cls = self.loss_cls(pred_map[..., 5:], target_map[..., 5:], weight=pos_mask)
        loss_conf = self.loss_conf(pred_map[..., 4], target_map[..., 4], weight=pos_and_neg_mask)
        loss_xy = self.loss_xy(pred_map[..., :2], target_map[..., :2], weight=pos_mask)
        loss_wh = self.loss_wh(pred_map[..., 2:4], target_map[..., 2:4], weight=pos_mask)
        return loss_cls, loss_conf, loss_xy, loss_wh

    def get_targets(self, anchor_list, responsible_flag_list, gt_bboxes_list, gt_labels_list):
        """yolo_head.py:464-498."""
        from .core.anchor import images_to_levels
        num_level_anchors = [a.size(0) for a in anchor_list[0]]
        res = [self._get_targets_single(a, f, b, l) for a, f, b, l in zip(anchor_list, responsible_flag_list, gt_bboxes_list, gt_labels_list)]
        return images_to_levels([r[0] for r in res], num_level_anchors), images_to_levels([r[1] for r in res], num_level_anchors)

    def _get_targets_single(self, anchors, responsible_flags, gt_bboxes, gt_labels):
        """yolo_head.py:500-560."""
        strides = torch.cat([torch.tensor(self.featmap_strides[i], device=gt_bboxes.device).repeat(len(anchors[i])) for i in range(len(anchors))])
        concat_anchors, concat_flags = torch.cat(anchors), torch.cat(responsible_flags)
        assert len(strides) == len(concat_anchors) == len(concat_flags)
        assign_result = self.assigner.assign(concat_anchors, concat_flags, gt_bboxes)
        sr = self.sampler.sample(assign_result, concat_anchors, gt_bboxes)
        target_map = concat_anchors.new_zeros(concat_anchors.size(0), self.num_attrib)
        target_map[sr.pos_inds, :4] = self.bbox_coder.encode(sr.pos_bboxes, sr.pos_gt_bboxes, strides[sr.pos_inds])
        target_map[sr.pos_inds, 4] = 1
        one_hot = nn.functional.one_hot(gt_labels, num_classes=self.num_classes).float()
        if self.one_hot_smoother != 0:
            one_hot = one_hot * (1 - self.one_hot_smoother) + self.one_hot_smoother / self.num_classes
        target_map[sr.pos_inds, 5:] = one_hot[sr.pos_assigned_gt_inds]
        neg_map = concat_anchors.new_zeros(concat_anchors.size(0), dtype=torch.uint8)
        neg_map[sr.neg_inds] = 1
        return target_map, neg_map

    def decode_level(self, pred_nhwc, level):
        """yolo_head.py:262-279 for one level on llv_yolo_decode: raw prediction map (n, h, w, A*(5+C)) fp32 NHWC on the GPU ->
        boxes (n, h*w*A, 4) decoded against the level's anchor grid (YOLOBBoxCoder.decode) and objectness (n, h*w*A) = sigmoid(attr 4)."""
        ops._need_gpu(pred_nhwc)
        n, h, w, c = pred_nhwc.shape
        assert c >= self.num_anchors * self.num_attrib and pred_nhwc.dtype == torch.float32 and pred_nhwc.is_contiguous()
        dev = pred_nhwc.device
        bx = torch.empty(n, h * w * self.num_anchors, 4, device=dev)
        cf = torch.empty(n, h * w * self.num_anchors, device=dev)
        wh = np.asarray(self.anchor_generator.base_sizes[level], dtype=np.float32).reshape(-1)
        whc = (C.c_float * len(wh))(*wh.tolist())
        check(lib().llv_yolo_decode(C.c_void_p(pred_nhwc.data_ptr()), n, h, w, self.num_anchors, self.num_attrib, c, self.featmap_strides[level],
                                    whc, C.c_void_p(bx.data_ptr()), C.c_void_p(cf.data_ptr()),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'llv_yolo_decode')
        return bx, cf

    def get_bboxes(self, pred_maps, img_metas, cfg=None, rescale=False, with_nms=True):
        """YOLOV3Head.get_bboxes / _get_bboxes (yolo_head.py:172-366) on raw prediction maps: list over levels of (N, A*(5+C), h, w)
        tensors on the GPU -> per image (dets (k, 5), labels (k,)). llv_yolo_decode per level, then the same selection + NMS the
        launch plan runs (YoloPlan.results)."""
        if not with_nms:
            raise LlvError('YOLOV3Head.get_bboxes: with_nms=False (ONNX export path) is outside the hot path')
        ops._need_gpu(*pred_maps)
        bs, dev = pred_maps[0].shape[0], pred_maps[0].device
        pred, boxes, confs = [], [], []
        for i, pm in enumerate(pred_maps):
            assert pm.shape[1] == self.num_anchors * self.num_attrib
            p = pm.detach().float().permute(0, 2, 3, 1).contiguous()
            bx, cf = self.decode_level(p, i)
            pred.append(p); boxes.append(bx); confs.append(cf)
        return _yolo_select_nms(self, pred, boxes, confs, self.num_anchors, self.num_attrib, bs, img_metas, rescale, self, cfg)


# ------------------------------------------------------------------------------------------------
# the HIP plan
# ------------------------------------------------------------------------------------------------
def _cm(b, sd, name, p, x, stride=1, pad=0, res=None, out=None, y_ld=None):
    """ConvModule = conv (no bias) + BN (folded) + LeakyReLU(0.1); with res: x + act(.) (ResBlock)."""
    pc = ops.pack_conv(sd[p + '.conv.weight'], None, stride, pad, bn=engine._bn(sd, p + '.bn'), eps=engine.BN_EPS)
    flags = CONV_LEAKY | ((CONV_RES_ADD | CONV_ACT_FIRST) if res is not None else 0)
    if out is None:
        xs = [x]
        n, h, w, cx = x.shape
        ho, wo = ops.conv_out_size(h, w, pc)
        out = b.buf(n, ho, wo, pc.cout)
        b.conv_desc(name, pc, xs, out, n, [(h, w, ho, wo, 0, 0)], x.stride(2), pc.cout, res, res.stride(2) if res is not None else 0,
                    flags, act_slope=SLOPE)
        return out
    n, h, w, _ = x.shape
    ho, wo = ops.conv_out_size(h, w, pc)
    b.conv_desc(name, pc, [x], out, n, [(h, w, ho, wo, 0, 0)], x.stride(2), y_ld, res, res.stride(2) if res is not None else 0, flags,
                act_slope=SLOPE)
    return out


def _concat_bwd(g, op):
    """Backward of the neck's torch.cat((upsampled, backbone stage), 1) (yolo_neck.py:123-124) in the training graph: the upsampled part
    goes back through the 2x nearest upsampling (sum of the four fine pixels), the stage's part is added to the stage's gradient."""
    from . import backward as B
    tmp, xf = op['xs']
    cat = op['y']
    gcat = g.gview(cat)
    oc = tmp.shape[-1]
    if train_engine_sptr(tmp) in g.req:
        B.upsample2x_bwd(gcat[..., :oc], g.gview(tmp))
    if train_engine_sptr(xf) in g.req:
        gx = g.gview(xf)
        n, h, w, c = xf.shape
        check(lib().llv_add_f16(gx.data_ptr(), gcat[..., oc:].data_ptr(), gx.data_ptr(), n * h * w, c, gx.stride(-2), gcat.stride(-2), gx.stride(-2),
                                C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'llv_add_f16')


def train_engine_sptr(t):
    return t.untyped_storage().data_ptr()


def build_yolo_body(b, sd, bb, neck, head, x_in, train=False):
    """Darknet -> YOLOV3Neck -> head convs on the builder; returns (per-level fp32 NHWC prediction maps, their (h, w)).
    Inference (train=False): the backbone stages the neck concatenates write straight into the concat buffer's upper channels and the
    upsampling into its lower ones. train=True: every conv writes a buffer of its own and the concatenation is an explicit pair of
    launches with a CUSTOM tape entry (the backward pass wants one gradient buffer per activation: train_engine.TrainGraph)."""
    layers, channels = ARCH[bb.depth]
    ns = neck.num_scales
    out_idx = list(bb.out_indices)
    assert len(out_idx) == ns
    feats = {}
    x = _cm(b, sd, 'backbone.conv1', 'backbone.conv1', x_in, 1, 1)
    for i, nres in enumerate(layers):
        q = f'backbone.conv_res_block{i + 1}'
        x = _cm(b, sd, q + '.conv', q + '.conv', x, 2, 1)
        for r in range(nres):
            y = _cm(b, sd, f'{q}.res{r}.conv1', f'{q}.res{r}.conv1', x, 1, 0)
            last = r == nres - 1 and (i + 1) in out_idx and out_idx.index(i + 1) < ns - 1
            if last and not train:
                # concat buffer of the neck level that consumes this stage: [upsampled out_c | this stage's in_c]
                lvl = ns - 1 - out_idx.index(i + 1)          # feats are reversed inside the neck
                oc = neck.out_channels[lvl]
                n_, h_, w_, c_ = x.shape
                cat = b.buf(n_, h_, w_, oc + c_)
                dst = cat[..., oc:]
                x = _cm(b, sd, f'{q}.res{r}.conv2', f'{q}.res{r}.conv2', y, 1, 1, res=x, out=dst, y_ld=oc + c_)
                feats[i + 1] = (x, cat)
            else:
                x = _cm(b, sd, f'{q}.res{r}.conv2', f'{q}.res{r}.conv2', y, 1, 1, res=x)
        if (i + 1) in out_idx and (i + 1) not in feats:
            feats[i + 1] = (x, None)

    # neck (yolo_neck.py:110-131)
    def detect(xin, p, tag):
        t = _cm(b, sd, tag + '.conv1', p + 'conv1', xin, 1, 0)
        t = _cm(b, sd, tag + '.conv2', p + 'conv2', t, 1, 1)
        t = _cm(b, sd, tag + '.conv3', p + 'conv3', t, 1, 0)
        t = _cm(b, sd, tag + '.conv4', p + 'conv4', t, 1, 1)
        return _cm(b, sd, tag + '.conv5', p + 'conv5', t, 1, 0)
    ordered = [feats[k] for k in out_idx]
    out = detect(ordered[-1][0], 'neck.detect1.', 'neck.detect1')
    outs = [out]
    for i, (xf, cat) in enumerate(reversed(ordered[:-1])):
        tmp = _cm(b, sd, f'neck.conv{i + 1}', f'neck.conv{i + 1}', out, 1, 0)
        n_, h_, w_, c_ = tmp.shape
        if cat is None:                          # training graph: the concat buffer is filled by two launches of its own
            cx = xf.shape[-1]
            cat = b.buf(n_, 2 * h_, 2 * w_, c_ + cx)
            zero = b.buf(*xf.shape)
            b.plan.add(f'neck.cat{i + 1}', lib().llv_add_f16,
                       (C.c_void_p(xf.data_ptr()), C.c_void_p(zero.data_ptr()), C.c_void_p(cat[..., c_:].data_ptr()), xf.numel() // cx, cx, cx, cx,
                        c_ + cx), 0.0, 4.0 * xf.numel())
        b.plan.add(f'neck.upsample{i + 1}', lib().llv_upsample2x_nhwc,
                   (C.c_void_p(tmp.data_ptr()), C.c_void_p(cat.data_ptr()), n_, h_, w_, c_, c_, cat.shape[-1]), 0.0,
                   2.0 * tmp.numel() * 5)
        b.plan.keep += [tmp, cat]
        if train:
            b.tape.append(dict(name=f'neck.cat{i + 1}', custom=True, xs=[tmp, xf], y=cat, bwd=_concat_bwd))
        out = detect(cat, f'neck.detect{i + 2}.', f'neck.detect{i + 2}')
        outs.append(out)
    # head (yolo_head.py:150-170): bridge 3x3 + 1x1 prediction (fp32 logits)
    pred, level_hw = [], []
    for i, f in enumerate(outs):
        t = _cm(b, sd, f'head.bridge{i}', f'bbox_head.convs_bridge.{i}', f, 1, 1)
        pc = ops.pack_conv(sd[f'bbox_head.convs_pred.{i}.weight'], sd[f'bbox_head.convs_pred.{i}.bias'], 1, 0)
        pred.append(b.conv(f'head.pred{i}', t, pc, CONV_OUT_F32))
        level_hw.append((t.shape[1], t.shape[2]))
    return pred, level_hw


class YoloPlan:
    """Whole YOLOv3 inference plan for a fixed (batch, H, W): Darknet-53 -> neck -> head convs -> decode; results() runs
    the selection + NMS."""

    def __init__(self, det, batch, height, width, device):
        bb, neck, head = det.backbone, det.neck, det.bbox_head
        if height % 32 or width % 32:
            raise LlvError('YOLOv3 input must be a multiple of 32 (Pad(size_divisor=32))')
        b = engine.Builder(device)
        self.builder, self.plan, self.batch, self.head, self.device = b, b.plan, batch, head, device
        sd = engine._cpu_sd(det.state_dict())
        self.x_in = b.buf(batch, height, width, 8)
        self.pred, self.level_hw = build_yolo_body(b, sd, bb, neck, head, self.x_in)
        self.A, self.na = head.num_anchors, head.num_attrib
        # decode every anchor of every level (llv_yolo_decode)
        self.boxes, self.conf, self._wh = [], [], []
        for i, pm in enumerate(self.pred):
            h, w = self.level_hw[i]
            nanch = h * w * self.A
            bx = torch.zeros(batch, nanch, 4, device=device)
            cf = torch.zeros(batch, nanch, device=device)
            wh = np.asarray(head.anchor_generator.base_sizes[i], dtype=np.float32).reshape(-1)
            whc = (C.c_float * len(wh))(*wh.tolist())
            self._wh.append(whc)
            st = head.featmap_strides[i]
            b.plan.add(f'decode{i}', lib().llv_yolo_decode, (C.c_void_p(pm.data_ptr()), batch, h, w, self.A, self.na, pm.shape[-1], st,
                                                            whc, C.c_void_p(bx.data_ptr()), C.c_void_p(cf.data_ptr())), 0.0,
                       4.0 * pm.numel())
            self.boxes.append(bx); self.conf.append(cf)
        self._nms = None

    def results(self, img_metas=None, rescale=False):
        """yolo_head.py:281-352 after the per-anchor decode: per-level top-k on objectness, gather, conf_thr, NMS."""
        return _yolo_select_nms(self.head, self.pred, self.boxes, self.conf, self.A, self.na, self.batch, img_metas, rescale, self)


def _yolo_select_nms(head, pred, boxes_l, conf_l, A, na, bs, img_metas, rescale, cache, cfg=None):
    """The part of YOLOV3Head._get_bboxes behind the per-anchor decode (yolo_head.py:281-352): per-level top-k on objectness,
    gather, conf_thr, multiclass NMS with the objectness as score_factors. pred: NHWC fp32 maps (bs, h, w, >= A*na) per level;
    boxes_l / conf_l: llv_yolo_decode outputs per level. `cache` keeps the NMS buffers between calls."""
    cfg = head.test_cfg if cfg is None else cfg
    nc = head.num_classes
    device = pred[0].device
    nms_pre = cfg.get('nms_pre', -1)
    thr = cfg.get('conf_thr', -1)
    # per level: llv_yolo_select (radix-select top-k on objectness, gather, sigmoid of the class logits, conf_thr, rescale) straight into
    # the concatenated candidate lists; buffers kept between calls
    ns = [int(c.shape[1]) for c in conf_l]
    ms = [min(nms_pre, n_) if nms_pre > 0 else n_ for n_ in ns]
    n_total = sum(ms)
    key = (bs, tuple(ns), tuple(ms), nc)
    sel = getattr(cache, '_sel', None)
    if sel is None or sel[0] != key:
        wsb = max(int(lib().llv_yolo_select_workspace_bytes(bs, n_)) for n_ in ns)
        sel = (key, torch.zeros(bs, n_total, 4, device=device), torch.zeros(bs, n_total, device=device), torch.zeros(bs, n_total, nc, device=device),
               torch.empty(max(wsb, 256), dtype=torch.uint8, device=device), torch.ones(bs, 4, device=device))
        cache._sel = sel
    _, boxes, confs, scores, sws, inv = sel
    use_sf = bool(rescale and img_metas is not None)
    if use_sf:
        inv.copy_(torch.tensor(np.stack([np.asarray(m['scale_factor'], dtype=np.float32).reshape(-1)[:4] * np.ones(4, np.float32) for m in img_metas])))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    row0 = 0
    for i, pm in enumerate(pred):
        check(lib().llv_yolo_select(pm.data_ptr(), boxes_l[i].data_ptr(), conf_l[i].data_ptr(), bs, ns[i], A, na, nc, pm.shape[-1], int(nms_pre),
                                    float(thr), inv.data_ptr() if use_sf else None, row0, n_total, boxes.data_ptr(), confs.data_ptr(), scores.data_ptr(),
                                    sws.data_ptr(), sws.numel(), stream), 'llv_yolo_select')
        row0 += ms[i]
    n = boxes.shape[1]
    nms_cfg = cfg['nms']
    iou_thr = nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr'))
    max_num = cfg['max_per_img']
    if getattr(cache, '_nms', None) is None or cache._nms[0] != (n, bs, max_num):
        wsb = int(lib().llv_multiclass_nms_batched_workspace_bytes(bs, n, nc, max_num))
        cache._nms = ((n, bs, max_num), torch.empty(max(wsb, 256), dtype=torch.uint8, device=device),
                      torch.zeros(bs, max_num, 5, device=device), torch.zeros(bs, max_num, dtype=torch.int64, device=device),
                      torch.zeros(bs, dtype=torch.int32, device=device), torch.zeros(bs, max_num, dtype=torch.int32, device=device))
    _, ws, dets, labels, count, keep = cache._nms
    check(lib().llv_multiclass_nms_batched_sf(boxes.data_ptr(), scores.data_ptr(), confs.data_ptr(), bs, n, nc, nc, cfg['score_thr'],
                                              iou_thr, max_num, dets.data_ptr(), labels.data_ptr(), count.data_ptr(),
                                              keep.data_ptr(), ws.data_ptr(), ws.numel(), stream), 'llv_multiclass_nms_batched_sf')
    cnt = count.cpu().tolist()
    return [(dets[i, :cnt[i]].clone(), labels[i, :cnt[i]].clone()) for i in range(bs)]


@DETECTORS.register_module()
class YOLOV3(SingleStageDetector):
    """mmdetection/mmdet/models/detectors/yolo.py"""

    def __init__(self, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__(backbone, neck, bbox_head, train_cfg, test_cfg, pretrained)

    def make_plan(self, batch, height, width, device, x_in=None):
        return YoloPlan(self, batch, height, width, device)

    def simple_test(self, img, img_metas, rescale=False):
        self._check_input(img)
        plan = self._get_plan(img)
        ops.pack_input(img.float().contiguous(), 8, out=plan.x_in)
        plan.plan.run()
        return [bbox2result(d, l, self.bbox_head.num_classes) for d, l in plan.results(img_metas, rescale)]

    def extract_feat(self, img):
        raise LlvError('YOLOV3.extract_feat: use simple_test (the plan keeps the features on the device)')

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        """single_stage.py:68-96 for YOLOv3: Darknet, YOLOV3Neck and the head convs as one training graph on the device
        (yolo_train.YoloTrainGraph), YOLOV3Head.loss on its prediction maps."""
        from .yolo_train import yolo_forward_train
        return yolo_forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore)
