"""RoI side of the SELSA two-stage video detector at test time (third piece), with the reference's
constructors and parameter names:
  SelsaAggregator   mmtracking/mmtrack/models/aggregators/selsa_aggregator.py:8-73
  SelsaBBoxHead     mmtracking/mmtrack/models/roi_heads/bbox_heads/selsa_bbox_head.py:8-84 (over mmdet ConvFCBBoxHead / BBoxHead.get_bboxes,
                    mmdetection/mmdet/models/roi_heads/bbox_heads/bbox_head.py:270-371)
  SelsaRoIHead      mmtracking/mmtrack/models/roi_heads/selsa_roi_head.py:61-78,113-187 (simple_test; SingleRoIExtractor or TemporalRoIAlign)
  TwoStageDetector / FasterRCNN, SELSA (mmtracking/mmtrack/models/vid/selsa.py) and the authors' SelsaFastDVDnetDetect
                    (mmtracking/mmtrack/models/vid/selsa_fastdvd_darkfarm.py:294-319,361-453)

What is native here: RoI features through `llv_roi_align`, detections through `llv_multiclass_nms_batched_ex` with class-specific
boxes. The fully connected layers and the attention of the aggregator are plain GEMMs and run as torch (rocBLAS / hipBLASLt) calls on
the device (fp16 operands, fp32 accumulation: `_lin`); the element-wise decode is torch on the device. No CPU path: CPU inputs raise. Training (`forward_train`, sampling,
losses): two_stage_train.py."""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import CONV_OUT_F32, LlvError, check, lib
from .core.bbox import bbox2result
from .registry import AGGREGATORS, HEADS, ROI_EXTRACTORS, build, build_aggregator
from .roi_heads import BBoxHeadMixin, _gpu, _lin, _wants_autograd, bbox2roi, linear_pc  # noqa: F401  (re-exported: selsa_plan, tests)


def selsa_attention(x_embed, ref_embed, ref_new, num_blocks):
    """softmax(x_embed_b . ref_embed_b^T / sqrt(d)) . ref_new_b for every attention block b (selsa_aggregator.py:44-71) on the
    convolution engine: two grouped 1x1 convolutions whose weights are the packed reference rows (llv_selsa_pack_kv) around
    llv_selsa_softmax. x_embed (N, C), ref_embed / ref_new (M, C) -> (N, C) fp16."""
    L = lib()
    n, c = x_embed.shape
    m = ref_embed.shape[0]
    nb, d = num_blocks, c // num_blocks
    if c % num_blocks or d % 8:
        raise LlvError(f'SelsaAggregator: {c} channels in {num_blocks} attention blocks (block width must be a multiple of 8)')
    dev = x_embed.device
    m_pad = ops.round_up(m, 32)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    xe = x_embed.to(torch.float16).contiguous()
    re, rn = ref_embed.to(torch.float16).contiguous(), ref_new.to(torch.float16).contiguous()
    kp = ops.round_up(d, 32)
    kw = torch.zeros((nb, m_pad, kp), device=dev, dtype=torch.float16)
    vw = torch.zeros((nb, ops.round_up(d, 16), m_pad), device=dev, dtype=torch.float16)
    check(L.llv_selsa_pack_kv(re.data_ptr(), kw.data_ptr(), m, 0, nb, d, c, m_pad, kp, 0, st), 'llv_selsa_pack_kv')
    check(L.llv_selsa_pack_kv(rn.data_ptr(), vw.data_ptr(), m, 0, nb, d, c, vw.shape[1], m_pad, 1, st), 'llv_selsa_pack_kv')
    pck = ops.PackedConv(w=kw, bias=torch.zeros(nb * m_pad, device=dev), cin=d, cout=m_pad, w_rows=m_pad, k_pad=kp,
                         kh=1, kw=1, stride=1, pad=0, groups=nb, true_macs_per_pixel=m * c)
    pcv = ops.PackedConv(w=vw, bias=torch.zeros(nb * d, device=dev), cin=m_pad, cout=d, w_rows=vw.shape[1], k_pad=m_pad, kh=1, kw=1, stride=1,
                         pad=0, groups=nb, true_macs_per_pixel=m * c)
    s = ops.conv2d(xe.view(1, n, 1, c), pck, flags=CONV_OUT_F32)                        # (1, n, 1, nb * m_pad) fp32
    p = torch.empty((1, n, 1, nb * m_pad), device=dev, dtype=torch.float16)
    check(L.llv_selsa_softmax(s.data_ptr(), p.data_ptr(), n, nb, m_pad, m, 0, None, 1.0 / d ** 0.5, st), 'llv_selsa_softmax')
    return ops.conv2d(p, pcv).view(n, c)


@AGGREGATORS.register_module()
class SelsaAggregator(nn.Module):
    def __init__(self, in_channels, num_attention_blocks=16):
        super().__init__()
        self.fc_embed = nn.Linear(in_channels, in_channels)
        self.ref_fc_embed = nn.Linear(in_channels, in_channels)
        self.fc = nn.Linear(in_channels, in_channels)
        self.ref_fc = nn.Linear(in_channels, in_channels)
        self.num_attention_blocks = num_attention_blocks

    def ref_side(self, ref_x):
        """The two projections of the reference features, (ref_fc_embed(ref_x), ref_fc(ref_x)), each [M, C]: they do not depend on the
        key frame, so a caller may keep them for reference frames that stay in the memo (SelsaBBoxHead.ref_branch)."""
        return _lin(self.ref_fc_embed, ref_x), _lin(self.ref_fc, ref_x)

    def forward_autograd(self, x, ref_x):
        """The same arithmetic with torch's differentiable ops (fp32 library GEMMs on the device): the training path, where autograd
        has to reach the four fully connected layers and both inputs (selsa_aggregator.py:29-73 as written)."""
        n, m, nb = x.shape[0], ref_x.shape[0], self.num_attention_blocks
        x_embed = self.fc_embed(x).view(n, nb, -1).permute(1, 0, 2)
        ref_embed = self.ref_fc_embed(ref_x).view(m, nb, -1).permute(1, 2, 0)
        weights = (torch.bmm(x_embed, ref_embed) / (x_embed.shape[-1] ** 0.5)).softmax(dim=2)
        ref_new = self.ref_fc(ref_x).view(m, nb, -1).permute(1, 0, 2)
        x_new = torch.bmm(weights, ref_new).permute(1, 0, 2).contiguous()
        return self.fc(x_new.view(n, -1))

    def forward(self, x, ref_x, ref_side=None):
        """x [N, C] key-frame proposal features, ref_x [M, C] reference ones -> [N, C] (selsa_aggregator.py:29-73)."""
        _gpu(x, ref_x)
        if torch.is_grad_enabled() and ref_side is None and (x.requires_grad or ref_x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self.forward_autograd(x, ref_x)
        n, m, nb = x.shape[0], ref_x.shape[0], self.num_attention_blocks
        r_embed, r_new = self.ref_side(ref_x) if ref_side is None else ref_side
        x_new = selsa_attention(_lin(self.fc_embed, x), r_embed, r_new, nb)
        return _lin(self.fc, x_new)


@HEADS.register_module()
class SelsaBBoxHead(BBoxHeadMixin, nn.Module):
    """ConvFCBBoxHead restricted to what the SELSA configs use: `num_shared_fcs` fully connected layers (no convs), one
    SelsaAggregator after each, fc_cls (num_classes + 1) and fc_reg (4 or 4 * num_classes)."""

    def __init__(self, aggregator, num_shared_convs=0, num_shared_fcs=2, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0, num_reg_fcs=0,
                 conv_out_channels=256, fc_out_channels=1024, with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7,
                 in_channels=256, num_classes=80, bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True, target_means=[0., 0., 0., 0.],
                                                                  target_stds=[0.1, 0.1, 0.2, 0.2]),
                 reg_class_agnostic=False, reg_decoded_bbox=False, loss_cls=None, loss_bbox=None, **kwargs):
        super().__init__()
        if num_shared_convs or num_cls_convs or num_cls_fcs or num_reg_convs or num_reg_fcs or with_avg_pool or not (with_cls and with_reg):
            raise NotImplementedError('SelsaBBoxHead: only the shared-fc form of the SELSA configs is built')
        self.num_shared_fcs, self.num_classes, self.reg_class_agnostic = num_shared_fcs, num_classes, reg_class_agnostic
        self.in_channels, self.fc_out_channels = in_channels, fc_out_channels
        self.roi_feat_size = (roi_feat_size, roi_feat_size) if isinstance(roi_feat_size, int) else tuple(roi_feat_size)
        self._build_losses(bbox_coder, loss_cls, loss_bbox, reg_decoded_bbox)
        last = in_channels * self.roi_feat_size[0] * self.roi_feat_size[1]
        self.shared_fcs = nn.ModuleList()
        for i in range(num_shared_fcs):
            self.shared_fcs.append(nn.Linear(last if i == 0 else fc_out_channels, fc_out_channels))
        self.fc_cls = nn.Linear(fc_out_channels, num_classes + 1)
        self.fc_reg = nn.Linear(fc_out_channels, 4 if reg_class_agnostic else 4 * num_classes)
        self.aggregator = nn.ModuleList([build_aggregator(aggregator) for _ in range(num_shared_fcs)])

    def init_weights(self):                      # bbox_head.py:93-101, convfc_bbox_head.py:119-127
        nn.init.normal_(self.fc_cls.weight, 0, 0.01); nn.init.constant_(self.fc_cls.bias, 0)
        nn.init.normal_(self.fc_reg.weight, 0, 0.001); nn.init.constant_(self.fc_reg.bias, 0)
        for m in self.shared_fcs:
            nn.init.xavier_uniform_(m.weight); nn.init.constant_(m.bias, 0)

    def ref_branch(self, ref_x):
        """Everything the head computes from the reference RoI features alone (selsa_bbox_head.py:60-84: the reference side of the loop
        never sees the key frame): per shared fc the pre-activation features [M, fc_out] and the aggregator's two projections of them.
        Rows are independent, so branches of disjoint row sets concatenate (`cat_branches`)."""
        _gpu(ref_x)
        ref_x = ref_x.flatten(1)
        out = []
        for i, fc in enumerate(self.shared_fcs):
            ref_x = _lin(fc, ref_x)
            out.append((ref_x,) + tuple(self.aggregator[i].ref_side(ref_x)))
            ref_x = F.relu(ref_x)
        return out

    @staticmethod
    def cat_branches(a, b):
        return [tuple(torch.cat((u, v), 0) for u, v in zip(la, lb)) for la, lb in zip(a, b)]

    def forward_autograd(self, x, ref_x):
        """selsa_bbox_head.py:60-84 with torch's differentiable ops (training): both RoI feature sets carry gradient back to the
        feature maps through RoIAlign's backward."""
        x, ref_x = x.flatten(1), ref_x.flatten(1)
        for i, fc in enumerate(self.shared_fcs):
            x = fc(x)
            ref_x = fc(ref_x)
            x = x + self.aggregator[i].forward_autograd(x, ref_x)
            ref_x = F.relu(ref_x)
            x = F.relu(x)
        return self.fc_cls(x), self.fc_reg(x)

    def forward(self, x, ref_x, ref_branch=None):
        """x [N, C, H, W] key-frame RoI features, ref_x [M, C, H, W] (or None with a precomputed `ref_branch` over the same M rows)
        -> (cls_score [N, classes+1], bbox_pred [N, 4*classes])."""
        _gpu(x, ref_x)
        if ref_branch is None and ref_x is not None and _wants_autograd(self, x, ref_x):
            return self.forward_autograd(x, ref_x)
        if ref_branch is None:
            ref_branch = self.ref_branch(ref_x)
        x = x.flatten(1)
        for i, fc in enumerate(self.shared_fcs):
            ref_pre, r_embed, r_new = ref_branch[i]
            x = _lin(fc, x)
            x = F.relu(x + self.aggregator[i](x, ref_pre, ref_side=(r_embed, r_new)))
        return _lin(self.fc_cls, x), _lin(self.fc_reg, x)


@HEADS.register_module()
class SelsaRoIHead(nn.Module):
    def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None, shared_head=None, train_cfg=None,
                 test_cfg=None):
        super().__init__()
        if mask_head is not None or shared_head is not None:
            raise NotImplementedError('SelsaRoIHead: mask / shared heads are not on this path')
        if bbox_roi_extractor.get('type') not in ('SingleRoIExtractor', 'TemporalRoIAlign'):
            raise NotImplementedError(f"bbox_roi_extractor {bbox_roi_extractor.get('type')}: SingleRoIExtractor and TemporalRoIAlign are built")
        self.bbox_roi_extractor = build(bbox_roi_extractor, ROI_EXTRACTORS)
        self.bbox_head = build(bbox_head, HEADS)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.bbox_assigner = self.bbox_sampler = None
        if train_cfg:                                    # base_roi_head.py init_assigner_sampler (standard_roi_head.py:18-25)
            from .registry import build_assigner, build_sampler
            self.bbox_assigner = build_assigner(train_cfg['assigner'])
            self.bbox_sampler = build_sampler(train_cfg['sampler'], context=self)

    with_bbox, with_mask = True, False

    def init_weights(self):
        self.bbox_head.init_weights()

    def _bbox_forward(self, x, ref_x, rois, ref_rois, static_refs=0, static_rows=0, cache=None):
        """selsa_roi_head.py:61-78. `static_refs` / `cache` (SELSA.simple_test, memo mode): the first `static_refs` images of ref_x and
        their RoIs are the memo's reference frames, unchanged for the whole video; their RoI features and the head's reference branch
        are computed once per video and kept in `cache` -- same kernels on the same inputs as recomputing them per key frame."""
        n_in = self.bbox_roi_extractor.num_inputs
        from .roi_extractors import TemporalRoIAlign
        if isinstance(self.bbox_roi_extractor, TemporalRoIAlign):       # selsa_roi_head.py:83-86: the key RoIs also see the reference maps
            bbox_feats = self.bbox_roi_extractor(x[:n_in], rois, ref_feats=ref_x[:n_in])
        else:
            bbox_feats = self.bbox_roi_extractor(x[:n_in], rois)
        if static_refs > 0 and cache is not None:
            # bbox2roi lists the RoIs image by image: the first static_rows rows belong to the static frames
            if 'ref_branch' not in cache:
                static_feats = self.bbox_roi_extractor(tuple(f[:static_refs] for f in ref_x[:n_in]), ref_rois[:static_rows])
                cache['ref_branch'] = self.bbox_head.ref_branch(static_feats)
            live_rois = ref_rois[static_rows:].clone()
            live_rois[:, 0] -= static_refs
            live_feats = self.bbox_roi_extractor(tuple(f[static_refs:] for f in ref_x[:n_in]), live_rois)
            branch = self.bbox_head.cat_branches(cache['ref_branch'], self.bbox_head.ref_branch(live_feats))
            cls_score, bbox_pred = self.bbox_head(bbox_feats, None, ref_branch=branch)
        else:
            ref_bbox_feats = self.bbox_roi_extractor(ref_x[:n_in], ref_rois)
            cls_score, bbox_pred = self.bbox_head(bbox_feats, ref_bbox_feats)
        return dict(cls_score=cls_score, bbox_pred=bbox_pred, bbox_feats=bbox_feats)

    def forward_train(self, x, ref_x, img_metas, proposal_list, ref_proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None):
        """selsa_roi_head.py:10-72: assign + sample the key frame's proposals, then _bbox_forward_train. x / ref_x: tuples of NCHW fp32
        feature maps that carry gradient (two_stage_train.TwoStageBody's outputs); the RoIs do not."""
        if self.bbox_assigner is None:
            raise LlvError('SelsaRoIHead.forward_train needs train_cfg (assigner, sampler, pos_weight)')
        num_imgs = len(img_metas)
        if gt_bboxes_ignore is None:
            gt_bboxes_ignore = [None for _ in range(num_imgs)]
        sampling_results = []
        for i in range(num_imgs):
            props = proposal_list[i].detach()
            assign_result = self.bbox_assigner.assign(props[:, :4].contiguous(), gt_bboxes[i], gt_bboxes_ignore[i], gt_labels[i])
            sampling_results.append(self.bbox_sampler.sample(assign_result, props, gt_bboxes[i], gt_labels[i]))
        losses = dict()
        bbox_results = self._bbox_forward_train(x, ref_x, sampling_results, ref_proposal_list, gt_bboxes, gt_labels)
        losses.update(bbox_results['loss_bbox'])
        return losses

    def _bbox_forward_train(self, x, ref_x, sampling_results, ref_proposal_list, gt_bboxes, gt_labels):
        """selsa_roi_head.py:94-111"""
        rois = bbox2roi([res.bboxes for res in sampling_results])
        ref_rois = bbox2roi([p.detach() for p in ref_proposal_list])
        bbox_results = self._bbox_forward(x, ref_x, rois, ref_rois)
        bbox_targets = self.bbox_head.get_targets(sampling_results, gt_bboxes, gt_labels, self.train_cfg)
        loss_bbox = self.bbox_head.loss(bbox_results['cls_score'], bbox_results['bbox_pred'], rois, *bbox_targets)
        bbox_results.update(loss_bbox=loss_bbox)
        return bbox_results

    def simple_test_bboxes(self, x, ref_x, proposals, ref_proposals, img_metas, rcnn_test_cfg, rescale=False, static_refs=0, cache=None):
        """selsa_roi_head.py:134-187."""
        rois, ref_rois = bbox2roi(proposals), bbox2roi(ref_proposals)
        if static_refs > 0 and cache is not None:
            res = self._bbox_forward(x, ref_x, rois, ref_rois, static_refs=static_refs,
                                     static_rows=sum(len(p) for p in ref_proposals[:static_refs]), cache=cache)
        else:
            res = self._bbox_forward(x, ref_x, rois, ref_rois)
        counts = tuple(len(p) for p in proposals)
        rois_s, cls_s, reg_s = rois.split(counts, 0), res['cls_score'].split(counts, 0), res['bbox_pred'].split(counts, 0)
        det_bboxes, det_labels = [], []
        for i in range(len(proposals)):
            d, l = self.bbox_head.get_bboxes(rois_s[i], cls_s[i], reg_s[i], img_metas[i]['img_shape'], img_metas[i]['scale_factor'],
                                             rescale=rescale, cfg=rcnn_test_cfg)
            det_bboxes.append(d); det_labels.append(l)
        return det_bboxes, det_labels

    def simple_test(self, x, ref_x, proposals_list, ref_proposals_list, img_metas, proposals=None, rescale=False, static_refs=0, cache=None):
        """selsa_roi_head.py:113-132: list per image of list[num_classes] of (k, 5) arrays."""
        det_bboxes, det_labels = self.simple_test_bboxes(x, ref_x, proposals_list, ref_proposals_list, img_metas, self.test_cfg, rescale,
                                                         static_refs, cache)
        return [bbox2result(det_bboxes[i], det_labels[i], self.bbox_head.num_classes) for i in range(len(det_bboxes))]


# ------------------------------------------------------------------------------------------------
# two-stage detector container and the SELSA video detector (test time)
# ------------------------------------------------------------------------------------------------
from .detectors import BaseDetector  # noqa: E402
from .registry import DETECTORS, MODELS, build_backbone, build_detector, build_head, build_neck  # noqa: E402
from .vid import BaseVideoDetector  # noqa: E402


@DETECTORS.register_module()
class TwoStageDetector(BaseDetector):
    """mmdetection/mmdet/models/detectors/two_stage.py:10-70 as far as the SELSA test path needs it: backbone, neck, rpn_head, roi_head
    built from the config (train_cfg / test_cfg split into the `rpn` / `rcnn` parts), `extract_feat` = backbone + neck."""

    def __init__(self, backbone, neck=None, rpn_head=None, roi_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        if rpn_head is not None:
            rh = dict(rpn_head)
            rh.update(train_cfg=train_cfg['rpn'] if train_cfg is not None else None, test_cfg=test_cfg['rpn'] if test_cfg is not None else None)
            self.rpn_head = build_head(rh)
        if roi_head is not None:
            rr = dict(roi_head)
            rr.update(train_cfg=train_cfg['rcnn'] if train_cfg is not None else None, test_cfg=test_cfg['rcnn'] if test_cfg is not None else None)
            self.roi_head = build_head(rr)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_weights(pretrained=pretrained)          # two_stage.py:48

    with_rpn = property(lambda self: hasattr(self, 'rpn_head') and self.rpn_head is not None)
    with_roi_head = property(lambda self: hasattr(self, 'roi_head') and self.roi_head is not None)

    def init_weights(self, pretrained=None):
        self.backbone.init_weights(pretrained)
        for m in (self.neck, getattr(self, 'rpn_head', None), getattr(self, 'roi_head', None)):
            if m is not None:
                m.init_weights()

    def extract_feat(self, img):
        """two_stage.py:80-95 of the reference tree: when the backbone returns MORE maps than the neck has convs (the authors'
        config: out_indices=(3, 3) with a one-conv ChannelMapper) the leading maps are handed back beside the neck's output, as
        (x[:ref_len], neck(x[ref_len:])); otherwise backbone + neck."""
        x = self.backbone(img)
        if self.neck is None:
            return x
        neck_len, backbone_len = len(self.neck.convs), len(self.backbone.out_indices)
        if neck_len != backbone_len:
            ref_len = backbone_len - neck_len
            return tuple(x[:ref_len]), self.neck(tuple(x[ref_len:]))
        return self.neck(x)

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None, proposals=None, **kwargs):
        """two_stage.py:115-176 for a StandardRoIHead (two_stage_train.faster_rcnn_forward_train). With a SELSA RoI head the detector is
        trained through the SELSA video detectors (SELSA / SelsaFastDVDnetDetect.forward_train: key + reference frames)."""
        from .two_stage_train import faster_rcnn_forward_train
        return faster_rcnn_forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_masks, proposals, **kwargs)

    def simple_test(self, img, img_metas, proposals=None, rescale=False):
        """two_stage.py:178-190: backbone + neck, RPN proposals, roi_head.simple_test -- for a StandardRoIHead. (SelsaRoIHead pools from
        reference frames as well and is driven by the SELSA video detectors.)"""
        from .roi_heads import StandardRoIHead
        if not isinstance(self.roi_head, StandardRoIHead):
            raise LlvError('the two-stage detector with a SELSA RoI head is driven by the SELSA video detector (key + reference frames)')
        self._check_input(img)
        x = self.extract_feat(img.float())
        proposal_list = self.rpn_head.simple_test_rpn(x, img_metas) if proposals is None else proposals
        return self.roi_head.simple_test(x, proposal_list, img_metas, rescale=rescale)


@DETECTORS.register_module()
class FasterRCNN(TwoStageDetector):
    """mmdetection/mmdet/models/detectors/faster_rcnn.py"""


@MODELS.register_module()
class SELSA(BaseVideoDetector):
    """Sequence-level semantics aggregation at test time (mmtracking/mmtrack/models/vid/selsa.py:14-56,172-330): the RoI features of the
    key frame's proposals attend to those of the reference frames' proposals. `extract_feats` keeps the reference-frame feature maps of
    a video in `self.memo` (both of the reference's modes: all references given with frame 0 when frame_stride < 1, or a sliding window
    refreshed every frame_stride frames), proposals come from the RPN on key and reference features, detections from SelsaRoIHead."""

    def __init__(self, detector, pretrains=None, frozen_modules=None, train_cfg=None, test_cfg=None):
        super().__init__()
        self.detector = build_detector(detector)
        assert hasattr(self.detector, 'roi_head'), 'selsa video detector only supports two stage detector'
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.memo = None
        self.init_weights(pretrains)
        if frozen_modules is not None:
            self.freeze_module(frozen_modules)

    def invalidate_plans(self):
        """Parameters may have changed (load_state_dict, train() / eval(), an optimizer step): besides the plans, the per-video memo
        holds feature maps, proposals and RoI-head reference branches computed with the old weights -- drop it, the next frame 0
        (or the next call, which then raises on the missing memo instead of mixing weights) rebuilds it."""
        super().invalidate_plans()
        self.memo = None

    def init_weights(self, pretrain):
        if pretrain is None:
            pretrain = dict()
        assert isinstance(pretrain, dict), '`pretrain` must be a dict.'
        if self.with_detector and pretrain.get('detector', False):
            self.init_module('detector', pretrain['detector'])

    cache_static_refs = True      # memo mode: keep the memo frames' proposals / RoI features / reference branch for the video (simple_test)
    use_key_plan = True           # memo mode, frames after frame 0: the whole step as one launch list / HIP graph (selsa_plan.SelsaKeyPlan)

    def _key_plan_step(self, img, img_metas, ref_img, proposals, rescale):
        """Frames 1.. of a video in memo mode (frame_stride < 1, nothing new to reference): one graph replay. Frame 0 (and every
        case the plan does not cover) goes through the module-level path below, which also fills the memo the plan is built from."""
        from . import selsa_plan
        meta = img_metas[0]
        if not (self.use_key_plan and self.cache_static_refs and proposals is None and ref_img is None and self.memo is not None
                and meta.get('frame_stride', -1) < 1 and meta.get('frame_id', -1) > 0 and img.shape[0] == 1):
            return None
        cache = self.memo.get('roi_cache') if isinstance(self.memo, dict) else None
        if not cache or 'ref_branch' not in cache or not selsa_plan.eligible(self):
            return None
        branch = cache['ref_branch']
        n, c, h, w = img.shape
        rows = tuple(int(t[1].shape[0]) for t in branch)
        shape = tuple(int(v) for v in meta['img_shape'])
        sfk = tuple(float(v) for v in np.asarray(meta.get('scale_factor', 1.0), dtype=np.float32).reshape(-1))
        key = ('key_plan', h, w, rows, shape, sfk, bool(rescale), str(img.device))
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = selsa_plan.SelsaKeyPlan(self, h, w, img.device, branch, meta, rescale)
            self.memo['key_plan_static'] = id(branch)
        elif self.memo.get('key_plan_static') != id(branch):        # a new video on a cached plan: its reference rows
            plan.set_static(branch)
            self.memo['key_plan_static'] = id(branch)
        dets, labels = plan.run(img)
        return dict(bbox_results=bbox2result(dets, labels, self.detector.roi_head.bbox_head.num_classes))

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, ref_img, ref_img_metas, ref_gt_bboxes=None, ref_gt_labels=None,
                      gt_instance_ids=None, gt_bboxes_ignore=None, gt_masks=None, proposals=None, ref_gt_instance_ids=None,
                      ref_gt_bboxes_ignore=None, ref_gt_masks=None, ref_proposals=None, **kwargs):
        """selsa.py:47-170 (two_stage_train.selsa_forward_train: every convolution on the conv engine's training graph, RoIAlign
        forward / backward kernels, the RoI head's GEMMs under autograd)."""
        from .two_stage_train import selsa_forward_train
        return selsa_forward_train(self, img, img_metas, gt_bboxes, gt_labels, ref_img, ref_img_metas, ref_gt_bboxes, ref_gt_labels,
                                   gt_bboxes_ignore=gt_bboxes_ignore, proposals=proposals, ref_proposals=ref_proposals)

    def extract_feats(self, img, img_metas, ref_img, ref_img_metas):
        """selsa.py:172-258."""
        frame_id = img_metas[0].get('frame_id', -1)
        assert frame_id >= 0
        num_left_ref_imgs = img_metas[0].get('num_left_ref_imgs', -1)
        frame_stride = img_metas[0].get('frame_stride', -1)
        if frame_stride < 1:
            if frame_id == 0:
                self.memo = dict(img_metas=list(ref_img_metas[0]), feats=list(self.detector.extract_feat(ref_img[0])))
            x = self.detector.extract_feat(img)
            ref_x = [torch.cat((r, xi), dim=0) for r, xi in zip(self.memo['feats'], x)]
            ref_metas = list(self.memo['img_metas']) + list(img_metas)
        else:
            if frame_id == 0:
                ref = self.detector.extract_feat(ref_img[0])
                self.memo = dict(img_metas=list(ref_img_metas[0]), feats=list(ref))
                x = [r[[num_left_ref_imgs]] for r in ref]
            elif frame_id % frame_stride == 0:
                assert ref_img is not None
                ref = self.detector.extract_feat(ref_img[0])
                x = []
                for i in range(len(ref)):
                    self.memo['feats'][i] = torch.cat((self.memo['feats'][i], ref[i]), dim=0)[1:]
                    x.append(self.memo['feats'][i][[num_left_ref_imgs]])
                self.memo['img_metas'] = (self.memo['img_metas'] + list(ref_img_metas[0]))[1:]
            else:
                assert ref_img is None
                x = self.detector.extract_feat(img)
            ref_x = [f.clone() for f in self.memo['feats']]
            for i in range(len(x)):
                ref_x[i][num_left_ref_imgs] = x[i][0]
            ref_metas = list(self.memo['img_metas'])
            ref_metas[num_left_ref_imgs] = img_metas[0]
        return tuple(x), img_metas, tuple(ref_x), ref_metas

    def simple_test(self, img, img_metas, ref_img=None, ref_img_metas=None, proposals=None, ref_proposals=None, rescale=False):
        """selsa.py:260-330. Returns {'bbox_results': list[num_classes] of (k, 5) arrays}."""
        self._check_input(img)
        fast = self._key_plan_step(img, img_metas, ref_img, proposals, rescale)
        if fast is not None:
            return fast
        if ref_img is not None:
            ref_img = ref_img[0]
        if ref_img_metas is not None:
            ref_img_metas = ref_img_metas[0]
        x, img_metas, ref_x, ref_img_metas = self.extract_feats(img, img_metas, ref_img, ref_img_metas)
        if proposals is None:
            proposal_list = self.detector.rpn_head.simple_test_rpn(x, img_metas)
            if img_metas[0].get('frame_stride', -1) < 1 and self.cache_static_refs:
                # ref_x = the memo's reference frames followed by the key frame (extract_feats above). The reference runs the RPN over
                # all of them for every key frame (selsa.py:312-315); the memo frames do not change during the video and the key
                # frame's proposals have just been computed, so their proposals are taken from the memo / from proposal_list: the
                # same kernels on the same inputs, computed once.
                if 'proposals' not in self.memo:
                    self.memo['proposals'] = self.detector.rpn_head.simple_test_rpn(tuple(self.memo['feats']), list(self.memo['img_metas']))
                ref_proposals_list = list(self.memo['proposals']) + list(proposal_list)
                # ... and so are those frames' RoI features and the head's reference branch over them (SelsaRoIHead._bbox_forward)
                outs = self.detector.roi_head.simple_test(x, ref_x, proposal_list, ref_proposals_list, img_metas, rescale=rescale,
                                                          static_refs=len(self.memo['proposals']), cache=self.memo.setdefault('roi_cache', {}))
                return dict(bbox_results=outs[0])
            ref_proposals_list = self.detector.rpn_head.simple_test_rpn(ref_x, ref_img_metas)
        else:
            proposal_list, ref_proposals_list = proposals, ref_proposals
        outs = self.detector.roi_head.simple_test(x, ref_x, proposal_list, ref_proposals_list, img_metas, rescale=rescale)
        return dict(bbox_results=outs[0])


@MODELS.register_module()
class SelsaFastDVDnetDetect(BaseVideoDetector):
    """The authors' own model (mmtracking/mmtrack/models/vid/selsa_fastdvd_darkfarm.py:17-45; config
    mmtracking/configs/vid/llvod/llvod_fastdvdnet_darkfarm.py): FastDVDnet in front of the two-stage SELSA detector, same constructor.

    Test time, adaptive stride (:294-319, the config's `test_with_adaptive_stride`): the four reference frames given with frame 0 stay
    in the memo for the whole video; per key frame the five frames [ref0, ref1, key, ref2, ref3] are denoised, each from the CYCLIC
    5-stack centred on it (:303-308: 20 DenBlocks in the reference; here one FastDVDnetPlan over the 9-frame cyclic sequence computes
    the 7 first-stage results it needs once: 12 DenBlocks), all five go through the detector's backbone + neck, the key frame's
    features are `x`, the other four + the key are the references; RPN on both, SelsaRoIHead (TemporalRoIAlign) on top.
    The fixed-stride branch (:321-357) is the plain SELSA memo on NOT denoised frames, as the reference has it.

    `plot_bbox` of the reference (:437-451, hard-wired True) draws the detections into ./result_fastdvdnet_out and, as a side
    effect, multiplies every returned box by img_metas[0]['scale_factor'][0]. The drawing is not reproduced (UI); the multiplication
    is, so that the returned arrays equal the reference's (`reference_box_rescale=False` turns it off)."""

    reference_box_rescale = True

    def __init__(self, detector, cleaner, pretrains=None, frozen_modules=None, train_cfg=None, test_cfg=None, loss_type=None):
        super().__init__()
        from .registry import build_cleaner
        self.detector = build_detector(detector)
        self.cleaner = build_cleaner(cleaner)
        assert hasattr(self.detector, 'roi_head'), 'selsa video detector only supports two stage detector'
        self.train_cfg, self.test_cfg, self.loss_type = train_cfg, test_cfg, loss_type
        self.memo = None
        self.init_weights(pretrains)
        if frozen_modules is not None:
            self.freeze_module(frozen_modules)

    def init_weights(self, pretrain):
        """selsa_fastdvd_darkfarm.py:47-63"""
        if pretrain is None:
            pretrain = dict()
        assert isinstance(pretrain, dict), '`pretrain` must be a dict.'
        if self.with_detector and pretrain.get('detector', False):
            self.init_module('detector', pretrain['detector'])
        if self.with_cleaner and pretrain.get('cleaner', False):
            self.init_module('cleaner', pretrain['cleaner'])

    def invalidate_plans(self):
        super().invalidate_plans()
        self.memo = None

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, ref_img, ref_img_metas, ref_gt_bboxes=None, ref_gt_labels=None,
                      gt_instance_ids=None, gt_bboxes_ignore=None, gt_masks=None, proposals=None, ref_gt_instance_ids=None,
                      ref_gt_bboxes_ignore=None, ref_gt_masks=None, ref_proposals=None, **kwargs):
        """selsa_fastdvd_darkfarm.py:65-227 (two_stage_train.selsa_fastdvd_forward_train; the cleaner is frozen)."""
        from .two_stage_train import selsa_fastdvd_forward_train
        return selsa_fastdvd_forward_train(self, img, img_metas, gt_bboxes, gt_labels, ref_img, ref_img_metas, ref_gt_bboxes, ref_gt_labels,
                                           gt_bboxes_ignore=gt_bboxes_ignore, proposals=proposals, ref_proposals=ref_proposals)

    def denoise_cyclic(self, noise_imgs):
        """noise_imgs (5, 3, H, W) fp32 on the GPU -> (5, 3, H, W): frame i denoised from the stack of frames (i-2 .. i+2) mod 5
        (selsa_fastdvd_darkfarm.py:303-308), through engine.FastDVDnetPlan on the 9-frame sequence [3, 4, 0, 1, 2, 3, 4, 0, 1]."""
        from . import engine
        n, c, h, w = noise_imgs.shape
        assert n == 5 and c == 3
        plan = self._plan(('den5', h, w, str(noise_imgs.device)), lambda: engine.FastDVDnetPlan(self.cleaner.state_dict(), 5, h, w, noise_imgs.device))
        idx = torch.tensor([3, 4, 0, 1, 2, 3, 4, 0, 1], device=noise_imgs.device)
        ops.pack_input(noise_imgs.float()[idx].contiguous(), 8, out=plan.frames)
        plan.plan.run()
        return ops.unpack_output(plan.out, 3)

    def extract_feats(self, img, img_metas, ref_img, ref_img_metas):
        """selsa_fastdvd_darkfarm.py:229-359."""
        frame_id = img_metas[0].get('frame_id', -1)
        assert frame_id >= 0
        num_left_ref_imgs = img_metas[0].get('num_left_ref_imgs', -1)
        frame_stride = img_metas[0].get('frame_stride', -1)
        if frame_stride < 1:
            if frame_id == 0:
                assert ref_img is not None and ref_img[0].shape[0] == 4, 'FastDVDnet support frames is 4.'
                self.memo = dict(img_metas=list(ref_img_metas[0]), ref_imgs=ref_img[0])
            refs = self.memo['ref_imgs']
            noise_imgs = torch.cat([refs[:2], img, refs[2:]], dim=0)
            den = self.denoise_cyclic(noise_imgs)
            _, all_x = self.detector.extract_feat(den)
            self.memo['denoising_img'] = den[[2]]
            x = [a[[2]] for a in all_x]
            ref_x = [torch.cat([a[:2], a[3:], a[[2]]], dim=0) for a in all_x]
            ref_metas = list(self.memo['img_metas']) + list(img_metas)
        else:
            if frame_id == 0:
                _, ref = self.detector.extract_feat(ref_img[0])
                self.memo = dict(img=ref_img[0], img_metas=list(ref_img_metas[0]), feats=list(ref))
                x = [r[[num_left_ref_imgs]] for r in ref]
            elif frame_id % frame_stride == 0:
                assert ref_img is not None
                ref_imgs = torch.cat([self.memo['img'], ref_img[0]], dim=0)[1:]
                _, ref = self.detector.extract_feat(ref_imgs)
                x = []
                for i in range(len(ref)):
                    self.memo['feats'][i] = ref[i]
                    x.append(ref[i][[num_left_ref_imgs]])
                self.memo['img_metas'] = (self.memo['img_metas'] + list(ref_img_metas[0]))[1:]
                self.memo['img'] = ref_imgs
            else:
                assert ref_img is None
                _, x = self.detector.extract_feat(img)       # the reference indexes ref_img (None) here and cannot run this branch
            ref_x = [f.clone() for f in self.memo['feats']]
            for i in range(len(x)):
                ref_x[i][num_left_ref_imgs] = x[i][0]
            ref_metas = list(self.memo['img_metas'])
            ref_metas[num_left_ref_imgs] = img_metas[0]
        return tuple(x), img_metas, tuple(ref_x), ref_metas

    def simple_test(self, img, img_metas, ref_img=None, ref_img_metas=None, proposals=None, ref_proposals=None, rescale=False):
        """selsa_fastdvd_darkfarm.py:361-453. Returns {'bbox_results': list[num_classes] of (k, 5) arrays}."""
        self._check_input(img)
        if ref_img is not None:
            ref_img = ref_img[0]
        if ref_img_metas is not None:
            ref_img_metas = ref_img_metas[0]
        x, img_metas, ref_x, ref_img_metas = self.extract_feats(img, img_metas, ref_img, ref_img_metas)
        if proposals is None:
            proposal_list = self.detector.rpn_head.simple_test_rpn(x, img_metas)
            ref_proposals_list = self.detector.rpn_head.simple_test_rpn(ref_x, ref_img_metas)
        else:
            proposal_list, ref_proposals_list = proposals, ref_proposals
        outs = self.detector.roi_head.simple_test(x, ref_x, proposal_list, ref_proposals_list, img_metas, rescale=rescale)
        res = outs[0]
        if self.reference_box_rescale:
            sf = float(np.asarray(img_metas[0]['scale_factor'], dtype=np.float32).reshape(-1)[0])
            res = [np.concatenate([r[:, :4] * sf, r[:, [4]]], axis=1) for r in res]
        return dict(bbox_results=res)

    def aug_test(self, imgs, img_metas, **kwargs):
        raise NotImplementedError('test-time augmentation is outside the hot path')
