"""The plain (single-frame) RoI side of the two-stage detectors, with the reference's constructors and parameter names:

  BBoxHead             mmdetection/mmdet/models/roi_heads/bbox_heads/bbox_head.py:10-371 (targets, loss, get_bboxes)
  ConvFCBBoxHead       mmdetection/mmdet/models/roi_heads/bbox_heads/convfc_bbox_head.py:9-160
  Shared2FCBBoxHead / Shared4Conv1FCBBoxHead   convfc_bbox_head.py:163-205
  StandardRoIHead      mmdetection/mmdet/models/roi_heads/standard_roi_head.py:10-270, test_mixins.py:50-170 (bbox branch)

This is the head of `mmtracking/configs/_base_/models/faster_rcnn_r50_dc5.py:38-62`, i.e. of the reference's own DFF / FGFA configs
(`configs/vid/dff/*.py`, `configs/vid/fgfa/*.py`): the two-stage branch of `DFF / FGFA / LLVOD.simple_test / forward_train`
(`dff.py:153-173,245-253`, `fgfa.py:165-185,344-352`, `llvod.py:124-145,225-233`) calls `roi_head.simple_test(x, proposal_list,
img_metas, rescale)` and `roi_head.forward_train(x, img_metas, proposal_list, gt_bboxes, gt_labels, ...)`.

Native: RoI features through `llv_roi_align` (+ `llv_roi_align_bwd`), fully connected layers at test time as 1x1 convolutions over rows
on the in-tree implicit-GEMM kernel (`linear_pc` / `_lin`), the RoI convolutions of the general ConvFC form on the conv engine,
class-specific decode `llv_delta2bbox`, detections `llv_multiclass_nms_batched_ex`. In training the head's layers are library GEMMs
under autograd on the device (as for SelsaBBoxHead, two_stage_train.py). No CPU path: CPU inputs raise."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import CONV_OUT_F32, CONV_RELU, LlvError
from .cnn import ConvModule
from .core.bbox import bbox2result
from .registry import HEADS, LOSSES, ROI_EXTRACTORS, build


def linear_pc(fc, roi_hw=None):
    """The packed weights of a fully connected layer as the convolution engine reads them: a 1x1 convolution over rows, or -- for the
    first shared fc, whose input is the flattened (C, ph, pw) RoI map -- a ph x pw "valid" convolution over the NHWC RoI features
    (the K index c*ph*pw + y*pw + x of the Linear weight is exactly conv weight [out][c][y][x]). Kept on the layer until weight or
    bias change."""
    w = fc.weight
    key = (w.data_ptr(), w._version, str(w.device), None if fc.bias is None else (fc.bias.data_ptr(), fc.bias._version), roi_hw)
    if getattr(fc, '_llv_pc_key', None) != key:
        wt = w.detach().float().cpu()
        if roi_hw is None:
            wt = wt.view(wt.shape[0], wt.shape[1], 1, 1)
        else:
            wt = wt.view(wt.shape[0], wt.shape[1] // (roi_hw[0] * roi_hw[1]), roi_hw[0], roi_hw[1])
        fc._llv_pc = ops.pack_conv(wt, None if fc.bias is None else fc.bias.detach().float().cpu(), 1, 0).to(w.device)
        fc._llv_pc_key = key
    return fc._llv_pc


def _lin(fc, x):
    """fc(x) on the in-tree implicit-GEMM kernel (fp16 operands, fp32 accumulation, fp32 out: the arithmetic type of the rest of the
    path): x (rows, K) -> (rows, out) fp32."""
    pc = linear_pc(fc)
    n, k = x.shape
    if k % 8:
        raise LlvError('fully connected layer: the input width must be a multiple of 8')
    if n == 0:
        return x.new_zeros((0, fc.out_features), dtype=torch.float32)
    x16 = x.to(torch.float16).contiguous().view(1, n, 1, k)
    y = ops.conv2d(x16, pc, flags=CONV_OUT_F32)
    return y.view(n, -1)[:, :fc.out_features]


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise LlvError('the RoI heads run on an MI355X only (RoIAlign / NMS kernels, device GEMMs); no CPU path')


def _wants_autograd(head, *xs):
    """Training call of a box head: gradients are being recorded and either an input carries one (the RoI features of a training graph)
    or the head itself is in train() mode with trainable layers. Everything else -- eval() mode, torch.no_grad() -- is the test-time
    path on the in-tree kernels."""
    return torch.is_grad_enabled() and (any(x is not None and x.requires_grad for x in xs) or
                                        (head.training and any(p.requires_grad for p in head.parameters())))


def bbox2roi(bbox_list):
    """mmdetection/mmdet/core/bbox/transforms.py:72-96."""
    rois_list = []
    for img_id, bboxes in enumerate(bbox_list):
        if bboxes.size(0) > 0:
            rois_list.append(torch.cat([bboxes.new_full((bboxes.size(0), 1), img_id), bboxes[:, :4]], dim=-1))
        else:
            rois_list.append(bboxes.new_zeros((0, 5)))
    return torch.cat(rois_list, 0)


class BBoxHeadMixin:
    """What every box head of this package shares with mmdet's BBoxHead: targets (bbox_head.py:103-207), loss (:209-268), decode + NMS
    (:270-371). Needs `num_classes, reg_class_agnostic, bbox_coder, loss_cls, loss_bbox` on the instance."""

    def _build_losses(self, bbox_coder, loss_cls, loss_bbox, reg_decoded_bbox):
        from .registry import build_bbox_coder
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.reg_decoded_bbox = reg_decoded_bbox
        if reg_decoded_bbox:
            raise NotImplementedError('bbox head: reg_decoded_bbox (IoU losses on decoded boxes) is not on this path')
        # bbox_head.py:28-36 defaults
        self.loss_cls = build(dict(loss_cls or dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)), LOSSES)
        self.loss_bbox = build(dict(loss_bbox or dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)), LOSSES)

    def _get_target_single(self, pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, cfg):
        """bbox_head.py:103-150"""
        from .training import _bbox2delta
        num_pos, num_neg = pos_bboxes.size(0), neg_bboxes.size(0)
        num_samples = num_pos + num_neg
        labels = pos_bboxes.new_full((num_samples,), self.num_classes, dtype=torch.long)
        label_weights = pos_bboxes.new_zeros(num_samples)
        bbox_targets = pos_bboxes.new_zeros(num_samples, 4)
        bbox_weights = pos_bboxes.new_zeros(num_samples, 4)
        if num_pos > 0:
            labels[:num_pos] = pos_gt_labels
            pw = cfg['pos_weight'] if isinstance(cfg, dict) else cfg.pos_weight
            label_weights[:num_pos] = 1.0 if pw <= 0 else pw
            bbox_targets[:num_pos, :] = _bbox2delta(pos_bboxes, pos_gt_bboxes, self.bbox_coder.means, self.bbox_coder.stds)
            bbox_weights[:num_pos, :] = 1
        if num_neg > 0:
            label_weights[-num_neg:] = 1.0
        return labels, label_weights, bbox_targets, bbox_weights

    def get_targets(self, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg, concat=True):
        """bbox_head.py:152-207"""
        outs = [self._get_target_single(r.pos_bboxes, r.neg_bboxes, r.pos_gt_bboxes, r.pos_gt_labels, rcnn_train_cfg) for r in sampling_results]
        cols = list(zip(*outs))
        if concat:
            return tuple(torch.cat(c, 0) for c in cols)
        return tuple(list(c) for c in cols)

    def loss(self, cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_weights, reduction_override=None):
        """bbox_head.py:209-268: softmax cross entropy over all sampled RoIs (llv_softmax_ce), box loss over the positives' own class."""
        from .losses import accuracy
        losses = dict()
        if cls_score is not None:
            avg_factor = max(torch.sum(label_weights > 0).float().item(), 1.)
            if cls_score.numel() > 0:
                losses['loss_cls'] = self.loss_cls(cls_score, labels, label_weights, avg_factor=avg_factor, reduction_override=reduction_override)
                losses['acc'] = accuracy(cls_score, labels)
        if bbox_pred is not None:
            pos_inds = (labels >= 0) & (labels < self.num_classes)
            if pos_inds.any():
                if self.reg_class_agnostic:
                    pos_bbox_pred = bbox_pred.view(bbox_pred.size(0), 4)[pos_inds]
                else:
                    pos_bbox_pred = bbox_pred.view(bbox_pred.size(0), -1, 4)[pos_inds, labels[pos_inds]]
                losses['loss_bbox'] = self.loss_bbox(pos_bbox_pred, bbox_targets[pos_inds], bbox_weights[pos_inds], avg_factor=bbox_targets.size(0),
                                                     reduction_override=reduction_override)
            else:
                losses['loss_bbox'] = bbox_pred[pos_inds].sum()
        return losses

    def _decode(self, rois, deltas, max_shape, scale_factor=None):
        """delta_xywh_bbox_coder.py:134-237 for (n, 4) rois and (n, 4k) deltas: one llv_delta2bbox launch (clip to max_shape, optional
        division by the scale factor)."""
        return ops.delta2bbox(rois, deltas, self.bbox_coder.means, self.bbox_coder.stds,
                              max_shape if self.bbox_coder.clip_border else None, 16 / 1000, self.bbox_coder.clip_border, scale_factor)

    def get_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False, cfg=None):
        """bbox_head.py:270-371 (one image): softmax, per-class delta2bbox clipped to the image, optional rescale, multiclass NMS."""
        _gpu(rois, cls_score, bbox_pred)
        scores = F.softmax(cls_score.float(), dim=-1)
        bboxes = self._decode(rois[:, 1:].float(), bbox_pred.float(), img_shape, scale_factor if rescale else None)
        if cfg is None:
            return bboxes, scores
        if bboxes.size(0) == 0:
            return bboxes.new_zeros((0, 5)), bboxes.new_zeros((0,), dtype=torch.long)
        return ops.multiclass_nms(bboxes.contiguous(), scores.contiguous(), cfg['score_thr'], cfg['nms'], cfg['max_per_img'])


def _roi_conv(conv_module, x):
    """One ConvModule (3x3, pad 1, ReLU) of a ConvFC head over RoI maps (k, C, ph, pw) on the conv engine."""
    from .rpn import _conv_nchw
    conv = conv_module.conv
    key = (conv.weight.data_ptr(), conv.weight._version, str(conv.weight.device))
    if getattr(conv_module, '_llv_pc_key', None) != key:
        conv_module._llv_pc = ops.pack_conv(conv.weight.detach().float().cpu(), None if conv.bias is None else conv.bias.detach().float().cpu(),
                                            1, conv.padding[0]).to(conv.weight.device)
        conv_module._llv_pc_key = key
    y = _conv_nchw(x, conv_module._llv_pc, CONV_RELU if conv_module.with_activation else 0)
    return y if y.shape[1] == conv.out_channels else y[:, :conv.out_channels]


@HEADS.register_module()
class BBoxHead(BBoxHeadMixin, nn.Module):
    """bbox_head.py:10-101: the simplest head, fc_cls and fc_reg on the flattened RoI map."""

    def __init__(self, with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7, in_channels=256, num_classes=80,
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True, target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2]),
                 reg_class_agnostic=False, reg_decoded_bbox=False, loss_cls=None, loss_bbox=None):
        super().__init__()
        if with_avg_pool or not (with_cls and with_reg):
            raise NotImplementedError('BBoxHead: with_avg_pool / a head without cls or reg output is not on this path')
        self.roi_feat_size = (roi_feat_size, roi_feat_size) if isinstance(roi_feat_size, int) else tuple(roi_feat_size)
        self.roi_feat_area = self.roi_feat_size[0] * self.roi_feat_size[1]
        self.in_channels, self.num_classes, self.reg_class_agnostic = in_channels, num_classes, reg_class_agnostic
        self._build_losses(bbox_coder, loss_cls, loss_bbox, reg_decoded_bbox)
        self.fc_cls = nn.Linear(in_channels * self.roi_feat_area, num_classes + 1)
        self.fc_reg = nn.Linear(in_channels * self.roi_feat_area, 4 if reg_class_agnostic else 4 * num_classes)

    def init_weights(self):
        nn.init.normal_(self.fc_cls.weight, 0, 0.01); nn.init.constant_(self.fc_cls.bias, 0)
        nn.init.normal_(self.fc_reg.weight, 0, 0.001); nn.init.constant_(self.fc_reg.bias, 0)

    def forward(self, x):
        _gpu(x)
        x = x.flatten(1)
        if _wants_autograd(self, x):
            return self.fc_cls(x), self.fc_reg(x)
        return _lin(self.fc_cls, x.float()), _lin(self.fc_reg, x.float())


@HEADS.register_module()
class ConvFCBBoxHead(BBoxHeadMixin, nn.Module):
    """shared convs -> shared fcs -> {cls convs -> cls fcs -> fc_cls, reg convs -> reg fcs -> fc_reg} (convfc_bbox_head.py:9-160).
    Parameter names as in the reference: shared_convs.{i}.conv, shared_fcs.{i}, cls_convs / cls_fcs / reg_convs / reg_fcs, fc_cls, fc_reg."""

    def __init__(self, num_shared_convs=0, num_shared_fcs=0, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0, num_reg_fcs=0,
                 conv_out_channels=256, fc_out_channels=1024, conv_cfg=None, norm_cfg=None, with_avg_pool=False, with_cls=True, with_reg=True,
                 roi_feat_size=7, in_channels=256, num_classes=80,
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True, target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2]),
                 reg_class_agnostic=False, reg_decoded_bbox=False, loss_cls=None, loss_bbox=None):
        super().__init__()
        assert num_shared_convs + num_shared_fcs + num_cls_convs + num_cls_fcs + num_reg_convs + num_reg_fcs > 0
        if num_cls_convs > 0 or num_reg_convs > 0:
            assert num_shared_fcs == 0
        if conv_cfg is not None or norm_cfg is not None:
            raise NotImplementedError('ConvFCBBoxHead with conv_cfg / norm_cfg is not on this path')
        if with_avg_pool or not (with_cls and with_reg):
            raise NotImplementedError('ConvFCBBoxHead: with_avg_pool / a head without cls or reg output is not on this path')
        self.with_avg_pool, self.with_cls, self.with_reg = with_avg_pool, with_cls, with_reg
        self.roi_feat_size = (roi_feat_size, roi_feat_size) if isinstance(roi_feat_size, int) else tuple(roi_feat_size)
        self.roi_feat_area = self.roi_feat_size[0] * self.roi_feat_size[1]
        self.in_channels, self.num_classes, self.reg_class_agnostic = in_channels, num_classes, reg_class_agnostic
        self.num_shared_convs, self.num_shared_fcs = num_shared_convs, num_shared_fcs
        self.num_cls_convs, self.num_cls_fcs, self.num_reg_convs, self.num_reg_fcs = num_cls_convs, num_cls_fcs, num_reg_convs, num_reg_fcs
        self.conv_out_channels, self.fc_out_channels = conv_out_channels, fc_out_channels
        self._build_losses(bbox_coder, loss_cls, loss_bbox, reg_decoded_bbox)
        self.shared_convs, self.shared_fcs, last = self._add_conv_fc_branch(num_shared_convs, num_shared_fcs, in_channels, True)
        self.shared_out_channels = last
        self.cls_convs, self.cls_fcs, self.cls_last_dim = self._add_conv_fc_branch(num_cls_convs, num_cls_fcs, last)
        self.reg_convs, self.reg_fcs, self.reg_last_dim = self._add_conv_fc_branch(num_reg_convs, num_reg_fcs, last)
        if num_shared_fcs == 0 and not with_avg_pool:
            if num_cls_fcs == 0:
                self.cls_last_dim *= self.roi_feat_area
            if num_reg_fcs == 0:
                self.reg_last_dim *= self.roi_feat_area
        self.fc_cls = nn.Linear(self.cls_last_dim, num_classes + 1)
        self.fc_reg = nn.Linear(self.reg_last_dim, 4 if reg_class_agnostic else 4 * num_classes)

    def _add_conv_fc_branch(self, num_convs, num_fcs, in_channels, is_shared=False):
        """convfc_bbox_head.py:82-117"""
        last = in_channels
        convs = nn.ModuleList()
        for i in range(num_convs):
            convs.append(ConvModule(last if i == 0 else self.conv_out_channels, self.conv_out_channels, 3, padding=1))
        if num_convs > 0:
            last = self.conv_out_channels
        fcs = nn.ModuleList()
        if num_fcs > 0:
            if (is_shared or self.num_shared_fcs == 0) and not self.with_avg_pool:
                last *= self.roi_feat_area
            for i in range(num_fcs):
                fcs.append(nn.Linear(last if i == 0 else self.fc_out_channels, self.fc_out_channels))
            last = self.fc_out_channels
        return convs, fcs, last

    def init_weights(self):
        """bbox_head.py:93-101, convfc_bbox_head.py:119-127"""
        nn.init.normal_(self.fc_cls.weight, 0, 0.01); nn.init.constant_(self.fc_cls.bias, 0)
        nn.init.normal_(self.fc_reg.weight, 0, 0.001); nn.init.constant_(self.fc_reg.bias, 0)
        for ml in (self.shared_fcs, self.cls_fcs, self.reg_fcs):
            for m in ml:
                nn.init.xavier_uniform_(m.weight); nn.init.constant_(m.bias, 0)

    def _branches(self, x, conv, lin):
        """convfc_bbox_head.py:129-160 with the layer calls abstracted (device kernels at test time, autograd in training)."""
        for c in self.shared_convs:
            x = conv(c, x)
        if self.num_shared_fcs > 0:
            x = x.flatten(1)
            for fc in self.shared_fcs:
                x = F.relu(lin(fc, x))
        x_cls = x_reg = x
        for c in self.cls_convs:
            x_cls = conv(c, x_cls)
        if x_cls.dim() > 2:
            x_cls = x_cls.flatten(1)
        for fc in self.cls_fcs:
            x_cls = F.relu(lin(fc, x_cls))
        for c in self.reg_convs:
            x_reg = conv(c, x_reg)
        if x_reg.dim() > 2:
            x_reg = x_reg.flatten(1)
        for fc in self.reg_fcs:
            x_reg = F.relu(lin(fc, x_reg))
        return lin(self.fc_cls, x_cls), lin(self.fc_reg, x_reg)

    def forward_autograd(self, x):
        """The same arithmetic with torch's differentiable ops on the device (training: the gradient reaches every layer and, through
        RoIAlign's backward, the feature map)."""
        def conv(c, t):
            t = F.conv2d(t, c.conv.weight, c.conv.bias, 1, c.conv.padding)
            return F.relu(t) if c.with_activation else t
        return self._branches(x, conv, lambda fc, t: fc(t))

    def forward(self, x):
        """x (k, C, ph, pw) RoI features on the GPU -> (cls_score (k, classes + 1), bbox_pred (k, 4 or 4 * classes)) fp32."""
        _gpu(x)
        if _wants_autograd(self, x):
            return self.forward_autograd(x)
        return self._branches(x.float(), _roi_conv, _lin)


@HEADS.register_module()
class Shared2FCBBoxHead(ConvFCBBoxHead):
    """convfc_bbox_head.py:163-176"""

    def __init__(self, fc_out_channels=1024, *args, **kwargs):
        super().__init__(num_shared_convs=0, num_shared_fcs=2, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0, num_reg_fcs=0,
                         fc_out_channels=fc_out_channels, *args, **kwargs)


@HEADS.register_module()
class Shared4Conv1FCBBoxHead(ConvFCBBoxHead):
    """convfc_bbox_head.py:179-192"""

    def __init__(self, fc_out_channels=1024, *args, **kwargs):
        super().__init__(num_shared_convs=4, num_shared_fcs=1, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0, num_reg_fcs=0,
                         fc_out_channels=fc_out_channels, *args, **kwargs)


@HEADS.register_module()
class StandardRoIHead(nn.Module):
    """standard_roi_head.py:10-270, bbox branch (the reference's video configs have no mask head)."""

    def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None, shared_head=None, train_cfg=None,
                 test_cfg=None):
        super().__init__()
        if mask_head is not None or shared_head is not None:
            raise NotImplementedError('StandardRoIHead: mask / shared heads are not on this path')
        self.bbox_roi_extractor = build(bbox_roi_extractor, ROI_EXTRACTORS)
        self.bbox_head = build(bbox_head, HEADS)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.bbox_assigner = self.bbox_sampler = None
        if train_cfg:                                    # standard_roi_head.py:18-25 init_assigner_sampler
            from .registry import build_assigner, build_sampler
            self.bbox_assigner = build_assigner(train_cfg['assigner'])
            self.bbox_sampler = build_sampler(train_cfg['sampler'], context=self)

    with_bbox, with_mask, with_shared_head = True, False, False

    def init_weights(self, pretrained=None):
        self.bbox_roi_extractor.init_weights()
        self.bbox_head.init_weights()

    def _bbox_forward(self, x, rois):
        """standard_roi_head.py:112-124"""
        n_in = self.bbox_roi_extractor.num_inputs
        bbox_feats = self.bbox_roi_extractor(x[:n_in], rois)
        cls_score, bbox_pred = self.bbox_head(bbox_feats)
        return dict(cls_score=cls_score, bbox_pred=bbox_pred, bbox_feats=bbox_feats)

    def forward_dummy(self, x, proposals):
        """standard_roi_head.py:50-68"""
        res = self._bbox_forward(x, bbox2roi([proposals]))
        return (res['cls_score'], res['bbox_pred'])

    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None, **kwargs):
        """standard_roi_head.py:70-126: assign + sample the proposals of every image, then the box head's loss. x: tuple of NCHW fp32
        feature maps that carry gradient (the training graph's output); the RoIs do not."""
        if self.bbox_assigner is None:
            raise LlvError('StandardRoIHead.forward_train needs train_cfg (assigner, sampler, pos_weight)')
        num_imgs = len(img_metas)
        if gt_bboxes_ignore is None:
            gt_bboxes_ignore = [None for _ in range(num_imgs)]
        sampling_results = []
        for i in range(num_imgs):
            props = proposal_list[i].detach()
            assign_result = self.bbox_assigner.assign(props[:, :4].contiguous(), gt_bboxes[i], gt_bboxes_ignore[i], gt_labels[i])
            sampling_results.append(self.bbox_sampler.sample(assign_result, props, gt_bboxes[i], gt_labels[i]))
        losses = dict()
        bbox_results = self._bbox_forward_train(x, sampling_results, gt_bboxes, gt_labels, img_metas)
        losses.update(bbox_results['loss_bbox'])
        return losses

    def _bbox_forward_train(self, x, sampling_results, gt_bboxes, gt_labels, img_metas):
        """standard_roi_head.py:126-139"""
        rois = bbox2roi([res.bboxes for res in sampling_results])
        bbox_results = self._bbox_forward(x, rois)
        bbox_targets = self.bbox_head.get_targets(sampling_results, gt_bboxes, gt_labels, self.train_cfg)
        loss_bbox = self.bbox_head.loss(bbox_results['cls_score'], bbox_results['bbox_pred'], rois, *bbox_targets)
        bbox_results.update(loss_bbox=loss_bbox)
        return bbox_results

    def simple_test_bboxes(self, x, img_metas, proposals, rcnn_test_cfg, rescale=False):
        """test_mixins.py:50-170 for unpadded per-image proposal lists: one head pass over all RoIs, per image softmax / class-specific
        decode / NMS. (The reference pads the lists of a batch to one length with zero rows and zeroes their scores; a video detector
        runs one image per GPU, where there is no padding.)"""
        rois = bbox2roi(proposals)
        res = self._bbox_forward(x, rois)
        counts = tuple(len(p) for p in proposals)
        rois_s, cls_s, reg_s = rois.split(counts, 0), res['cls_score'].split(counts, 0), res['bbox_pred'].split(counts, 0)
        det_bboxes, det_labels = [], []
        for i in range(len(proposals)):
            d, l = self.bbox_head.get_bboxes(rois_s[i], cls_s[i], reg_s[i], img_metas[i]['img_shape'], img_metas[i]['scale_factor'],
                                             rescale=rescale, cfg=rcnn_test_cfg)
            det_bboxes.append(d); det_labels.append(l)
        return det_bboxes, det_labels

    def simple_test(self, x, proposal_list, img_metas, proposals=None, rescale=False):
        """standard_roi_head.py:238-268: list per image of list[num_classes] of (k, 5) arrays."""
        det_bboxes, det_labels = self.simple_test_bboxes(x, img_metas, proposal_list, self.test_cfg, rescale=rescale)
        return [bbox2result(det_bboxes[i], det_labels[i], self.bbox_head.num_classes) for i in range(len(det_bboxes))]
