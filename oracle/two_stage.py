"""TEST INFRASTRUCTURE (oracle): the training step of the two-stage SELSA video detectors, plain torch
fp32 on the CPU under autograd. Only tests/ may import this file.

  SELSA.forward_train               mmtracking/mmtrack/models/vid/selsa.py:133-170
  SelsaFastDVDnetDetect.forward_train   mmtracking/mmtrack/models/vid/selsa_fastdvd_darkfarm.py:147-227
  RPNHead.loss                      mmdetection/mmdet/models/dense_heads/rpn_head.py:49-80 -> AnchorHead.loss (anchor_head.py:425-493)
                                    with a real sampler (sampling = True: avg_factor = #pos + #neg)
  RandomSampler                     mmdetection/mmdet/core/bbox/samplers/random_sampler.py:8-80, base_sampler.py:35-101
  AssignResult.add_gt_              mmdetection/mmdet/core/bbox/assigners/assign_result.py:193-204
  SelsaRoIHead.forward_train        mmtracking/mmtrack/models/roi_heads/selsa_roi_head.py:10-111
  BBoxHead.get_targets / loss       mmdetection/mmdet/models/roi_heads/bbox_heads/bbox_head.py:103-268
  binary_cross_entropy              mmdetection/mmdet/models/losses/cross_entropy_loss.py:43-96
  ConvFCBBoxHead.forward            mmdetection/mmdet/models/roi_heads/bbox_heads/convfc_bbox_head.py:129-160 (shared-fc form: Shared2FCBBoxHead)
  StandardRoIHead.simple_test / forward_train   mmdetection/mmdet/models/roi_heads/standard_roi_head.py:70-139,238-268, test_mixins.py:50-170
  DFF / FGFA / LLVOD two-stage branches         mmtracking/mmtrack/models/vid/dff.py:139-173,245-253, fgfa.py:149-185,344-352, llvod.py:124-145

Parity unpinned for the sampler (the reference holds no vector for it): RandomSampler draws with torch.randperm when it has more
candidates than it wants, which no two implementations reproduce; the tests size `num` so that every candidate is taken (the
`len(inds) <= num_expected` branch), where the result is deterministic."""
import torch
import torch.nn.functional as F

from .anchors import anchor_inside_flags
from .bbox import bbox2delta
from .roi import roi_align
from .selsa import bbox2roi, selsa_bbox_head_forward, temporal_roi_align
from .training import accuracy, cross_entropy, l1_loss, smooth_l1_loss, unmap, weight_reduce_loss


class RandomSampler:
    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True):
        self.num, self.pos_fraction, self.neg_pos_ub, self.add_gt_as_proposals = num, pos_fraction, neg_pos_ub, add_gt_as_proposals

    @staticmethod
    def _choice(gallery, num):
        return gallery[torch.randperm(gallery.numel())[:num]]

    def sample(self, gt_inds, labels, bboxes, gt_bboxes, gt_labels=None):
        """-> (pos_inds, neg_inds, bboxes (gts in front when added), gt_inds, labels) (base_sampler.py:35-101)."""
        bboxes = bboxes[:, :4]
        if self.add_gt_as_proposals and len(gt_bboxes) > 0:
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            gt_inds = torch.cat([torch.arange(1, len(gt_labels) + 1, dtype=torch.long), gt_inds])
            if labels is not None:
                labels = torch.cat([gt_labels, labels])
        num_expected_pos = int(self.num * self.pos_fraction)
        pos = torch.nonzero(gt_inds > 0, as_tuple=False).reshape(-1)
        if pos.numel() > num_expected_pos:
            pos = self._choice(pos, num_expected_pos)
        pos = pos.unique()
        num_expected_neg = self.num - pos.numel()
        if self.neg_pos_ub >= 0:
            num_expected_neg = min(num_expected_neg, int(self.neg_pos_ub * max(1, pos.numel())))
        neg = torch.nonzero(gt_inds == 0, as_tuple=False).reshape(-1)
        if neg.numel() > num_expected_neg:
            neg = self._choice(neg, num_expected_neg)
        neg = neg.unique()
        return pos, neg, bboxes, gt_inds, labels


def binary_cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None):
    """cross_entropy_loss.py:43-96: labels in [0, C) -> one-hot, background (C) -> all zero."""
    if pred.dim() != label.dim():
        c = pred.size(-1)
        bin_labels = label.new_full((label.size(0), c), 0)
        inds = torch.nonzero((label >= 0) & (label < c), as_tuple=False).reshape(-1)
        if inds.numel() > 0:
            bin_labels[inds, label[inds]] = 1
        weight = None if weight is None else weight.view(-1, 1).expand(weight.size(0), c)
        label = bin_labels
    if weight is not None:
        weight = weight.float()
    loss = F.binary_cross_entropy_with_logits(pred, label.float(), reduction='none')
    return weight_reduce_loss(loss, weight, reduction=reduction, avg_factor=avg_factor)


def rpn_loss(cls_scores, bbox_preds, gt_bboxes, img_metas, anchor_generator, assigner, sampler, allowed_border=-1, pos_weight=-1,
             means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), smooth_l1_beta=1.0 / 9.0, box_loss='smooth_l1'):
    """RPNHead.loss: one class (label 0 = object, 1 = background), sigmoid cross entropy + (smooth) L1 over sampled anchors."""
    featmap_sizes = [c.size()[-2:] for c in cls_scores]
    anchors = anchor_generator.grid_anchors(featmap_sizes)
    num_level = [a.size(0) for a in anchors]
    flat = torch.cat(anchors)
    per_img = []
    for i, meta in enumerate(img_metas):
        flags = torch.cat(anchor_generator.valid_flags(featmap_sizes, meta['pad_shape']))
        inside = anchor_inside_flags(flat, flags, meta['img_shape'][:2], allowed_border)
        anc = flat[inside]
        ar = assigner.assign(anc, gt_bboxes[i], None, None)
        pos, neg, _, gt_inds, _ = sampler.sample(ar.gt_inds, None, anc, gt_bboxes[i])
        n = anc.shape[0]
        bt, bw = torch.zeros_like(anc), torch.zeros_like(anc)
        labels = anc.new_full((n,), 1, dtype=torch.long)
        lw = anc.new_zeros(n)
        if len(pos) > 0:
            bt[pos] = bbox2delta(anc[pos], gt_bboxes[i][gt_inds[pos] - 1], means, stds)
            bw[pos] = 1.0
            labels[pos] = 0
            lw[pos] = 1.0 if pos_weight <= 0 else pos_weight
        if len(neg) > 0:
            lw[neg] = 1.0
        tot = flat.size(0)
        per_img.append((unmap(labels, tot, inside, fill=1), unmap(lw, tot, inside), unmap(bt, tot, inside), unmap(bw, tot, inside), pos, neg))
    num_total = sum(max(r[4].numel(), 1) for r in per_img) + sum(max(r[5].numel(), 1) for r in per_img)

    def to_levels(k):
        t = torch.stack([r[k] for r in per_img], 0)
        out, s = [], 0
        for n in num_level:
            out.append(t[:, s:s + n])
            s += n
        return out
    losses_cls, losses_bbox = [], []
    for cls, reg, lab, lw, bt, bw in zip(cls_scores, bbox_preds, to_levels(0), to_levels(1), to_levels(2), to_levels(3)):
        cls = cls.permute(0, 2, 3, 1).reshape(-1, 1)
        losses_cls.append(binary_cross_entropy(cls, lab.reshape(-1), lw.reshape(-1), 'mean', num_total))
        reg = reg.permute(0, 2, 3, 1).reshape(-1, 4)
        if box_loss == 'l1':
            losses_bbox.append(l1_loss(reg, bt.reshape(-1, 4), bw.reshape(-1, 4), 'mean', num_total))
        else:
            losses_bbox.append(smooth_l1_loss(reg, bt.reshape(-1, 4), bw.reshape(-1, 4), smooth_l1_beta, 'mean', num_total))
    return dict(loss_rpn_cls=losses_cls, loss_rpn_bbox=losses_bbox)


def bbox_head_targets(samples, gt_bboxes, num_classes, pos_weight=-1, means=(0., 0., 0., 0.), stds=(0.2, 0.2, 0.2, 0.2)):
    """bbox_head.py:103-207 over the images' (pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels)."""
    cols = []
    for pos_b, neg_b, pos_gt, pos_lab in samples:
        num_pos, num_neg = pos_b.size(0), neg_b.size(0)
        n = num_pos + num_neg
        labels = pos_b.new_full((n,), num_classes, dtype=torch.long)
        lw, bt, bw = pos_b.new_zeros(n), pos_b.new_zeros(n, 4), pos_b.new_zeros(n, 4)
        if num_pos > 0:
            labels[:num_pos] = pos_lab
            lw[:num_pos] = 1.0 if pos_weight <= 0 else pos_weight
            bt[:num_pos] = bbox2delta(pos_b, pos_gt, means, stds)
            bw[:num_pos] = 1
        if num_neg > 0:
            lw[-num_neg:] = 1.0
        cols.append((labels, lw, bt, bw))
    return tuple(torch.cat(c, 0) for c in zip(*cols))


def bbox_head_loss(cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, num_classes, box_loss='l1', beta=1.0):
    """bbox_head.py:209-268 (class-specific regression)."""
    losses = dict()
    avg = max(torch.sum(label_weights > 0).float().item(), 1.)
    losses['loss_cls'] = cross_entropy(cls_score, labels, label_weights, 'mean', avg)
    losses['acc'] = accuracy(cls_score, labels)
    pos = (labels >= 0) & (labels < num_classes)
    if pos.any():
        pp = bbox_pred.view(bbox_pred.size(0), -1, 4)[pos, labels[pos]]
        fn = (lambda p, t, w, a: l1_loss(p, t, w, 'mean', a)) if box_loss == 'l1' else (lambda p, t, w, a: smooth_l1_loss(p, t, w, beta, 'mean', a))
        losses['loss_bbox'] = fn(pp, bbox_targets[pos], bbox_weights[pos], bbox_targets.size(0))
    else:
        losses['loss_bbox'] = bbox_pred[pos].sum()
    return losses


def selsa_roi_head_forward_train(sd, p_head, x, ref_x, proposals, ref_proposals, gt_bboxes, gt_labels, assigner, sampler, num_classes,
                                 featmap_stride=16, num_shared_fcs=2, num_attention_blocks=16, box_loss='l1', p_ext=None,
                                 num_temporal_blocks=4, pos_weight=-1):
    """selsa_roi_head.py:10-111 for one key image. x (1, C, h, w), ref_x (M, C, h, w) carry the graph; proposals do not.
    p_ext: parameter prefix of a TemporalRoIAlign extractor (None = SingleRoIExtractor)."""
    props = proposals.detach()
    ar = assigner.assign(props[:, :4], gt_bboxes, None, gt_labels)
    pos, neg, boxes, gt_inds, _ = sampler.sample(ar.gt_inds, ar.labels, props, gt_bboxes, gt_labels)
    pos_b, neg_b = boxes[pos], boxes[neg]
    pos_gt = gt_bboxes[gt_inds[pos] - 1]
    pos_lab = gt_labels[gt_inds[pos] - 1]
    rois = bbox2roi([torch.cat([pos_b, neg_b])])
    ref_rois = bbox2roi([p.detach() for p in ref_proposals])
    if p_ext is None:
        feats = roi_align(x, rois, 7, 1.0 / featmap_stride, 2, True)
    else:
        feats = temporal_roi_align(sd, p_ext, x, rois, ref_x, featmap_stride, num_temporal_blocks)
    ref_feats = roi_align(ref_x, ref_rois, 7, 1.0 / featmap_stride, 2, True)
    cls_score, bbox_pred = selsa_bbox_head_forward(sd, p_head, feats, ref_feats, num_shared_fcs, num_attention_blocks)
    targets = bbox_head_targets([(pos_b, neg_b, pos_gt, pos_lab)], gt_bboxes, num_classes, pos_weight)
    losses = bbox_head_loss(cls_score, bbox_pred, *targets, num_classes, box_loss)
    return losses, dict(rois=rois, cls_score=cls_score, bbox_pred=bbox_pred, targets=targets)


# ------------------------------------------------------------------------------------------------
# the plain two-stage head of the reference's DFF / FGFA configs (faster_rcnn_r50_dc5.py:38-62)
# ------------------------------------------------------------------------------------------------
def shared_fc_bbox_head_forward(sd, p, x, num_shared_fcs=2):
    """ConvFCBBoxHead.forward with shared fcs only: flatten -> (fc, ReLU) x n -> fc_cls, fc_reg."""
    x = x.flatten(1)
    for i in range(num_shared_fcs):
        x = F.relu(F.linear(x, sd[f'{p}shared_fcs.{i}.weight'], sd[f'{p}shared_fcs.{i}.bias']))
    return F.linear(x, sd[p + 'fc_cls.weight'], sd[p + 'fc_cls.bias']), F.linear(x, sd[p + 'fc_reg.weight'], sd[p + 'fc_reg.bias'])


def standard_roi_head_simple_test(sd, p, x, proposals, img_shape, scale_factor, test_cfg, featmap_stride=16, num_shared_fcs=2,
                                  means=(0., 0., 0., 0.), stds=(0.2, 0.2, 0.2, 0.2), rescale=False):
    """One image: x (1, C, h, w), proposals (k, 5) -> (dets (m, 5), labels (m,), cls_score, bbox_pred). BBoxHead.get_bboxes
    (bbox_head.py:270-371): softmax, class-specific delta2bbox clipped to img_shape, optional rescale, multiclass NMS."""
    from .bbox import delta2bbox
    from .nms import multiclass_nms
    rois = bbox2roi([proposals])
    feats = roi_align(x, rois, 7, 1.0 / featmap_stride, 2, True)
    cls_score, bbox_pred = shared_fc_bbox_head_forward(sd, p, feats, num_shared_fcs)
    scores = F.softmax(cls_score, dim=-1)
    bboxes = delta2bbox(rois[:, 1:], bbox_pred, means, stds, max_shape=img_shape)
    if rescale:
        bboxes = bboxes / bboxes.new_tensor(scale_factor).repeat(bboxes.size(-1) // 4)
    dets, labels = multiclass_nms(bboxes, scores, test_cfg['score_thr'], test_cfg['nms'], test_cfg['max_per_img'])
    return dets, labels, cls_score, bbox_pred


def standard_roi_head_forward_train(sd, p_head, x, proposals, gt_bboxes, gt_labels, assigner, sampler, num_classes, featmap_stride=16,
                                    num_shared_fcs=2, box_loss='smooth_l1', pos_weight=-1, stds=(0.2, 0.2, 0.2, 0.2)):
    """standard_roi_head.py:70-139 for one image; x (1, C, h, w) carries the graph, the proposals do not."""
    props = proposals.detach()
    ar = assigner.assign(props[:, :4], gt_bboxes, None, gt_labels)
    pos, neg, boxes, gt_inds, _ = sampler.sample(ar.gt_inds, ar.labels, props, gt_bboxes, gt_labels)
    pos_b, neg_b = boxes[pos], boxes[neg]
    pos_gt = gt_bboxes[gt_inds[pos] - 1]
    pos_lab = gt_labels[gt_inds[pos] - 1]
    rois = bbox2roi([torch.cat([pos_b, neg_b])])
    feats = roi_align(x, rois, 7, 1.0 / featmap_stride, 2, True)
    cls_score, bbox_pred = shared_fc_bbox_head_forward(sd, p_head, feats, num_shared_fcs)
    targets = bbox_head_targets([(pos_b, neg_b, pos_gt, pos_lab)], gt_bboxes, num_classes, pos_weight, stds=stds)
    losses = bbox_head_loss(cls_score, bbox_pred, *targets, num_classes, box_loss)
    return losses, dict(rois=rois, cls_score=cls_score, bbox_pred=bbox_pred, targets=targets)
