"""Training-side restatement (targets + losses of the dense head), fp32 CPU. Test infrastructure only.

  MaxIoUAssigner                mmdetection/mmdet/core/bbox/assigners/max_iou_assigner.py:60-212
  PseudoSampler                 mmdetection/mmdet/core/bbox/samplers/pseudo_sampler.py:23-41
  AnchorHead targets / loss     mmdetection/mmdet/models/dense_heads/anchor_head.py:175-493
  py_sigmoid_focal_loss         mmdetection/mmdet/models/losses/focal_loss.py:11-56 (the reference's own CPU path)
  l1 / smooth-l1 / weighting    mmdetection/mmdet/models/losses/smooth_l1_loss.py:8-45, utils.py:6-54
"""
import torch
import torch.nn.functional as F

from .anchors import anchor_inside_flags
from .bbox import bbox2delta, bbox_overlaps


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class MaxIoUAssigner:
    """max_iou_assigner.py:9-212"""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1, iou_calculator=None):
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all, self.ignore_iof_thr = gt_max_assign_all, ignore_iof_thr
        self.ignore_wrt_candidates, self.match_low_quality = ignore_wrt_candidates, match_low_quality

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        overlaps = bbox_overlaps(gt_bboxes, bboxes)
        if (self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0
                and bboxes.numel() > 0):
            if self.ignore_wrt_candidates:
                ignore_overlaps = bbox_overlaps(bboxes, gt_bboxes_ignore, mode='iof')
                ignore_max_overlaps, _ = ignore_overlaps.max(dim=1)
            else:
                ignore_overlaps = bbox_overlaps(gt_bboxes_ignore, bboxes, mode='iof')
                ignore_max_overlaps, _ = ignore_overlaps.max(dim=0)
            overlaps[:, ignore_max_overlaps > self.ignore_iof_thr] = -1
        return self.assign_wrt_overlaps(overlaps, gt_labels)

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        num_gts, num_bboxes = overlaps.size(0), overlaps.size(1)
        assigned_gt_inds = overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            max_overlaps = overlaps.new_zeros((num_bboxes,))
            if num_gts == 0:
                assigned_gt_inds[:] = 0
            assigned_labels = None if gt_labels is None else overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
            return AssignResult(num_gts, assigned_gt_inds, max_overlaps, labels=assigned_labels)
        max_overlaps, argmax_overlaps = overlaps.max(dim=0)
        gt_max_overlaps, gt_argmax_overlaps = overlaps.max(dim=1)
        if isinstance(self.neg_iou_thr, float):
            assigned_gt_inds[(max_overlaps >= 0) & (max_overlaps < self.neg_iou_thr)] = 0
        elif isinstance(self.neg_iou_thr, tuple):
            assert len(self.neg_iou_thr) == 2
            assigned_gt_inds[(max_overlaps >= self.neg_iou_thr[0]) & (max_overlaps < self.neg_iou_thr[1])] = 0
        pos_inds = max_overlaps >= self.pos_iou_thr
        assigned_gt_inds[pos_inds] = argmax_overlaps[pos_inds] + 1
        if self.match_low_quality:
            for i in range(num_gts):
                if gt_max_overlaps[i] >= self.min_pos_iou:
                    if self.gt_max_assign_all:
                        assigned_gt_inds[overlaps[i, :] == gt_max_overlaps[i]] = i + 1
                    else:
                        assigned_gt_inds[gt_argmax_overlaps[i]] = i + 1
        if gt_labels is not None:
            assigned_labels = assigned_gt_inds.new_full((num_bboxes,), -1)
            pos = torch.nonzero(assigned_gt_inds > 0, as_tuple=False).squeeze()
            if pos.numel() > 0:
                assigned_labels[pos] = gt_labels[assigned_gt_inds[pos] - 1]
        else:
            assigned_labels = None
        return AssignResult(num_gts, assigned_gt_inds, max_overlaps, labels=assigned_labels)


def reduce_loss(loss, reduction):
    if reduction == 'none':
        return loss
    return loss.mean() if reduction == 'mean' else loss.sum()


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """losses/utils.py:27-54"""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return reduce_loss(loss, reduction)
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def l1_loss(pred, target, weight=None, reduction='mean', avg_factor=None):
    """smooth_l1_loss.py:33-45 under @weighted_loss (utils.py:57-101)"""
    assert pred.size() == target.size() and target.numel() > 0
    return weight_reduce_loss(torch.abs(pred - target), weight, reduction, avg_factor)


def smooth_l1_loss(pred, target, weight=None, beta=1.0, reduction='mean', avg_factor=None):
    assert beta > 0
    diff = torch.abs(pred - target)
    loss = torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


def py_sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None):
    """focal_loss.py:11-56. target: one-hot (N, C) float, as FocalLoss.forward builds it on the CPU path (:164-168)."""
    pred_sigmoid = pred.sigmoid()
    target = target.type_as(pred)
    pt = (1 - pred_sigmoid) * target + pred_sigmoid * (1 - target)
    focal_weight = (alpha * target + (1 - alpha) * (1 - target)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, target, reduction='none') * focal_weight
    if weight is not None:
        if weight.shape != loss.shape:
            if weight.size(0) == loss.size(0):
                weight = weight.view(-1, 1)
            else:
                assert weight.numel() == loss.numel()
                weight = weight.view(loss.size(0), -1)
        assert weight.ndim == loss.ndim
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


def focal_loss_labels(pred, labels, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None, loss_weight=1.0):
    """FocalLoss.forward on the CPU (focal_loss.py:160-181): labels (N,) int64 with C = background."""
    num_classes = pred.size(1)
    target = F.one_hot(labels, num_classes=num_classes + 1)[:, :num_classes]
    return loss_weight * py_sigmoid_focal_loss(pred, target, weight, gamma, alpha, reduction, avg_factor)


def cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None):
    """mmdetection/mmdet/models/losses/cross_entropy_loss.py:9-40 (softmax cross entropy; pinned by the reference's
    tests/test_metrics/test_losses.py:7-31: 40. with class weights [0.8, 0.2], 200. without)."""
    loss = F.cross_entropy(pred, label, weight=class_weight, reduction='none')
    if weight is not None:
        weight = weight.float()
    return weight_reduce_loss(loss, weight=weight, reduction=reduction, avg_factor=avg_factor)


def accuracy(pred, target, topk=1, thresh=None):
    """mmdetection/mmdet/models/losses/accuracy.py:5-50 (pinned by test_losses.py:81-139)."""
    single = isinstance(topk, int)
    topk = (topk,) if single else topk
    maxk = max(topk)
    if pred.size(0) == 0:
        accu = [pred.new_tensor(0.) for _ in topk]
        return accu[0] if single else accu
    assert pred.ndim == 2 and target.ndim == 1 and pred.size(0) == target.size(0) and maxk <= pred.size(1)
    pred_value, pred_label = pred.topk(maxk, dim=1)
    pred_label = pred_label.t()
    correct = pred_label.eq(target.view(1, -1).expand_as(pred_label))
    if thresh is not None:
        correct = correct & (pred_value > thresh).t()
    res = [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / pred.size(0)) for k in topk]
    return res[0] if single else res


def unmap(data, count, inds, fill=0):
    """mmdetection/mmdet/core/utils/misc.py:31-42"""
    if data.dim() == 1:
        ret = data.new_full((count,), fill)
        ret[inds.type(torch.bool)] = data
    else:
        ret = data.new_full((count,) + data.size()[1:], fill)
        ret[inds.type(torch.bool), :] = data
    return ret


def get_targets_single(flat_anchors, valid_flags, gt_bboxes, gt_labels, img_meta, num_classes, assigner,
                       allowed_border=-1, pos_weight=-1, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), reg_decoded_bbox=False):
    """anchor_head.py:175-271 with PseudoSampler (FocalLoss => no sampling)."""
    inside_flags = anchor_inside_flags(flat_anchors, valid_flags, img_meta['img_shape'][:2], allowed_border)
    if not inside_flags.any():
        return (None,) * 6
    anchors = flat_anchors[inside_flags, :]
    ar = assigner.assign(anchors, gt_bboxes, None, gt_labels)
    pos_inds = torch.nonzero(ar.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
    neg_inds = torch.nonzero(ar.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
    n = anchors.shape[0]
    bbox_targets = torch.zeros_like(anchors)
    bbox_weights = torch.zeros_like(anchors)
    labels = anchors.new_full((n,), num_classes, dtype=torch.long)
    label_weights = anchors.new_zeros(n, dtype=torch.float)
    if len(pos_inds) > 0:
        pos_assigned = ar.gt_inds[pos_inds] - 1
        if reg_decoded_bbox:                                 # anchor_head.py:246-249
            bbox_targets[pos_inds, :] = gt_bboxes[pos_assigned, :]
        else:
            bbox_targets[pos_inds, :] = bbox2delta(anchors[pos_inds], gt_bboxes[pos_assigned, :], means, stds)
        bbox_weights[pos_inds, :] = 1.0
        labels[pos_inds] = gt_labels[pos_assigned]
        label_weights[pos_inds] = 1.0 if pos_weight <= 0 else pos_weight
    if len(neg_inds) > 0:
        label_weights[neg_inds] = 1.0
    tot = flat_anchors.size(0)
    return (unmap(labels, tot, inside_flags, fill=num_classes), unmap(label_weights, tot, inside_flags),
            unmap(bbox_targets, tot, inside_flags), unmap(bbox_weights, tot, inside_flags), pos_inds, neg_inds)


def iou_family_loss(kind, pred, target, weight=None, avg_factor=None, eps=None, linear=False):
    """mmdetection/mmdet/models/losses/iou_loss.py: iou_loss :14-38, giou_loss :88-104, diou_loss :109-154, ciou_loss :159-211
    and the shared class forward (:214-436): (n,4) weights are averaged to (n,), reduction 'mean' with avg_factor."""
    import math
    from .bbox import bbox_overlaps
    if weight is not None and not torch.any(weight > 0):
        return (pred * weight).sum()
    if weight is not None and weight.dim() > 1:
        weight = weight.mean(-1)
    if kind == 'iou':
        ious = bbox_overlaps(pred, target, is_aligned=True).clamp(min=1e-6 if eps is None else eps)
        loss = 1 - ious if linear else -ious.log()
    elif kind == 'giou':
        loss = 1 - bbox_overlaps(pred, target, mode='giou', is_aligned=True, eps=1e-7 if eps is None else eps)
    else:
        e = 1e-7 if eps is None else eps
        lt = torch.max(pred[:, :2], target[:, :2])
        rb = torch.min(pred[:, 2:], target[:, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[:, 0] * wh[:, 1]
        ap = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
        ag = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
        ious = overlap / (ap + ag - overlap + e)
        x1y1 = torch.min(pred[:, :2], target[:, :2])
        x2y2 = torch.max(pred[:, 2:], target[:, 2:])
        ewh = (x2y2 - x1y1).clamp(min=0)
        c2 = ewh[:, 0] ** 2 + ewh[:, 1] ** 2 + e
        left = ((target[:, 0] + target[:, 2]) - (pred[:, 0] + pred[:, 2])) ** 2 / 4
        right = ((target[:, 1] + target[:, 3]) - (pred[:, 1] + pred[:, 3])) ** 2 / 4
        rho2 = left + right
        if kind == 'diou':
            loss = 1 - (ious - rho2 / c2)
        else:
            w1, h1 = pred[:, 2] - pred[:, 0], pred[:, 3] - pred[:, 1] + e
            w2, h2 = target[:, 2] - target[:, 0], target[:, 3] - target[:, 1] + e
            v = 4 / math.pi ** 2 * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
            loss = 1 - (ious - (rho2 / c2 + v ** 2 / (1 - ious + v)))
    return weight_reduce_loss(loss, weight, 'mean', avg_factor)


def retina_loss(cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, anchor_generator, num_classes,
                assigner=None, gamma=2.0, alpha=0.25, box_loss='l1'):
    """AnchorHead.loss (anchor_head.py:425-493) for RetinaHead: FocalLoss + L1Loss, avg_factor = sum max(#pos, 1)."""
    assigner = assigner or MaxIoUAssigner(0.5, 0.4, min_pos_iou=0, ignore_iof_thr=-1)
    featmap_sizes = [c.size()[-2:] for c in cls_scores]
    anchors = anchor_generator.grid_anchors(featmap_sizes)
    num_level = [a.size(0) for a in anchors]
    flat = torch.cat(anchors)
    per_img = []
    for i, meta in enumerate(img_metas):
        flags = torch.cat(anchor_generator.valid_flags(featmap_sizes, meta['pad_shape']))
        per_img.append(get_targets_single(flat, flags, gt_bboxes[i], gt_labels[i], meta, num_classes, assigner,
                                          reg_decoded_bbox=box_loss != 'l1'))
    num_total_pos = sum(max(r[4].numel(), 1) for r in per_img)

    def to_levels(k):
        t = torch.stack([r[k] for r in per_img], 0)
        out, s = [], 0
        for n in num_level:
            out.append(t[:, s:s + n])
            s += n
        return out
    labels_l, lw_l, bt_l, bw_l = to_levels(0), to_levels(1), to_levels(2), to_levels(3)
    losses_cls, losses_bbox = [], []
    for cls, reg, lab, lw, bt, bw in zip(cls_scores, bbox_preds, labels_l, lw_l, bt_l, bw_l):
        cls = cls.permute(0, 2, 3, 1).reshape(-1, num_classes)
        losses_cls.append(focal_loss_labels(cls, lab.reshape(-1), lw.reshape(-1), gamma, alpha, 'mean', num_total_pos))
        reg = reg.permute(0, 2, 3, 1).reshape(-1, 4)
        if box_loss == 'l1':
            losses_bbox.append(l1_loss(reg, bt.reshape(-1, 4), bw.reshape(-1, 4), 'mean', num_total_pos))
        else:                                                # reg_decoded_bbox, anchor_head.py:404-409
            from .bbox import delta2bbox
            anc = anchors[len(losses_bbox)].repeat(cls_scores[0].shape[0], 1)
            dec = delta2bbox(anc, reg, max_shape=None)
            losses_bbox.append(iou_family_loss(box_loss, dec, bt.reshape(-1, 4), bw.reshape(-1, 4), num_total_pos))
    return dict(loss_cls=losses_cls, loss_bbox=losses_bbox)
