"""MaxIoUAssigner / PseudoSampler with the reference's interface
(mmdetection/mmdet/core/bbox/assigners/max_iou_assigner.py:9-212, samplers/pseudo_sampler.py:8-41,
assigners/assign_result.py, samplers/sampling_result.py). The k x n IoU matrix, both arg-maxes and the
low-quality match run in llv_iou_assign (csrc/train_ops.hip) without materialising the matrix."""
import torch

from .. import ops
from .._lib import LlvError
from ..registry import BBOX_ASSIGNERS, BBOX_SAMPLERS


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    @property
    def num_preds(self):
        return len(self.gt_inds)

    def add_gt_(self, gt_labels):
        """assign_result.py:193-204: the ground-truth boxes, put in front of the proposals, are assigned to themselves."""
        self_inds = torch.arange(1, len(gt_labels) + 1, dtype=torch.long, device=gt_labels.device)
        self.gt_inds = torch.cat([self_inds, self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(len(gt_labels)), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


class SamplingResult:
    """sampling_result.py:7-60"""

    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            assert self.pos_assigned_gt_inds.numel() == 0
            self.pos_gt_bboxes = torch.empty_like(gt_bboxes).view(-1, 4)
        else:
            if len(gt_bboxes.shape) < 2:
                gt_bboxes = gt_bboxes.view(-1, 4)
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None

    @property
    def bboxes(self):
        """sampling_result.py:63-65"""
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner(object):
    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1,
                 iou_calculator=dict(type='BboxOverlaps2D')):
        if not isinstance(neg_iou_thr, float):
            raise NotImplementedError('tuple neg_iou_thr is outside the dense-head hot path')
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all, self.ignore_iof_thr = gt_max_assign_all, ignore_iof_thr
        self.ignore_wrt_candidates, self.match_low_quality = ignore_wrt_candidates, match_low_quality
        self.gpu_assign_thr = gpu_assign_thr

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        if self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0:
            raise NotImplementedError('ignore regions (ignore_iof_thr > 0) are outside the hot path (config uses -1)')
        n, k = bboxes.shape[0], gt_bboxes.shape[0]
        if n == 0:
            gt_inds = bboxes.new_full((0,), -1, dtype=torch.long)
            labels = None if gt_labels is None else bboxes.new_full((0,), -1, dtype=torch.long)
            return AssignResult(k, gt_inds, bboxes.new_zeros((0,)), labels)
        gt_inds, max_ov = ops.iou_assign(gt_bboxes[:, :4], bboxes[:, :4], None, self.pos_iou_thr, self.neg_iou_thr,
                                         self.min_pos_iou, self.match_low_quality, self.gt_max_assign_all)
        if gt_labels is not None:
            labels = gt_inds.new_full((n,), -1)
            pos = torch.nonzero(gt_inds > 0, as_tuple=False).squeeze(-1)
            if pos.numel() > 0:
                labels[pos] = gt_labels[gt_inds[pos] - 1]
        else:
            labels = None
        return AssignResult(k, gt_inds, max_ov, labels)


def _bbox_overlaps(b1, b2, eps=1e-6):
    """BboxOverlaps2D, mode 'iou' (iou2d_calculator.py:43-159), (k, 4) x (n, 4) -> (k, n), on the boxes' device."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    ov = wh[..., 0] * wh[..., 1]
    union = torch.max(a1[:, None] + a2[None, :] - ov, ov.new_tensor([eps]))
    return ov / union


@BBOX_ASSIGNERS.register_module()
class GridAssigner(object):
    """mmdetection/mmdet/core/bbox/assigners/grid_assigner.py:9-155 (the YOLOv3 head's assigner): as MaxIoUAssigner, but only the
    anchors of the grid cell a ground-truth centre falls in (`box_responsible_flags`) can become positive. Tensor operations on the
    boxes' device (the (k, n) overlap matrix is tiny next to the convolutions that produced the predictions)."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, iou_calculator=dict(type='BboxOverlaps2D')):
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou, self.gt_max_assign_all = pos_iou_thr, neg_iou_thr, min_pos_iou, gt_max_assign_all

    def assign(self, bboxes, box_responsible_flags, gt_bboxes, gt_labels=None):
        num_gts, num_bboxes = gt_bboxes.size(0), bboxes.size(0)
        overlaps = _bbox_overlaps(gt_bboxes[:, :4], bboxes[:, :4])
        assigned_gt_inds = overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            max_overlaps = overlaps.new_zeros((num_bboxes,))
            if num_gts == 0:
                assigned_gt_inds[:] = 0
            labels = None if gt_labels is None else overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
            return AssignResult(num_gts, assigned_gt_inds, max_overlaps, labels)
        max_overlaps, _ = overlaps.max(dim=0)
        if isinstance(self.neg_iou_thr, float):
            assigned_gt_inds[(max_overlaps >= 0) & (max_overlaps <= self.neg_iou_thr)] = 0
        else:
            assert len(self.neg_iou_thr) == 2
            assigned_gt_inds[(max_overlaps > self.neg_iou_thr[0]) & (max_overlaps <= self.neg_iou_thr[1])] = 0
        flags = box_responsible_flags.type(torch.bool)
        overlaps[:, ~flags] = -1.
        max_overlaps, argmax_overlaps = overlaps.max(dim=0)
        gt_max_overlaps, gt_argmax_overlaps = overlaps.max(dim=1)
        pos_inds = (max_overlaps > self.pos_iou_thr) & flags
        assigned_gt_inds[pos_inds] = argmax_overlaps[pos_inds] + 1
        for i in range(num_gts):
            if gt_max_overlaps[i] > self.min_pos_iou:
                if self.gt_max_assign_all:
                    assigned_gt_inds[(overlaps[i, :] == gt_max_overlaps[i]) & flags] = i + 1
                elif flags[gt_argmax_overlaps[i]]:
                    assigned_gt_inds[gt_argmax_overlaps[i]] = i + 1
        if gt_labels is not None:
            labels = assigned_gt_inds.new_full((num_bboxes,), -1)
            pos = torch.nonzero(assigned_gt_inds > 0, as_tuple=False).squeeze(-1)
            if pos.numel() > 0:
                labels[pos] = gt_labels[assigned_gt_inds[pos] - 1]
        else:
            labels = None
        return AssignResult(num_gts, assigned_gt_inds, max_overlaps, labels)


@BBOX_SAMPLERS.register_module()
class PseudoSampler(object):
    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        gt_flags = bboxes.new_zeros(bboxes.shape[0], dtype=torch.uint8)
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)


@BBOX_SAMPLERS.register_module()
class RandomSampler(object):
    """samplers/random_sampler.py:8-80 over base_sampler.py:12-101 (the RPN / R-CNN samplers of the two-stage configs): up to
    num * pos_fraction positives and the rest negatives, each drawn with torch.randperm on the boxes' device when there are more
    than wanted; ground-truth boxes optionally join the proposals first."""

    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        self.num, self.pos_fraction, self.neg_pos_ub, self.add_gt_as_proposals = num, pos_fraction, neg_pos_ub, add_gt_as_proposals
        self.pos_sampler = self.neg_sampler = self

    @staticmethod
    def random_choice(gallery, num):
        assert len(gallery) >= num
        perm = torch.randperm(gallery.numel(), device=gallery.device)[:num]
        return gallery[perm]

    def _sample_pos(self, assign_result, num_expected, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False)
        if pos_inds.numel() != 0:
            pos_inds = pos_inds.squeeze(1)
        return pos_inds if pos_inds.numel() <= num_expected else self.random_choice(pos_inds, num_expected)

    def _sample_neg(self, assign_result, num_expected, **kwargs):
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False)
        if neg_inds.numel() != 0:
            neg_inds = neg_inds.squeeze(1)
        return neg_inds if len(neg_inds) <= num_expected else self.random_choice(neg_inds, num_expected)

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, **kwargs):
        if len(bboxes.shape) < 2:
            bboxes = bboxes[None, :]
        bboxes = bboxes[:, :4]
        gt_flags = bboxes.new_zeros((bboxes.shape[0],), dtype=torch.uint8)
        if self.add_gt_as_proposals and len(gt_bboxes) > 0:
            if gt_labels is None:
                raise ValueError('gt_labels must be given when add_gt_as_proposals is True')
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            assign_result.add_gt_(gt_labels)
            gt_flags = torch.cat([bboxes.new_ones(gt_bboxes.shape[0], dtype=torch.uint8), gt_flags])
        num_expected_pos = int(self.num * self.pos_fraction)
        pos_inds = self.pos_sampler._sample_pos(assign_result, num_expected_pos, bboxes=bboxes, **kwargs).unique()
        num_expected_neg = self.num - pos_inds.numel()
        if self.neg_pos_ub >= 0:
            num_expected_neg = min(num_expected_neg, int(self.neg_pos_ub * max(1, pos_inds.numel())))
        neg_inds = self.neg_sampler._sample_neg(assign_result, num_expected_neg, bboxes=bboxes, **kwargs).unique()
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)
