"""Training-side head code (targets, losses). Built after the inference path; see DESIGN.md scope table."""
from ._lib import LlvError


def anchor_head_loss(head, *args, **kwargs):
    raise LlvError('AnchorHead.loss: the training path (row a11-a13 of the scope table) is not built yet in this round; '
                   'the focal-loss and IoU-assign kernels exist (ops.sigmoid_focal_loss_*, ops.iou_assign)')


def single_stage_forward_train(det, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
    raise LlvError('SingleStageDetector.forward_train: conv backward kernels are not built yet in this round')
