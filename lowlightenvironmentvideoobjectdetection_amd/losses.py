"""Loss modules with the reference's constructors (mmdetection/mmdet/models/losses/{focal_loss,smooth_l1_loss,
iou_loss,utils}.py). FocalLoss runs the llv_sigmoid_focal_loss kernels; the reductions are a handful of
scalars and stay in torch."""
import torch
import torch.nn as nn

from . import ops
from .registry import LOSSES


def reduce_loss(loss, reduction):
    """losses/utils.py:6-24"""
    if reduction == 'none':
        return loss
    if reduction == 'mean':
        return loss.mean()
    if reduction == 'sum':
        return loss.sum()
    raise ValueError(reduction)


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """losses/utils.py:27-54"""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        loss = reduce_loss(loss, reduction)
    else:
        if reduction == 'mean':
            loss = loss.sum() / avg_factor
        elif reduction != 'none':
            raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


class _SigmoidFocal(torch.autograd.Function):
    """mmcv.ops.sigmoid_focal_loss forward/backward on the llv kernels (per-element, reduction outside)."""

    @staticmethod
    def forward(ctx, pred, target, gamma, alpha):
        ctx.save_for_backward(pred, target)
        ctx.gamma, ctx.alpha = gamma, alpha
        return ops.sigmoid_focal_loss_fwd(pred, target, gamma, alpha)

    @staticmethod
    def backward(ctx, grad_out):
        pred, target = ctx.saved_tensors
        g = ops.sigmoid_focal_loss_bwd(pred, target, ctx.gamma, ctx.alpha)
        return g * grad_out, None, None, None


def sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None):
    """focal_loss.py:59-98"""
    loss = _SigmoidFocal.apply(pred.contiguous().float(), target, gamma, alpha)
    if weight is not None:
        if weight.shape != loss.shape:
            if weight.size(0) == loss.size(0):
                weight = weight.view(-1, 1)
            else:
                assert weight.numel() == loss.numel()
                weight = weight.view(loss.size(0), -1)
        assert weight.ndim == loss.ndim
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class FocalLoss(nn.Module):
    """focal_loss.py:101-181"""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha, self.reduction, self.loss_weight = use_sigmoid, gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * sigmoid_focal_loss(pred, target, weight, gamma=self.gamma, alpha=self.alpha,
                                                     reduction=reduction, avg_factor=avg_factor)


def _weighted(elementwise):
    def wrapper(pred, target, weight=None, reduction='mean', avg_factor=None, **kwargs):
        return weight_reduce_loss(elementwise(pred, target, **kwargs), weight, reduction, avg_factor)
    return wrapper


@_weighted
def l1_loss(pred, target):
    """smooth_l1_loss.py:33-45"""
    assert pred.size() == target.size() and target.numel() > 0
    return torch.abs(pred - target)


@_weighted
def smooth_l1_loss(pred, target, beta=1.0):
    """smooth_l1_loss.py:8-29"""
    assert beta > 0
    assert pred.size() == target.size() and target.numel() > 0
    diff = torch.abs(pred - target)
    return torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)


@LOSSES.register_module()
class L1Loss(nn.Module):
    """smooth_l1_loss.py:100-139"""

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * l1_loss(pred, target, weight, reduction=reduction, avg_factor=avg_factor)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    """smooth_l1_loss.py:48-97"""

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * smooth_l1_loss(pred, target, weight, beta=self.beta, reduction=reduction,
                                                 avg_factor=avg_factor, **kwargs)


# ----------------------------------------------------------------------------------------------
# IoU-family box losses (mmdetection/mmdet/models/losses/iou_loss.py:14-436). Inputs are decoded boxes
# (reg_decoded_bbox=True, anchor_head.py:404-409); elementwise on (n, 4) tensors that live on the device.
# ----------------------------------------------------------------------------------------------
def _aligned_overlaps(pred, target, mode, eps):
    """bbox_overlaps(..., is_aligned=True) (iou2d_calculator.py:115-135,151-159)"""
    area1 = (pred[..., 2] - pred[..., 0]) * (pred[..., 3] - pred[..., 1])
    area2 = (target[..., 2] - target[..., 0]) * (target[..., 3] - target[..., 1])
    lt = torch.max(pred[..., :2], target[..., :2])
    rb = torch.min(pred[..., 2:], target[..., 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = area1 + area2 - overlap
    e = union.new_tensor([eps])
    union = torch.max(union, e)
    ious = overlap / union
    if mode == 'iou':
        return ious
    elt = torch.min(pred[..., :2], target[..., :2])
    erb = torch.max(pred[..., 2:], target[..., 2:])
    ewh = (erb - elt).clamp(min=0)
    earea = torch.max(ewh[..., 0] * ewh[..., 1], e)
    return ious - (earea - union) / earea


@_weighted
def iou_loss(pred, target, linear=False, eps=1e-6):
    """iou_loss.py:14-38"""
    ious = _aligned_overlaps(pred, target, 'iou', 1e-6).clamp(min=eps)
    return 1 - ious if linear else -ious.log()


@_weighted
def bounded_iou_loss(pred, target, beta=0.2, eps=1e-3):
    """iou_loss.py:43-83"""
    pcx, pcy = (pred[:, 0] + pred[:, 2]) * 0.5, (pred[:, 1] + pred[:, 3]) * 0.5
    pw, ph = pred[:, 2] - pred[:, 0], pred[:, 3] - pred[:, 1]
    with torch.no_grad():
        tcx, tcy = (target[:, 0] + target[:, 2]) * 0.5, (target[:, 1] + target[:, 3]) * 0.5
        tw, th = target[:, 2] - target[:, 0], target[:, 3] - target[:, 1]
    dx, dy = tcx - pcx, tcy - pcy
    ldx = 1 - torch.max((tw - 2 * dx.abs()) / (tw + 2 * dx.abs() + eps), torch.zeros_like(dx))
    ldy = 1 - torch.max((th - 2 * dy.abs()) / (th + 2 * dy.abs() + eps), torch.zeros_like(dy))
    ldw = 1 - torch.min(tw / (pw + eps), pw / (tw + eps))
    ldh = 1 - torch.min(th / (ph + eps), ph / (th + eps))
    comb = torch.stack([ldx, ldy, ldw, ldh], dim=-1).view(ldx.size(0), -1)
    return torch.where(comb < beta, 0.5 * comb * comb / beta, comb - 0.5 * beta)


@_weighted
def giou_loss(pred, target, eps=1e-7):
    """iou_loss.py:88-104"""
    return 1 - _aligned_overlaps(pred, target, 'giou', eps)


def _diou_terms(pred, target, eps):
    lt = torch.max(pred[:, :2], target[:, :2])
    rb = torch.min(pred[:, 2:], target[:, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    ap = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    ag = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    ious = overlap / (ap + ag - overlap + eps)
    ewh = (torch.max(pred[:, 2:], target[:, 2:]) - torch.min(pred[:, :2], target[:, :2])).clamp(min=0)
    c2 = ewh[:, 0] ** 2 + ewh[:, 1] ** 2 + eps
    rho2 = ((target[:, 0] + target[:, 2]) - (pred[:, 0] + pred[:, 2])) ** 2 / 4 + \
        ((target[:, 1] + target[:, 3]) - (pred[:, 1] + pred[:, 3])) ** 2 / 4
    return ious, rho2, c2


@_weighted
def diou_loss(pred, target, eps=1e-7):
    """iou_loss.py:109-154"""
    ious, rho2, c2 = _diou_terms(pred, target, eps)
    return 1 - (ious - rho2 / c2)


@_weighted
def ciou_loss(pred, target, eps=1e-7):
    """iou_loss.py:159-211"""
    import math
    ious, rho2, c2 = _diou_terms(pred, target, eps)
    w1, h1 = pred[:, 2] - pred[:, 0], pred[:, 3] - pred[:, 1] + eps
    w2, h2 = target[:, 2] - target[:, 0], target[:, 3] - target[:, 1] + eps
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    return 1 - (ious - (rho2 / c2 + v ** 2 / (1 - ious + v)))


class _BoxLoss(nn.Module):
    """Shared forward of IoULoss / GIoULoss / DIoULoss / CIoULoss (iou_loss.py:214-436): all-zero weights short-cut,
    (n, 4) weights averaged to (n,), reduction override."""
    fn = None

    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0, **extra):
        super().__init__()
        self.eps, self.reduction, self.loss_weight, self.extra = eps, reduction, loss_weight, extra

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if (weight is not None) and (not torch.any(weight > 0)) and (reduction != 'none'):
            return (pred * weight).sum()  # 0
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        return self.loss_weight * type(self).fn(pred, target, weight, eps=self.eps, reduction=reduction, avg_factor=avg_factor,
                                                **self.extra, **kwargs)


@LOSSES.register_module()
class IoULoss(_BoxLoss):
    fn = staticmethod(iou_loss)

    def __init__(self, linear=False, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__(eps, reduction, loss_weight, linear=linear)
        self.linear = linear


@LOSSES.register_module()
class GIoULoss(_BoxLoss):
    fn = staticmethod(giou_loss)


@LOSSES.register_module()
class DIoULoss(_BoxLoss):
    fn = staticmethod(diou_loss)


@LOSSES.register_module()
class CIoULoss(_BoxLoss):
    fn = staticmethod(ciou_loss)


@LOSSES.register_module()
class BoundedIoULoss(nn.Module):
    """iou_loss.py:271-302"""

    def __init__(self, beta=0.2, eps=1e-3, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.eps, self.reduction, self.loss_weight = beta, eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        if weight is not None and not torch.any(weight > 0):
            return (pred * weight).sum()  # 0
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * bounded_iou_loss(pred, target, weight, beta=self.beta, eps=self.eps, reduction=reduction,
                                                   avg_factor=avg_factor, **kwargs)


class _SoftmaxCE(torch.autograd.Function):
    """F.cross_entropy(..., reduction='none') forward/backward on the llv kernels (one wavefront per row)."""

    @staticmethod
    def forward(ctx, pred, label, class_weight):
        ctx.save_for_backward(pred, label, class_weight)
        return ops.softmax_ce_fwd(pred, label, class_weight)

    @staticmethod
    def backward(ctx, grad_out):
        pred, label, cw = ctx.saved_tensors
        return ops.softmax_ce_bwd(pred, label, grad_out, cw), None, None


def cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None):
    """cross_entropy_loss.py:9-40"""
    loss = _SoftmaxCE.apply(pred.contiguous().float(), label, class_weight)
    if weight is not None:
        weight = weight.float()
    return weight_reduce_loss(loss, weight=weight, reduction=reduction, avg_factor=avg_factor)


def _expand_onehot_labels(labels, label_weights, label_channels):
    """cross_entropy_loss.py:43-56"""
    bin_labels = labels.new_full((labels.size(0), label_channels), 0)
    inds = torch.nonzero((labels >= 0) & (labels < label_channels), as_tuple=False).squeeze()
    if inds.numel() > 0:
        bin_labels[inds, labels[inds]] = 1
    bin_label_weights = None if label_weights is None else label_weights.view(-1, 1).expand(label_weights.size(0), label_channels)
    return bin_labels, bin_label_weights


def binary_cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None):
    """cross_entropy_loss.py:59-91 (element-wise torch on the device, like the a12 box losses)."""
    if pred.dim() != label.dim():
        label, weight = _expand_onehot_labels(label, weight, pred.size(-1))
    if weight is not None:
        weight = weight.float()
    loss = nn.functional.binary_cross_entropy_with_logits(pred, label.float(), pos_weight=class_weight, reduction='none')
    return weight_reduce_loss(loss, weight, reduction=reduction, avg_factor=avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """mmdetection/mmdet/models/losses/cross_entropy_loss.py:139-214. use_sigmoid=True -> binary_cross_entropy (:58-91; the YOLOv3
    head's class / objectness / centre terms), default -> softmax cross_entropy (:9-40) on llv_softmax_ce_{fwd,bwd} (the RPN / bbox heads
    of the two-stage path; pinned by the reference's test_losses.py:7-31). Mask cross entropy is not on this path."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert (use_sigmoid is False) or (use_mask is False)
        self.use_sigmoid, self.use_mask, self.reduction, self.class_weight, self.loss_weight = use_sigmoid, use_mask, reduction, class_weight, loss_weight
        if use_mask:
            self.cls_criterion = None
        else:
            self.cls_criterion = binary_cross_entropy if use_sigmoid else cross_entropy

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        if self.use_mask:
            raise NotImplementedError('CrossEntropyLoss: mask cross entropy is outside the path (no mask heads)')
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        cw = cls_score.new_tensor(self.class_weight) if self.class_weight is not None else None
        return self.loss_weight * self.cls_criterion(cls_score, label, weight, class_weight=cw, reduction=reduction,
                                                     avg_factor=avg_factor, **kwargs)


def accuracy(pred, target, topk=1, thresh=None):
    """mmdetection/mmdet/models/losses/accuracy.py:5-50 (logging only; a top-k over <= 81 columns, torch on the device)."""
    assert isinstance(topk, (int, tuple))
    if isinstance(topk, int):
        topk, return_single = (topk, ), True
    else:
        return_single = False
    maxk = max(topk)
    if pred.size(0) == 0:
        accu = [pred.new_tensor(0.) for _ in range(len(topk))]
        return accu[0] if return_single else accu
    assert pred.ndim == 2 and target.ndim == 1
    assert pred.size(0) == target.size(0)
    assert maxk <= pred.size(1), f'maxk {maxk} exceeds pred dimension {pred.size(1)}'
    pred_value, pred_label = pred.topk(maxk, dim=1)
    pred_label = pred_label.t()
    correct = pred_label.eq(target.view(1, -1).expand_as(pred_label))
    if thresh is not None:
        correct = correct & (pred_value > thresh).t()
    res = []
    for k in topk:
        correct_k = correct[:k].reshape(-1).float().sum(0, keepdim=True)
        res.append(correct_k.mul_(100.0 / pred.size(0)))
    return res[0] if return_single else res


class Accuracy(nn.Module):
    """accuracy.py:53-78"""

    def __init__(self, topk=(1, ), thresh=None):
        super().__init__()
        self.topk, self.thresh = topk, thresh

    def forward(self, pred, target):
        return accuracy(pred, target, self.topk, self.thresh)


@LOSSES.register_module()
class MSELoss(nn.Module):
    """mmdetection/mmdet/models/losses/mse_loss.py:8-49."""

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        loss = nn.functional.mse_loss(pred, target, reduction='none')
        return self.loss_weight * weight_reduce_loss(loss, weight, self.reduction, avg_factor)
