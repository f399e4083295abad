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


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """Constructor only (AnchorHead's default loss_cls names it, anchor_head.py:49-52); the RetinaNet path uses FocalLoss."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        self.use_sigmoid, self.reduction, self.loss_weight = use_sigmoid, reduction, loss_weight

    def forward(self, *a, **k):
        raise NotImplementedError('CrossEntropyLoss is outside the dense-head hot path (RetinaNet uses FocalLoss)')
