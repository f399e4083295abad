"""Seeded synthetic weights for benchmarks and tests (no checkpoints can be fetched here).

The reference's own init (resnet.py:601-618 zero_init_residual, retina_head.py:83-91 std=0.01) makes every
residual branch and almost every logit constant, which would leave NMS idle and hide numerical problems;
these helpers give every layer non-trivial, reproducible statistics instead."""
import math

import torch
import torch.nn as nn


def seeded_weights(model, seed=0, cls_bias=-2.0):
    """He-normal convs, BatchNorm with non-trivial affine + running statistics, classification bias cls_bias."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, nn.Conv2d):
                fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                std = math.sqrt(2.0 / fan_in)
                if name.endswith('retina_cls') or name.endswith('retina_reg'):
                    std *= 0.5
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
                    if name.endswith('retina_cls'):
                        m.bias.add_(cls_bias)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
    if hasattr(model, 'invalidate_plans'):
        model.invalidate_plans()
    return model


def calibrate_head_on_device(det, plan, target_cls_std=1.2, target_cls_mean=-3.0, target_reg_std=0.25):
    """Rescale retina_cls / retina_reg from the statistics of one device pass so that logits look like a trained
    detector's (scores spread over (0,1), a few thousand above score_thr; box deltas O(0.1)). plan must have been
    run on a representative frame. The caller rebuilds its plans afterwards."""
    head = det.bbox_head
    nc, nr = head.num_anchors * head.cls_out_channels, head.num_anchors * 4
    cls = plan.cls_buf[:, :nc].float()
    reg = plan.reg_buf[:, :nr].float()
    with torch.no_grad():
        cs = target_cls_std / (cls.std().item() + 1e-12)
        b = head.retina_cls.bias
        mean_wo_bias = (cls.mean().item() - b.mean().item()) * cs
        head.retina_cls.weight.mul_(cs)
        b.copy_(torch.full_like(b, target_cls_mean - mean_wo_bias) + (b - b.mean()) * cs)
        rs = target_reg_std / (reg.std().item() + 1e-12)
        head.retina_reg.weight.mul_(rs)
        head.retina_reg.bias.mul_(rs)
    det.invalidate_plans()
    return det
