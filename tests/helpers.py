"""Shared test helpers: reference-style configs, seeded weight randomisation, synthetic dark frames."""
import math

import numpy as np
import torch
import torch.nn as nn


def retinanet_cfg(num_classes=80, depth=50, in_channels=(256, 512, 1024, 2048), stacked_convs=4, feat=256, score_thr=0.05):
    """mmdetection/configs/_base_/models/retinanet_r50_fpn.py:2-60 (pretrained=None: no network here)."""
    return dict(
        type='RetinaNet', pretrained=None,
        backbone=dict(type='ResNet', depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'),
        neck=dict(type='FPN', in_channels=list(in_channels), out_channels=feat, start_level=1, add_extra_convs='on_input',
                  num_outs=5),
        bbox_head=dict(type='RetinaHead', num_classes=num_classes, in_channels=feat, stacked_convs=stacked_convs,
                       feat_channels=feat,
                       anchor_generator=dict(type='AnchorGenerator', octave_base_scale=4, scales_per_octave=3,
                                             ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128]),
                       bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0],
                                       target_stds=[1.0, 1.0, 1.0, 1.0]),
                       loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                       loss_bbox=dict(type='L1Loss', loss_weight=1.0)),
        train_cfg=None,
        test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=score_thr, nms=dict(type='nms', iou_threshold=0.5),
                      max_per_img=100))


def randomize(model, seed=0, cls_bias=-2.0):
    """Seeded weights that exercise every layer: He-normal convs, non-trivial BN statistics (the reference's
    zero_init_residual would silence the residual branches), a classification bias that lets candidates
    pass score_thr so NMS does real work (SURVEY.md 8d 'busy' variant)."""
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
            elif isinstance(getattr(m, 'weight', None), nn.Parameter) and m.weight.dim() == 4:
                # a module that owns a conv weight without being an nn.Conv2d (ModulatedDCNPack): left alone it would keep its UNSEEDED
                # initialisation and make a comparison differ from run to run
                fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan_in))
                if getattr(m, 'bias', None) is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
    if hasattr(model, 'invalidate_plans'):
        model.invalidate_plans()
    return model


def calibrate_cls(det, img, target_std=1.2, target_mean=-3.0):
    """Rescale retina_cls so that, on `img`, the logits have the spread of a trained detector (scores spread
    over (0, 1) instead of saturating): otherwise the top-100 detections are all ~1.0 and their membership is
    decided by rounding noise. Uses the fp32 oracle on the CPU; deterministic."""
    from oracle.detector import retina_head_forward, retinanet_extract_feat
    sd = det.state_dict()
    with torch.no_grad():
        cls, reg = retina_head_forward(sd, retinanet_extract_feat(sd, img, det.backbone.depth), det.bbox_head.stacked_convs)
        # box deltas of a trained head are O(0.1); random ones reach the +-log(1000/16) clamp where exp() turns a
        # 1e-3 relative logit difference into pixels
        rflat = torch.cat([r.flatten() for r in reg])
        rs = 0.25 / (rflat.std().item() + 1e-12)
        det.bbox_head.retina_reg.weight.mul_(rs)
        det.bbox_head.retina_reg.bias.mul_(rs)
        flat = torch.cat([c.flatten() for c in cls])
        b = det.bbox_head.retina_cls.bias
        w = det.bbox_head.retina_cls.weight
        scale = target_std / (flat.std().item() + 1e-12)
        mean_wo_bias = (flat.mean().item() - b.mean().item()) * scale
        w.mul_(scale)
        b.copy_(torch.full_like(b, target_mean - mean_wo_bias) + (b - b.mean()) * scale)
    det.invalidate_plans()
    return det


def smooth_image(n, h, w, seed=0):
    """Low-pass noise + rectangles, normalised-image statistics (zero mean, unit-ish variance), NCHW fp32."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(n, 3, h // 8 + 1, w // 8 + 1, generator=g)
    img = torch.nn.functional.interpolate(base, size=(h, w), mode='bilinear', align_corners=False)
    for k in range(6):
        y0, x0 = int(torch.randint(0, h - 8, (1,), generator=g)), int(torch.randint(0, w - 8, (1,), generator=g))
        hh, ww = int(torch.randint(6, max(h // 3, 8), (1,), generator=g)), int(torch.randint(6, max(w // 3, 8), (1,), generator=g))
        img[:, :, y0:y0 + hh, x0:x0 + ww] += torch.randn(1, 3, 1, 1, generator=g)
    img += 0.15 * torch.randn(n, 3, h, w, generator=g)
    return img.contiguous()


def img_meta(h, w, pad_h=None, pad_w=None, scale=1.0):
    pad_h, pad_w = pad_h or h, pad_w or w
    return dict(filename='synthetic.png', ori_filename='synthetic.png', ori_shape=(h, w, 3), img_shape=(h, w, 3),
                pad_shape=(pad_h, pad_w, 3), scale_factor=np.array([scale] * 4, dtype=np.float32), flip=False,
                flip_direction=None, img_norm_cfg=dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True))


def match_detections(res, ref, iou_thr=0.9, score_tol=0.03):
    """Fraction of the oracle's detections (per class arrays of (k,5)) that the device result reproduces."""
    tot, hit = 0, 0
    for a, b in zip(res, ref):
        tot += len(b)
        if len(a) == 0 or len(b) == 0:
            continue
        a_t, b_t = torch.from_numpy(np.ascontiguousarray(a)), torch.from_numpy(np.ascontiguousarray(b))
        lt = torch.max(a_t[:, None, :2], b_t[None, :, :2])
        rb = torch.min(a_t[:, None, 2:4], b_t[None, :, 2:4])
        wh = (rb - lt).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        aa = (a_t[:, 2] - a_t[:, 0]) * (a_t[:, 3] - a_t[:, 1])
        ab = (b_t[:, 2] - b_t[:, 0]) * (b_t[:, 3] - b_t[:, 1])
        iou = inter / (aa[:, None] + ab[None, :] - inter).clamp(min=1e-6)
        ds = (a_t[:, None, 4] - b_t[None, :, 4]).abs()
        ok = (iou > iou_thr) & (ds < score_tol)
        hit += int(ok.any(0).sum())
    return hit / max(tot, 1), tot


def detection_parity_report(res, ref, iou_thr=0.9, score_tol=0.03):
    """Distribution of the differences between the device's detections `res` and the oracle's `ref` (per-class lists of (k, 5)
    arrays): each oracle detection is paired with the same-class device detection of highest IoU. Returns a dict with
      matched / total, box corner error (pixels) and |score difference| over the matched pairs (max, mean, p95), and what became of
      the unmatched oracle detections: 'score_only' (IoU > thr, score off by > tol), 'shifted' (0.5 < IoU <= thr: a neighbouring
      anchor's box won the NMS), 'absent' (no device box of that class overlaps: it fell below score_thr / out of the top max_per_img),
      plus 'extra' = device detections that pair with no oracle detection."""
    corner, dscore = [], []
    out = dict(total=0, matched=0, score_only=0, shifted=0, absent=0, extra=0)
    for a, b in zip(res, ref):
        a = np.asarray(a, dtype=np.float64).reshape(-1, 5)
        b = np.asarray(b, dtype=np.float64).reshape(-1, 5)
        out['total'] += len(b)
        used = np.zeros(len(a), dtype=bool)
        for r in b:
            if len(a) == 0:
                out['absent'] += 1
                continue
            lt = np.maximum(a[:, :2], r[:2])
            rb = np.minimum(a[:, 2:4], r[2:4])
            wh = np.clip(rb - lt, 0, None)
            inter = wh[:, 0] * wh[:, 1]
            iou = inter / np.maximum((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (r[2] - r[0]) * (r[3] - r[1]) - inter, 1e-6)
            j = int(np.argmax(iou))
            ds = abs(a[j, 4] - r[4])
            if iou[j] > iou_thr and ds < score_tol:
                out['matched'] += 1
                used[j] = True
                corner.append(float(np.abs(a[j, :4] - r[:4]).max()))
                dscore.append(float(ds))
            elif iou[j] > iou_thr:
                out['score_only'] += 1
                used[j] = True
            elif iou[j] > 0.5:
                out['shifted'] += 1
                used[j] = True
            else:
                out['absent'] += 1
        out['extra'] += int((~used).sum())
    c, d = np.asarray(corner or [0.0]), np.asarray(dscore or [0.0])
    out.update(frac=out['matched'] / max(out['total'], 1),
               corner_px=dict(max=float(c.max()), mean=float(c.mean()), p95=float(np.percentile(c, 95))),
               score=dict(max=float(d.max()), mean=float(d.mean()), p95=float(np.percentile(d, 95))))
    return out


def grad_metrics(g, r):
    """A device gradient `g` against the oracle's `r` for one parameter tensor -> (max-abs error / max-abs of r, cosine, relative L2 error,
    max-abs of r). The max-abs bound alone cannot tell fp16 rounding (cosine ~ 1, error spread evenly) from a mis-routed gradient of a
    small tensor (cosine far from 1); the training-parity tests assert all three."""
    g, r = g.detach().double().cpu().flatten(), r.detach().double().cpu().flatten()
    scale = r.abs().max().item()
    nr = r.norm().item()
    cos = float(torch.dot(g, r) / (g.norm() * r.norm()).clamp(min=1e-300))
    return (g - r).abs().max().item() / max(scale, 1e-300), cos, (g - r).norm().item() / max(nr, 1e-300), scale
