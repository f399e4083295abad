"""TEST INFRASTRUCTURE (oracle): restatement of mmcv.ops.roi_align (pool_mode='avg'), the op behind the two-stage path's
`roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=2)` (mmtracking/configs/vid/llvod/*.py -> mmdet
SingleRoIExtractor, mmdetection/mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:50-108).
mmcv-full (>=1.2.4,<=1.4.0) is un-vendored: the algorithm below follows its published roi_align kernel
(roi_align_forward_cuda_kernel / bilinear_interpolate). Parity unpinned (no reference vector); tests/test_roi.py pins it with
properties: a constant map pools to the constant, a linear ramp pools to its value at the bin centre, samples outside the
map count as zero. Only tests/ may import this file."""
import math

import torch


def _bilinear(feat, y, x):
    """feat (C, H, W); scalar sample position. mmcv bilinear_interpolate."""
    c, h, w = feat.shape
    if y < -1.0 or y > h or x < -1.0 or x > w:
        return feat.new_zeros(c)
    y = max(y, 0.0)
    x = max(x, 0.0)
    yl, xl = int(y), int(x)
    if yl >= h - 1:
        yh = yl = h - 1
        y = float(yl)
    else:
        yh = yl + 1
    if xl >= w - 1:
        xh = xl = w - 1
        x = float(xl)
    else:
        xh = xl + 1
    ly, lx = y - yl, x - xl
    hy, hx = 1.0 - ly, 1.0 - lx
    return hy * hx * feat[:, yl, xl] + hy * lx * feat[:, yl, xh] + ly * hx * feat[:, yh, xl] + ly * lx * feat[:, yh, xh]


def roi_align(feat, rois, output_size, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    """feat (N, C, H, W) fp32; rois (K, 5) = (batch index, x1, y1, x2, y2) -> (K, C, ph, pw)."""
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    k = rois.shape[0]
    out = feat.new_zeros(k, feat.shape[1], ph, pw)
    off = 0.5 if aligned else 0.0
    for r in range(k):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = [float(v) * spatial_scale - off for v in rois[r, 1:]]
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / ph, rw / pw
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        cnt = max(gh * gw, 1)
        for oy in range(ph):
            for ox in range(pw):
                acc = feat.new_zeros(feat.shape[1])
                for iy in range(gh):
                    y = y1 + oy * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        x = x1 + ox * bw + (ix + 0.5) * bw / gw
                        acc = acc + _bilinear(feat[b], y, x)
                out[r, :, oy, ox] = acc / cnt
    return out


def map_roi_levels(rois, num_levels, finest_scale=56):
    """single_level_roi_extractor.py:33-53: level = floor(log2(sqrt(w*h) / finest_scale + 1e-6)), clamped."""
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lvls = torch.floor(torch.log2(scale / finest_scale + 1e-6))
    return lvls.clamp(min=0, max=num_levels - 1).long()
