"""Host-side anchor bookkeeping with the reference's interface (mmdetection/mmdet/core/anchor/).

Only the base anchors (a handful of floats per level) are computed here, with torch on the host exactly as
anchor_generator.py:142-185 does; the per-cell grid is never materialised for inference -- the decode
kernel adds the integer shift to the base anchor on the fly (csrc/decode.hip). grid_anchors()/valid_flags()
exist for the training-target path and API compatibility.
"""
import numpy as np
import torch

from ..registry import ANCHOR_GENERATORS


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


@ANCHOR_GENERATORS.register_module()
class AnchorGenerator(object):
    """Same constructor as mmdet's AnchorGenerator (anchor_generator.py:57-109)."""

    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0.):
        if center_offset != 0:
            assert centers is None, f'center cannot be set when center_offset!=0, {centers} is given.'
        if not (0 <= center_offset <= 1):
            raise ValueError(f'center_offset should be in range [0, 1], {center_offset} is given.')
        if centers is not None:
            assert len(centers) == len(strides)
        self.strides = [_pair(stride) for stride in strides]
        self.base_sizes = [min(stride) for stride in self.strides] if base_sizes is None else base_sizes
        assert len(self.base_sizes) == len(self.strides)
        assert ((octave_base_scale is not None and scales_per_octave is not None) ^ (scales is not None)), \
            'scales and octave_base_scale with scales_per_octave cannot be set at the same time'
        if scales is not None:
            self.scales = torch.Tensor(scales)
        elif octave_base_scale is not None and scales_per_octave is not None:
            octave_scales = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
            self.scales = torch.Tensor(octave_scales * octave_base_scale)
        else:
            raise ValueError('Either scales or octave_base_scale with scales_per_octave should be set')
        self.octave_base_scale = octave_base_scale
        self.scales_per_octave = scales_per_octave
        self.ratios = torch.Tensor(ratios)
        self.scale_major = scale_major
        self.centers = centers
        self.center_offset = center_offset
        self.base_anchors = self.gen_base_anchors()

    @property
    def num_base_anchors(self):
        return [base_anchors.size(0) for base_anchors in self.base_anchors]

    @property
    def num_levels(self):
        return len(self.strides)

    def gen_base_anchors(self):
        return [self.gen_single_level_base_anchors(bs, self.scales, self.ratios,
                                                   self.centers[i] if self.centers is not None else None)
                for i, bs in enumerate(self.base_sizes)]

    def gen_single_level_base_anchors(self, base_size, scales, ratios, center=None):
        w, h = base_size, base_size
        if center is None:
            x_center, y_center = self.center_offset * w, self.center_offset * h
        else:
            x_center, y_center = center
        h_ratios = torch.sqrt(ratios)
        w_ratios = 1 / h_ratios
        if self.scale_major:
            ws = (w * w_ratios[:, None] * scales[None, :]).view(-1)
            hs = (h * h_ratios[:, None] * scales[None, :]).view(-1)
        else:
            ws = (w * scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (h * scales[:, None] * h_ratios[None, :]).view(-1)
        return torch.stack([x_center - 0.5 * ws, y_center - 0.5 * hs, x_center + 0.5 * ws, y_center + 0.5 * hs], dim=-1)

    def grid_anchors(self, featmap_sizes, device='cuda'):
        assert self.num_levels == len(featmap_sizes)
        out = []
        for i in range(self.num_levels):
            feat_h, feat_w = featmap_sizes[i]
            sx, sy = self.strides[i]
            base = self.base_anchors[i].to(device)
            shift_x = torch.arange(0, feat_w, device=device) * sx
            shift_y = torch.arange(0, feat_h, device=device) * sy
            xx = shift_x.repeat(feat_h)
            yy = shift_y.view(-1, 1).repeat(1, feat_w).view(-1)
            shifts = torch.stack([xx, yy, xx, yy], dim=-1).type_as(base)
            out.append((base[None, :, :] + shifts[:, None, :]).view(-1, 4))
        return out

    def valid_flags(self, featmap_sizes, pad_shape, device='cuda'):
        assert self.num_levels == len(featmap_sizes)
        out = []
        for i in range(self.num_levels):
            sx, sy = self.strides[i]
            feat_h, feat_w = featmap_sizes[i]
            h, w = pad_shape[:2]
            valid_h = min(int(np.ceil(h / sy)), feat_h)
            valid_w = min(int(np.ceil(w / sx)), feat_w)
            vx = torch.zeros(feat_w, dtype=torch.bool, device=device)
            vy = torch.zeros(feat_h, dtype=torch.bool, device=device)
            vx[:valid_w] = 1
            vy[:valid_h] = 1
            valid = vx.repeat(feat_h) & vy.view(-1, 1).repeat(1, feat_w).view(-1)
            out.append(valid[:, None].expand(valid.size(0), self.num_base_anchors[i]).contiguous().view(-1))
        return out

    def __repr__(self):
        return (f'{self.__class__.__name__}(strides={self.strides}, ratios={self.ratios}, scales={self.scales}, '
                f'base_sizes={self.base_sizes}, scale_major={self.scale_major}, num_levels={self.num_levels})')


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    """mmdet/core/anchor/utils.py:26-46"""
    img_h, img_w = img_shape[:2]
    if allowed_border >= 0:
        return valid_flags & (flat_anchors[:, 0] >= -allowed_border) & (flat_anchors[:, 1] >= -allowed_border) & \
            (flat_anchors[:, 2] < img_w + allowed_border) & (flat_anchors[:, 3] < img_h + allowed_border)
    return valid_flags


def images_to_levels(target, num_levels):
    """mmdet/core/anchor/utils.py:4-23"""
    target = torch.stack(target, 0)
    level_targets = []
    start = 0
    for n in num_levels:
        end = start + n
        level_targets.append(target[:, start:end])
        start = end
    return level_targets
