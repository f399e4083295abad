"""AnchorGenerator restated (mmdetection/mmdet/core/anchor/anchor_generator.py). Test infrastructure only."""
import numpy as np
import torch


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class AnchorGenerator:
    """anchor_generator.py:9-346. Same constructor arguments, same float32 arithmetic order."""

    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0.):
        # anchor_generator.py:69-109
        if center_offset != 0:
            assert centers is None
        if not (0 <= center_offset <= 1):
            raise ValueError('center_offset should be in range [0, 1]')
        if centers is not None:
            assert len(centers) == len(strides)
        self.strides = [_pair(s) for s in strides]
        self.base_sizes = [min(s) for s in self.strides] if base_sizes is None else base_sizes
        assert len(self.base_sizes) == len(self.strides)
        assert ((octave_base_scale is not None and scales_per_octave is not None) ^ (scales is not None))
        if scales is not None:
            self.scales = torch.Tensor(scales)
        else:
            octave_scales = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
            self.scales = torch.Tensor(octave_scales * octave_base_scale)
        self.ratios = torch.Tensor(ratios)
        self.scale_major = scale_major
        self.centers = centers
        self.center_offset = center_offset
        self.base_anchors = self.gen_base_anchors()

    @property
    def num_base_anchors(self):
        return [b.size(0) for b in self.base_anchors]

    @property
    def num_levels(self):
        return len(self.strides)

    def gen_base_anchors(self):
        # anchor_generator.py:121-141
        out = []
        for i, base_size in enumerate(self.base_sizes):
            center = self.centers[i] if self.centers is not None else None
            out.append(self.gen_single_level_base_anchors(base_size, self.scales, self.ratios, center))
        return out

    def gen_single_level_base_anchors(self, base_size, scales, ratios, center=None):
        # anchor_generator.py:142-185
        w = h = base_size
        if center is None:
            x_center = self.center_offset * w
            y_center = self.center_offset * h
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

    def grid_anchors(self, featmap_sizes, device='cpu'):
        # anchor_generator.py:207-270: order (y, x, anchor)
        assert self.num_levels == len(featmap_sizes)
        out = []
        for i in range(self.num_levels):
            feat_h, feat_w = featmap_sizes[i]
            stride = self.strides[i]
            shift_x = torch.arange(0, feat_w) * stride[0]
            shift_y = torch.arange(0, feat_h) * stride[1]
            xx = shift_x.repeat(len(shift_y))
            yy = shift_y.view(-1, 1).repeat(1, len(shift_x)).view(-1)
            shifts = torch.stack([xx, yy, xx, yy], dim=-1).type_as(self.base_anchors[i])
            out.append((self.base_anchors[i][None, :, :] + shifts[:, None, :]).view(-1, 4))
        return out

    def valid_flags(self, featmap_sizes, pad_shape, device='cpu'):
        # anchor_generator.py:272-328
        assert self.num_levels == len(featmap_sizes)
        out = []
        for i in range(self.num_levels):
            stride = self.strides[i]
            feat_h, feat_w = featmap_sizes[i]
            h, w = pad_shape[:2]
            valid_h = min(int(np.ceil(h / stride[1])), feat_h)
            valid_w = min(int(np.ceil(w / stride[0])), feat_w)
            vx = torch.zeros(feat_w, dtype=torch.bool)
            vy = torch.zeros(feat_h, dtype=torch.bool)
            vx[:valid_w] = 1
            vy[:valid_h] = 1
            xx = vx.repeat(feat_h)
            yy = vy.view(-1, 1).repeat(1, feat_w).view(-1)
            valid = xx & yy
            n = self.num_base_anchors[i]
            out.append(valid[:, None].expand(valid.size(0), n).contiguous().view(-1))
        return out


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    """mmdetection/mmdet/core/anchor/utils.py:26-46"""
    img_h, img_w = img_shape[:2]
    if allowed_border >= 0:
        return valid_flags & (flat_anchors[:, 0] >= -allowed_border) & (flat_anchors[:, 1] >= -allowed_border) & \
            (flat_anchors[:, 2] < img_w + allowed_border) & (flat_anchors[:, 3] < img_h + allowed_border)
    return valid_flags
