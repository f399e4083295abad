"""RoI feature extraction for the two-stage path (first piece): `SingleRoIExtractor` with the reference's
constructor (mmdetection/mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:9-108, base_roi_extractor.py:17-83) over
`llv_roi_align`. Every pyramid level is packed to NHWC fp16 once per call; the RoIs of a level go through one launch; the result comes
back as (k, C, ph, pw) fp32 like mmcv's RoIAlign output under `force_fp32`. The RPN / RoI heads that would call it are not built yet."""
import torch
import torch.nn as nn

from . import ops
from ._lib import LlvError
from .registry import ROI_EXTRACTORS


class RoIAlign(nn.Module):
    """mmcv.ops.RoIAlign's arguments (output_size, spatial_scale, sampling_ratio, pool_mode, aligned)."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True, use_torchvision=False):
        super().__init__()
        if pool_mode != 'avg':
            raise NotImplementedError("RoIAlign pool_mode 'max' is not on this path")
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale, self.sampling_ratio, self.aligned = float(spatial_scale), int(sampling_ratio), bool(aligned)

    def forward(self, feat, rois):
        """feat (N, C, H, W) fp32 on the GPU, rois (k, 5) -> (k, C, ph, pw) fp32."""
        if not feat.is_cuda:
            raise LlvError('RoIAlign runs on an MI355X only (llv_roi_align)')
        c = feat.shape[1]
        if c % 8:
            raise LlvError('RoIAlign: channel count must be a multiple of 8')
        x = ops.pack_input(feat.float().contiguous(), c)
        out = ops.roi_align(x, rois.float(), self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned, out_f32=True)
        return out.permute(0, 3, 1, 2).contiguous()


@ROI_EXTRACTORS.register_module()
class SingleRoIExtractor(nn.Module):
    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56):
        super().__init__()
        cfg = dict(roi_layer)
        layer_type = cfg.pop('type')
        if layer_type != 'RoIAlign':
            raise NotImplementedError(f'roi_layer {layer_type}: only RoIAlign is built')
        self.roi_layers = nn.ModuleList([RoIAlign(spatial_scale=1 / s, **cfg) for s in featmap_strides])   # base_roi_extractor.py:32-55
        self.out_channels, self.featmap_strides, self.finest_scale = out_channels, featmap_strides, finest_scale

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def init_weights(self):
        pass

    def map_roi_levels(self, rois, num_levels):
        """single_level_roi_extractor.py:33-53."""
        scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
        target_lvls = torch.floor(torch.log2(scale / self.finest_scale + 1e-6))
        return target_lvls.clamp(min=0, max=num_levels - 1).long()

    def roi_rescale(self, rois, scale_factor):
        """base_roi_extractor.py:57-83."""
        cx, cy = (rois[:, 1] + rois[:, 3]) * 0.5, (rois[:, 2] + rois[:, 4]) * 0.5
        w, h = (rois[:, 3] - rois[:, 1]) * scale_factor, (rois[:, 4] - rois[:, 2]) * scale_factor
        return torch.stack((rois[:, 0], cx - w * 0.5, cy - h * 0.5, cx + w * 0.5, cy + h * 0.5), dim=-1)

    def forward(self, feats, rois, roi_scale_factor=None):
        """feats: tuple of (N, C, H_l, W_l); rois (k, 5) = (batch index, x1, y1, x2, y2) -> (k, C, ph, pw)
        (single_level_roi_extractor.py:55-108)."""
        out_size = self.roi_layers[0].output_size
        num_levels = len(feats)
        roi_feats = feats[0].new_zeros(rois.size(0), self.out_channels, *out_size)
        if num_levels == 1:
            return roi_feats if len(rois) == 0 else self.roi_layers[0](feats[0], rois)
        target_lvls = self.map_roi_levels(rois, num_levels)
        if roi_scale_factor is not None:
            rois = self.roi_rescale(rois, roi_scale_factor)
        for i in range(num_levels):
            inds = (target_lvls == i).nonzero(as_tuple=False).squeeze(1)
            if inds.numel() > 0:
                roi_feats[inds] = self.roi_layers[i](feats[i], rois[inds])
        return roi_feats
