"""RoI feature extraction for the two-stage path (first piece): `SingleRoIExtractor` with the reference's
constructor (mmdetection/mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:9-108, base_roi_extractor.py:17-83) over
`llv_roi_align`. Every pyramid level is packed to NHWC fp16 once per call; the RoIs of a level go through one launch; the result comes
back as (k, C, ph, pw) fp32 like mmcv's RoIAlign output under `force_fp32`. Callers: `selsa.SelsaRoIHead` (proposals from `rpn.RPNHead`);
`TemporalRoIAlign` below is the extractor of the authors' own model."""
import ctypes as C

import torch
import torch.nn as nn

from . import ops
from ._lib import CONV_OUT_F32, LlvError, check, lib
from .planned import PlannedModule
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
        """feat (N, C, H, W) fp32 on the GPU, rois (k, 5) -> (k, C, ph, pw) fp32. Differentiable with respect to `feat` (training
        of the two-stage heads: llv_roi_align_bwd); the RoIs carry no gradient, as in mmcv."""
        if not feat.is_cuda:
            raise LlvError('RoIAlign runs on an MI355X only (llv_roi_align)')
        c = feat.shape[1]
        if c % 8:
            raise LlvError('RoIAlign: channel count must be a multiple of 8')
        if feat.requires_grad and torch.is_grad_enabled():
            return _RoIAlignFn.apply(feat, rois.detach().float().contiguous(), self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)
        return self.forward_nhwc(feat, rois).permute(0, 3, 1, 2).contiguous()

    def forward_nhwc(self, feat, rois):
        """The same RoI features as the kernel writes them: (k, ph, pw, C) fp32 (TemporalRoIAlign works on rows of C channels)."""
        if not feat.is_cuda:
            raise LlvError('RoIAlign runs on an MI355X only (llv_roi_align)')
        c = feat.shape[1]
        if c % 8:
            raise LlvError('RoIAlign: channel count must be a multiple of 8')
        x = ops.pack_input(feat.float().contiguous(), c)
        return ops.roi_align(x, rois.float(), self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned, out_f32=True)


class _RoIAlignFn(torch.autograd.Function):
    """mmcv RoIAlignFunction (forward llv_roi_align on the fp16 NHWC copy of the map, backward llv_roi_align_bwd: fp32 atomics)."""

    @staticmethod
    def forward(ctx, feat, rois, output_size, spatial_scale, sampling_ratio, aligned):
        n, c, h, w = feat.shape
        x = ops.pack_input(feat.detach().float().contiguous(), c)
        out = ops.roi_align(x, rois, output_size, spatial_scale, sampling_ratio, aligned, out_f32=True)        # (k, ph, pw, c)
        ctx.save_for_backward(rois)
        ctx.geom = (n, c, h, w, output_size, spatial_scale, sampling_ratio, aligned)
        return out.permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def backward(ctx, grad):
        (rois,) = ctx.saved_tensors
        n, c, h, w, (ph, pw), scale, ratio, aligned = ctx.geom
        g = grad.float().permute(0, 2, 3, 1).contiguous()
        dfeat = torch.zeros((n, h, w, c), device=grad.device, dtype=torch.float32)
        check(lib().llv_roi_align_bwd(g.data_ptr(), rois.data_ptr(), dfeat.data_ptr(), n, h, w, c, rois.shape[0], ph, pw, float(scale), int(ratio),
                                      1 if aligned else 0, _stream()), 'llv_roi_align_bwd')
        return dfeat.permute(0, 3, 1, 2).contiguous(), None, None, None, None, None


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


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@ROI_EXTRACTORS.register_module()
class TemporalRoIAlign(SingleRoIExtractor, PlannedModule):
    """mmtracking/mmtrack/models/roi_heads/roi_extractors/temporal_roi_align.py:9-207 (the extractor of the authors' own model,
    configs/vid/llvod/llvod_fastdvdnet_darkfarm.py:49-59), same constructor and parameter names (`embed_network.conv.*`).

    forward(feats, rois, ref_feats=None): RoIAlign features of the key frame; with `ref_feats` they are blended with the Most Similar
    RoI features of every reference frame (last pyramid level): cosine similarity of every RoI point with every reference pixel, the
    `num_most_similar_points` best pixels per reference image, softmax of their cosines, weighted sum of their features; then
    temporal attention over {key, references} per attention block (or the plain mean with 0 blocks).

    On the device: the similarity maps are ONE launch of the convolution engine (a 1x1 convolution of the L2-normalised reference
    maps whose `weights` are the L2-normalised RoI points, fp32 out: llv_l2norm_rows_f16 writes them straight into the packed
    weight layout), the pick + blend is llv_msra_top2_gather, the embed network a 3x3 convolution over the 7x7 RoI maps of all
    images in one launch, the attention llv_troi_attention_fuse. num_most_similar_points must be 2 (the reference's default and the
    authors' setting)."""

    def __init__(self, num_most_similar_points=2, num_temporal_attention_blocks=4, *args, **kwargs):
        SingleRoIExtractor.__init__(self, *args, **kwargs)
        self._plans = {}
        if num_most_similar_points != 2:
            raise NotImplementedError('TemporalRoIAlign: num_most_similar_points = 2 (the reference default) is what the device kernel selects')
        self.num_most_similar_points = num_most_similar_points
        self.num_temporal_attention_blocks = num_temporal_attention_blocks
        if num_temporal_attention_blocks > 0:
            from .cnn import ConvModule
            self.embed_network = ConvModule(self.out_channels, self.out_channels, 3, padding=1, conv_cfg=None, norm_cfg=None, act_cfg=None)

    def most_similar_roi_align(self, roi_rows, ref_feats, return_picks=False):
        """roi_rows (rows, C) fp32 = the RoI points (roi, y, x) x channels; ref_feats (img_n, C, H, W) fp32 -> (img_n, rows, C) fp16
        (temporal_roi_align.py:98-168)."""
        L = lib()
        dev = roi_rows.device
        rows, c = roi_rows.shape
        img_n, _, h, w = ref_feats.shape
        if c % 32:
            raise LlvError('TemporalRoIAlign: channel count must be a multiple of 32')
        ref = ops.pack_input(ref_feats.float().contiguous(), c)                       # (img_n, h, w, c) fp16, un-normalised
        refn = torch.empty_like(ref)
        check(L.llv_l2norm_rows_f16(ref.data_ptr(), 0, refn.data_ptr(), img_n * h * w, c, c, c, None, _stream()), 'llv_l2norm_rows_f16')
        cout, w_rows = ops.round_up(rows, 4), ops.round_up(rows, 16)
        wbuf = torch.zeros((1, w_rows, c), device=dev, dtype=torch.float16)           # the packed-weight layout of a 1x1 conv: [row][k = channel]
        check(L.llv_l2norm_rows_f16(roi_rows.data_ptr(), 1, wbuf.data_ptr(), rows, c, c, c, None, _stream()), 'llv_l2norm_rows_f16')
        pc = ops.PackedConv(w=wbuf, bias=torch.zeros(cout, device=dev), cin=c, cout=cout, w_rows=w_rows, k_pad=c, kh=1, kw=1, stride=1, pad=0,
                            true_macs_per_pixel=rows * c)
        sim = ops.conv2d(refn, pc, flags=CONV_OUT_F32)                                # (img_n, h, w, cout) fp32 cosines
        out = torch.empty((img_n, rows, c), device=dev, dtype=torch.float16)
        idx = torch.empty((img_n, rows, 2), device=dev, dtype=torch.int32) if return_picks else None
        wts = torch.empty((img_n, rows, 2), device=dev, dtype=torch.float32) if return_picks else None
        check(L.llv_msra_top2_gather(sim.data_ptr(), ref.data_ptr(), out.data_ptr(), idx.data_ptr() if idx is not None else None,
                                     wts.data_ptr() if wts is not None else None, img_n, h * w, rows, cout, c, c, _stream()),
              'llv_msra_top2_gather')
        return (out, idx, wts, sim) if return_picks else out

    def _embed_pc(self, dev):
        def make():
            conv = self.embed_network.conv
            return ops.pack_conv(conv.weight, conv.bias, 1, 1).to(dev)
        return self._plan(('embed', str(dev)), make)

    def temporal_attentional_feature_aggregation(self, x_all, roi_hw):
        """x_all (n_img, rows, C) fp16, image 0 = the key frame's RoI features -> (rows, C) fp32 (temporal_roi_align.py:46-96)."""
        n_img, rows, c = x_all.shape
        ph, pw = roi_hw
        imgs = x_all.view(n_img * (rows // (ph * pw)), ph, pw, c)
        emb = ops.conv2d(imgs, self._embed_pc(x_all.device))
        out = torch.empty((rows, c), device=x_all.device, dtype=torch.float32)
        check(lib().llv_troi_attention_fuse(x_all.data_ptr(), emb.data_ptr(), out.data_ptr(), n_img, rows, c, self.num_temporal_attention_blocks,
                                            _stream()), 'llv_troi_attention_fuse')
        return out

    def forward_autograd(self, feats, rois, ref_feats):
        """temporal_roi_align.py:46-207 for training: RoIAlign through llv_roi_align / llv_roi_align_bwd, the similarity search, the
        embed convolution over 7x7 maps and the attention as torch's differentiable ops on the device (a few thousand RoI points:
        the gradient reaches the key map through RoIAlign, the reference maps through the gathered pixels and the cosines)."""
        import torch.nn.functional as F
        ph, pw = self.roi_layers[0].output_size
        roi_feats = self.roi_layers[0](feats[0], rois)                                # (k, C, ph, pw)
        k, c = roi_feats.shape[:2]
        ref = ref_feats[-1]
        img_n, _, h, w = ref.shape
        re = (roi_feats / roi_feats.norm(p=2, dim=1, keepdim=True)).permute(0, 2, 3, 1).reshape(-1, c)
        fe = (ref / ref.norm(p=2, dim=1, keepdim=True)).permute(1, 0, 2, 3).reshape(c, -1)
        cos = re.mm(fe).view(-1, img_n, h * w)
        values, indices = cos.topk(k=self.num_most_similar_points, dim=2, largest=True)
        weights = values.softmax(dim=2)
        ref_r = ref.permute(2, 3, 0, 1).reshape(-1, img_n, c)
        picked = [(ref_r[indices[:, i], i, :] * weights[:, i].unsqueeze(-1)).sum(dim=1) for i in range(img_n)]
        ref_roi = torch.stack(picked, 0).view(img_n, k, ph, pw, c).permute(0, 1, 4, 2, 3)
        x = torch.cat((roi_feats.unsqueeze(0), ref_roi), dim=0)                       # (1 + img_n, k, C, ph, pw)
        nb = self.num_temporal_attention_blocks
        if nb <= 0:
            return x.mean(dim=0)
        n = x.shape[0]
        conv = self.embed_network.conv
        e = F.conv2d(x.reshape(n * k, c, ph, pw), conv.weight, conv.bias, 1, 1)
        ce = e.shape[1]
        e = e.view(n, k, nb, -1, ph, pw)
        a = torch.sum(e * e[[0]], dim=3, keepdim=True) / (float(ce / nb) ** 0.5)
        a = a.expand(-1, -1, -1, int(ce / nb), -1, -1).contiguous().view(n, k, ce, ph, pw).softmax(dim=0)
        return (x * a).sum(dim=0)

    def forward(self, feats, rois, roi_scale_factor=None, ref_feats=None):
        if ref_feats is None:          # proposals of the reference images: plain RoIAlign (temporal_roi_align.py:184-187)
            return SingleRoIExtractor.forward(self, feats, rois, roi_scale_factor)
        if torch.is_grad_enabled() and len(feats) == 1 and rois.size(0) > 0 and (feats[0].requires_grad or ref_feats[-1].requires_grad):
            return self.forward_autograd(feats, rois, ref_feats)
        if len(feats) != 1:
            raise LlvError('TemporalRoIAlign with ref_feats: one feature level (featmap_strides of the SELSA configs: [16])')
        # roi_scale_factor: with one level the reference returns before its roi_rescale (single_level_roi_extractor.py:73-76), so
        # the argument has no effect there either
        ph, pw = self.roi_layers[0].output_size
        k = rois.size(0)
        if k == 0:
            return feats[0].new_zeros(0, self.out_channels, ph, pw)
        roi_nhwc = self.roi_layers[0].forward_nhwc(feats[0], rois)                    # (k, ph, pw, C) fp32
        c = roi_nhwc.shape[-1]
        roi_rows = roi_nhwc.view(k * ph * pw, c)
        ref_roi = self.most_similar_roi_align(roi_rows, ref_feats[-1])                # (img_n, rows, C) fp16
        x_all = torch.cat([roi_rows.to(torch.float16).unsqueeze(0), ref_roi], dim=0).contiguous()
        if self.num_temporal_attention_blocks > 0:
            out = self.temporal_attentional_feature_aggregation(x_all, (ph, pw))
        else:
            out = x_all.float().mean(dim=0)
        return out.view(k, ph, pw, c).permute(0, 3, 1, 2).contiguous()
