"""TEST INFRASTRUCTURE (oracle): the RoI side of the SELSA two-stage detector at test time.
  SelsaAggregator.forward        mmtracking/mmtrack/models/aggregators/selsa_aggregator.py:29-73
  SelsaBBoxHead.forward          mmtracking/mmtrack/models/roi_heads/bbox_heads/selsa_bbox_head.py:26-84
  SelsaRoIHead.simple_test_bboxes / _bbox_forward   mmtracking/mmtrack/models/roi_heads/selsa_roi_head.py:61-78,134-187
  BBoxHead.get_bboxes            mmdetection/mmdet/models/roi_heads/bbox_heads/bbox_head.py:270-371
  bbox2roi                       mmdetection/mmdet/core/bbox/transforms.py:72-96
  TemporalRoIAlign               mmtracking/mmtrack/models/roi_heads/roi_extractors/temporal_roi_align.py:46-207
  SelsaFastDVDnetDetect          mmtracking/mmtrack/models/vid/selsa_fastdvd_darkfarm.py:229-319,321-420 (test time, adaptive stride)
Plain torch fp32 on the CPU. Only tests/ may import this file."""
import torch
import torch.nn.functional as F

from .bbox import delta2bbox
from .nms import multiclass_nms
from .roi import roi_align


def selsa_aggregator(sd, p, x, ref_x, num_attention_blocks=16):
    n, m = x.shape[0], ref_x.shape[0]
    xe = F.linear(x, sd[p + 'fc_embed.weight'], sd[p + 'fc_embed.bias']).view(n, num_attention_blocks, -1).permute(1, 0, 2)
    re = F.linear(ref_x, sd[p + 'ref_fc_embed.weight'], sd[p + 'ref_fc_embed.bias']).view(m, num_attention_blocks, -1).permute(1, 2, 0)
    w = (torch.bmm(xe, re) / (xe.shape[-1] ** 0.5)).softmax(dim=2)
    rn = F.linear(ref_x, sd[p + 'ref_fc.weight'], sd[p + 'ref_fc.bias']).view(m, num_attention_blocks, -1).permute(1, 0, 2)
    xn = torch.bmm(w, rn).permute(1, 0, 2).contiguous()
    return F.linear(xn.view(n, -1), sd[p + 'fc.weight'], sd[p + 'fc.bias'])


def selsa_bbox_head_forward(sd, p, x, ref_x, num_shared_fcs=2, num_attention_blocks=16):
    x, ref_x = x.flatten(1), ref_x.flatten(1)
    for i in range(num_shared_fcs):
        w, b = sd[f'{p}shared_fcs.{i}.weight'], sd[f'{p}shared_fcs.{i}.bias']
        x, ref_x = F.linear(x, w, b), F.linear(ref_x, w, b)
        x = x + selsa_aggregator(sd, f'{p}aggregator.{i}.', x, ref_x, num_attention_blocks)
        ref_x, x = F.relu(ref_x), F.relu(x)
    return F.linear(x, sd[p + 'fc_cls.weight'], sd[p + 'fc_cls.bias']), F.linear(x, sd[p + 'fc_reg.weight'], sd[p + 'fc_reg.bias'])


def bbox2roi(bbox_list):
    rois = []
    for i, b in enumerate(bbox_list):
        rois.append(torch.cat([b.new_full((b.size(0), 1), i), b[:, :4]], dim=-1) if b.size(0) > 0 else b.new_zeros((0, 5)))
    return torch.cat(rois, 0)


def selsa_roi_head_simple_test(sd, p, x, ref_x, proposals, ref_proposals, img_shape, scale_factor, test_cfg, featmap_stride=16,
                               num_shared_fcs=2, num_attention_blocks=16, means=(0., 0., 0., 0.), stds=(0.2, 0.2, 0.2, 0.2), rescale=False):
    """One key image with proposals (k, 5), reference images with their proposals; single-level RoIAlign(7, sampling_ratio 2)."""
    rois, ref_rois = bbox2roi([proposals]), bbox2roi(ref_proposals)
    feats = roi_align(x, rois, 7, 1.0 / featmap_stride, 2, True)
    ref_feats = roi_align(ref_x, ref_rois, 7, 1.0 / featmap_stride, 2, True)
    cls_score, bbox_pred = selsa_bbox_head_forward(sd, p, feats, ref_feats, num_shared_fcs, num_attention_blocks)
    scores = F.softmax(cls_score, dim=-1)
    bboxes = delta2bbox(rois[:, 1:], bbox_pred, means, stds, max_shape=img_shape)
    if rescale:
        bboxes = bboxes / bboxes.new_tensor(scale_factor).repeat(bboxes.size(-1) // 4)
    dets, labels = multiclass_nms(bboxes, scores, test_cfg['score_thr'], test_cfg['nms'], test_cfg['max_per_img'])
    return dets, labels, cls_score, bbox_pred


def most_similar_roi_align(roi_feats, ref_feats, top_k=2):
    """temporal_roi_align.py:98-168: roi_feats (k, C, 7, 7), ref_feats (img_n, C, H, W) -> (img_n, k, C, 7, 7), plus the picked pixel
    indices and their softmax weights (rows = k*49 in (roi, y, x) order)."""
    k, c, ph, pw = roi_feats.shape
    img_n, _, h, w = ref_feats.shape
    re = (roi_feats / roi_feats.norm(p=2, dim=1, keepdim=True)).permute(0, 2, 3, 1).contiguous().view(-1, c)
    fe = (ref_feats / ref_feats.norm(p=2, dim=1, keepdim=True)).permute(1, 0, 2, 3).contiguous().view(c, -1)
    cos = re.mm(fe).view(-1, img_n, h * w)
    values, indices = cos.topk(k=top_k, dim=2, largest=True)
    weights = values.softmax(dim=2)
    ref_r = ref_feats.permute(2, 3, 0, 1).contiguous().view(-1, img_n, c)
    out = []
    for i in range(img_n):
        out.append((ref_r[indices[:, i], i, :] * weights[:, i].unsqueeze(-1)).sum(dim=1))
    out = torch.stack(out, 0).view(img_n, k, ph, pw, c).permute(0, 1, 4, 2, 3)
    return out, indices.permute(1, 0, 2), weights.permute(1, 0, 2), cos.permute(1, 0, 2)


def temporal_attention(sd, p, x, ref_x, num_blocks):
    """temporal_roi_align.py:46-96: x (1, k, C, 7, 7), ref_x (img_n, k, C, 7, 7) -> (k, C, 7, 7)."""
    x = torch.cat((x, ref_x), dim=0)
    n, k, _, ph, pw = x.shape
    e = F.conv2d(x.view(n * k, -1, ph, pw), sd[p + 'embed_network.conv.weight'], sd[p + 'embed_network.conv.bias'], 1, 1)
    ce = e.shape[1]
    e = e.view(n, k, num_blocks, -1, ph, pw)
    a = torch.sum(e * e[[0]], dim=3, keepdim=True) / (float(ce / num_blocks) ** 0.5)
    a = a.expand(-1, -1, -1, int(ce / num_blocks), -1, -1).contiguous().view(n, k, ce, ph, pw).softmax(dim=0)
    return (x * a).sum(dim=0)


def temporal_roi_align(sd, p, feat, rois, ref_feat=None, featmap_stride=16, num_blocks=4):
    """TemporalRoIAlign.forward (temporal_roi_align.py:170-207) on one feature level."""
    rf = roi_align(feat, rois, 7, 1.0 / featmap_stride, 2, True)
    if ref_feat is None:
        return rf
    ref_rf = most_similar_roi_align(rf, ref_feat)[0]
    if num_blocks > 0:
        return temporal_attention(sd, p, rf.unsqueeze(0), ref_rf, num_blocks)
    return torch.cat((rf.unsqueeze(0), ref_rf), dim=0).mean(dim=0)


def selsa_troi_roi_head_simple_test(sd, p_head, p_ext, x, ref_x, proposals, ref_proposals, img_shape, scale_factor, test_cfg, featmap_stride=16,
                                    num_shared_fcs=3, num_attention_blocks=16, num_temporal_blocks=4, means=(0., 0., 0., 0.),
                                    stds=(0.2, 0.2, 0.2, 0.2), rescale=False):
    """SelsaRoIHead.simple_test_bboxes with the TemporalRoIAlign extractor (selsa_roi_head.py:80-97,134-187): the key frame's RoIs see
    the reference feature maps, the reference frames' RoIs are plain RoIAlign."""
    rois, ref_rois = bbox2roi([proposals]), bbox2roi(ref_proposals)
    feats = temporal_roi_align(sd, p_ext, x, rois, ref_x, featmap_stride, num_temporal_blocks)
    ref_feats = roi_align(ref_x, ref_rois, 7, 1.0 / featmap_stride, 2, True)
    cls_score, bbox_pred = selsa_bbox_head_forward(sd, p_head, feats, ref_feats, num_shared_fcs, num_attention_blocks)
    scores = F.softmax(cls_score, dim=-1)
    bboxes = delta2bbox(rois[:, 1:], bbox_pred, means, stds, max_shape=img_shape)
    if rescale:
        bboxes = bboxes / bboxes.new_tensor(scale_factor).repeat(bboxes.size(-1) // 4)
    dets, labels = multiclass_nms(bboxes, scores, test_cfg['score_thr'], test_cfg['nms'], test_cfg['max_per_img'])
    return dets, labels, cls_score, bbox_pred, feats
