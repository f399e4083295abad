"""TEST INFRASTRUCTURE (oracle): the RoI side of the SELSA two-stage detector at test time.
  SelsaAggregator.forward        mmtracking/mmtrack/models/aggregators/selsa_aggregator.py:29-73
  SelsaBBoxHead.forward          mmtracking/mmtrack/models/roi_heads/bbox_heads/selsa_bbox_head.py:26-84
  SelsaRoIHead.simple_test_bboxes / _bbox_forward   mmtracking/mmtrack/models/roi_heads/selsa_roi_head.py:61-78,134-187
  BBoxHead.get_bboxes            mmdetection/mmdet/models/roi_heads/bbox_heads/bbox_head.py:270-371
  bbox2roi                       mmdetection/mmdet/core/bbox/transforms.py:72-96
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
