"""TEST INFRASTRUCTURE (oracle): proposal generation of the two-stage path.
  ChannelMapper.forward   mmdetection/mmdet/models/necks/channel_mapper.py:58-74 (ConvModule: conv + ReLU)
  RPNHead.forward_single  mmdetection/mmdet/models/dense_heads/rpn_head.py:43-50
  RPNHead._get_bboxes     rpn_head.py:88-234: per level sigmoid -> sort, keep nms_pre -> decode (max_shape = img_shape) ->
                          batched_nms with the LEVEL as the class id (mmcv, un-vendored: oracle/nms.py) -> dets[:max_per_img]
Only tests/ may import this file."""
import torch
import torch.nn.functional as F

from .detector import decode_candidates
from .nms import batched_nms


def channel_mapper_forward(sd, inputs, prefix='neck.'):
    return tuple(F.relu(F.conv2d(x, sd[f'{prefix}convs.{i}.conv.weight'], sd.get(f'{prefix}convs.{i}.conv.bias'),
                                 padding=(sd[f'{prefix}convs.{i}.conv.weight'].shape[-1] - 1) // 2)) for i, x in enumerate(inputs))


def rpn_head_forward(sd, feats, prefix='rpn_head.'):
    cls, reg = [], []
    for x in feats:
        t = F.relu(F.conv2d(x, sd[prefix + 'rpn_conv.weight'], sd[prefix + 'rpn_conv.bias'], padding=1))
        cls.append(F.conv2d(t, sd[prefix + 'rpn_cls.weight'], sd[prefix + 'rpn_cls.bias']))
        reg.append(F.conv2d(t, sd[prefix + 'rpn_reg.weight'], sd[prefix + 'rpn_reg.bias']))
    return cls, reg


def rpn_get_bboxes(cls_scores, bbox_preds, anchor_generator, bbox_coder, img_shape, cfg):
    """One image. Returns proposals (k, 5) sorted by score."""
    nms_pre = cfg.get('nms_pre', -1)
    boxes, scores, _ = decode_candidates(cls_scores, bbox_preds, anchor_generator, bbox_coder, img_shape, [1, 1, 1, 1], 1, nms_pre, False)
    scores = scores[:, 0]
    ids = torch.cat([torch.full((min(nms_pre, c.shape[1] * c.shape[2] * c.shape[3]) if nms_pre > 0 else c.shape[1] * c.shape[2] * c.shape[3],),
                                i, dtype=torch.long) for i, c in enumerate(cls_scores)])
    if cfg.get('min_bbox_size', 0) > 0:
        w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
        ok = (w >= cfg['min_bbox_size']) & (h >= cfg['min_bbox_size'])
        boxes, scores, ids = boxes[ok], scores[ok], ids[ok]
    nms_cfg = cfg['nms'] if 'nms' in cfg else dict(type='nms', iou_threshold=cfg['nms_thr'])
    dets, _ = batched_nms(boxes, scores, ids, nms_cfg)
    return dets[:cfg.get('max_per_img', cfg.get('max_num'))]
