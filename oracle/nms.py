"""NMS restated: mmdet multiclass_nms + mmcv (1.3.0, un-vendored) batched_nms / nms. Test infrastructure only.

mmcv is absent from /root/reference (pinned >=1.2.4,<=1.4.0 by mmdetection/mmdet/__init__.py:18-26; the
authors ran 1.3.0). Its published algorithm, restated here:
  nms(boxes, scores, iou_threshold, offset=0): order = scores.sort(descending); walk the order, keep a box
      unless an earlier kept box has IoU > iou_threshold with it; areas (x2-x1+offset)*(y2-y1+offset);
      returns dets = cat(boxes[keep], scores[keep]) and keep, in score-descending order.
  batched_nms(boxes, scores, idxs, nms_cfg): boxes_for_nms = boxes + idxs.to(boxes) * (boxes.max() + 1);
      one nms() call when fewer than split_thr (10000) boxes, else one call per class and the union
      re-sorted by score. The two branches keep the same set.
No reference test pins NMS ("parity unpinned"); the only in-tree evidence is the docstring example at
mmdetection/mmdet/models/dense_heads/anchor_head.py:527-551 (shapes only).
Tie rule: stable sort, i.e. (score desc, index asc).
"""
import numpy as np
import torch


def nms(boxes, scores, iou_threshold, offset=0):
    """-> (dets (k,5) float32, keep (k,) int64). float32 IEEE arithmetic throughout."""
    b = boxes.detach().cpu().numpy().astype(np.float32).reshape(-1, 4)
    s = scores.detach().cpu().numpy().astype(np.float32).reshape(-1)
    n = b.shape[0]
    if n == 0:
        return torch.zeros((0, 5)), torch.zeros((0,), dtype=torch.long)
    order = np.argsort(-s.astype(np.float64), kind='stable')  # score desc, index asc on ties
    thr = np.float32(iou_threshold)
    off = np.float32(offset)
    bs = b[order]
    x1, y1, x2, y2 = bs[:, 0], bs[:, 1], bs[:, 2], bs[:, 3]
    areas = (x2 - x1 + off) * (y2 - y1 + off)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    zero = np.float32(0)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(order[i])
        if i + 1 >= n:
            break
        xx1 = np.maximum(x1[i], x1[i + 1:])
        yy1 = np.maximum(y1[i], y1[i + 1:])
        xx2 = np.minimum(x2[i], x2[i + 1:])
        yy2 = np.minimum(y2[i], y2[i + 1:])
        w = np.maximum(xx2 - xx1 + off, zero)
        h = np.maximum(yy2 - yy1 + off, zero)
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > thr
    keep = torch.from_numpy(np.asarray(keep, dtype=np.int64))
    dets = torch.cat([torch.from_numpy(b)[keep], torch.from_numpy(s)[keep, None]], dim=1)
    return dets, keep


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.batched_nms (1.3.0). Returns (dets (k,5), keep (k,))."""
    nms_cfg_ = dict(nms_cfg)
    class_agnostic = nms_cfg_.pop('class_agnostic', class_agnostic)
    if class_agnostic:
        boxes_for_nms = boxes
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        boxes_for_nms = boxes + offsets[:, None]
    nms_type = nms_cfg_.pop('type', 'nms')
    assert nms_type == 'nms'
    split_thr = nms_cfg_.pop('split_thr', 10000)
    iou_thr = nms_cfg_.pop('iou_threshold', nms_cfg_.pop('iou_thr', None))
    if boxes_for_nms.shape[0] < split_thr:
        dets, keep = nms(boxes_for_nms, scores, iou_thr)
        boxes = boxes[keep]
        scores = dets[:, -1]
    else:
        total_mask = scores.new_zeros(scores.size(), dtype=torch.bool)
        for cid in torch.unique(idxs):
            mask = (idxs == cid).nonzero(as_tuple=False).view(-1)
            dets, keep = nms(boxes_for_nms[mask], scores[mask], iou_thr)
            total_mask[mask[keep]] = True
        keep = total_mask.nonzero(as_tuple=False).view(-1)
        keep = keep[torch.sort(scores[keep], descending=True, stable=True)[1]]
        boxes = boxes[keep]
        scores = scores[keep]
    return torch.cat([boxes, scores[:, None]], -1), keep


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None, return_inds=False):
    """mmdetection/mmdet/core/post_processing/bbox_nms.py:7-93"""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
    scores = multi_scores[:, :-1]
    labels = torch.arange(num_classes, dtype=torch.long)
    labels = labels.view(1, -1).expand_as(scores)
    bboxes = bboxes.reshape(-1, 4)
    scores = scores.reshape(-1)
    labels = labels.reshape(-1)
    valid_mask = scores > score_thr
    if score_factors is not None:
        score_factors = score_factors.view(-1, 1).expand(multi_scores.size(0), num_classes).reshape(-1)
        scores = scores * score_factors
    inds = valid_mask.nonzero(as_tuple=False).squeeze(1)
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    if bboxes.numel() == 0:
        if return_inds:
            return bboxes, labels, inds
        return bboxes, labels
    dets, keep = batched_nms(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets = dets[:max_num]
        keep = keep[:max_num]
    if return_inds:
        return dets, labels[keep], inds[keep]   # flat (i*C + c) indices, for index-exact tests
    return dets, labels[keep]
