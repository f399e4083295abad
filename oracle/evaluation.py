"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reference's VOC / ImageNet-style mAP evaluation.

Follows /root/reference/mmdetection/mmdet/core/evaluation/
  bbox_overlaps.py:4-48        bbox_overlaps (numpy, float32, eps 1e-6)
  mean_ap.py:12-57             average_precision ('area' / '11points')
  mean_ap.py:60-151            tpfp_imagenet (ImageNet DET/VID rule: gt shifted by -1 for the IoU, per-gt threshold
                               min(w*h / ((w+10)*(h+10)), thr), a detection may fall through to the next best *uncovered* gt)
  mean_ap.py:154-244           tpfp_default (PASCAL rule: the detection's best gt; second hit on a covered gt is a false positive)
  mean_ap.py:247-265           get_cls_results
  mean_ap.py:268-400           eval_map

Only tests/ may import this file. Parity unpinned: the reference holds no known-answer test for mean_ap.py; the
tests check this restatement on hand-computed cases and the device kernel against it on random ones.

One order is left open by the reference and fixed here: `np.argsort(-scores)` (quicksort) gives ties in score an
unspecified order; this file and the product use the stable order (equal scores: lower index first; across images:
image order), `kind='stable'`.
"""
import numpy as np


def bbox_overlaps(bboxes1, bboxes2, mode='iou', eps=1e-6):
    """bbox_overlaps.py:4-48 (the row/column exchange there only changes the loop direction, not a value)."""
    assert mode in ('iou', 'iof')
    b1 = bboxes1.astype(np.float32)
    b2 = bboxes2.astype(np.float32)
    rows, cols = b1.shape[0], b2.shape[0]
    ious = np.zeros((rows, cols), dtype=np.float32)
    if rows * cols == 0:
        return ious
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    for i in range(rows):
        xs = np.maximum(b1[i, 0], b2[:, 0])
        ys = np.maximum(b1[i, 1], b2[:, 1])
        xe = np.minimum(b1[i, 2], b2[:, 2])
        ye = np.minimum(b1[i, 3], b2[:, 3])
        overlap = np.maximum(xe - xs, 0) * np.maximum(ye - ys, 0)
        union = area1[i] + area2 - overlap if mode == 'iou' else np.full(cols, area1[i], np.float32)
        ious[i, :] = overlap / np.maximum(union, np.float32(eps))
    return ious


def average_precision(recalls, precisions, mode='area'):
    """mean_ap.py:12-57."""
    no_scale = recalls.ndim == 1
    if no_scale:
        recalls, precisions = recalls[None, :], precisions[None, :]
    assert recalls.shape == precisions.shape and recalls.ndim == 2
    ns = recalls.shape[0]
    ap = np.zeros(ns, dtype=np.float32)
    if mode == 'area':
        mrec = np.hstack((np.zeros((ns, 1), recalls.dtype), recalls, np.ones((ns, 1), recalls.dtype)))
        mpre = np.hstack((np.zeros((ns, 1), recalls.dtype), precisions, np.zeros((ns, 1), recalls.dtype)))
        for i in range(mpre.shape[1] - 1, 0, -1):
            mpre[:, i - 1] = np.maximum(mpre[:, i - 1], mpre[:, i])
        for i in range(ns):
            ind = np.where(mrec[i, 1:] != mrec[i, :-1])[0]
            ap[i] = np.sum((mrec[i, ind + 1] - mrec[i, ind]) * mpre[i, ind + 1])
    elif mode == '11points':
        for i in range(ns):
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                precs = precisions[i, recalls[i, :] >= thr]
                ap[i] += precs.max() if precs.size > 0 else 0
            ap /= 11            # inside the scale loop, as in the reference (mean_ap.py:44-49)
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return ap[0] if no_scale else ap


def _empty_gt(det_bboxes, area_ranges, tp, fp):
    if area_ranges == [(None, None)]:
        fp[...] = 1
    else:
        a = (det_bboxes[:, 2] - det_bboxes[:, 0]) * (det_bboxes[:, 3] - det_bboxes[:, 1])
        for i, (lo, hi) in enumerate(area_ranges):
            fp[i, (a >= lo) & (a < hi)] = 1
    return tp, fp


def tpfp_default(det_bboxes, gt_bboxes, gt_bboxes_ignore=None, iou_thr=0.5, area_ranges=None):
    """mean_ap.py:154-244."""
    if gt_bboxes_ignore is None:
        gt_bboxes_ignore = np.empty((0, 4), dtype=np.float32)
    gt_ignore = np.concatenate((np.zeros(gt_bboxes.shape[0], dtype=bool), np.ones(gt_bboxes_ignore.shape[0], dtype=bool)))
    gt = np.vstack((gt_bboxes, gt_bboxes_ignore))
    m, n = det_bboxes.shape[0], gt.shape[0]
    if area_ranges is None:
        area_ranges = [(None, None)]
    tp = np.zeros((len(area_ranges), m), dtype=np.float32)
    fp = np.zeros((len(area_ranges), m), dtype=np.float32)
    if n == 0:
        return _empty_gt(det_bboxes, area_ranges, tp, fp)
    if m == 0:
        return tp, fp
    ious = bbox_overlaps(det_bboxes, gt)
    ious_max, ious_argmax = ious.max(axis=1), ious.argmax(axis=1)
    order = np.argsort(-det_bboxes[:, -1], kind='stable')
    for k, (lo, hi) in enumerate(area_ranges):
        covered = np.zeros(n, dtype=bool)
        if lo is None:
            area_ignore = np.zeros_like(gt_ignore, dtype=bool)
        else:
            ga = (gt[:, 2] - gt[:, 0]) * (gt[:, 3] - gt[:, 1])
            area_ignore = (ga < lo) | (ga >= hi)
        for i in order:
            if ious_max[i] >= iou_thr:
                j = ious_argmax[i]
                if not (gt_ignore[j] or area_ignore[j]):
                    if not covered[j]:
                        covered[j] = True
                        tp[k, i] = 1
                    else:
                        fp[k, i] = 1
            elif lo is None:
                fp[k, i] = 1
            else:
                b = det_bboxes[i, :4]
                a = (b[2] - b[0]) * (b[3] - b[1])
                if lo <= a < hi:
                    fp[k, i] = 1
    return tp, fp


def tpfp_imagenet(det_bboxes, gt_bboxes, gt_bboxes_ignore=None, default_iou_thr=0.5, area_ranges=None):
    """mean_ap.py:60-151."""
    if gt_bboxes_ignore is None:
        gt_bboxes_ignore = np.empty((0, 4), dtype=np.float32)
    gt_ignore = np.concatenate((np.zeros(gt_bboxes.shape[0], dtype=bool), np.ones(gt_bboxes_ignore.shape[0], dtype=bool)))
    gt = np.vstack((gt_bboxes, gt_bboxes_ignore))
    m, n = det_bboxes.shape[0], gt.shape[0]
    if area_ranges is None:
        area_ranges = [(None, None)]
    tp = np.zeros((len(area_ranges), m), dtype=np.float32)
    fp = np.zeros((len(area_ranges), m), dtype=np.float32)
    if n == 0:
        return _empty_gt(det_bboxes, area_ranges, tp, fp)
    if m == 0:
        return tp, fp
    ious = bbox_overlaps(det_bboxes, gt - 1)
    gw, gh = gt[:, 2] - gt[:, 0], gt[:, 3] - gt[:, 1]
    thrs = np.minimum((gw * gh) / ((gw + 10.0) * (gh + 10.0)), default_iou_thr)
    order = np.argsort(-det_bboxes[:, -1], kind='stable')
    for k, (lo, hi) in enumerate(area_ranges):
        covered = np.zeros(n, dtype=bool)
        if lo is None:
            area_ignore = np.zeros_like(gt_ignore, dtype=bool)
        else:
            ga = gw * gh
            area_ignore = (ga < lo) | (ga >= hi)
        for i in order:
            best, j_best = -1, -1
            for j in range(n):
                if covered[j]:
                    continue
                if ious[i, j] >= thrs[j] and ious[i, j] > best:
                    best, j_best = ious[i, j], j
            if j_best >= 0:
                covered[j_best] = True
                if not (gt_ignore[j_best] or area_ignore[j_best]):
                    tp[k, i] = 1
            elif lo is None:
                fp[k, i] = 1
            else:
                b = det_bboxes[i, :4]
                a = (b[2] - b[0]) * (b[3] - b[1])
                if lo <= a < hi:
                    fp[k, i] = 1
    return tp, fp


def get_cls_results(det_results, annotations, class_id):
    """mean_ap.py:247-265."""
    cls_dets = [img_res[class_id] for img_res in det_results]
    cls_gts, cls_ign = [], []
    for ann in annotations:
        cls_gts.append(ann['bboxes'][ann['labels'] == class_id, :])
        if ann.get('labels_ignore', None) is not None:
            cls_ign.append(ann['bboxes_ignore'][ann['labels_ignore'] == class_id, :])
        else:
            cls_ign.append(np.empty((0, 4), dtype=np.float32))
    return cls_dets, cls_gts, cls_ign


def eval_map(det_results, annotations, scale_ranges=None, iou_thr=0.5, dataset=None, tpfp_fn=None):
    """mean_ap.py:268-400 without the process pool and the table printing. Returns (mean_ap, eval_results)."""
    assert len(det_results) == len(annotations)
    num_scales = len(scale_ranges) if scale_ranges is not None else 1
    num_classes = len(det_results[0])
    area_ranges = [(rg[0] ** 2, rg[1] ** 2) for rg in scale_ranges] if scale_ranges is not None else None
    if tpfp_fn is None:
        tpfp_fn = tpfp_imagenet if dataset in ['det', 'vid'] else tpfp_default
    eval_results = []
    for c in range(num_classes):
        cls_dets, cls_gts, cls_ign = get_cls_results(det_results, annotations, c)
        tpfp = [tpfp_fn(d, g, ig, iou_thr, area_ranges) for d, g, ig in zip(cls_dets, cls_gts, cls_ign)]
        tp, fp = tuple(zip(*tpfp))
        num_gts = np.zeros(num_scales, dtype=int)
        for bbox in cls_gts:
            if area_ranges is None:
                num_gts[0] += bbox.shape[0]
            else:
                ga = (bbox[:, 2] - bbox[:, 0]) * (bbox[:, 3] - bbox[:, 1])
                for k, (lo, hi) in enumerate(area_ranges):
                    num_gts[k] += np.sum((ga >= lo) & (ga < hi))
        dets = np.vstack(cls_dets)
        order = np.argsort(-dets[:, -1], kind='stable')
        tp = np.cumsum(np.hstack(tp)[:, order], axis=1)
        fp = np.cumsum(np.hstack(fp)[:, order], axis=1)
        eps = np.finfo(np.float32).eps
        recalls = tp / np.maximum(num_gts[:, None], eps)
        precisions = tp / np.maximum(tp + fp, eps)
        if scale_ranges is None:
            recalls, precisions, num_gts = recalls[0, :], precisions[0, :], num_gts.item()
        ap = average_precision(recalls, precisions, 'area' if dataset != 'voc07' else '11points')
        eval_results.append(dict(num_gts=num_gts, num_dets=dets.shape[0], recall=recalls, precision=precisions, ap=ap))
    if scale_ranges is not None:
        all_ap = np.vstack([r['ap'] for r in eval_results])
        all_gts = np.vstack([r['num_gts'] for r in eval_results])
        mean_ap = [all_ap[all_gts[:, i] > 0, i].mean() if np.any(all_gts[:, i] > 0) else 0.0 for i in range(num_scales)]
    else:
        aps = [r['ap'] for r in eval_results if r['num_gts'] > 0]
        mean_ap = np.array(aps).mean().item() if aps else 0.0
    return mean_ap, eval_results
