"""TEST INFRASTRUCTURE (oracle): restatement of the COCO detection protocol for boxes, as the reference evaluates video
detection (mmtracking/mmtrack/datasets/coco_video_dataset.py:374-461 -> mmdet CocoDataset.evaluate,
mmdetection/mmdet/datasets/coco.py:347-548 -> pycocotools.cocoeval.COCOeval).

`pycocotools` (mmdet requirement `pycocotools`, unpinned; 2.0.x) is a third-party dependency that is neither under
/root/reference nor installed here. Its published algorithm is restated below from cocoeval.py:
  computeIoU   dets of an (image, category) sorted by score, truncated to maxDets[-1]; maskUtils.iou on (x, y, w, h) boxes in
               double precision; for an `iscrowd` ground truth the union is the detection's area
  evaluateImg  ignore = gt.ignore (= iscrowd) or area outside the range; gts stably sorted by ignore; per IoU threshold greedy
               matching in score order: skip matched non-crowd gts, stop at the first ignored gt once a regular one is matched,
               keep the gt with the largest IoU >= min(t, 1-1e-10) (a later one on equality); unmatched detections outside the
               area range are ignored
  accumulate   per (category, area, maxDet): merge-sort all detections by score, cumulative tp / fp over the non-ignored ones,
               precision made monotone from the right, sampled at 101 recall thresholds with searchsorted(side='left')
  summarize    AP = mean of precision > -1 over (IoU thresholds, recalls, categories); AR = mean of recall > -1
with mmdet's settings on top: maxDets = proposal_nums = (100, 300, 1000), AP summarised at maxDets[2], the key names of
coco.py:491-506. Parity unpinned: no reference test holds a vector for this path; tests/test_coco_eval.py checks hand-computed
cases. Only tests/ may import this file.
"""
import numpy as np

AREA_RANGES = [(0 ** 2, 1e5 ** 2), (0 ** 2, 32 ** 2), (32 ** 2, 96 ** 2), (96 ** 2, 1e5 ** 2)]      # all, small, medium, large


def _iou(d, g, crowd):
    w = min(d[0] + d[2], g[0] + g[2]) - max(d[0], g[0])
    h = min(d[1] + d[3], g[1] + g[3]) - max(d[1], g[1])
    if w <= 0 or h <= 0:
        return 0.0
    inter = w * h
    return inter / (d[2] * d[3] if crowd else d[2] * d[3] + g[2] * g[3] - inter)


def evaluate_img(dt, gt, a_rng, max_det, iou_thrs):
    """dt: list of dict(bbox xywh, score, area); gt: list of dict(bbox xywh, area, iscrowd, ignore). cocoeval.py evaluateImg."""
    if len(gt) == 0 and len(dt) == 0:
        return None
    gt_ig0 = [bool(g['ignore']) or g['area'] < a_rng[0] or g['area'] > a_rng[1] for g in gt]
    gtind = np.argsort(np.array(gt_ig0, dtype=np.int64), kind='mergesort')
    gt = [gt[i] for i in gtind]
    gt_ig = np.array([gt_ig0[i] for i in gtind], dtype=bool)
    dtind = np.argsort([-d['score'] for d in dt], kind='mergesort')
    dt = [dt[i] for i in dtind[:max_det]]
    T, G, D = len(iou_thrs), len(gt), len(dt)
    gtm = np.zeros((T, G), dtype=bool)
    dtm = np.zeros((T, D), dtype=bool)
    dt_ig = np.zeros((T, D), dtype=bool)
    for ti, t in enumerate(iou_thrs):
        for di, d in enumerate(dt):
            iou = min([t, 1 - 1e-10])
            m = -1
            for gi, g in enumerate(gt):
                if gtm[ti, gi] and not g['iscrowd']:
                    continue
                if m > -1 and not gt_ig[m] and gt_ig[gi]:
                    break
                v = _iou(d['bbox'], g['bbox'], g['iscrowd'])
                if v < iou:
                    continue
                iou, m = v, gi
            if m == -1:
                continue
            dt_ig[ti, di] = gt_ig[m]
            dtm[ti, di] = True
            gtm[ti, m] = True
    out_of_range = np.array([d['area'] < a_rng[0] or d['area'] > a_rng[1] for d in dt], dtype=bool).reshape(1, D)
    dt_ig = np.logical_or(dt_ig, np.logical_and(~dtm, np.repeat(out_of_range, T, 0)))
    return dict(dtm=dtm, dt_ig=dt_ig, scores=[d['score'] for d in dt], gt_ig=gt_ig)


def coco_bbox_eval(det_results, annotations, iou_thrs=None, max_dets=(100, 300, 1000), area_ranges=AREA_RANGES):
    """det_results: per image, per class (k, 5) arrays (x1, y1, x2, y2, score). annotations: per image dict(bboxes (n,4) xyxy, labels,
    optional areas, bboxes_ignore / labels_ignore (crowd regions), areas_ignore). Returns dict(precision [T,R,K,A,M], recall [T,K,A,M],
    stats (12,), and mmdet's bbox_* keys)."""
    iou_thrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True) if iou_thrs is None else np.asarray(iou_thrs, float)
    rec_thrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
    n_img, K = len(det_results), len(det_results[0])
    T, R, A, M = len(iou_thrs), len(rec_thrs), len(area_ranges), len(max_dets)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))

    def xywh(b):
        return [float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])]
    for k in range(K):
        cells = []
        for i in range(n_img):
            ann = annotations[i]
            dt = [dict(bbox=xywh(r), score=float(r[4]), area=float((r[2] - r[0]) * (r[3] - r[1]))) for r in det_results[i][k]]
            gt = []
            sel = np.nonzero(ann['labels'] == k)[0]
            for j in sel:
                b = ann['bboxes'][j]
                ar = float(ann['areas'][j]) if 'areas' in ann else float((b[2] - b[0]) * (b[3] - b[1]))
                gt.append(dict(bbox=xywh(b), area=ar, iscrowd=0, ignore=0))
            if ann.get('labels_ignore', None) is not None:
                for j in np.nonzero(ann['labels_ignore'] == k)[0]:
                    b = ann['bboxes_ignore'][j]
                    ar = float(ann['areas_ignore'][j]) if 'areas_ignore' in ann else float((b[2] - b[0]) * (b[3] - b[1]))
                    gt.append(dict(bbox=xywh(b), area=ar, iscrowd=1, ignore=1))
            cells.append((dt, gt))
        for a, rng in enumerate(area_ranges):
            ev = [evaluate_img(dt, gt, rng, max_dets[-1], iou_thrs) for dt, gt in cells]
            ev = [e for e in ev if e is not None]
            if not ev:
                continue
            for mi, md in enumerate(max_dets):
                scores = np.concatenate([np.asarray(e['scores'][:md], dtype=float) for e in ev])
                inds = np.argsort(-scores, kind='mergesort')
                dtm = np.concatenate([e['dtm'][:, :md] for e in ev], axis=1)[:, inds]
                dt_ig = np.concatenate([e['dt_ig'][:, :md] for e in ev], axis=1)[:, inds]
                gt_ig = np.concatenate([e['gt_ig'] for e in ev])
                npig = np.count_nonzero(~gt_ig)
                if npig == 0:
                    continue
                tps = np.logical_and(dtm, ~dt_ig)
                fps = np.logical_and(~dtm, ~dt_ig)
                tp_sum = np.cumsum(tps, axis=1).astype(float)
                fp_sum = np.cumsum(fps, axis=1).astype(float)
                for t in range(T):
                    tp, fp = tp_sum[t], fp_sum[t]
                    nd = len(tp)
                    rc = tp / npig
                    pr = tp / (fp + tp + np.spacing(1))
                    q = np.zeros((R,))
                    recall[t, k, a, mi] = rc[-1] if nd else 0
                    pr = pr.tolist()
                    for i in range(nd - 1, 0, -1):
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    idx = np.searchsorted(rc, rec_thrs, side='left')
                    for ri, pi in enumerate(idx):
                        if pi < nd:
                            q[ri] = pr[pi]
                    precision[t, :, k, a, mi] = q
    return summarize(precision, recall, iou_thrs, max_dets)


def summarize(precision, recall, iou_thrs, max_dets):
    def _s(ap, iou=None, a=0, m=len(max_dets) - 1):
        if ap:
            s = precision[:, :, :, a, m]
            if iou is not None:
                s = s[np.where(np.isclose(iou_thrs, iou))[0]]
        else:
            s = recall[:, :, a, m]
            if iou is not None:
                s = s[np.where(np.isclose(iou_thrs, iou))[0]]
        return -1.0 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))
    last = len(max_dets) - 1
    stats = np.array([_s(1), _s(1, .5), _s(1, .75), _s(1, a=1), _s(1, a=2), _s(1, a=3),
                      _s(0, m=0), _s(0, m=min(1, last)), _s(0, m=last), _s(0, a=1), _s(0, a=2), _s(0, a=3)])
    out = dict(precision=precision, recall=recall, stats=stats)
    for name, i in [('mAP', 0), ('mAP_50', 1), ('mAP_75', 2), ('mAP_s', 3), ('mAP_m', 4), ('mAP_l', 5)]:
        out[f'bbox_{name}'] = float(f'{stats[i]:.3f}')                      # coco.py:536-541
    out['bbox_mAP_copypaste'] = ' '.join(f'{stats[i]:.3f}' for i in range(6))
    return out
