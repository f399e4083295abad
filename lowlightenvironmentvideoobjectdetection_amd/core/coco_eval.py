"""COCO-protocol box evaluation (the metric `CocoVideoDataset.evaluate(metric='bbox')` reports,
mmtracking/mmtrack/datasets/coco_video_dataset.py:374-461 -> mmdetection/mmdet/datasets/coco.py:347-548 -> pycocotools COCOeval),
with the per-image matching on the device.

`coco_bbox_eval(det_results, annotations, ...)` takes what the reference's datasets hold in memory (per image the detector's
`bbox2result` lists; per image `dict(bboxes, labels[, areas][, bboxes_ignore, labels_ignore, areas_ignore])`, crowd regions being the
`*_ignore` entries as `CocoDataset._parse_ann_info` files them, coco.py:118-175) and returns mmdet's keys (`bbox_mAP`, `bbox_mAP_50`,
`bbox_mAP_75`, `bbox_mAP_s/m/l`, `bbox_mAP_copypaste`) plus the COCOeval arrays (`precision` [T,R,K,A,M], `recall` [T,K,A,M], `stats`).
COCOeval.evaluateImg — a Python triple loop per (image, category, area range) — is one `llv_coco_match` launch over all
(category, image) cells, one lane per (area range, IoU threshold); `accumulate` / `summarize` are a few numpy passes on the host.
mmdet's settings: maxDets = proposal_nums = (100, 300, 1000), AP summarised at the last of them.

No CPU path: raises LlvError without libllv.so or without a GPU.
"""
import ctypes as C

import numpy as np
import torch

from .._lib import LlvError, check, lib

AREA_RANGES = [(0 ** 2, 1e5 ** 2), (0 ** 2, 32 ** 2), (32 ** 2, 96 ** 2), (96 ** 2, 1e5 ** 2)]      # all, small, medium, large
MAX_GT_PER_CELL = 256


def _xywh(b):
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4)
    return np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1)


def coco_bbox_eval(det_results, annotations, iou_thrs=None, max_dets=(100, 300, 1000), area_ranges=AREA_RANGES):
    if not torch.cuda.is_available():
        raise LlvError('coco_bbox_eval matches detections to ground truth on the MI355X (llv_coco_match); there is no CPU path')
    dev = torch.device('cuda', torch.cuda.current_device())
    iou_thrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True) if iou_thrs is None \
        else np.asarray(iou_thrs, dtype=np.float64)
    rec_thrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
    max_dets = tuple(int(m) for m in max_dets)
    n_img, K = len(det_results), len(det_results[0])
    T, R, A, M = len(iou_thrs), len(rec_thrs), len(area_ranges), len(max_dets)
    if A * T > 64:
        raise LlvError('coco_bbox_eval: at most 64 (area range, IoU threshold) pairs per launch')

    # ---- cells, category-major: detections in score-descending order (stable), truncated to the largest maxDets
    d_box, d_score, d_pos, det_cnt = [], [], [], []
    g_box, g_area, g_crowd, gt_cnt, cell_used = [], [], [], [], []
    for k in range(K):
        for i in range(n_img):
            r = np.asarray(det_results[i][k], dtype=np.float64).reshape(-1, 5)
            order = np.argsort(-r[:, 4], kind='mergesort')[:max_dets[-1]]
            r = r[order]
            d_box.append(_xywh(r[:, :4])); d_score.append(r[:, 4]); d_pos.append(np.arange(len(r))); det_cnt.append(len(r))
            ann = annotations[i]
            sel = np.nonzero(np.asarray(ann['labels']) == k)[0]
            gb = _xywh(np.asarray(ann['bboxes'], dtype=np.float64).reshape(-1, 4)[sel])
            ga = np.asarray(ann['areas'], dtype=np.float64)[sel] if 'areas' in ann else gb[:, 2] * gb[:, 3]
            gc = np.zeros(len(sel), np.uint8)
            if ann.get('labels_ignore', None) is not None:
                si = np.nonzero(np.asarray(ann['labels_ignore']) == k)[0]
                ib = _xywh(np.asarray(ann['bboxes_ignore'], dtype=np.float64).reshape(-1, 4)[si])
                ia = np.asarray(ann['areas_ignore'], dtype=np.float64)[si] if 'areas_ignore' in ann else ib[:, 2] * ib[:, 3]
                gb, ga, gc = np.vstack([gb, ib]), np.concatenate([ga, ia]), np.concatenate([gc, np.ones(len(si), np.uint8)])
            g_box.append(gb); g_area.append(ga); g_crowd.append(gc); gt_cnt.append(len(gb))
            cell_used.append(len(r) > 0 or len(gb) > 0)              # evaluateImg returns None for an empty cell
    n_cells = K * n_img
    det_off = np.zeros(n_cells + 1, np.int32); det_off[1:] = np.cumsum(det_cnt)
    gt_off = np.zeros(n_cells + 1, np.int32); gt_off[1:] = np.cumsum(gt_cnt)
    total_d, total_g = int(det_off[-1]), int(gt_off[-1])
    if gt_cnt and max(gt_cnt) > MAX_GT_PER_CELL:
        raise LlvError(f'coco_bbox_eval: {max(gt_cnt)} ground-truth boxes of one category in one image (kernel limit {MAX_GT_PER_CELL})')
    dets = np.ascontiguousarray(np.vstack(d_box)) if total_d else np.zeros((1, 4))
    scores = np.concatenate(d_score) if total_d else np.zeros(0)
    pos = np.concatenate(d_pos) if total_d else np.zeros(0, np.int64)
    gts = np.ascontiguousarray(np.vstack(g_box)) if total_g else np.zeros((1, 4))
    gareas = np.ascontiguousarray(np.concatenate(g_area)) if total_g else np.zeros(1)
    gcrowd = np.ascontiguousarray(np.concatenate(g_crowd)) if total_g else np.zeros(1, np.uint8)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_dets, d_doff, d_gts, d_ga, d_gc, d_goff = t(dets), t(det_off), t(gts), t(gareas), t(gcrowd), t(gt_off)
    d_rng = t(np.asarray(area_ranges, dtype=np.float64).reshape(-1, 2))
    d_thr = t(iou_thrs.astype(np.float64))
    matched = torch.zeros((A, T, max(total_d, 1)), dtype=torch.uint8, device=dev)
    ignored = torch.zeros((A, T, max(total_d, 1)), dtype=torch.uint8, device=dev)
    gt_valid = torch.zeros((A, n_cells), dtype=torch.int32, device=dev)
    # ignore flag of a ground truth = its iscrowd flag (cocoeval.py _prepare), so the crowd array is passed for both
    check(lib().llv_coco_match(d_dets.data_ptr(), d_doff.data_ptr(), d_gts.data_ptr(), d_ga.data_ptr(), d_gc.data_ptr(), d_gc.data_ptr(),
                               d_goff.data_ptr(), n_cells, max(gt_cnt) if gt_cnt else 0, d_rng.data_ptr(), A, d_thr.data_ptr(), T,
                               total_d, matched.data_ptr(), ignored.data_ptr(), gt_valid.data_ptr(),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'llv_coco_match')
    matched = matched.cpu().numpy().astype(bool)[:, :, :total_d]
    ignored = ignored.cpu().numpy().astype(bool)[:, :, :total_d]
    gt_valid = gt_valid.cpu().numpy()

    # ---- accumulate (cocoeval.py accumulate): per category, area range and maxDet
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    used = np.asarray(cell_used, dtype=bool)
    for k in range(K):
        c0, c1 = k * n_img, (k + 1) * n_img
        if not used[c0:c1].any():
            continue
        lo, hi = det_off[c0], det_off[c1]
        for a in range(A):
            npig = int(gt_valid[a, c0:c1].sum())
            if npig == 0:
                continue
            for mi, md in enumerate(max_dets):
                keep = pos[lo:hi] < md
                sc = scores[lo:hi][keep]
                inds = np.argsort(-sc, kind='mergesort')
                dtm = matched[a][:, lo:hi][:, keep][:, inds]
                dtig = ignored[a][:, lo:hi][:, keep][:, inds]
                tp_sum = np.cumsum(np.logical_and(dtm, ~dtig), axis=1).astype(np.float64)
                fp_sum = np.cumsum(np.logical_and(~dtm, ~dtig), axis=1).astype(np.float64)
                nd = tp_sum.shape[1]
                for ti in range(T):
                    tp, fp = tp_sum[ti], fp_sum[ti]
                    rc = tp / npig
                    pr = tp / (fp + tp + np.spacing(1))
                    recall[ti, k, a, mi] = rc[-1] if nd else 0
                    pr = np.maximum.accumulate(pr[::-1])[::-1]                   # monotone from the right
                    idx = np.searchsorted(rc, rec_thrs, side='left')
                    q = np.zeros((R,))
                    ok = idx < nd
                    q[ok] = pr[idx[ok]]
                    precision[ti, :, k, a, mi] = q
    return _summarize(precision, recall, iou_thrs, max_dets)


def _summarize(precision, recall, iou_thrs, max_dets):
    """cocoeval.py summarize / _summarizeDets with mmdet's maxDets (coco.py:454-456) and its result keys (coco.py:491-541)."""
    last = len(max_dets) - 1

    def one(ap, iou=None, a=0, m=last):
        s = precision[:, :, :, a, m] if ap else recall[:, :, a, m]
        if iou is not None:
            s = s[np.where(np.isclose(iou_thrs, iou))[0]]
        s = s[s > -1]
        return -1.0 if s.size == 0 else float(np.mean(s))
    stats = np.array([one(1), one(1, .5), one(1, .75), one(1, a=1), one(1, a=2), one(1, a=3),
                      one(0, m=0), one(0, m=min(1, last)), one(0, m=last), one(0, a=1), one(0, a=2), one(0, a=3)])
    out = dict(precision=precision, recall=recall, stats=stats)
    for name, i in [('mAP', 0), ('mAP_50', 1), ('mAP_75', 2), ('mAP_s', 3), ('mAP_m', 4), ('mAP_l', 5)]:
        out[f'bbox_{name}'] = float(f'{stats[i]:.3f}')
    out['bbox_mAP_copypaste'] = ' '.join(f'{stats[i]:.3f}' for i in range(6))
    return out
