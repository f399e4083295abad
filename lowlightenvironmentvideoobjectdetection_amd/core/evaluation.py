"""mAP evaluation with the reference's call surface, matching on the device.

Mirror of mmdetection/mmdet/core/evaluation/mean_ap.py: `eval_map(det_results, annotations, scale_ranges, iou_thr,
dataset, logger, tpfp_fn, nproc)` -> `(mean_ap, eval_results)` with the same per-class dicts (`num_gts`, `num_dets`,
`recall`, `precision`, `ap`), `average_precision`, `print_map_summary`. The reference marks true / false positives per
class and image in a 4-process pool of Python loops (mean_ap.py:326-340); here all num_classes x num_images cells go
to the device in one `llv_eval_tpfp` launch (one wavefront per cell, csrc/eval.hip), and only the final per-class
cumulative sums and the AP integral (a few operations per detection) run in numpy on the host as in the reference.

No CPU path: without libllv.so or without a GPU, eval_map raises LlvError.
Score ties are ordered lower index first (image order, then detection order); the reference's argsort leaves them open.
"""
import ctypes as C
import logging

import numpy as np
import torch

from .._lib import LlvError, check, lib

MAX_DET_PER_CELL, MAX_GT_PER_CELL = 2048, 4096


def average_precision(recalls, precisions, mode='area'):
    """mean_ap.py:12-57: area under the monotone precision envelope, or the 11-point VOC07 average."""
    no_scale = recalls.ndim == 1
    if no_scale:
        recalls, precisions = recalls[np.newaxis, :], precisions[np.newaxis, :]
    assert recalls.shape == precisions.shape and recalls.ndim == 2
    num_scales = recalls.shape[0]
    ap = np.zeros(num_scales, dtype=np.float32)
    if mode == 'area':
        pad0 = np.zeros((num_scales, 1), dtype=recalls.dtype)
        mrec = np.hstack((pad0, recalls, pad0 + 1))
        mpre = np.hstack((pad0, precisions, pad0))
        mpre = np.maximum.accumulate(mpre[:, ::-1], axis=1)[:, ::-1]       # running max from the right
        for i in range(num_scales):
            step = np.nonzero(mrec[i, 1:] != mrec[i, :-1])[0]
            ap[i] = np.sum((mrec[i, step + 1] - mrec[i, step]) * mpre[i, step + 1])
    elif mode == '11points':
        for i in range(num_scales):
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                above = precisions[i, recalls[i, :] >= thr]
                ap[i] += above.max() if above.size > 0 else 0
            ap /= 11        # the reference divides inside the loop over scales (mean_ap.py:44-49); kept
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return ap[0] if no_scale else ap


def _dev():
    if not torch.cuda.is_available():
        raise LlvError('eval_map marks true/false positives on the MI355X (llv_eval_tpfp); there is no CPU path')
    return torch.device('cuda', torch.cuda.current_device())


def tpfp_cells(cell_dets, cell_gts, cell_gts_ignore, iou_thr=0.5, area_ranges=None, mode='default'):
    """tpfp_default / tpfp_imagenet (mean_ap.py:154-244 / :60-151) for a list of cells in one launch.

    cell_dets[i] (m_i, 5), cell_gts[i] (n_i, 4), cell_gts_ignore[i] (k_i, 4) float arrays.
    Returns [(tp_i, fp_i)], each (num_scales, m_i) float32 -- what the reference's function returns per call."""
    dev = _dev()
    ncell = len(cell_dets)
    assert len(cell_gts) == ncell and len(cell_gts_ignore) == ncell
    det_off = np.zeros(ncell + 1, dtype=np.int32)
    gt_off = np.zeros(ncell + 1, dtype=np.int32)
    det_off[1:] = np.cumsum([d.shape[0] for d in cell_dets])
    gt_off[1:] = np.cumsum([g.shape[0] + ig.shape[0] for g, ig in zip(cell_gts, cell_gts_ignore)])
    total_d, total_g = int(det_off[-1]), int(gt_off[-1])
    nscale = len(area_ranges) if area_ranges else 1
    if total_d == 0:
        return [(np.zeros((nscale, 0), np.float32), np.zeros((nscale, 0), np.float32)) for _ in range(ncell)]
    max_d = int(np.diff(det_off).max())
    max_g = int(np.diff(gt_off).max()) if ncell else 0
    if max_d > MAX_DET_PER_CELL or max_g > MAX_GT_PER_CELL:
        raise LlvError(f'eval_map: {max_d} detections / {max_g} ground-truth boxes of one class in one image exceed the '
                       f'kernel limits ({MAX_DET_PER_CELL} / {MAX_GT_PER_CELL})')
    dets = np.ascontiguousarray(np.vstack([d.reshape(-1, 5) for d in cell_dets]).astype(np.float32))
    parts, ign = [], []
    for g, ig in zip(cell_gts, cell_gts_ignore):
        parts += [g.reshape(-1, 4), ig.reshape(-1, 4)]
        ign += [np.zeros(g.shape[0], np.uint8), np.ones(ig.shape[0], np.uint8)]
    gts = np.ascontiguousarray(np.vstack(parts).astype(np.float32)) if total_g else np.zeros((1, 4), np.float32)
    ign = np.ascontiguousarray(np.concatenate(ign)) if total_g else np.zeros(1, np.uint8)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_dets, d_doff, d_gts, d_goff, d_ign = t(dets), t(det_off), t(gts), t(gt_off), t(ign)
    if area_ranges:
        rg = torch.tensor([[float(lo), float(hi)] for lo, hi in area_ranges], dtype=torch.float64, device=dev)
        rg_ptr, n_rg = rg.data_ptr(), len(area_ranges)
    else:
        rg, rg_ptr, n_rg = None, None, 0
    tp = torch.zeros((nscale, total_d), dtype=torch.uint8, device=dev)
    fp = torch.zeros((nscale, total_d), dtype=torch.uint8, device=dev)
    check(lib().llv_eval_tpfp(d_dets.data_ptr(), d_doff.data_ptr(), d_gts.data_ptr(), d_goff.data_ptr(), d_ign.data_ptr(),
                              ncell, max_d, max_g, rg_ptr, n_rg, float(iou_thr), 1 if mode == 'imagenet' else 0, total_d,
                              tp.data_ptr(), fp.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
          'llv_eval_tpfp')
    tp_h, fp_h = tp.cpu().numpy().astype(np.float32), fp.cpu().numpy().astype(np.float32)
    return [(tp_h[:, det_off[i]:det_off[i + 1]], fp_h[:, det_off[i]:det_off[i + 1]]) for i in range(ncell)]


def tpfp_default(det_bboxes, gt_bboxes, gt_bboxes_ignore=None, iou_thr=0.5, area_ranges=None):
    """mean_ap.py:154-244 for one image (one-cell launch; eval_map batches all cells instead)."""
    ig = gt_bboxes_ignore if gt_bboxes_ignore is not None else np.empty((0, 4), np.float32)
    return tpfp_cells([det_bboxes], [gt_bboxes], [ig], iou_thr, area_ranges, 'default')[0]


def tpfp_imagenet(det_bboxes, gt_bboxes, gt_bboxes_ignore=None, default_iou_thr=0.5, area_ranges=None):
    """mean_ap.py:60-151 for one image."""
    ig = gt_bboxes_ignore if gt_bboxes_ignore is not None else np.empty((0, 4), np.float32)
    return tpfp_cells([det_bboxes], [gt_bboxes], [ig], default_iou_thr, area_ranges, 'imagenet')[0]


def get_cls_results(det_results, annotations, class_id):
    """mean_ap.py:247-265."""
    cls_dets = [img_res[class_id] for img_res in det_results]
    cls_gts, cls_gts_ignore = [], []
    for ann in annotations:
        cls_gts.append(ann['bboxes'][ann['labels'] == class_id, :])
        if ann.get('labels_ignore', None) is not None:
            cls_gts_ignore.append(ann['bboxes_ignore'][ann['labels_ignore'] == class_id, :])
        else:
            cls_gts_ignore.append(np.empty((0, 4), dtype=np.float32))
    return cls_dets, cls_gts, cls_gts_ignore


def eval_map(det_results, annotations, scale_ranges=None, iou_thr=0.5, dataset=None, logger=None, tpfp_fn=None, nproc=4):
    """mean_ap.py:268-400. det_results: per image, per class (k, 5) arrays (the detectors' `bbox2result` lists);
    annotations: per image dict(bboxes (n,4), labels (n,), optional bboxes_ignore / labels_ignore).
    `tpfp_fn`: None (by `dataset`: 'det' / 'vid' -> ImageNet rule, else PASCAL rule), or this module's tpfp_default /
    tpfp_imagenet; other callables are refused (the matching runs on the device). `nproc` is accepted and unused."""
    assert len(det_results) == len(annotations)
    num_imgs = len(det_results)
    num_scales = len(scale_ranges) if scale_ranges is not None else 1
    num_classes = len(det_results[0])
    area_ranges = [(rg[0] ** 2, rg[1] ** 2) for rg in scale_ranges] if scale_ranges is not None else None
    if tpfp_fn is None:
        mode = 'imagenet' if dataset in ['det', 'vid'] else 'default'
    elif tpfp_fn is tpfp_default:
        mode = 'default'
    elif tpfp_fn is tpfp_imagenet:
        mode = 'imagenet'
    elif not callable(tpfp_fn):
        raise ValueError(f'tpfp_fn has to be a function or None, but got {tpfp_fn}')
    else:
        raise LlvError('eval_map: only tpfp_default / tpfp_imagenet of this module run on the device')

    # every (class, image) cell, class-major: one launch
    per_class = [get_cls_results(det_results, annotations, c) for c in range(num_classes)]
    cells = tpfp_cells([d for cd, _, _ in per_class for d in cd], [g for _, cg, _ in per_class for g in cg],
                       [ig for _, _, ci in per_class for ig in ci], iou_thr, area_ranges, mode)
    eval_results = []
    for c in range(num_classes):
        cls_dets, cls_gts, _ = per_class[c]
        tp, fp = tuple(zip(*cells[c * num_imgs:(c + 1) * num_imgs]))
        num_gts = np.zeros(num_scales, dtype=int)
        for bbox in cls_gts:
            if area_ranges is None:
                num_gts[0] += bbox.shape[0]
            else:
                gt_areas = (bbox[:, 2] - bbox[:, 0]) * (bbox[:, 3] - bbox[:, 1])
                for k, (min_area, max_area) in enumerate(area_ranges):
                    num_gts[k] += np.sum((gt_areas >= min_area) & (gt_areas < max_area))
        cls_dets = np.vstack(cls_dets)
        num_dets = cls_dets.shape[0]
        order = np.argsort(-cls_dets[:, -1], kind='stable')
        tp = np.cumsum(np.hstack(tp)[:, order], axis=1)
        fp = np.cumsum(np.hstack(fp)[:, order], axis=1)
        eps = np.finfo(np.float32).eps
        recalls = tp / np.maximum(num_gts[:, np.newaxis], eps)
        precisions = tp / np.maximum(tp + fp, eps)
        if scale_ranges is None:
            recalls, precisions, num_gts = recalls[0, :], precisions[0, :], num_gts.item()
        ap = average_precision(recalls, precisions, 'area' if dataset != 'voc07' else '11points')
        eval_results.append({'num_gts': num_gts, 'num_dets': num_dets, 'recall': recalls, 'precision': precisions, 'ap': ap})
    if scale_ranges is not None:
        all_ap = np.vstack([r['ap'] for r in eval_results])
        all_num_gts = np.vstack([r['num_gts'] for r in eval_results])
        mean_ap = [all_ap[all_num_gts[:, i] > 0, i].mean() if np.any(all_num_gts[:, i] > 0) else 0.0
                   for i in range(num_scales)]
    else:
        aps = [r['ap'] for r in eval_results if r['num_gts'] > 0]
        mean_ap = np.array(aps).mean().item() if aps else 0.0
    print_map_summary(mean_ap, eval_results, dataset, area_ranges, logger=logger)
    return mean_ap, eval_results


def _log(msg, logger):
    """mmcv.utils.print_log: None -> print, 'silent' -> nothing, Logger -> info, str -> logging.getLogger(str)."""
    if logger is None:
        print(msg)
    elif isinstance(logger, logging.Logger):
        logger.info(msg)
    elif logger == 'silent':
        pass
    elif isinstance(logger, str):
        logging.getLogger(logger).info(msg)
    else:
        raise TypeError(f'logger should be either a logging.Logger object, str, "silent" or None, but got {type(logger)}')


def print_map_summary(mean_ap, results, dataset=None, scale_ranges=None, logger=None):
    """mean_ap.py:403-469 (plain-text table; terminaltables is not a dependency here)."""
    if logger == 'silent':
        return
    num_scales = len(results[0]['ap']) if isinstance(results[0]['ap'], np.ndarray) else 1
    if scale_ranges is not None:
        assert len(scale_ranges) == num_scales
    num_classes = len(results)
    recalls = np.zeros((num_scales, num_classes), dtype=np.float32)
    aps = np.zeros((num_scales, num_classes), dtype=np.float32)
    num_gts = np.zeros((num_scales, num_classes), dtype=int)
    for i, r in enumerate(results):
        if r['recall'].size > 0:
            recalls[:, i] = np.array(r['recall'], ndmin=2)[:, -1]
        aps[:, i] = r['ap']
        num_gts[:, i] = r['num_gts']
    if dataset is None or isinstance(dataset, str):
        label_names = [str(i) for i in range(num_classes)]          # class-name tables (class_names.py) are not carried
    else:
        label_names = list(dataset)
    if not isinstance(mean_ap, list):
        mean_ap = [mean_ap]
    for i in range(num_scales):
        if scale_ranges is not None:
            _log(f'Scale range {scale_ranges[i]}', logger)
        rows = [['class', 'gts', 'dets', 'recall', 'ap']]
        rows += [[str(label_names[j]), str(num_gts[i, j]), str(results[j]['num_dets']), f'{recalls[i, j]:.3f}', f'{aps[i, j]:.3f}']
                 for j in range(num_classes)]
        rows.append(['mAP', '', '', '', f'{mean_ap[i]:.3f}'])
        wid = [max(len(r[c]) for r in rows) for c in range(5)]
        line = '+' + '+'.join('-' * (w + 2) for w in wid) + '+'
        fmt = lambda r: '| ' + ' | '.join(v.ljust(w) for v, w in zip(r, wid)) + ' |'
        out = [line, fmt(rows[0]), line] + [fmt(r) for r in rows[1:-1]] + [line, fmt(rows[-1]), line]
        _log('\n' + '\n'.join(out), logger)
