"""`CocoVideoDataset.evaluate` -- the glue between the video detectors' results and the COCO box metric:

  CocoVideoDataset.load_annotations / get_ann_info / evaluate   mmtracking/mmtrack/datasets/coco_video_dataset.py:41-76,318-372,374-461
  CocoDataset._parse_ann_info / evaluate                        mmdetection/mmdet/datasets/coco.py:118-175,347-548
  pycocotools COCOeval (un-vendored)                             its matching is core.coco_eval.coco_bbox_eval (llv_coco_match on the device)

Only what evaluation needs is here: the annotation file (COCO-video JSON: `images` with `video_id` / `frame_id`, `annotations` with
`bbox` = x, y, w, h, `area`, `category_id`, `iscrowd`, `instance_id`, `categories`) held in memory, the per-frame ground truth as the
reference files it, and `evaluate(results, metric='bbox')` with the reference's result keys. The data-loading side (pipelines, samplers,
image decoding) is outside SURVEY section 8; `metric='track'` (CLEAR MOT of the trackers, which are out of scope) raises."""
import json

import numpy as np

from .core.coco_eval import coco_bbox_eval


class CocoVideoDataset:
    CLASSES = None

    def __init__(self, ann_file, classes=None, load_as_video=True, test_mode=True):
        data = ann_file if isinstance(ann_file, dict) else json.load(open(ann_file))
        self.load_as_video, self.test_mode = load_as_video, test_mode
        cats = list(data.get('categories', []))
        self.CLASSES = tuple(classes) if classes is not None else (self.CLASSES or tuple(c['name'] for c in cats))
        # coco.py:57-58: category ids in the order of CLASSES
        by_name = {c['name']: c['id'] for c in cats}
        self.cat_ids = [by_name[n] for n in self.CLASSES if n in by_name]
        self.cat2label = {cat_id: i for i, cat_id in enumerate(self.cat_ids)}
        images = list(data.get('images', []))
        if load_as_video:
            # coco_video_dataset.py:62-76: videos in order, the frames of a video in frame order
            vids = {}
            for im in images:
                vids.setdefault(im.get('video_id', 0), []).append(im)
            images = [im for v in sorted(vids) for im in sorted(vids[v], key=lambda t: t.get('frame_id', 0))]
        self.data_infos = []
        for im in images:
            info = dict(im)
            info['filename'] = im.get('file_name', '')
            self.data_infos.append(info)
        self.img_ids = [im['id'] for im in self.data_infos]
        self._anns = {}
        for a in data.get('annotations', []):
            self._anns.setdefault(a['image_id'], []).append(a)

    def __len__(self):
        return len(self.data_infos)

    def get_ann_info(self, idx):
        """coco.py:118-175 + coco_video_dataset.py:318-372: the frame's ground truth as training and the evaluators read it. Degenerate
        boxes, boxes outside the image, unknown categories and `ignore`d annotations are dropped; crowd regions go to bboxes_ignore."""
        info = self.data_infos[idx]
        gt_bboxes, gt_labels, gt_ignore, ids = [], [], [], []
        for ann in self._anns.get(info['id'], []):
            if ann.get('ignore', False):
                continue
            x1, y1, w, h = ann['bbox']
            inter_w = max(0, min(x1 + w, info['width']) - max(x1, 0))
            inter_h = max(0, min(y1 + h, info['height']) - max(y1, 0))
            if inter_w * inter_h == 0:
                continue
            if ann.get('area', w * h) <= 0 or w < 1 or h < 1:
                continue
            if ann['category_id'] not in self.cat_ids:
                continue
            bbox = [x1, y1, x1 + w, y1 + h]
            if ann.get('iscrowd', False):
                gt_ignore.append(bbox)
            else:
                gt_bboxes.append(bbox)
                gt_labels.append(self.cat2label[ann['category_id']])
                ids.append(ann.get('instance_id', len(ids)))
        out = dict(bboxes=np.array(gt_bboxes, dtype=np.float32).reshape(-1, 4), labels=np.array(gt_labels, dtype=np.int64),
                   bboxes_ignore=np.array(gt_ignore, dtype=np.float32).reshape(-1, 4), masks=[], seg_map=info['filename'].replace('jpg', 'png'))
        out['instance_ids'] = np.array(ids).astype(np.int64) if self.load_as_video else np.arange(len(gt_labels))
        return out

    def _eval_annotations(self):
        """What COCOeval holds as ground truth: every annotation of a known category of every image (the file as it is -- `_parse_ann_info`'s
        filters apply to training, not to the evaluator), `area` from the file, crowd regions as the ignore set."""
        out = []
        for info in self.data_infos:
            b, l, a, bi, li, ai = [], [], [], [], [], []
            for ann in self._anns.get(info['id'], []):
                if ann['category_id'] not in self.cat2label:
                    continue
                x1, y1, w, h = ann['bbox']
                box, lab, area = [x1, y1, x1 + w, y1 + h], self.cat2label[ann['category_id']], ann.get('area', w * h)
                if ann.get('iscrowd', 0):
                    bi.append(box); li.append(lab); ai.append(area)
                else:
                    b.append(box); l.append(lab); a.append(area)
            out.append(dict(bboxes=np.array(b, dtype=np.float64).reshape(-1, 4), labels=np.array(l, dtype=np.int64), areas=np.array(a, dtype=np.float64),
                            bboxes_ignore=np.array(bi, dtype=np.float64).reshape(-1, 4), labels_ignore=np.array(li, dtype=np.int64),
                            areas_ignore=np.array(ai, dtype=np.float64)))
        return out

    def evaluate(self, results, metric=['bbox', 'track'], logger=None,
                 bbox_kwargs=dict(classwise=False, proposal_nums=(100, 300, 1000), iou_thrs=None, metric_items=None), track_kwargs=None):
        """coco_video_dataset.py:374-461 -> coco.py:347-548. results: {'bbox_results': [per frame: list[num_classes] of (k, 5)]} (what
        the video detectors' simple_test returns, collected over the dataset) or that list itself. Returns {'bbox_mAP', 'bbox_mAP_50',
        'bbox_mAP_75', 'bbox_mAP_s', 'bbox_mAP_m', 'bbox_mAP_l', 'bbox_mAP_copypaste'} (or the `metric_items` subset)."""
        if isinstance(metric, list):
            metrics = metric
        elif isinstance(metric, str):
            metrics = [metric]
        else:
            raise TypeError('metric must be a list or a str.')
        for m in metrics:
            if m not in ['bbox', 'segm', 'track']:
                raise KeyError(f'metric {m} is not supported.')
        if 'track' in metrics:
            raise NotImplementedError("metric 'track' (CLEAR MOT of the trackers) is outside the detection hot path; pass metric='bbox'")
        if 'segm' in metrics:
            raise NotImplementedError("metric 'segm': no mask head on this path")
        if isinstance(results, dict):
            dets = results['bbox_results']
        elif isinstance(results, list):
            dets = results
        else:
            raise TypeError('Results must be a dict or a list.')
        assert len(dets) == len(self), f'The length of results is not equal to the dataset len: {len(dets)} != {len(self)}'   # coco.py:323
        kw = dict(bbox_kwargs or {})
        res = coco_bbox_eval(dets, self._eval_annotations(), iou_thrs=kw.get('iou_thrs'), max_dets=tuple(kw.get('proposal_nums', (100, 300, 1000))))
        items = kw.get('metric_items') or ['mAP', 'mAP_50', 'mAP_75', 'mAP_s', 'mAP_m', 'mAP_l']
        out = {f'bbox_{k}': res[f'bbox_{k}'] for k in items}
        out['bbox_mAP_copypaste'] = res['bbox_mAP_copypaste']
        if kw.get('classwise', False):                  # coco.py:500-528: AP per category = mean of its precision over IoU and recall
            pr = res['precision']
            last = pr.shape[-1] - 1
            per = []
            for k, name in enumerate(self.CLASSES[:pr.shape[2]]):
                p = pr[:, :, k, 0, last]
                p = p[p > -1]
                per.append((name, float(np.mean(p)) if p.size else float('nan')))
            out['bbox_classwise'] = per
        return out
