"""COCO-protocol box evaluation: the oracle (oracle/coco_eval.py, restatement of pycocotools' COCOeval as mmdet drives it) on
hand-computed cases (CPU), the device path (core/coco_eval.py -> llv_coco_match) against the oracle (GPU): equal precision /
recall arrays. pycocotools is not installed here and the reference holds no vector for this path: parity unpinned."""
import numpy as np
import pytest
import torch


def _b(x, y, w, h, s=None):
    return [x, y, x + w, y + h] + ([] if s is None else [s])


E5 = np.zeros((0, 5), np.float32)


def test_oracle_coco_ap_by_hand():
    from oracle.coco_eval import coco_bbox_eval
    # one image, one category, two 100x100 gts (area 10^4: 'large'); detections in score order:
    #   .9 exact hit on gt0; .8 far away; .7 on gt1 with IoU 0.62
    ann = [dict(bboxes=np.array([_b(0, 0, 100, 100), _b(300, 0, 100, 100)], np.float64), labels=np.array([0]  * 2))]
    det = [[np.array([_b(0, 0, 100, 100, .9), _b(600, 600, 50, 50, .8), _b(300, 0, 100, 62, .7)], np.float64)]]
    out = coco_bbox_eval(det, ann)
    # IoU thresholds .5 .55 .6: tp fp tp -> recall .5 .5 1, precision 1 .5 2/3 -> envelope 1 2/3 2/3;
    # 101 recall points: r <= .5 (51 of them) read 1, the other 50 read 2/3
    ap_lo = (51 * 1.0 + 50 * 2 / 3) / 101
    # thresholds .65 .. .95: the third detection is a false positive: recall stops at .5 -> 51 points at 1, 50 at 0
    ap_hi = 51 / 101
    assert out['stats'][0] == pytest.approx((3 * ap_lo + 7 * ap_hi) / 10, abs=1e-9)
    assert out['stats'][1] == pytest.approx(ap_lo, abs=1e-9) and out['stats'][2] == pytest.approx(ap_hi, abs=1e-9)
    assert out['stats'][8] == pytest.approx((3 * 1.0 + 7 * 0.5) / 10, abs=1e-9)          # AR at the last maxDets
    assert out['stats'][3] == -1 and out['stats'][4] == -1                               # no small / medium ground truth
    # 'large' (area >= 96^2): both gts count; the .8 box (2500 px^2) and, above IoU .6, the unmatched .7 box (6200 px^2) lie outside
    # the range and are ignored instead of counting as false positives: AP 1 at the three low thresholds, 51/101 at the seven others
    assert out['stats'][5] == pytest.approx((3 * 1.0 + 7 * 51 / 101) / 10, abs=1e-9)
    assert out['bbox_mAP'] == float(f"{out['stats'][0]:.3f}") and out['bbox_mAP_copypaste'].count(' ') == 5


def test_oracle_coco_crowd_area_and_maxdets_rules():
    from oracle.coco_eval import coco_bbox_eval, evaluate_img
    thr = np.array([0.5])
    gt = [dict(bbox=[0, 0, 10, 10], area=100, iscrowd=0, ignore=0), dict(bbox=[50, 50, 40, 40], area=1600, iscrowd=1, ignore=1)]
    dt = [dict(bbox=[50, 50, 10, 10], score=.9, area=100),      # inside the crowd region: IoU = inter / det area = 1 -> matched, ignored
          dict(bbox=[60, 60, 10, 10], score=.8, area=100),      # the same crowd region again: crowd gts can be matched repeatedly
          dict(bbox=[0, 0, 10, 10], score=.7, area=100),        # the regular gt: true positive
          dict(bbox=[0, 1, 10, 10], score=.6, area=100)]        # the regular gt is taken: false positive
    e = evaluate_img(dt, gt, (0, 1e10), 100, thr)
    assert e['dtm'][0].tolist() == [True, True, True, False] and e['dt_ig'][0].tolist() == [True, True, False, False]
    assert e['gt_ig'].tolist() == [False, True]
    # area range 'medium' (1024 .. 9216): the 100 px^2 gt is ignored, so is its match; unmatched small detections are ignored too
    e = evaluate_img(dt, gt, (32 ** 2, 96 ** 2), 100, thr)
    assert e['dt_ig'][0].tolist() == [True, True, True, True]
    # maxDet truncation keeps the best-scoring detections
    e = evaluate_img(dt, gt, (0, 1e10), 2, thr)
    assert e['scores'] == [.9, .8] and e['dtm'].shape == (1, 2)
    # a stable sort puts ignored gts last; a detection that reached a regular gt does not look at ignored ones (the `break`)
    gt2 = [dict(bbox=[0, 0, 10, 10], area=100, iscrowd=1, ignore=1), dict(bbox=[0, 0, 10, 9], area=90, iscrowd=0, ignore=0)]
    e = evaluate_img([dict(bbox=[0, 0, 10, 10], score=.9, area=100)], gt2, (0, 1e10), 100, thr)
    assert e['dtm'][0].tolist() == [True] and e['dt_ig'][0].tolist() == [False]       # IoU .9 with the regular gt wins over IoU 1 with the crowd
    # whole evaluation with a crowd region given the way mmdet's datasets hold it (bboxes_ignore)
    ann = [dict(bboxes=np.array([_b(0, 0, 10, 10)], np.float64), labels=np.array([0]),
                bboxes_ignore=np.array([_b(50, 50, 40, 40)], np.float64), labels_ignore=np.array([0]))]
    det = [[np.array([_b(50, 50, 10, 10, .9), _b(0, 0, 10, 10, .7), _b(0, 1, 10, 10, .6)], np.float64)]]
    out = coco_bbox_eval(det, ann)
    assert out['stats'][1] == pytest.approx(1.0)          # at IoU .5: ignored, tp, fp -> precision 1 up to recall 1
    assert out['stats'][3] == pytest.approx(out['stats'][0])   # the one counted gt is 'small'


def test_product_coco_eval_has_no_cpu_path():
    from lowlightenvironmentvideoobjectdetection_amd._lib import LlvError
    from lowlightenvironmentvideoobjectdetection_amd.core import coco_eval
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(LlvError):
        coco_eval.coco_bbox_eval([[E5]], [dict(bboxes=np.zeros((0, 4)), labels=np.zeros(0, int))])


def _random_set(seed, n_img, K):
    g = np.random.RandomState(seed)
    dets, anns = [], []
    for _ in range(n_img):
        n = g.randint(0, 8)
        xy = g.rand(n, 2) * 500
        wh = np.exp(g.rand(n, 2) * 4.5 + 1.5)                 # 4 .. 400 px: small, medium and large objects
        gt = np.hstack([xy, xy + wh])
        labels = g.randint(0, K, n)
        k = g.randint(0, 3)
        xy2 = g.rand(k, 2) * 400
        ann = dict(bboxes=gt, labels=labels, areas=(wh[:, 0] * wh[:, 1]) * (0.5 + 0.5 * g.rand(n)),       # segmentation area < box area
                   bboxes_ignore=np.hstack([xy2, xy2 + g.rand(k, 2) * 150 + 20]), labels_ignore=g.randint(0, K, k))
        anns.append(ann)
        per = []
        for c in range(K):
            src = gt[labels == c]
            rows = []
            for _ in range(g.randint(0, 12)):
                if len(src) and g.rand() < 0.7:
                    b = src[g.randint(len(src))] + g.randn(4) * g.choice([0.5, 3.0, 12.0])
                    b[2:] = np.maximum(b[2:], b[:2] + 1)
                elif k and g.rand() < 0.3:
                    p = ann['bboxes_ignore'][g.randint(k)][:2] + g.rand(2) * 20
                    b = np.concatenate([p, p + g.rand(2) * 30 + 3])
                else:
                    p = g.rand(2) * 500
                    b = np.concatenate([p, p + np.exp(g.rand(2) * 4 + 1.5)])
                rows.append(np.concatenate([b, [np.round(g.rand(), 2)]]))      # two-digit scores: ties across images
            per.append(np.array(rows, np.float64).reshape(-1, 5))
        dets.append(per)
    return dets, anns


@pytest.mark.gpu
@pytest.mark.parametrize('seed,n_img,K,max_dets', [(1, 60, 5, (100, 300, 1000)), (2, 25, 3, (1, 10, 100)), (3, 40, 1, (5,))])
def test_device_coco_eval_equals_oracle(dev, seed, n_img, K, max_dets):
    from lowlightenvironmentvideoobjectdetection_amd.core.coco_eval import coco_bbox_eval
    from oracle import coco_eval as ref
    dets, anns = _random_set(seed, n_img, K)
    got = coco_bbox_eval(dets, anns, max_dets=max_dets)
    exp = ref.coco_bbox_eval(dets, anns, max_dets=max_dets)
    assert np.array_equal(got['precision'], exp['precision'])
    assert np.array_equal(got['recall'], exp['recall'])
    assert np.array_equal(got['stats'], exp['stats'])
    assert exp['stats'][1] > 0.05 and got['bbox_mAP_copypaste'] == exp['bbox_mAP_copypaste']


@pytest.mark.gpu
def test_device_coco_eval_edge_cases(dev):
    from lowlightenvironmentvideoobjectdetection_amd.core.coco_eval import coco_bbox_eval
    from oracle import coco_eval as ref
    # no detections at all; no ground truth at all; a category that never occurs
    ann = [dict(bboxes=np.array([_b(0, 0, 50, 50)], np.float64), labels=np.array([1]))]
    out = coco_bbox_eval([[E5, E5, E5]], ann)
    assert out['stats'][0] == 0.0 and out['stats'][8] == 0.0
    out = coco_bbox_eval([[np.array([_b(0, 0, 5, 5, .5)], np.float64)]], [dict(bboxes=np.zeros((0, 4)), labels=np.zeros(0, int))])
    assert out['stats'][0] == -1.0
    ann = [dict(bboxes=np.array([_b(0, 0, 50, 50)], np.float64), labels=np.array([0]))]
    det = [[np.array([_b(0, 0, 50, 50, .5)], np.float64), E5]]
    got, exp = coco_bbox_eval(det, ann), ref.coco_bbox_eval(det, ann)
    assert got['stats'][0] == pytest.approx(1.0, abs=1e-12) and np.array_equal(got['precision'], exp['precision'])   # tp / (tp + fp + eps)
