"""mAP evaluation: the oracle (oracle/evaluation.py, numpy restatement of
mmdetection/mmdet/core/evaluation/mean_ap.py) against hand-computed cases on the CPU; the device path
(core/evaluation.py -> llv_eval_tpfp) against the oracle on the GPU, bit-exact marks and equal AP.

The reference holds no known-answer test for mean_ap.py (parity unpinned); the numbers below are worked out by hand
from the rules in mean_ap.py:60-244 and the precision/recall definitions in :349-366."""
import numpy as np
import pytest
import torch


def _box(x, y, w, h, s=None):
    return [x, y, x + w, y + h] + ([] if s is None else [s])


# ---------------------------------------------------------------- oracle, CPU
def test_oracle_average_precision_by_hand():
    from oracle.evaluation import average_precision
    # detections in score order: TP, FP, TP with 2 gts -> recall .5 .5 1, precision 1 .5 2/3
    r = np.array([0.5, 0.5, 1.0]); p = np.array([1.0, 0.5, 2 / 3])
    # area: envelope 1 on (0,.5], 2/3 on (.5,1] -> .5*1 + .5*2/3
    assert average_precision(r, p, 'area') == pytest.approx(0.5 + 1 / 3, abs=1e-6)
    # 11 points: thresholds 0..0.5 -> max precision 1 (6 points), 0.6..1.0 -> 2/3 (5 points)
    assert average_precision(r, p, '11points') == pytest.approx((6 * 1.0 + 5 * 2 / 3) / 11, abs=1e-6)
    # two scales at once
    ap2 = average_precision(np.stack([r, r]), np.stack([p, np.ones(3)]), 'area')
    assert ap2.shape == (2,) and ap2[1] == pytest.approx(1.0)


def test_oracle_tpfp_default_rules():
    from oracle.evaluation import tpfp_default
    gts = np.array([_box(10, 10, 20, 20), _box(100, 100, 40, 40)], np.float32)
    ign = np.array([_box(200, 200, 30, 30)], np.float32)
    dets = np.array([
        _box(10, 10, 20, 20, 0.9),      # exact hit on gt0 -> tp
        _box(11, 11, 20, 20, 0.8),      # IoU 361/439 = .82 with gt0, already covered -> fp (PASCAL rule)
        _box(100, 100, 40, 30, 0.7),    # IoU .75 with gt1 -> tp
        _box(200, 200, 30, 30, 0.6),    # hits the ignored box -> neither
        _box(300, 300, 10, 10, 0.5),    # hits nothing -> fp
        _box(100, 100, 40, 10, 0.4),    # IoU .25 with gt1 (< .5) -> fp
    ], np.float32)
    tp, fp = tpfp_default(dets, gts, ign, 0.5)
    assert tp.tolist() == [[1, 0, 1, 0, 0, 0]] and fp.tolist() == [[0, 1, 0, 0, 1, 1]]
    # area ranges: gt0 is 400 px^2 (small), gt1 1600 (medium); detections outside the range that match nothing are dropped
    tp, fp = tpfp_default(dets, gts, ign, 0.5, [(0, 1024), (1024, 1e10)])
    assert tp.tolist() == [[1, 0, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0]]
    # scale 0: det1 -> covered gt0 -> fp; det2 matches out-of-range gt1 -> neither; det4 (100 px^2) fp; det5 (400) fp
    # scale 1: det0/det1 match out-of-range gt0 -> neither; det4/det5 are small unmatched -> dropped
    assert fp.tolist() == [[0, 1, 0, 0, 1, 1], [0, 0, 0, 0, 0, 0]]
    # no gt at all: everything is a false positive
    tp, fp = tpfp_default(dets, np.zeros((0, 4), np.float32), None, 0.5)
    assert tp.sum() == 0 and fp.tolist() == [[1] * 6]


def test_oracle_tpfp_imagenet_rules():
    from oracle.evaluation import tpfp_imagenet
    # two overlapping gts; IoUs are taken against gt - 1 (mean_ap.py:104)
    gts = np.array([_box(11, 11, 40, 40), _box(21, 11, 40, 40)], np.float32)
    dets = np.array([
        _box(10, 10, 40, 40, 0.9),      # == gt0 - 1 -> IoU 1 -> tp, gt0 covered
        _box(12, 10, 40, 40, 0.8),      # IoU .905 with gt0-1 (covered), .667 with gt1-1 -> falls through to gt1: tp (ImageNet rule)
        _box(14, 10, 40, 40, 0.7),      # both covered -> fp
    ], np.float32)
    tp, fp = tpfp_imagenet(dets, gts, None, 0.5)
    assert tp.tolist() == [[1, 1, 0]] and fp.tolist() == [[0, 0, 1]]
    # small objects get a lower threshold: 10x10 gt -> min(100/400, .5) = .25
    g = np.array([_box(101, 101, 10, 10)], np.float32)
    d = np.array([_box(100, 100, 10, 4, 0.9)], np.float32)      # IoU 40/100 = .4 >= .25
    tp, fp = tpfp_imagenet(d, g, None, 0.5)
    assert tp.tolist() == [[1]] and fp.tolist() == [[0]]
    # an ignored gt is consumed by its match, which is then neither tp nor fp (mean_ap.py:138-142)
    tp, fp = tpfp_imagenet(d, np.zeros((0, 4), np.float32), g, 0.5)
    assert tp.tolist() == [[0]] and fp.tolist() == [[0]]


def test_oracle_eval_map_by_hand():
    from oracle.evaluation import eval_map
    e = np.zeros((0, 5), np.float32)
    # class 0: image A gts {a0, a1}, image B gt {b0}; class 1: no gt anywhere (skipped in the mean); class 2: one gt, no det
    ann = [dict(bboxes=np.array([_box(0, 0, 10, 10), _box(50, 50, 10, 10), _box(80, 0, 10, 10)], np.float32), labels=np.array([0, 0, 2])),
           dict(bboxes=np.array([_box(20, 20, 10, 10)], np.float32), labels=np.array([0]))]
    det = [[np.array([_box(0, 0, 10, 10, 0.9), _box(30, 30, 5, 5, 0.6)], np.float32), np.array([_box(1, 1, 5, 5, 0.3)], np.float32), e],
           [np.array([_box(20, 20, 10, 10, 0.8), _box(60, 60, 5, 5, 0.7)], np.float32), e, e]]
    # class 0 in score order: .9 tp, .8 tp, .7 fp, .6 fp; 3 gts -> recall 1/3 2/3 2/3 2/3, precision 1 1 2/3 1/2 -> AP = 2/3
    m, res = eval_map(det, ann, iou_thr=0.5)
    assert res[0]['ap'] == pytest.approx(2 / 3, abs=1e-6) and res[0]['num_gts'] == 3 and res[0]['num_dets'] == 4
    assert res[1]['num_gts'] == 0 and res[2]['ap'] == 0.0 and res[2]['num_gts'] == 1
    assert m == pytest.approx((2 / 3 + 0.0) / 2, abs=1e-6)          # classes without gt are left out of the mean
    np.testing.assert_allclose(res[0]['recall'], [1 / 3, 2 / 3, 2 / 3, 2 / 3], rtol=1e-6)
    np.testing.assert_allclose(res[0]['precision'], [1, 1, 2 / 3, 0.5], rtol=1e-6)


def test_product_eval_has_no_cpu_path():
    from lowlightenvironmentvideoobjectdetection_amd._lib import LlvError
    from lowlightenvironmentvideoobjectdetection_amd.core import evaluation
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(LlvError):
        evaluation.eval_map([[np.zeros((1, 5), np.float32)]], [dict(bboxes=np.zeros((1, 4), np.float32), labels=np.array([0]))])


# ---------------------------------------------------------------- device vs oracle, GPU
def _random_eval_set(seed, num_imgs, num_classes, ties=False, with_ignore=True):
    g = np.random.RandomState(seed)
    det_results, annotations = [], []
    for _ in range(num_imgs):
        n = g.randint(0, 9)
        xy = g.rand(n, 2) * 400
        wh = g.rand(n, 2) * 120 + 4
        gt = np.hstack([xy, xy + wh]).astype(np.float32)
        labels = g.randint(0, num_classes, n)
        ann = dict(bboxes=gt, labels=labels)
        if with_ignore:
            k = g.randint(0, 3)
            xy2 = g.rand(k, 2) * 400
            ann['bboxes_ignore'] = np.hstack([xy2, xy2 + g.rand(k, 2) * 80 + 4]).astype(np.float32)
            ann['labels_ignore'] = g.randint(0, num_classes, k)
        annotations.append(ann)
        per_cls = []
        for c in range(num_classes):
            src = gt[labels == c]
            m = g.randint(0, 14)
            rows = []
            for _ in range(m):
                if len(src) and g.rand() < 0.7:          # jittered copy of a gt (several per gt: duplicates)
                    b = src[g.randint(len(src))] + g.randn(4) * g.choice([1.0, 6.0, 20.0])
                else:
                    p = g.rand(2) * 400
                    b = np.concatenate([p, p + g.rand(2) * 100 + 2])
                s = np.round(g.rand(), 1) if ties else g.rand()
                rows.append(np.concatenate([b, [s]]))
            per_cls.append(np.array(rows, np.float32).reshape(-1, 5))
        det_results.append(per_cls)
    return det_results, annotations


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['default', 'imagenet'])
@pytest.mark.parametrize('ranges', [None, [(0, 1024), (1024, 9216), (9216, 1e10)]])
def test_device_tpfp_marks_bit_exact(dev, mode, ranges):
    from lowlightenvironmentvideoobjectdetection_amd.core import evaluation as ev
    from oracle import evaluation as ref
    det_results, annotations = _random_eval_set(5 + (mode == 'imagenet'), 40, 6, ties=True)
    fn_ref = ref.tpfp_imagenet if mode == 'imagenet' else ref.tpfp_default
    cells_d, cells_g, cells_i = [], [], []
    for c in range(6):
        d, g, ig = ref.get_cls_results(det_results, annotations, c)
        cells_d += d; cells_g += g; cells_i += ig
    got = ev.tpfp_cells(cells_d, cells_g, cells_i, 0.5, ranges, mode)
    n_tp = 0
    for (tp, fp), d, g, ig in zip(got, cells_d, cells_g, cells_i):
        tr, fr = fn_ref(d, g, ig, 0.5, ranges)
        assert np.array_equal(tp, tr) and np.array_equal(fp, fr)
        n_tp += int(tr.sum())
    assert n_tp > 50


@pytest.mark.gpu
@pytest.mark.parametrize('dataset,scale_ranges', [(None, None), ('vid', None), ('voc07', None), (None, [(0, 32), (32, 96), (96, 1e5)])])
def test_device_eval_map_equals_oracle(dev, dataset, scale_ranges):
    from lowlightenvironmentvideoobjectdetection_amd.core import evaluation as ev
    from oracle import evaluation as ref
    det_results, annotations = _random_eval_set(11, 120, 8)
    m, res = ev.eval_map(det_results, annotations, scale_ranges=scale_ranges, iou_thr=0.5, dataset=dataset, logger='silent')
    m_ref, res_ref = ref.eval_map(det_results, annotations, scale_ranges=scale_ranges, iou_thr=0.5, dataset=dataset)
    assert np.array_equal(np.asarray(m), np.asarray(m_ref))
    assert np.all(np.asarray(m_ref) > 0.05)
    for a, b in zip(res, res_ref):
        assert np.array_equal(a['recall'], b['recall']) and np.array_equal(a['precision'], b['precision'])
        assert np.array_equal(np.asarray(a['ap']), np.asarray(b['ap'])) and a['num_dets'] == b['num_dets']


@pytest.mark.gpu
def test_device_eval_edge_cases(dev, capsys):
    from lowlightenvironmentvideoobjectdetection_amd.core import evaluation as ev
    from oracle import evaluation as ref
    e5, e4 = np.zeros((0, 5), np.float32), np.zeros((0, 4), np.float32)
    # no detections at all
    m, res = ev.eval_map([[e5, e5]], [dict(bboxes=np.array([_box(0, 0, 5, 5)], np.float32), labels=np.array([1]))], logger='silent')
    assert m == 0.0 and res[1]['num_gts'] == 1 and res[1]['num_dets'] == 0
    # no ground truth at all: every detection is a false positive, mAP 0
    d = np.array([_box(0, 0, 5, 5, 0.5), _box(1, 1, 5, 5, 0.4)], np.float32)
    m, res = ev.eval_map([[d]], [dict(bboxes=e4, labels=np.zeros(0, int))], logger='silent')
    assert m == 0.0 and res[0]['precision'].tolist() == [0.0, 0.0]
    # one-cell wrappers equal the oracle's functions; 300 detections on 70 gts exercises the multi-pass loops
    g = np.random.RandomState(3)
    xy = g.rand(70, 2) * 300
    gts = np.hstack([xy, xy + g.rand(70, 2) * 60 + 5]).astype(np.float32)
    dd = np.hstack([gts[g.randint(0, 70, 300)] + g.randn(300, 4) * 4, g.rand(300, 1)]).astype(np.float32)
    for fd, fr in [(ev.tpfp_default, ref.tpfp_default), (ev.tpfp_imagenet, ref.tpfp_imagenet)]:
        tp, fp = fd(dd, gts[:60], gts[60:], 0.5)
        tr, frr = fr(dd, gts[:60], gts[60:], 0.5)
        assert np.array_equal(tp, tr) and np.array_equal(fp, frr) and tr.sum() > 20
    # the summary table prints one row per class and the mAP
    ev.eval_map([[d]], [dict(bboxes=np.array([_box(0, 0, 5, 5)], np.float32), labels=np.array([0]))], dataset=['fox'])
    out = capsys.readouterr().out
    assert 'fox' in out and 'mAP' in out and '1.000' in out
