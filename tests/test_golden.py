"""Committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle):
the CPU tests check that the oracle still reproduces them; the GPU tests check the HIP path against them."""
import os

import numpy as np
import pytest
import torch

from helpers import calibrate_cls, img_meta, match_detections, randomize, retinanet_cfg, smooth_image

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    return np.load(os.path.join(G, name))


def test_oracle_reproduces_golden_nms():
    from oracle.nms import multiclass_nms
    z = _load('nms_full_geometry.npz')
    d, l, k = multiclass_nms(torch.from_numpy(z['boxes']), torch.from_numpy(z['scores']), 0.05,
                             dict(type='nms', iou_threshold=0.5), 100, return_inds=True)
    assert int(z['n_candidates']) > 2000
    assert np.array_equal(k.numpy(), z['keep_flat']) and np.array_equal(l.numpy(), z['labels']) and np.array_equal(d.numpy(), z['dets'])


def test_oracle_reproduces_golden_decode():
    from oracle.anchors import AnchorGenerator
    from oracle.bbox import DeltaXYWHBBoxCoder
    from oracle.detector import decode_candidates
    z = _load('decode_small.npz')
    cls = [torch.from_numpy(z[f'cls{i}']) for i in range(3)]
    reg = [torch.from_numpy(z[f'reg{i}']) for i in range(3)]
    ag = AnchorGenerator(strides=[8, 16, 32], ratios=[0.5, 1.0, 2.0], octave_base_scale=4, scales_per_octave=3)
    b, s, idx = decode_candidates(cls, reg, ag, DeltaXYWHBBoxCoder(), (150, 224, 3), np.array([1.5, 1.25, 1.5, 1.25], np.float32), 6, 1000, True)
    assert np.array_equal(idx.numpy(), z['anchors'])
    assert np.allclose(b.numpy(), z['boxes'], rtol=1e-6, atol=1e-4) and np.allclose(s.numpy(), z['scores'], rtol=1e-6, atol=1e-7)


def test_oracle_reproduces_golden_fastdvdnet():
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    from oracle.cleaner import fastdvdnet_forward
    z = _load('fastdvdnet_small.npz')
    cl = randomize(M.build_cleaner(dict(type='FastDVDnet')), seed=9)
    y = fastdvdnet_forward(cl.state_dict(), torch.from_numpy(z['x']))
    assert np.allclose(y.numpy(), z['y'], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_gpu_nms_matches_golden(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    z = _load('nms_full_geometry.npz')
    d, l, k = ops.multiclass_nms(torch.from_numpy(z['boxes']).to(dev), torch.from_numpy(z['scores']).to(dev), 0.05,
                                 dict(type='nms', iou_threshold=0.5), 100, return_inds=True)
    assert np.array_equal(k.cpu().numpy(), z['keep_flat'])          # index-exact
    assert np.array_equal(l.cpu().numpy(), z['labels']) and np.array_equal(d.cpu().numpy(), z['dets'])


@pytest.mark.gpu
def test_gpu_decode_matches_golden(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from lowlightenvironmentvideoobjectdetection_amd.core.anchor import AnchorGenerator
    z = _load('decode_small.npz')
    sizes, C, A = [(20, 28), (10, 14), (5, 7)], 6, 9
    cls = torch.cat([torch.from_numpy(z[f'cls{i}']).permute(0, 2, 3, 1).reshape(-1, A * C) for i in range(3)]).contiguous().to(dev)
    reg = torch.cat([torch.from_numpy(z[f'reg{i}']).permute(0, 2, 3, 1).reshape(-1, A * 4) for i in range(3)]).contiguous().to(dev)
    ag = AnchorGenerator(strides=[8, 16, 32], ratios=[0.5, 1.0, 2.0], octave_base_scale=4, scales_per_octave=3)
    offs = [0, 560, 700]
    plan = ops.make_decode_plan(cls, reg, sizes, [8, 16, 32], offs, offs, A * C, A * 4, C, A, ag.base_anchors, 1000,
                                (0., 0., 0., 0.), (1., 1., 1., 1.), 16 / 1000, (150, 224, 3), [1.5, 1.25, 1.5, 1.25], True)
    boxes, scores = ops.retina_decode_topk(plan)
    torch.cuda.synchronize()
    same = plan.cand_anchor.cpu().numpy() == z['anchors']
    assert same.mean() > 0.999
    assert np.allclose(boxes.cpu().numpy()[same], z['boxes'][same], rtol=1e-5, atol=2e-3)
    assert np.allclose(scores.cpu().numpy()[same], z['scores'][same, :C], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_gpu_fastdvdnet_matches_golden(dev):
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    z = _load('fastdvdnet_small.npz')
    cl = randomize(M.build_cleaner(dict(type='FastDVDnet')), seed=9).to(dev).eval()
    y = cl(torch.from_numpy(z['x']).to(dev)).cpu().numpy()
    assert np.abs(y - z['y']).max() / np.abs(z['y']).max() < 1e-2        # fp16 storage / fp32 accumulate vs fp32


@pytest.mark.gpu
def test_gpu_retinanet_matches_golden(dev):
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    z = _load('retinanet_small.npz')
    det = randomize(M.build_detector(retinanet_cfg(num_classes=4, depth=18, in_channels=(64, 128, 256, 512), feat=64)), seed=5, cls_bias=-1.5)
    with torch.no_grad():      # the calibrated head weights are part of the fixture (no oracle call on the GPU box)
        det.bbox_head.retina_cls.weight.copy_(torch.from_numpy(z['cls_w']))
        det.bbox_head.retina_cls.bias.copy_(torch.from_numpy(z['cls_b']))
        det.bbox_head.retina_reg.weight.copy_(torch.from_numpy(z['reg_w']))
        det.bbox_head.retina_reg.bias.copy_(torch.from_numpy(z['reg_b']))
    det = det.to(dev).eval()
    img = torch.from_numpy(z['img']).to(dev)
    feats = det.extract_feat(img)
    for i, f in enumerate(feats):
        ref = z[f'feat{i}'].astype(np.float32)
        assert np.abs(f.cpu().numpy() - ref).max() / np.abs(ref).max() < 1.5e-2
    res = det(img=[img], img_metas=[[img_meta(96, 128)]], return_loss=False)[0]
    ref = [z[f'det{c}'] for c in range(4)]
    frac, tot = match_detections(res, ref)
    assert tot > 10 and frac >= 0.85


def _eval_fixture():
    import sys
    GOLD = G
    sys.path.insert(0, GOLD)
    from make_golden_eval import unflatten
    z = np.load(os.path.join(GOLD, 'eval_small.npz'))
    det_results, anns = unflatten(z['dets'], z['gts'], z['ign'], int(z['n_img']), int(z['K']))
    return z, det_results, anns


EVAL_CASES = [('voc', dict()), ('vid', dict(dataset='vid')), ('voc07', dict(dataset='voc07')),
              ('scales', dict(scale_ranges=[(0, 32), (32, 96), (96, 1e5)]))]


def test_oracle_reproduces_golden_eval_and_dcn():
    from oracle import coco_eval, evaluation
    from oracle.dcn import modulated_deform_conv2d
    z, det_results, anns = _eval_fixture()
    for name, kw in EVAL_CASES:
        m, res = evaluation.eval_map(det_results, anns, iou_thr=0.5, **kw)
        np.testing.assert_allclose(np.asarray(m, np.float64), z[f'map_{name}'], rtol=0, atol=1e-7)
        np.testing.assert_allclose(np.stack([np.atleast_1d(r['ap']) for r in res]), z[f'ap_{name}'], rtol=0, atol=1e-7)
    c = coco_eval.coco_bbox_eval([[d.astype(np.float64) for d in im] for im in det_results], anns)
    np.testing.assert_allclose(c['stats'], z['coco_stats'], rtol=0, atol=1e-12)
    d = np.load(os.path.join(G, 'dcn_small.npz'))
    dg = int(d['dg'])
    om = torch.from_numpy(d['om'])
    y = modulated_deform_conv2d(torch.from_numpy(d['x']), om[:, :2 * dg * 9], torch.sigmoid(om[:, 2 * dg * 9:]), torch.from_numpy(d['w']),
                                torch.from_numpy(d['b']), 1, 1, 1, 1, dg)
    np.testing.assert_allclose(y.numpy(), d['y'], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_gpu_eval_and_dcn_match_golden(dev):
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from lowlightenvironmentvideoobjectdetection_amd.core import coco_eval, evaluation
    z, det_results, anns = _eval_fixture()
    for name, kw in EVAL_CASES:
        m, res = evaluation.eval_map(det_results, anns, iou_thr=0.5, logger='silent', **kw)
        assert np.array_equal(np.asarray(m, np.float64), z[f'map_{name}'])
        assert np.array_equal(np.stack([np.atleast_1d(r['ap']) for r in res]).astype(np.float64), z[f'ap_{name}'])
    c = coco_eval.coco_bbox_eval([[d.astype(np.float64) for d in im] for im in det_results], anns)
    assert np.array_equal(c['stats'], z['coco_stats'])
    d = np.load(os.path.join(G, 'dcn_small.npz'))
    dg, co = int(d['dg']), d['w'].shape[0]
    x = ops.pack_input(torch.from_numpy(d['x']).to(dev), d['x'].shape[1])
    om = torch.from_numpy(d['om']).permute(0, 2, 3, 1).contiguous().to(dev)
    pc = ops.pack_dcn_weight(torch.from_numpy(d['w']), torch.from_numpy(d['b'])).to(dev)
    y = ops.unpack_output(ops.modulated_deform_conv2d(x, om, pc, 3, 3, 1, 1, 1, dg), co).cpu().numpy()
    assert np.abs(y - d['y']).max() / np.abs(d['y']).max() < 3e-3


# ---- the reference's own input fixtures (mmdetection/tests/test_onnx/data/*.pkl -> tests/golden/onnx_head_inputs.npz, see
# tests/golden/make_golden_onnx_inputs.py). They hold inputs only; the reference runs them through get_bboxes with the head of
# test_head.py:20-52 (4 classes, 9 anchors, img_shape (128, 128), score_thr 0.05, nms 0.5, max_per_img 100, NO nms_pre: every one of
# the 12,276 anchors is a candidate) and compares torch with onnxruntime.
ONNX_TEST_CFG = dict(min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100)


def _onnx_retina_inputs():
    z = _load('onnx_head_inputs.npz')
    return [torch.from_numpy(z[f'retina_cls_{i}']) for i in range(5)], [torch.from_numpy(z[f'retina_bbox_{i}']) for i in range(5)]


@pytest.mark.parametrize('nms_pre', [-1, 1000])
def test_oracle_on_reference_get_bboxes_fixture(nms_pre):
    from oracle.anchors import AnchorGenerator
    from oracle.bbox import DeltaXYWHBBoxCoder
    from oracle.detector import decode_candidates, get_bboxes
    cls, reg = _onnx_retina_inputs()
    ag = AnchorGenerator(strides=[8, 16, 32, 64, 128], ratios=[0.5, 1.0, 2.0], octave_base_scale=4, scales_per_octave=3)
    cfg = dict(ONNX_TEST_CFG, nms_pre=nms_pre)
    b, s, idx = decode_candidates(cls, reg, ag, DeltaXYWHBBoxCoder(), (128, 128, 2), 1, 4, nms_pre)
    n = 9 * (32 * 32 + 16 * 16 + 8 * 8 + 4 * 4 + 2 * 2)
    assert b.shape[0] == (n if nms_pre < 0 else 1000 + 1000 + 576 + 144 + 36) and s.shape[1] == 5
    assert float(b.min()) >= 0 and float(b.max()) <= 128 and float(s[:, :4].min()) > 0.5        # rand logits in (0, 1): sigmoid > 0.5
    d, l, k = get_bboxes(cls, reg, ag, DeltaXYWHBBoxCoder(), (128, 128, 2), 1, 4, cfg, return_inds=True)
    assert d.shape == (100, 5) and bool((d[:-1, 4] >= d[1:, 4]).all()) and len(set(k.tolist())) == 100
    assert bool((d[:, 4] > 0.7).all())                           # the 100 best of >= 11,000 candidates near sigmoid(1)


@pytest.mark.gpu
@pytest.mark.parametrize('nms_pre', [-1, 1000])
def test_gpu_get_bboxes_on_reference_fixture(dev, nms_pre):
    """The reference's fixture through llv_retina_decode_topk + llv_multiclass_nms: candidates equal the oracle's (boxes 1e-5, scores
    1e-6), kept indices and labels index-exact. nms_pre = -1 is the reference test's own setting (12,276 boxes per class list)."""
    from lowlightenvironmentvideoobjectdetection_amd import ops
    from oracle.anchors import AnchorGenerator
    from oracle.bbox import DeltaXYWHBBoxCoder
    from oracle.detector import decode_candidates
    from oracle.nms import multiclass_nms as oracle_mnms
    cls, reg = _onnx_retina_inputs()
    sizes, strides, A, C = [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)], [8, 16, 32, 64, 128], 9, 4
    ag = AnchorGenerator(strides=strides, ratios=[0.5, 1.0, 2.0], octave_base_scale=4, scales_per_octave=3)
    b_ref, s_ref, idx_ref = decode_candidates(cls, reg, ag, DeltaXYWHBBoxCoder(), (128, 128, 2), 1, C, nms_pre)
    cls_buf = torch.cat([c.permute(0, 2, 3, 1).reshape(-1, A * C) for c in cls]).contiguous().to(dev)
    reg_buf = torch.cat([r.permute(0, 2, 3, 1).reshape(-1, A * 4) for r in reg]).contiguous().to(dev)
    offs, o = [], 0
    for h, w in sizes:
        offs.append(o)
        o += h * w
    plan = ops.make_decode_plan(cls_buf, reg_buf, sizes, strides, offs, offs, A * C, A * 4, C, A, ag.base_anchors, nms_pre,
                                (0., 0., 0., 0.), (1., 1., 1., 1.), 16 / 1000, (128, 128, 2), (1., 1., 1., 1.), False)
    boxes, scores = ops.retina_decode_topk(plan)
    torch.cuda.synchronize()
    got_idx = plan.cand_anchor.cpu().long()
    same = got_idx == idx_ref
    assert plan.n_cand == b_ref.shape[0] and float(same.float().mean()) > 0.998       # ties of the device expf within 1e-6 may swap
    sel = same.nonzero().view(-1)
    torch.testing.assert_close(boxes.cpu()[sel], b_ref[sel], rtol=1e-5, atol=2e-3)
    torch.testing.assert_close(scores.cpu()[sel], s_ref[sel, :C], rtol=1e-5, atol=1e-6)
    # NMS on IDENTICAL inputs (the oracle's candidates): index-exact
    d_ref, l_ref, k_ref = oracle_mnms(b_ref, s_ref, 0.05, ONNX_TEST_CFG['nms'], 100, return_inds=True)
    d, l, k = ops.multiclass_nms(b_ref.to(dev), s_ref.to(dev), 0.05, ONNX_TEST_CFG['nms'], 100, return_inds=True)
    assert torch.equal(k.cpu(), k_ref) and torch.equal(l.cpu(), l_ref) and torch.equal(d.cpu(), d_ref)
    assert ops.guard_trips() == 0


@pytest.mark.gpu
def test_gpu_yolo_get_bboxes_on_reference_fixture(dev):
    """mmdetection/tests/test_onnx/data/yolov3_head_get_bboxes.pkl (3 torch.rand maps (1, 27, s, s): 3 anchors x (5 + 4 classes)) through
    llv_yolo_decode + NMS with score factors against oracle/yolo.py, head of test_head.py:134-179 (conf_thr 0.005, score_thr 0.05,
    nms 0.45, max_per_img 100)."""
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    from oracle.yolo import yolo_get_bboxes
    z = _load('onnx_head_inputs.npz')
    maps = [torch.from_numpy(z[f'yolo_pred_{i}']) for i in range(3)]
    tc = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, conf_thr=0.005, nms=dict(type='nms', iou_threshold=0.45), max_per_img=100)
    base_sizes = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
    ref_d, ref_l = yolo_get_bboxes(maps, base_sizes, [32, 16, 8], 4, tc)[0]
    head = registry.build_head(dict(type='YOLOV3Head', num_classes=4, in_channels=[1, 1, 1], out_channels=[16, 8, 4], test_cfg=tc)).to(dev)
    # the reference's strides are [32, 16, 8] for maps of size 32, 16, 8 (its own, inconsistent, toy geometry): keep it
    d, l = head.get_bboxes([m.to(dev) for m in maps], [dict(img_shape=(128, 128, 3), scale_factor=1.0)])[0]
    d, l = d.cpu(), l.cpu()
    assert d.shape == ref_d.shape and torch.equal(l, ref_l)
    torch.testing.assert_close(d, ref_d, rtol=1e-5, atol=1e-3)
