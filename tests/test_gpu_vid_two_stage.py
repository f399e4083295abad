"""The two-stage branch of the flow / aggregation video detectors -- the one the reference's own configs take (configs/vid/dff/*.py,
configs/vid/fgfa/*.py on _base_/models/faster_rcnn_r50_dc5.py): DFF / FGFA / LLVOD around FasterRCNN(DC5 ResNet, ChannelMapper, RPNHead,
StandardRoIHead + Shared2FCBBoxHead), at reduced width, on the device against the composed fp32 oracle:

  test time   dff.py:184-217,245-253   fgfa.py:196-284,344-352   llvod.py:225-233   standard_roi_head.py:238-268, test_mixins.py:50-170
  training    dff.py:139-173           fgfa.py:149-185           llvod.py:87-145    standard_roi_head.py:70-139, bbox_head.py:103-268

plus the reference's own property tests of the box heads (mmdetection/tests/test_models/test_roi_heads/test_bbox_head.py:10-66,
mmtracking/tests/test_models/test_heads.py:78-140: empty ground truth => box loss 0, class loss > 0).

Both sides see the SAME proposals where logits are compared (the oracle's, through the reference's `proposals=` override at test time and
in place of RPNHead.get_bboxes in training): the device RPN has its own parity test (tests/test_rpn.py)."""
import numpy as np
import pytest
import torch

from helpers import grad_metrics, img_meta, match_detections, randomize, smooth_image

pytestmark = pytest.mark.gpu

NCLS, C, FC = 6, 32, 128
H, W = 128, 192
TEST_RCNN = dict(score_thr=0.0001, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100)
TRAIN_CFG = dict(
    rpn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, ignore_iof_thr=-1),
             sampler=dict(type='RandomSampler', num=4096, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
             allowed_border=0, pos_weight=-1, debug=False),
    rpn_proposal=dict(nms_across_levels=False, nms_pre=600, nms_post=60, max_num=60, nms_thr=0.7, min_bbox_size=0),
    rcnn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=-1),
              sampler=dict(type='RandomSampler', num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True),
              pos_weight=-1, debug=False))
GTB = [torch.tensor([[10., 12., 90., 100.], [60., 30., 150., 120.], [100., 8., 180., 60.]])]
GTL = [torch.tensor([1, 3, 0])]
DEPTH_KW = dict(depth=18, out_indices=(3,), strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2))
RPN_CFG = dict(nms_pre=600, max_per_img=60, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0)


def _det_cfg(train=False):
    """mmtracking/configs/_base_/models/faster_rcnn_r50_dc5.py at reduced width (ResNet-18 DC5, 32-channel neck, 128-wide head)."""
    return dict(
        type='FasterRCNN', pretrained=None,
        backbone=dict(type='ResNet', depth=18, num_stages=4, out_indices=(3,), strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'),
        neck=dict(type='ChannelMapper', in_channels=[512], out_channels=C, kernel_size=3),
        rpn_head=dict(type='RPNHead', in_channels=C, feat_channels=C,
                      anchor_generator=dict(type='AnchorGenerator', scales=[4, 8, 16, 32], ratios=[0.5, 1.0, 2.0], strides=[16]),
                      bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
                      loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                      loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0)),
        roi_head=dict(type='StandardRoIHead',
                      bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=2),
                                              out_channels=C, featmap_strides=[16]),
                      bbox_head=dict(type='Shared2FCBBoxHead', in_channels=C, fc_out_channels=FC, roi_feat_size=7, num_classes=NCLS,
                                     bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.2, 0.2, 0.2, 0.2]),
                                     reg_class_agnostic=False, loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                                     loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0))),
        train_cfg=TRAIN_CFG if train else None,
        test_cfg=dict(rpn=dict(nms_across_levels=False, nms_pre=600, nms_post=60, max_num=60, nms_thr=0.7, min_bbox_size=0), rcnn=dict(TEST_RCNN)))


def _build(kind, seed, train=False, freeze_motion=True):
    import lowlightenvironmentvideoobjectdetection_amd.models as M
    flow = dict(type='FlowNetSimple', img_scale_factor=0.5)
    agg = dict(type='EmbedAggregator', num_convs=1, channels=C, kernel_size=3)
    fm = dict(frozen_modules=['motion']) if freeze_motion else {}
    cfg = dict(dff=dict(type='DFF', detector=_det_cfg(train), motion=flow, test_cfg=dict(key_frame_interval=10), **fm),
               fgfa=dict(type='FGFA', detector=_det_cfg(train), motion=flow, aggregator=agg, train_cfg=dict(add_noise=False), **fm),
               llvod=dict(type='LLVOD', detector=_det_cfg(train), aggregator=agg, train_cfg=dict(add_noise=False), memo_len=4),
               frcnn=_det_cfg(train))[kind]
    model = M.build_detector(cfg) if kind == 'frcnn' else M.build_model(cfg)
    det = model if kind == 'frcnn' else model.detector
    randomize(det.backbone, seed=seed)
    randomize(det.neck, seed=seed + 1)
    randomize(det.rpn_head, seed=seed + 2)
    if hasattr(model, 'aggregator'):
        randomize(model.aggregator, seed=seed + 3)
    g = torch.Generator().manual_seed(seed + 4)
    with torch.no_grad():
        det.rpn_head.rpn_cls.bias.add_(-1.0)
        det.rpn_head.rpn_reg.weight.mul_(0.3)
        for n, p in det.roi_head.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * ((1.5 / p.shape[1]) ** 0.5 if p.dim() == 2 else 0.1))
        det.roi_head.bbox_head.fc_reg.weight.mul_(0.3)
        det.roi_head.bbox_head.fc_cls.weight.mul_(3.0)      # peaked softmax, as a trained head's: flat scores make the top-100 cut a coin toss
        if hasattr(model, 'motion'):
            randomize(model.motion, seed=seed + 5)
            for name, p in model.motion.named_parameters():
                if 'deconv' in name or 'upsample_flow' in name:
                    p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(len(name))) * (1.0 / (p.shape[0] * 4) ** 0.5))
                if name.startswith('predict_flow'):
                    p.mul_(0.02)
    model.invalidate_plans()
    return model


def _sd(model, grad=False):
    sd = {k: v.detach().clone().float().cpu() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    if grad:
        for n in names:
            sd[n].requires_grad_(True)
    return sd, names


def _feats(sd, img, p='detector.'):
    from oracle.detector import resnet_forward
    from oracle.rpn import channel_mapper_forward
    return channel_mapper_forward(sd, resnet_forward(sd, img, prefix=p + 'backbone.', **DEPTH_KW)[-1:], p + 'neck.')[0]


def _props(sd, f, p='detector.', hw=(H, W)):
    from oracle.anchors import AnchorGenerator
    from oracle.bbox import DeltaXYWHBBoxCoder
    from oracle.rpn import rpn_get_bboxes, rpn_head_forward
    ag = AnchorGenerator(strides=[16], ratios=[0.5, 1.0, 2.0], scales=[4, 8, 16, 32])
    with torch.no_grad():
        cls, reg = rpn_head_forward(sd, [f.detach()], p + 'rpn_head.')
        return rpn_get_bboxes(cls, reg, ag, DeltaXYWHBBoxCoder(), (hw[0], hw[1], 3), RPN_CFG)


def _sub(sd, p):
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


def _hook_head(roi_head):
    cap = {}
    orig = type(roi_head)._bbox_forward

    def hooked(x_, rois_, _o=orig, _m=roi_head):
        r = _o(_m, x_, rois_)
        cap.update(r, rois=rois_)
        return r
    roi_head._bbox_forward = hooked
    return cap


def _check_frame(model, dev, sd, x_ref, out, cap, props, tag, p='detector.'):
    """Device run (oracle proposals on both sides) against the oracle RoI head on the oracle's feature map x_ref, stage by stage:
    logits / deltas of every RoI, decoded class-specific boxes and softmax scores before NMS (tight), the device's detections ==
    the ORACLE's multiclass NMS applied to the device's own pre-NMS tensors (exact: the NMS kernel is index-exact), and the fraction of
    the oracle's final detections reproduced end to end (reported; greedy NMS at IoU 0.5 over 60 heavily overlapping random-weight
    proposals x 6 classes flips on last-bit score differences and every flip changes which boxes survive)."""
    from oracle.bbox import delta2bbox
    from oracle.nms import multiclass_nms as oracle_mnms
    from oracle.two_stage import standard_roi_head_simple_test
    det = model.detector if hasattr(model, 'detector') else model
    with torch.no_grad():
        d_ref, l_ref, cls_ref, reg_ref = standard_roi_head_simple_test(sd, p + 'roi_head.bbox_head.', x_ref, props, (H, W, 3), [1, 1, 1, 1], TEST_RCNN)
    e_cls = (cap['cls_score'].cpu() - cls_ref).abs().max().item() / cls_ref.abs().max().item()
    e_reg = (cap['bbox_pred'].cpu() - reg_ref).abs().max().item() / reg_ref.abs().max().item()
    assert e_cls < 5e-3 and e_reg < 5e-3, (tag, e_cls, e_reg)
    bh = det.roi_head.bbox_head
    with torch.no_grad():
        b_dev, s_dev = bh.get_bboxes(cap['rois'], cap['cls_score'], cap['bbox_pred'], (H, W, 3), [1, 1, 1, 1], cfg=None)
    rois_ref = torch.cat([props.new_zeros(len(props), 1), props[:, :4]], 1)
    b_ref = delta2bbox(rois_ref[:, 1:], reg_ref, (0., 0., 0., 0.), (0.2, 0.2, 0.2, 0.2), max_shape=(H, W, 3))
    e_box = (b_dev.cpu() - b_ref).abs().max().item()
    e_sc = (s_dev.cpu() - torch.softmax(cls_ref, -1)).abs().max().item()
    assert e_box < 0.5 and e_sc < 5e-3, (tag, e_box, e_sc)
    d2, l2 = oracle_mnms(b_dev.cpu(), s_dev.cpu(), TEST_RCNN['score_thr'], TEST_RCNN['nms'], TEST_RCNN['max_per_img'])
    want = [d2[l2 == c].numpy() for c in range(NCLS)]
    for c in range(NCLS):
        assert out[c].shape == want[c].shape and np.array_equal(out[c], want[c]), (tag, c)
    ref = [d_ref[l_ref == c].numpy() for c in range(NCLS)]
    frac, n = match_detections(out, ref, iou_thr=0.9, score_tol=0.05)
    print(f'{tag}: logits {e_cls:.2e}, deltas {e_reg:.2e} of their scale; boxes {e_box:.3f} px, scores {e_sc:.1e}; NMS on the device tensors '
          f'exact; {frac:.2f} of {n} oracle detections reproduced end to end')
    assert n >= 30 and abs(sum(len(o) for o in out) - len(d_ref)) <= 3, (tag, frac, n)
    return n


# ------------------------------------------------------------------------------------------------ property tests of the heads
def _dummy_bbox_sampling(proposal_list, gt_bboxes, gt_labels):
    """mmdetection/tests/test_models/test_roi_heads/utils.py:6-37"""
    from lowlightenvironmentvideoobjectdetection_amd.registry import build_assigner, build_sampler
    a = build_assigner(dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=-1))
    s = build_sampler(dict(type='RandomSampler', num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True))
    out = []
    for i in range(1):
        ar = a.assign(proposal_list[i], gt_bboxes[i], None, gt_labels[i])
        out.append(s.sample(ar, proposal_list[i], gt_bboxes[i], gt_labels[i]))
    return out


@pytest.mark.parametrize('kind', ['BBoxHead', 'Shared2FCBBoxHead', 'SelsaBBoxHead'])
def test_bbox_head_loss_empty_and_non_empty_truth(dev, kind):
    """test_bbox_head.py:10-66 (BBoxHead(in_channels=8, roi_feat_size=3)) and test_heads.py:78-140 (SelsaBBoxHead(num_shared_fcs=2,
    in_channels=8, fc_out_channels=16, roi_feat_size=3, SelsaAggregator(16, 4 blocks))) restated: one proposal; with empty ground
    truth the class loss is > 0 and the box loss == 0, with one matching box both are > 0."""
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    from lowlightenvironmentvideoobjectdetection_amd.roi_heads import bbox2roi
    cfg = dict(BBoxHead=dict(type='BBoxHead', in_channels=8, roi_feat_size=3),
               Shared2FCBBoxHead=dict(type='Shared2FCBBoxHead', in_channels=8, roi_feat_size=3, fc_out_channels=16),
               SelsaBBoxHead=dict(type='SelsaBBoxHead', num_shared_fcs=2, in_channels=8, fc_out_channels=16, roi_feat_size=3,
                                  aggregator=dict(type='SelsaAggregator', in_channels=16, num_attention_blocks=4)))[kind]
    torch.manual_seed(0)
    head = registry.build_head(cfg).to(dev)
    proposal_list = [torch.tensor([[23.6667, 23.8757, 228.6326, 153.8874]], device=dev)]
    target_cfg = dict(pos_weight=1)
    for gt_b, gt_l, box_zero in ((torch.empty((0, 4)), torch.empty((0,), dtype=torch.long), True),
                                 (torch.tensor([[23.6667, 23.8757, 238.6326, 151.8874]]), torch.tensor([2]), False)):
        gt_bboxes, gt_labels = [gt_b.to(dev)], [gt_l.to(dev)]
        sampling = _dummy_bbox_sampling(proposal_list, gt_bboxes, gt_labels)
        labels, lw, bt, bw = head.get_targets(sampling, gt_bboxes, gt_labels, target_cfg)
        num = sum(len(r.bboxes) for r in sampling)
        rois = bbox2roi([r.bboxes for r in sampling])
        feats = torch.rand(num, 8, 3, 3, device=dev)
        if kind == 'SelsaBBoxHead':
            cls, reg = head(feats, torch.rand(2 * num, 8, 3, 3, device=dev))
        else:
            cls, reg = head(feats)
        assert cls.shape == (num, 81) and reg.shape == (num, 320)
        losses = head.loss(cls, reg, rois, labels, lw, bt, bw)
        assert float(losses.get('loss_cls', 0)) > 0, 'cls-loss should be non-zero'
        if box_zero:
            assert float(losses.get('loss_bbox', 0)) == 0, 'empty gt loss should be zero'
        else:
            assert float(losses.get('loss_bbox', 0)) > 0, 'box-loss should be non-zero'
            sum(v for k, v in losses.items() if 'loss' in k).backward()
            assert head.fc_reg.weight.grad is not None and head.fc_reg.weight.grad.abs().max().item() > 0


def test_bbox_head_get_bboxes_shapes(dev):
    """test_bbox_head.py:69-84 (the unbatched half; the batched call is the padding form of test_mixins.py, not built): class-agnostic
    head, cfg=None returns the decoded boxes and the softmax scores; no RoIs give empty results."""
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    head = registry.build_head(dict(type='BBoxHead', reg_class_agnostic=True))
    for n in (2, 0):
        rois, cls, reg = torch.rand(n, 5, device=dev), torch.rand(n, 6, device=dev), torch.rand(n, 4, device=dev)
        b, s = head.get_bboxes(rois, cls, reg, None, 2.0, rescale=True)
        assert b.shape == reg.shape and s.shape == cls.shape


# ------------------------------------------------------------------------------------------------ test time
def test_faster_rcnn_simple_test_vs_oracle(dev):
    """TwoStageDetector.simple_test (two_stage.py:178-190) with the StandardRoIHead: single image, oracle proposals on both sides."""
    model = _build('frcnn', 301)
    sd, _ = _sd(model)
    img = smooth_image(1, H, W, seed=302)
    with torch.no_grad():
        f = _feats(sd, img, '')
    p = _props(sd, f, '')
    model = model.to(dev).eval()
    cap = _hook_head(model.roi_head)
    with torch.no_grad():
        out = model(return_loss=False, img=[img.to(dev)], img_metas=[[img_meta(H, W)]], proposals=[[p.to(dev)]])
    assert len(out) == 1 and len(out[0]) == NCLS
    _check_frame(model, dev, sd, f, out[0], cap, p, 'faster r-cnn', p='')
    with torch.no_grad():
        own = model(return_loss=False, img=[img.to(dev)], img_metas=[[img_meta(H, W)]])
    assert len(own[0]) == NCLS and sum(len(r) for r in own[0]) >= 30
    # rescale=True divides the boxes by the scale factor (bbox_head.py:340-347)
    meta2 = img_meta(H, W, scale=2.0)
    with torch.no_grad():
        o2 = model(return_loss=False, img=[img.to(dev)], img_metas=[[meta2]], proposals=[[p.to(dev)]], rescale=True)
    a, b = np.concatenate(out[0]), np.concatenate(o2[0])
    assert a.shape == b.shape and np.allclose(a[:, :4] / 2.0, b[:, :4], atol=1e-3) and np.allclose(a[:, 4], b[:, 4])


def test_dff_two_stage_key_and_warped_frames_vs_oracle(dev):
    """DFF.simple_test (dff.py:184-253): frame 0 is a key frame (features cached), frames 1 and 2 warp the cached feature map along the
    FlowNetSimple flow of (frame, key frame); RPN + StandardRoIHead on the result."""
    from oracle.flownet import flownet_forward
    from oracle.temporal import flow_warp_feats
    model = _build('dff', 311)
    sd, _ = _sd(model)
    f0 = smooth_image(1, H, W, seed=312)
    g = torch.Generator().manual_seed(313)
    frames = [f0] + [torch.roll(f0, shifts=(2 * t, 3 * t), dims=(2, 3)) + 0.02 * torch.randn(1, 3, H, W, generator=g) for t in (1, 2)]
    with torch.no_grad():
        x0 = _feats(sd, f0)
    model = model.to(dev).eval()
    cap = _hook_head(model.detector.roi_head)
    tot = 0
    for t, fr in enumerate(frames):
        meta = dict(img_meta(H, W), frame_id=t)
        with torch.no_grad():
            if t == 0:
                x = x0
            else:
                flow = flownet_forward(_sub(sd, 'motion.'), torch.cat((fr, f0), 1), meta, 0.5, 5.0)
                assert 0.05 < flow.abs().max().item() < 40
                x = flow_warp_feats(x0, flow)
        p = _props(sd, x)
        with torch.no_grad():
            out = model(return_loss=False, img=[fr.to(dev)], img_metas=[[meta]], proposals=[[p.to(dev)]])['bbox_results']
        tot += _check_frame(model, dev, sd, x, out, cap, p, f'dff frame {t}')
        if t == 1:                                  # the whole chain with the device's own proposals
            with torch.no_grad():
                own = model(return_loss=False, img=[fr.to(dev)], img_metas=[[meta]])['bbox_results']
            assert len(own) == NCLS and sum(len(r) for r in own) >= 30
    assert tot > 100


@pytest.mark.parametrize('fixed', [False, True])
def test_fgfa_two_stage_vs_oracle(dev, fixed):
    """FGFA.simple_test (fgfa.py:196-352) with three reference frames, both branches of extract_feats: adaptive stride (references given
    with frame 0 stay; stack = cat(x, warped references)) and fixed stride (window slides by one frame per frame_stride; the key frame's
    features replace slot num_left_ref_imgs of the warped stack)."""
    from oracle.flownet import flownet_forward
    from oracle.temporal import embed_aggregate, flow_warp_feats
    model = _build('fgfa', 321)
    sd, _ = _sd(model)
    nr, left = 3, 1
    clip = smooth_image(1, H, W, seed=322)
    g = torch.Generator().manual_seed(323)
    clip = torch.cat([torch.roll(clip, shifts=(t, 2 * t), dims=(2, 3)) + 0.02 * torch.randn(1, 3, H, W, generator=g) for t in range(6)], 0)
    asd, msd = _sub(sd, 'aggregator.'), _sub(sd, 'motion.')
    model = model.to(dev).eval()
    cap = _hook_head(model.detector.roi_head)
    tot = 0
    if fixed:
        # frame t's window = clip[t : t + 3], key frame = clip[t + left]
        steps = [(t, clip[t + left:t + left + 1], clip[0:3] if t == 0 else clip[t + 2:t + 3]) for t in range(3)]
    else:
        steps = [(t, clip[3 + t:4 + t], clip[0:3] if t == 0 else None) for t in range(2)]
    memo_img = None
    for t, img, refs in steps:
        meta = dict(img_meta(H, W), frame_id=t, num_left_ref_imgs=left if fixed else -1, frame_stride=1 if fixed else -1)
        with torch.no_grad():
            if t == 0:
                memo_img = refs
            elif fixed:
                memo_img = torch.cat((memo_img, refs), 0)[1:]
            memo_f = _feats(sd, memo_img)
            x = memo_f[[left]] if fixed else _feats(sd, img)
            flows = flownet_forward(msd, torch.cat((img.repeat(nr, 1, 1, 1), memo_img), 1), meta, 0.5, 5.0)
            stack = flow_warp_feats(memo_f, flows)
            if fixed:
                stack[left] = x[0]
            else:
                stack = torch.cat((x, stack), 0)
            xa = embed_aggregate(asd, x, stack, 1)
        p = _props(sd, xa)
        kw = {} if refs is None else dict(ref_img=[refs.unsqueeze(0).to(dev)], ref_img_metas=[[[img_meta(H, W)] * refs.shape[0]]])
        with torch.no_grad():
            out = model(return_loss=False, img=[img.to(dev)], img_metas=[[meta]], proposals=[[p.to(dev)]], **kw)['bbox_results']
        tot += _check_frame(model, dev, sd, xa, out, cap, p, f"fgfa {'fixed' if fixed else 'adaptive'} frame {t}")
    assert tot > 60


def test_llvod_two_stage_sliding_window_vs_oracle(dev):
    """LLVOD around the two-stage detector: frames join a window of the last memo_len = 4 feature maps; per frame
    aggregator(x_key, the others) (llvod.py:201-206), RPN + StandardRoIHead (llvod.py:225-233)."""
    from oracle.temporal import embed_aggregate
    model = _build('llvod', 331)
    sd, _ = _sd(model)
    clip = smooth_image(1, H, W, seed=332)
    g = torch.Generator().manual_seed(333)
    clip = torch.cat([torch.roll(clip, shifts=(t, 2 * t), dims=(2, 3)) + 0.02 * torch.randn(1, 3, H, W, generator=g) for t in range(6)], 0)
    with torch.no_grad():
        f_all = _feats(sd, clip)
    model = model.to(dev).eval()
    cap = _hook_head(model.detector.roi_head)
    # frame 0 arrives with two reference frames (clip[0:2]) and is clip[2]; frames 1.. are clip[3], clip[4], clip[5]
    for t in range(4):
        idx = 2 + t
        lo = max(0, idx + 1 - 4)
        with torch.no_grad():
            xa = embed_aggregate(_sub(sd, 'aggregator.'), f_all[[idx]], f_all[lo:idx], 1)
        p = _props(sd, xa)
        meta = dict(img_meta(H, W), frame_id=t)
        kw = dict(ref_img=[clip[0:2].unsqueeze(0).to(dev)]) if t == 0 else {}
        with torch.no_grad():
            out = model(return_loss=False, img=[clip[idx:idx + 1].to(dev)], img_metas=[[meta]], proposals=[[p.to(dev)]], **kw)['bbox_results']
        _check_frame(model, dev, sd, xa, out, cap, p, f'llvod frame {t}')


# ------------------------------------------------------------------------------------------------ training
def _oracle_train(kind, model, img, refs, meta, flow_grad=False, hw=(H, W)):
    from oracle.anchors import AnchorGenerator
    from oracle.flownet import flownet_forward
    from oracle.rpn import rpn_head_forward
    from oracle.temporal import embed_aggregate, flow_warp_feats
    from oracle.training import MaxIoUAssigner
    from oracle.two_stage import RandomSampler, rpn_loss, standard_roi_head_forward_train
    sd, names = _sd(model, grad=True)
    asd, msd = _sub(sd, 'aggregator.'), _sub(sd, 'motion.')
    nr = refs.shape[0]
    import contextlib
    ng = contextlib.nullcontext if flow_grad else torch.no_grad
    if kind == 'dff':
        with ng():
            flow = flownet_forward(msd, torch.cat((img, refs[:1]), 1), meta, 0.5, 5.0)
        x = flow_warp_feats(_feats(sd, refs[:1]), flow)
    elif kind == 'fgfa':
        with ng():
            flows = flownet_forward(msd, torch.cat((img.repeat(nr, 1, 1, 1), refs), 1), meta, 0.5, 5.0)
        f = _feats(sd, torch.cat((img, refs), 0))
        x = embed_aggregate(asd, f[[0]], flow_warp_feats(f[1:], flows), 1)
    else:
        idx = nr // 2
        f = _feats(sd, torch.cat((refs[:idx], img, refs[idx:]), 0))
        x = embed_aggregate(asd, f[[idx]], torch.cat((f[:idx], f[idx + 1:]), 0), 1)
    ag = AnchorGenerator(strides=[16], ratios=[0.5, 1.0, 2.0], scales=[4, 8, 16, 32])
    cls, reg = rpn_head_forward(sd, [x], 'detector.rpn_head.')
    losses = rpn_loss(cls, reg, GTB, [meta], ag, MaxIoUAssigner(0.7, 0.3, 0.3, ignore_iof_thr=-1), RandomSampler(4096, 0.5, -1, False), allowed_border=0)
    losses = dict(loss_rpn_cls=sum(losses['loss_rpn_cls']), loss_rpn_bbox=sum(losses['loss_rpn_bbox']))
    p_key = _props(sd, x, hw=hw)
    roi_losses, _ = standard_roi_head_forward_train(sd, 'detector.roi_head.bbox_head.', x, p_key, GTB[0], GTL[0],
                                                    MaxIoUAssigner(0.5, 0.5, 0.5, ignore_iof_thr=-1), RandomSampler(512, 0.25, -1, True), NCLS)
    losses.update(roi_losses)
    sum(v for k, v in losses.items() if 'loss' in k).backward()
    return losses, p_key, {n: sd[n].grad for n in names}


@pytest.mark.parametrize('kind,nr', [('dff', 1), ('fgfa', 3), ('llvod', 4)])
def test_video_two_stage_forward_train_matches_oracle(dev, kind, nr):
    """DFF / FGFA / LLVOD.forward_train around the DC5 Faster R-CNN: the four loss terms + accuracy and every parameter gradient of
    backbone (layer2.. : frozen_stages=1), ChannelMapper, aggregator, RPN and Shared2FC head against autograd over the oracle.
    Per tensor: max-abs error relative to the tensor's own scale (the bound of the other training tests) AND cosine / relative L2,
    which tell a mis-routed gradient (cosine far from 1) from fp16 rounding (cosine ~ 1, error spread evenly)."""
    model = _build(kind, 341 + nr, train=True)
    base = smooth_image(1, H, W, seed=351)
    g = torch.Generator().manual_seed(352)
    img = base + 0.05 * torch.randn(1, 3, H, W, generator=g)
    refs = torch.cat([torch.roll(base, shifts=(i + 1, 2 * i - 1), dims=(2, 3)) + 0.05 * torch.randn(1, 3, H, W, generator=g) for i in range(nr)], 0)
    meta = dict(img_meta(H, W), is_video_data=True, frame_id=3)
    ref_losses, p_key, ref_grads = _oracle_train(kind, model, img, refs, meta)
    model = model.to(dev).train()
    model.detector.rpn_head.get_bboxes = lambda *a, **k: [p_key.to(dev)]
    losses = model(img=img.to(dev), img_metas=[meta], gt_bboxes=[b.to(dev) for b in GTB], gt_labels=[l.to(dev) for l in GTL],
                   ref_img=refs.unsqueeze(0).to(dev), ref_img_metas=[[meta] * nr], ref_gt_bboxes=None, ref_gt_labels=None, return_loss=True)
    assert set(losses) == {'loss_rpn_cls', 'loss_rpn_bbox', 'loss_cls', 'acc', 'loss_bbox'}
    for k, v in ref_losses.items():
        got = losses[k]
        got = sum(got) if isinstance(got, (list, tuple)) else got
        print(f'{kind} {k}: {float(got):.5f} vs {float(v):.5f}')
        assert abs(float(got) - float(v)) <= 2e-2 * max(abs(float(v)), 1e-2), (k, float(got), float(v))
    sum(sum(v) if isinstance(v, (list, tuple)) else v for k, v in losses.items() if 'loss' in k).backward()
    params = dict(model.named_parameters())
    top = max(v.abs().max().item() for v in ref_grads.values() if v is not None)
    rows = []
    for k, r in ref_grads.items():
        frozen = k.startswith(('detector.backbone.conv1', 'detector.backbone.bn1', 'detector.backbone.layer1.', 'motion.'))
        if frozen or r is None:
            assert params[k].grad is None or params[k].grad.abs().max().item() == 0, k
            continue
        gk = params[k].grad
        assert gk is not None, k
        if r.abs().max().item() < 1e-6 * top:
            assert gk.abs().max().item() < 1e-4 * top, k
            continue
        rows.append((k,) + grad_metrics(gk, r))
    rows.sort(key=lambda t: -t[1])
    print(kind, 'largest max-abs errors:', [(k, round(e, 3), round(c, 5), round(l2, 3)) for k, e, c, l2, s in rows[:6]])
    print(kind, 'lowest cosines:', [(k, round(c, 5)) for k, e, c, l2, s in sorted(rows, key=lambda t: t[2])[:4]])
    for k, e, c, l2, s in rows:
        big = s > 1e-4 * top
        assert e < (8e-2 if big else 1.6e-1), (k, e, s)          # first run: <= 4.6 %
        assert c > (0.999 if big else 0.99), (k, c, s)             # first run: >= 0.9996
        assert l2 < (4e-2 if big else 1e-1), (k, l2, s)            # first run: <= 2.2 %
    assert len(rows) > 40, len(rows)
    for k in ('detector.roi_head.bbox_head.shared_fcs.0.weight', 'detector.rpn_head.rpn_cls.weight', 'detector.neck.convs.0.conv.weight',
              'detector.backbone.layer4.1.conv2.weight', 'detector.backbone.layer2.0.conv1.weight'):
        assert params[k].grad is not None and params[k].grad.abs().max().item() > 0, k
    if kind != 'dff':
        assert params['aggregator.embed_convs.0.conv.weight'].grad.abs().max().item() > 0


def test_dff_two_stage_trains_the_flow_network_too(dev):
    """Without frozen_modules=['motion'] (the reference's configs) the two-stage DFF step also fills the flow network's gradients,
    through d(loss)/d(flow) of the feature warp: all 33 FlowNetSimple tensors against autograd over the oracle (bounds of the dense-head
    form of this test, tests/test_gpu_vid_train.py::test_dff_trains_the_flow_network_too)."""
    from test_gpu_vid_train import _flownet
    model = _build('dff', 361, train=True, freeze_motion=False)
    model.motion.load_state_dict(_flownet(7).state_dict())
    model.invalidate_plans()
    Ww = 256
    base = smooth_image(1, H, Ww, seed=362)
    g = torch.Generator().manual_seed(363)
    img = base + 0.05 * torch.randn(1, 3, H, Ww, generator=g)
    refs = torch.roll(base, shifts=(2, -3), dims=(2, 3)) + 0.05 * torch.randn(1, 3, H, Ww, generator=g)
    meta = dict(img_meta(H, Ww), is_video_data=True, frame_id=3)
    ref_losses, p_key, ref_grads = _oracle_train('dff', model, img, refs, meta, flow_grad=True, hw=(H, Ww))
    model = model.to(dev).train()
    model.detector.rpn_head.get_bboxes = lambda *a, **k: [p_key.to(dev)]
    losses = model(img=img.to(dev), img_metas=[meta], gt_bboxes=[b.to(dev) for b in GTB], gt_labels=[l.to(dev) for l in GTL],
                   ref_img=refs.unsqueeze(0).to(dev), ref_img_metas=[[meta]], ref_gt_bboxes=None, ref_gt_labels=None, return_loss=True)
    tot = lambda v: sum(v) if isinstance(v, (list, tuple)) else v
    for k, v in ref_losses.items():
        assert abs(float(tot(losses[k])) - float(v)) <= 2e-2 * max(abs(float(v)), 1e-2), (k, float(tot(losses[k])), float(v))
    sum(tot(v) for k, v in losses.items() if 'loss' in k).backward()
    params = dict(model.named_parameters())
    rows = []
    for k, r in ref_grads.items():
        if not k.startswith('motion.') or r is None or r.abs().max().item() < 1e-10:
            continue
        assert params[k].grad is not None, k
        rows.append((k,) + grad_metrics(params[k].grad, r))
    rows.sort(key=lambda t: -t[1])
    print('motion:', [(k, round(e, 3), round(c, 4), round(l2, 3)) for k, e, c, l2, s in rows[:8]])
    assert len(rows) == 33
    assert sum(1 for r in rows if r[1] < 1.5e-1) >= 28 and all(r[1] < 3e-1 for r in rows), rows[:6]
    assert sum(1 for r in rows if r[2] > 0.99) >= 28 and all(r[2] > 0.95 for r in rows), sorted(rows, key=lambda t: t[2])[:6]


def test_faster_rcnn_forward_train_matches_oracle(dev):
    """TwoStageDetector.forward_train (two_stage.py:115-176) on one still image: the DC5 Faster R-CNN's four losses and every trainable
    gradient against autograd over the oracle (the video detectors' shared tail, without the temporal part)."""
    from oracle.anchors import AnchorGenerator
    from oracle.rpn import rpn_head_forward
    from oracle.training import MaxIoUAssigner
    from oracle.two_stage import RandomSampler, rpn_loss, standard_roi_head_forward_train
    model = _build('frcnn', 371, train=True)
    img = smooth_image(1, H, W, seed=372)
    meta = img_meta(H, W)
    sd, names = _sd(model, grad=True)
    x = _feats(sd, img, '')
    cls, reg = rpn_head_forward(sd, [x], 'rpn_head.')
    ag = AnchorGenerator(strides=[16], ratios=[0.5, 1.0, 2.0], scales=[4, 8, 16, 32])
    ref = rpn_loss(cls, reg, GTB, [meta], ag, MaxIoUAssigner(0.7, 0.3, 0.3, ignore_iof_thr=-1), RandomSampler(4096, 0.5, -1, False), allowed_border=0)
    ref = dict(loss_rpn_cls=sum(ref['loss_rpn_cls']), loss_rpn_bbox=sum(ref['loss_rpn_bbox']))
    p_key = _props(sd, x, '')
    roi_losses, _ = standard_roi_head_forward_train(sd, 'roi_head.bbox_head.', x, p_key, GTB[0], GTL[0], MaxIoUAssigner(0.5, 0.5, 0.5, ignore_iof_thr=-1),
                                                    RandomSampler(512, 0.25, -1, True), NCLS)
    ref.update(roi_losses)
    sum(v for k, v in ref.items() if 'loss' in k).backward()
    model = model.to(dev).train()
    model.rpn_head.get_bboxes = lambda *a, **k: [p_key.to(dev)]
    losses = model(img=img.to(dev), img_metas=[meta], gt_bboxes=[b.to(dev) for b in GTB], gt_labels=[l.to(dev) for l in GTL], return_loss=True)
    tot = lambda v: sum(v) if isinstance(v, (list, tuple)) else v
    for k, v in ref.items():
        assert abs(float(tot(losses[k])) - float(v)) <= 2e-2 * max(abs(float(v)), 1e-2), (k, float(tot(losses[k])), float(v))
    sum(tot(v) for k, v in losses.items() if 'loss' in k).backward()
    params = dict(model.named_parameters())
    top = max(sd[n].grad.abs().max().item() for n in names if sd[n].grad is not None)
    rows = []
    for k in names:
        r = sd[k].grad
        if k.startswith(('backbone.conv1', 'backbone.bn1', 'backbone.layer1.')) or r is None:
            assert params[k].grad is None or params[k].grad.abs().max().item() == 0, k
            continue
        if r.abs().max().item() < 1e-6 * top:
            continue
        rows.append((k,) + grad_metrics(params[k].grad, r))
    rows.sort(key=lambda t: -t[1])
    print('faster r-cnn largest max-abs errors:', [(k, round(e, 3), round(c, 5), round(l2, 3)) for k, e, c, l2, s in rows[:5]])
    for k, e, c, l2, s in rows:
        big = s > 1e-4 * top
        assert e < (8e-2 if big else 1.6e-1) and c > (0.999 if big else 0.99) and l2 < (4e-2 if big else 1e-1), (k, e, c, l2)
    assert len(rows) > 40
