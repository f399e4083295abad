"""Proposal generation of the two-stage path: ChannelMapper + RPNHead on the device against the oracle
(oracle/rpn.py). RPN settings of mmtracking/configs/vid/llvod/llvod_insert_plugins_l34_i1234_vid_a7s3.py:37-52,129-135."""
import numpy as np
import pytest
import torch

from helpers import img_meta, randomize

RPN = dict(type='RPNHead', in_channels=64, feat_channels=64,
           anchor_generator=dict(type='AnchorGenerator', scales=[4, 8, 16, 32], ratios=[0.5, 1.0, 2.0], strides=[16]),
           bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
           loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
           loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
           test_cfg=dict(nms_across_levels=False, nms_pre=6000, nms_post=300, max_num=300, nms_thr=0.7, min_bbox_size=0))


def _build():
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    neck = registry.build_neck(dict(type='ChannelMapper', in_channels=[128], out_channels=64, kernel_size=3))
    head = registry.build_head(dict(RPN))
    return neck, head


def test_rpn_and_channel_mapper_build_with_reference_parameter_names():
    neck, head = _build()
    assert set(neck.state_dict()) == {'convs.0.conv.weight', 'convs.0.conv.bias'}
    assert set(head.state_dict()) == {'rpn_conv.weight', 'rpn_conv.bias', 'rpn_cls.weight', 'rpn_cls.bias', 'rpn_reg.weight', 'rpn_reg.bias'}
    assert head.num_anchors == 12 and head.state_dict()['rpn_cls.weight'].shape == (12, 64, 1, 1)
    assert head.state_dict()['rpn_reg.weight'].shape == (48, 64, 1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize('fpn', [False, True])
def test_rpn_proposals_vs_oracle(dev, fpn):
    """Single level (the SELSA configs: stride 16, 12 anchors, nms_pre 6000, 300 proposals) and a 3-level pyramid (batched_nms by
    level): the proposal lists must agree box for box. Device convs are fp16: proposals are matched by IoU > 0.95 and score within 0.02,
    and >= 90 % of the oracle's must be reproduced (NMS at 0.7 on near-duplicate anchors amplifies last-bit logit differences)."""
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    from oracle.anchors import AnchorGenerator
    from oracle.bbox import DeltaXYWHBBoxCoder
    from oracle.rpn import channel_mapper_forward, rpn_get_bboxes, rpn_head_forward
    from helpers import match_detections
    if fpn:
        cfg = dict(RPN, anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32]),
                   test_cfg=dict(nms_pre=500, max_per_img=200, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0))
        neck = registry.build_neck(dict(type='ChannelMapper', in_channels=[32, 64, 128], out_channels=64, kernel_size=3))
        ag = AnchorGenerator(strides=[8, 16, 32], ratios=[0.5, 1.0, 2.0], scales=[8])
        shapes = [(32, 24, 40), (64, 12, 20), (128, 6, 10)]
    else:
        cfg = dict(RPN)
        neck = registry.build_neck(dict(type='ChannelMapper', in_channels=[128], out_channels=64, kernel_size=3))
        ag = AnchorGenerator(strides=[16], ratios=[0.5, 1.0, 2.0], scales=[4, 8, 16, 32])
        shapes = [(128, 24, 40)]
    head = registry.build_head(cfg)
    randomize(neck, seed=61)
    randomize(head, seed=62)
    with torch.no_grad():
        head.rpn_cls.bias.add_(-1.0)
        head.rpn_reg.weight.mul_(0.3)
    g = torch.Generator().manual_seed(63)
    feats = [torch.randn(1, c, h, w, generator=g) for c, h, w in shapes]
    H, W = shapes[0][1] * (8 if fpn else 16), shapes[0][2] * (8 if fpn else 16)
    sd_n = {'neck.' + k: v.detach().float() for k, v in neck.state_dict().items()}
    sd_h = {'rpn_head.' + k: v.detach().float() for k, v in head.state_dict().items()}
    with torch.no_grad():
        x = channel_mapper_forward(sd_n, feats)
        cls, reg = rpn_head_forward(sd_h, x)
        ref = rpn_get_bboxes(cls, reg, ag, DeltaXYWHBBoxCoder(), (H, W, 3), dict(cfg['test_cfg']))
    neck, head = neck.to(dev), head.to(dev)
    xs = neck(tuple(f.to(dev) for f in feats))
    for a, b in zip(xs, x):
        assert (a.cpu() - b).abs().max().item() < 1e-2 * b.abs().max().item()
    props = head.simple_test_rpn(xs, [img_meta(H, W)])[0].cpu()
    assert props.shape[1] == 5 and len(ref) > 50 and abs(len(props) - len(ref)) <= max(3, len(ref) // 20)
    assert torch.all(props[:-1, 4] >= props[1:, 4])                       # best first
    frac, tot = match_detections([props.numpy()], [ref.numpy()], iou_thr=0.95, score_tol=0.02)
    assert frac >= 0.9, f'{frac:.3f} of {tot} oracle proposals reproduced'


def test_rpn_softmax_head_is_refused():
    """rpn_head.py:128-137 has a softmax branch (use_sigmoid=False: 2 channels per anchor); the device decode reads one sigmoid logit
    per anchor, so such a head is refused -- at construction (AnchorHead) and again in get_bboxes -- instead of decoding misaligned
    logits."""
    import lowlightenvironmentvideoobjectdetection_amd.models  # noqa: F401
    from lowlightenvironmentvideoobjectdetection_amd import registry
    from lowlightenvironmentvideoobjectdetection_amd._lib import LlvError
    with pytest.raises(NotImplementedError, match='softmax'):
        registry.build_head(dict(RPN, loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)))
    head = registry.build_head(dict(RPN))
    head.use_sigmoid_cls, head.cls_out_channels = False, 2          # a head that slipped past the constructor
    with pytest.raises(LlvError, match='sigmoid'):
        head.get_bboxes([torch.zeros(1, 24, 4, 4)], [torch.zeros(1, 48, 4, 4)], [img_meta(64, 64)])
