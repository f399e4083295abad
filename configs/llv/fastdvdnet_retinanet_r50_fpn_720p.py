# Enhance + detect on 1280x720 low-light video (BASELINE.json configs[1]): FastDVDnet over a 5-frame window feeding a
# RetinaNet R-50+FPN. The detector sits under `detector=` as in mmtracking's configs
# (mmtracking/configs/_base_/models/retinanet_r50_fpn.py).
_base_ = ['../_base_/default_runtime.py']
model = dict(
    type='FastDVDnetDetect',
    pretrains=None,
    detector=dict(
        type='RetinaNet',
        pretrained=None,
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'),
        neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                  add_extra_convs='on_input', num_outs=5),
        bbox_head=dict(
            type='RetinaHead', num_classes=80, in_channels=256, stacked_convs=4, feat_channels=256,
            anchor_generator=dict(type='AnchorGenerator', octave_base_scale=4, scales_per_octave=3,
                                  ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128]),
            bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
            loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
            loss_bbox=dict(type='L1Loss', loss_weight=1.0)),
        train_cfg=None,
        test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5),
                      max_per_img=100)),
    cleaner=dict(type='FastDVDnet'),
    frozen_modules='cleaner')
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
test_pipeline = [
    dict(type='SeqBrighten', m=0.25),
    dict(type='SeqNormalize', **img_norm_cfg),
    dict(type='SeqPad', size_divisor=32),
]
data = dict(samples_per_gpu=1, test=dict(ref_img_sampler=dict(num_ref_imgs=4, frame_range=[-2, 2])))
