_base_ = ['../_base_/models/retinanet_r50_fpn.py', '../_base_/default_runtime.py']
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
test_pipeline = [dict(type='Resize', img_scale=(1280, 720), keep_ratio=True), dict(type='Normalize', **img_norm_cfg),
                 dict(type='Pad', size_divisor=32)]
