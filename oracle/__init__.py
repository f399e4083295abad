"""oracle/ -- CPU restatement of the reference's hot path. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only
as the checker / the timed CPU baseline -- never as part of the product path. The product package
(lowlightenvironmentvideoobjectdetection_amd) does not import it and has no CPU fallback.

What it restates (paths relative to /root/reference; each function cites file:line):
  anchors.py      mmdetection/mmdet/core/anchor/anchor_generator.py
  bbox.py         mmdetection/mmdet/core/bbox/coder/delta_xywh_bbox_coder.py, iou_calculators/iou2d_calculator.py
  nms.py          mmdetection/mmdet/core/post_processing/bbox_nms.py + mmcv 1.3.0 ops.nms / batched_nms
  detector.py     mmdet backbones/resnet.py, necks/fpn.py, dense_heads/{retina_head,anchor_head}.py
  cleaner.py      mmtracking/mmtrack/models/cleaners/FastDVDnet.py
  temporal.py     mmtracking/mmtrack/models/aggregators/embed_aggregator.py, core/motion/flow.py
  training.py     mmdet assigners/max_iou_assigner.py, losses/{focal_loss,smooth_l1_loss,utils}.py, anchor_head.py loss
  preprocess.py   mmtrack datasets/pipelines/transforms.py (Brighten), mmdet pipelines (Normalize, Pad),
                  mmtrack core/utils/add_noise.py (synthetic dark-clip noise model)

Arithmetic is float32 torch / numpy on the CPU: the heavy ops are the very torch.nn.functional kernels
the reference dispatches to on its CPU path. The reference package itself cannot be imported here
(`import mmdet` needs mmcv, which is absent: ModuleNotFoundError, not a permission denial), so the
restatement is pinned by the reference's own known-answer tests (tests/test_oracle_*.py cite them):
anchors, delta2bbox, MaxIoUAssigner, bbox_overlaps and weighted-loss doctests are PINNED;
NMS and focal loss have no vectors in the reference ("parity unpinned" for those two: mmcv is
un-vendored third-party code, restated from its published algorithm).

Tie rule (the reference inherits torch/mmcv's unspecified tie order): every sort here is
(score descending, index ascending), implemented with stable sorts.
"""
