from .anchor import AnchorGenerator, anchor_inside_flags, images_to_levels  # noqa: F401
from .assign import AssignResult, MaxIoUAssigner, PseudoSampler, SamplingResult  # noqa: F401
from .bbox import DeltaXYWHBBoxCoder, bbox2result  # noqa: F401
from .evaluation import average_precision, eval_map, print_map_summary, tpfp_default, tpfp_imagenet  # noqa: F401
from .coco_eval import coco_bbox_eval  # noqa: F401
