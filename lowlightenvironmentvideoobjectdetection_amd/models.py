"""Importing this module registers every class of the hot path in the reference's registries."""
from . import core  # noqa: F401
from . import losses  # noqa: F401
from . import backbones, necks, dense_heads, detectors, cleaners, aggregators, motion, vid, yolo, roi_extractors, rpn, roi_heads, selsa  # noqa: F401
from .registry import (build_backbone, build_neck, build_head, build_loss, build_detector, build_cleaner,  # noqa: F401
                       build_aggregator, build_motion, build_model)
