"""MI355X-native low-light video object detection hot path (enhance-CNN -> RetinaNet -> decode -> NMS).

Host side mirrors the reference's registry/config surface (mmdet 2.10 / mmtrack 0.5 fork); the
arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI of include/llv.h (libllv.so).
"""
__version__ = '0.1.0'
