"""Box coder object and result formatting with the reference's interface
(mmdetection/mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:10-84, transforms.py:99-116).

The coder here only carries its parameters to the device kernels (decode: csrc/decode.hip); it holds no
arithmetic of its own."""
import numpy as np
import torch

from ..registry import BBOX_CODERS


@BBOX_CODERS.register_module()
class DeltaXYWHBBoxCoder(object):
    def __init__(self, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.), clip_border=True):
        self.means = tuple(target_means)
        self.stds = tuple(target_stds)
        self.clip_border = clip_border


def bbox2result(bboxes, labels, num_classes):
    """transforms.py:99-116: split (k,5) detections by label into num_classes float32 arrays (D2H copy)."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes = bboxes.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]
