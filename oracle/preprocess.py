"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the reference's test-time image pipeline.

  Brighten / SeqBrighten   mmtracking/mmtrack/datasets/pipelines/transforms.py:376-432 (gain from the first frame, shared)
  Normalize                mmdetection/mmdet/datasets/pipelines/transforms.py:547-575 -> mmcv.imnormalize:
                           float32(img), BGR->RGB, subtract mean, multiply by 1/std (float64 reciprocal)
  Pad                      mmdetection/mmdet/datasets/pipelines/transforms.py:476-524 (zeros to a multiple of size_divisor)
"""
import numpy as np


def brighten_clip(frames_u8, m=0.25):
    """frames_u8: (T,H,W,3) uint8 BGR. One gain for the clip, computed on frame 0 exactly as the reference does."""
    assert frames_u8.dtype == np.uint8
    t, h, w, c = frames_u8.shape
    amp = m * (h * w * 3) / np.sum(frames_u8[0].astype(np.float32) / 255.0)
    return np.stack([np.clip(f * amp, 0, 255.0).astype(np.uint8) for f in frames_u8]), amp


def normalize(img_u8, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), to_rgb=True):
    img = img_u8.astype(np.float32)
    if to_rgb:
        img = img[..., ::-1]
    mean = np.float64(np.array(mean).reshape(1, -1))
    stdinv = 1 / np.float64(np.array(std).reshape(1, -1))
    return ((img - mean.astype(np.float32)) * stdinv.astype(np.float32)).astype(np.float32)   # cv2.subtract / cv2.multiply on float32


def pad(img, size_divisor=32):
    h, w = img.shape[:2]
    hp, wp = (h + size_divisor - 1) // size_divisor * size_divisor, (w + size_divisor - 1) // size_divisor * size_divisor
    out = np.zeros((hp, wp) + img.shape[2:], img.dtype)
    out[:h, :w] = img
    return out


def preprocess_clip(frames_u8, m=0.25, size_divisor=32):
    """-> (T, Hp, Wp, 3) float32 RGB normalised, zero padded."""
    b, _ = brighten_clip(frames_u8, m)
    return np.stack([pad(normalize(f), size_divisor) for f in b])
