"""Generates tests/golden/fox_noise_frames.npz: nine consecutive frames of the reference's in-tree dark, noisy JPEG sequence
mmtracking/demo/videos/fox_noise_n10-l025/ (220 frames of 480 x 360, the authors' own low-light demo input), decoded to uint8 and
stored in BGR order, the order mmcv.imread / LoadImageFromFile hands to the pipeline (mmdetection/mmdet/datasets/pipelines/loading.py:54-66).
Run in the build container (the GPU box has no /root/reference):

    python tests/golden/make_golden_fox_frames.py

The fixture is data (decoded pixels), not reference source. JPEG decoding is libjpeg's in both PIL and OpenCV; the frames are what the
reference's loader would deliver up to the decoder's IDCT variant (at most +-1 grey level on isolated pixels)."""
import os

import numpy as np
from PIL import Image

SRC = '/root/reference/mmtracking/demo/videos/fox_noise_n10-l025'
FIRST, COUNT = 100, 9

if __name__ == '__main__':
    names = sorted(os.listdir(SRC))[FIRST:FIRST + COUNT]
    frames = np.stack([np.asarray(Image.open(os.path.join(SRC, n)).convert('RGB'))[..., ::-1] for n in names]).astype(np.uint8)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fox_noise_frames.npz')
    np.savez_compressed(out, frames_bgr=frames, first=np.int64(FIRST), names=np.array(names))
    print(out, frames.shape, frames.dtype, 'mean', frames.mean(), os.path.getsize(out), 'bytes')
