"""Test-time driver of the video detectors, with ONE clip sharded BY FRAME across the ranks (BASELINE.json
north_star: "clips shard by frame across the 8 GPUs").

What it mirrors in the reference (paths relative to /root/reference/mmtracking/mmtrack):
  * `multi_gpu_test`                       apis/test.py:77-122  (loop `model(return_loss=False, rescale=True, **data)`, collect to rank 0)
  * `DistributedVideoSampler`              datasets/samplers/distributed_video_sampler.py:24-45  (each rank a contiguous run of WHOLE videos)
  * `CocoVideoDataset.ref_img_sampling`    datasets/coco_video_dataset.py:198-219  (which reference frames accompany a test frame)
  * `VideoCollect` metas                   datasets/pipelines/formatting.py:252-255 (frame_id, is_video_data, num_left_ref_imgs, frame_stride)

The reference can only distribute whole videos, because a video detector keeps per-video state (`self.memo`). Here a single clip is
cut into `world` contiguous segments (dist_utils.shard_clip) and every rank gets, with the FIRST frame of its segment, exactly the
reference frames that rebuild the state the one-rank run holds at that frame (the "halo"):

  FastDVDnetDetect   the frame's own 2+2 neighbours (sliding window): nothing is carried, the halo is 2 frames on either side
  DFF                no halo: segments start on a key frame (boundaries are multiples of test_cfg.key_frame_interval)
  FGFA, fixed stride the (left + right + 1)-frame window of the first frame instead of the one new reference; boundaries are multiples
                     of the stride
  FGFA / SELSA, adaptive stride   the same `num_ref_imgs` frames spread over the whole clip on every rank (they never change)
  LLVOD              the up-to memo_len - 1 frames before the segment, pushed into the ring at the slots the one-rank run uses

so that every frame goes through the same launches on the same pixels as in the one-rank run: the concatenated per-rank results
are BIT-IDENTICAL to it (tests/test_clip_shard.py on CPU/gloo for the feed logic at world 2 and 8, tests/test_gpu_shard.py on the
device for the detections). No data-path collective; the only exchange is the result gather (dist_utils.collect_results).
"""
from collections import defaultdict

import torch

from . import dist_utils
from ._lib import LlvError

TEST_METHODS = ('test_with_adaptive_stride', 'test_with_fix_stride', 'sliding', 'none')


def clip_window(t, num_frames, left, right, stride=1):
    """Frame ids of the fixed-stride reference window of test frame t (coco_video_dataset.py:204-215): slot j holds frame
    (t // stride + j - left) * stride clamped into the clip -- frame 0 fills the slots before the clip's start, the last frame those
    past its end, exactly what the reference's memo contains at frame t after its one-frame-per-stride slides."""
    k = t // stride
    return [min(max((k + j - left) * stride, 0), num_frames - 1) for j in range(left + right + 1)]


def adaptive_ref_ids(num_frames, num_ref_imgs):
    """test_with_adaptive_stride (coco_video_dataset.py:198-203): num_ref_imgs frames spread evenly over the whole video."""
    if num_ref_imgs < 2:
        return [0] * num_ref_imgs
    stride = float(num_frames - 1) / (num_ref_imgs - 1)
    return [int(round(i * stride)) for i in range(num_ref_imgs)]


def _norm_range(frame_range):
    if isinstance(frame_range, int):
        assert frame_range >= 0, 'frame_range can not be a negative value.'
        return -frame_range, frame_range
    assert len(frame_range) == 2 and frame_range[0] <= 0 <= frame_range[1]
    return int(frame_range[0]), int(frame_range[1])


class ClipTestFeed:
    """What the reference's test dataset + pipeline deliver for one video, frame by frame, for the frames [start, stop) of a clip
    that is resident in memory: dict(img=[(1,3,H,W)], img_metas=[[meta]], ref_img=[(1,N,3,H,W)]) (fgfa.py:330-333; `ref_img` absent
    when the sampler gives the frame no reference).

    frames: (T, 3, H, W) tensor, or any object with `__len__` and `__getitem__(list of ids) -> (n, 3, H, W)` (a rank then only ever
    touches its segment and halo). ref_img_sampler: the test sampler of the config (num_ref_imgs, frame_range, stride, method); method
    'sliding' = the frame's own neighbours frame_range[0] .. frame_range[1] without the frame itself, clamped into the clip (the
    per-frame windows FastDVDnetDetect slides), 'none' = no references (DFF).

    start > 0: the first frame carries the references that rebuild the video detector's state (module docstring); `start` must be
    a multiple of the stride. Global frame ids are kept in the metas (`frame_id`), so key-frame schedules do not shift."""

    def __init__(self, frames, img_meta, ref_img_sampler=None, start=0, stop=None, device=None):
        s = dict(ref_img_sampler or dict(method='none'))
        self.method = s.get('method', 'none')
        if self.method not in TEST_METHODS:
            raise LlvError(f'ClipTestFeed: ref_img_sampler method {self.method!r} (test-time methods: {TEST_METHODS})')
        self.frames, self.meta, self.device = frames, dict(img_meta), device
        self.T = len(frames)
        self.start, self.stop = int(start), self.T if stop is None else int(stop)
        if not 0 <= self.start <= self.stop <= self.T:
            raise LlvError(f'ClipTestFeed: segment [{start}, {stop}) outside the clip of {self.T} frames')
        self.left, self.right = 0, 0
        self.stride = int(s.get('stride', 1))
        if self.method != 'none':
            lo, hi = _norm_range(s.get('frame_range', 0))
            self.left, self.right = -lo, hi
            self.num_ref_imgs = s.get('num_ref_imgs', hi - lo)
            if 'test_with' in self.method and hi - lo != self.num_ref_imgs:     # coco_video_dataset.py:146-154
                self.num_ref_imgs = hi - lo
        if self.method == 'test_with_fix_stride' and self.start < self.stop and self.start % self.stride:
            raise LlvError(f'ClipTestFeed: segment start {self.start} is not a multiple of the sampling stride {self.stride}')

    def __len__(self):
        return self.stop - self.start

    def halo(self):
        """Frame ids this segment reads: (lowest, highest + 1)."""
        ids = set()
        for t in range(self.start, self.stop):
            ids.add(t)
            ids.update(self.ref_ids(t, first=(t == self.start)))
        return (min(ids), max(ids) + 1) if ids else (self.start, self.start)

    def ref_ids(self, t, first):
        """Frame ids of `ref_img` for test frame t (sorted by frame id like the reference's list, coco_video_dataset.py:227);
        `first`: t is the first frame this detector instance sees (frame 0 of the video, or the first frame of a later segment)."""
        if self.method == 'none':
            return []
        if self.method == 'sliding':
            return [min(max(t + d, 0), self.T - 1) for d in range(-self.left, self.right + 1) if d != 0]
        if self.method == 'test_with_adaptive_stride':
            return adaptive_ref_ids(self.T, self.num_ref_imgs) if first else []
        # test_with_fix_stride
        if first:
            return clip_window(t, self.T, self.left, self.right, self.stride)
        if t % self.stride == 0:
            return [min(t + self.right * self.stride, self.T - 1)]
        return []

    def _take(self, ids):
        f = self.frames
        x = f[ids] if not torch.is_tensor(f) else f[torch.as_tensor(ids, dtype=torch.long)]
        return x.to(self.device) if self.device is not None else x

    def __iter__(self):
        for t in range(self.start, self.stop):
            meta = dict(self.meta, frame_id=t, is_video_data=True)
            if self.method == 'test_with_fix_stride':                        # coco_video_dataset.py:216-218
                meta.update(num_left_ref_imgs=self.left, frame_stride=self.stride)
            data = dict(img=[self._take([t])], img_metas=[[meta]])
            ids = self.ref_ids(t, first=(t == self.start))
            if ids:
                data['ref_img'] = [self._take(ids).unsqueeze(0)]
                data['ref_img_metas'] = [[[dict(self.meta, frame_id=i, is_video_data=True) for i in ids]]]
            yield t, data


def shard_alignment(model, ref_img_sampler=None):
    """Segment boundaries must fall where the one-rank run would itself rebuild / slide its state: multiples of DFF's
    key_frame_interval, multiples of the fixed sampling stride."""
    align = 1
    tc = getattr(model, 'test_cfg', None) or {}
    if type(model).__name__ == 'DFF' or 'key_frame_interval' in tc:
        align = int(tc.get('key_frame_interval', 10))
    s = ref_img_sampler or {}
    if s.get('method') == 'test_with_fix_stride':
        st = int(s.get('stride', 1))
        align = align * st // _gcd(align, st)
    return align


def _gcd(a, b):
    while b:
        a, b = b, a % b
    return a


def prepare_segment(model, feed):
    """Forget the previous video and, for the ring-buffer detector (LLVOD), line the ring up with the one-rank run: that run keeps
    frame f in slot f % memo_len, and the weighted sum over the frames is taken in slot order, so a segment that starts at frame s
    pushes its warm-up frames max(0, s - memo_len + 1) .. s - 1 (ClipTestFeed delivers them as the first frame's ref_img) beginning
    at the slot the first of them has there."""
    if hasattr(model, 'reset_video'):
        model.reset_video()
    else:
        model.memo = None
    if hasattr(model, 'memo_len') and feed.start > 0:
        first = max(0, feed.start - model.memo_len + 1)
        model.memo = dict(next=first % model.memo_len, valid=0)


class _RingFeed(ClipTestFeed):
    """LLVOD's plain sliding ring (vid.LLVOD): no sampler; the first frame of a later segment brings the frames before it."""

    def __init__(self, frames, img_meta, memo_len, start=0, stop=None, device=None):
        super().__init__(frames, img_meta, dict(method='none'), start, stop, device)
        self.memo_len = memo_len

    def ref_ids(self, t, first):
        return list(range(max(0, t - self.memo_len + 1), t)) if (first and t > 0) else []


def make_feed(model, frames, img_meta, ref_img_sampler, start, stop, device=None):
    if hasattr(model, 'memo_len') and type(model).__name__ == 'LLVOD':
        return _RingFeed(frames, img_meta, model.memo_len, start, stop, device)
    return ClipTestFeed(frames, img_meta, ref_img_sampler, start, stop, device)


def single_gpu_test(model, feed, rescale=True):
    """apis/test.py:13-74 without the drawing: run every frame of the feed through the model; dict of lists like the reference's."""
    results = defaultdict(list)
    prepare_segment(model, feed)
    for _, data in feed:
        with torch.no_grad():
            result = model(return_loss=False, rescale=rescale, **data)
        for k, v in result.items():
            results[k].append(v)
    return results


def sharded_clip_test(model, frames, img_meta, ref_img_sampler=None, rank=None, world=None, rescale=True, device=None):
    """One clip, every rank its contiguous segment of frames (+ halo), results gathered on rank 0 in frame order -- multi_gpu_test
    (apis/test.py:77-122) with the frame, not the video, as the unit of distribution. Returns dict(bbox_results=[per frame]) on
    rank 0, None elsewhere. world == 1 is the plain single-GPU loop over the whole clip."""
    if rank is None or world is None:
        rank, world = dist_utils.get_dist_info()
    T = len(frames)
    start, stop = dist_utils.shard_clip(T, rank, world, align=shard_alignment(model, ref_img_sampler))
    feed = make_feed(model, frames, img_meta, ref_img_sampler, start, stop, device)
    part = single_gpu_test(model, feed, rescale) if stop > start else defaultdict(list)
    if world == 1:
        return dict(part)
    return dist_utils.collect_result_dicts(part)
