"""Frame-sharded single clip, host logic on the CPU.

`apis.ClipTestFeed` plays the reference's test dataset for one clip segment; `apis.sharded_clip_test` runs a video detector over the
segment a rank owns and gathers the results. The product's detectors need the MI355X, so here the feed drives TRACE detectors: small
stand-ins that replay the reference's per-video memo control flow (the `extract_feats` of fgfa.py:196-284 / dff.py:184-217 /
selsa_fastdvd_darkfarm.py:294-301 and the ring of vid.LLVOD) on frame IDS -- each frame tensor carries its id -- and return as
'bbox_results' the complete state the real model's arithmetic depends on (key frame, ordered window, which frames were extracted
how). Equal traces for world = 1 and the concatenation over world = 2 / 8 (real gloo process groups) mean that every frame sees the
same inputs in the same order on the same code path: what makes the device results bit-identical (tests/test_gpu_shard.py)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ids(x):
    """frame ids of a (n, 3, h, w) / (1, n, 3, h, w) tensor whose pixels all hold the frame id"""
    x = x[0] if x.dim() == 5 else x
    return [int(v) for v in x[:, 0, 0, 0].tolist()]


def _frames(T):
    return torch.arange(T, dtype=torch.float32).view(T, 1, 1, 1).expand(T, 3, 2, 2).contiguous()


class _Trace:
    test_cfg = None

    def __init__(self):
        self.memo = None

    def __call__(self, return_loss=False, rescale=True, img=None, img_metas=None, ref_img=None, ref_img_metas=None):
        assert return_loss is False and len(img) == 1 and len(img_metas) == 1 and len(img_metas[0]) == 1
        return self.simple_test(img[0], img_metas[0], ref_img=ref_img, ref_img_metas=ref_img_metas)


class TraceFGFA(_Trace):
    """fgfa.py:196-284: adaptive stride keeps the references of frame 0; fixed stride slides the window every `frame_stride` frames and
    takes the key frame's features out of slot num_left_ref_imgs."""

    def simple_test(self, img, img_metas, ref_img=None, ref_img_metas=None):
        m = img_metas[0]
        frame_id, stride, left = m['frame_id'], m.get('frame_stride', -1), m.get('num_left_ref_imgs', -1)
        key = _ids(img)[0]
        assert key == frame_id
        if stride < 1:
            if frame_id == 0 or self.memo is None:
                self.memo = dict(ids=_ids(ref_img[0]))
            else:
                assert ref_img is None
            x = ('own', key)
        else:
            if frame_id == 0 or self.memo is None:
                self.memo = dict(ids=_ids(ref_img[0]))
                x = ('memo', self.memo['ids'][left])
            elif frame_id % stride == 0:
                new = _ids(ref_img[0])
                assert len(new) == 1
                self.memo['ids'] = (self.memo['ids'] + new)[1:]
                x = ('memo', self.memo['ids'][left])
            else:
                assert ref_img is None
                x = ('own', key)
        return dict(bbox_results=(key, x, tuple(self.memo['ids'])))


class TraceDFF(_Trace):
    """dff.py:184-217: key frames every key_frame_interval frames cache image + features, the others warp them."""
    test_cfg = dict(key_frame_interval=4)

    def simple_test(self, img, img_metas, ref_img=None, ref_img_metas=None):
        frame_id = img_metas[0]['frame_id']
        assert ref_img is None
        if frame_id % self.test_cfg['key_frame_interval'] == 0 or self.memo is None:
            self.memo = dict(key=_ids(img)[0])
            return dict(bbox_results=(frame_id, 'key', self.memo['key']))
        return dict(bbox_results=(frame_id, 'warp', self.memo['key']))


TraceDFF.__name__ = 'DFF'


class TraceFastDVD(_Trace):
    """vid.FastDVDnetDetect with its own references every frame: window [ref0, ref1, img, ref2, ref3]."""

    def simple_test(self, img, img_metas, ref_img=None, ref_img_metas=None):
        r = _ids(ref_img[0])
        assert len(r) == 4
        return dict(bbox_results=(img_metas[0]['frame_id'], tuple(r[:2] + _ids(img) + r[2:])))


class TraceRing(_Trace):
    """vid.LLVOD: ring of the last memo_len frames, slot = push order; the result lists (slot, frame) pairs in slot order, the order
    in which llv_embed_aggregate_ring sums."""
    memo_len = 5

    def simple_test(self, img, img_metas, ref_img=None, ref_img_metas=None):
        frame_id = img_metas[0]['frame_id']
        if self.memo is None or frame_id == 0:
            self.memo = dict(next=0, valid=0)
        self.memo.setdefault('slots', {})
        for f in (_ids(ref_img[0]) if ref_img is not None else []) + _ids(img):
            self.memo['slots'][self.memo['next']] = f
            self.memo['next'] = (self.memo['next'] + 1) % self.memo_len
            self.memo['valid'] = min(self.memo['valid'] + 1, self.memo_len)
        return dict(bbox_results=(frame_id, self.memo['valid'], tuple(sorted(self.memo['slots'].items()))))


TraceRing.__name__ = 'LLVOD'

CASES = {
    'fgfa_fix1': (TraceFGFA, dict(num_ref_imgs=8, frame_range=[-3, 5], stride=1, method='test_with_fix_stride')),
    'fgfa_fix3': (TraceFGFA, dict(num_ref_imgs=4, frame_range=[-2, 2], stride=3, method='test_with_fix_stride')),
    'fgfa_adaptive': (TraceFGFA, dict(num_ref_imgs=6, frame_range=[-3, 3], method='test_with_adaptive_stride')),
    'dff': (TraceDFF, None),
    'fastdvd': (TraceFastDVD, dict(num_ref_imgs=4, frame_range=[-2, 2], method='sliding')),
    'llvod': (TraceRing, None),
}


def _run(case, T, rank, world):
    sys.path.insert(0, ROOT)
    from lowlightenvironmentvideoobjectdetection_amd import apis
    cls, sampler = CASES[case]
    return apis.sharded_clip_test(cls(), _frames(T), dict(img_shape=(2, 2, 3)), sampler, rank=rank, world=world)


# ------------------------------------------------------------------------------------------------ pure host logic
def test_shard_clip_partitions_the_clip():
    from lowlightenvironmentvideoobjectdetection_amd.dist_utils import shard_clip
    for T in (0, 1, 7, 32, 33, 256, 1001):
        for world in (1, 2, 3, 8):
            for align in (1, 3, 10):
                segs = [shard_clip(T, r, world, align) for r in range(world)]
                assert segs[0][0] == 0 and segs[-1][1] == T
                for (a0, a1), (b0, b1) in zip(segs, segs[1:]):
                    assert a1 == b0 and a0 <= a1
                assert all(s % align == 0 for s, _ in segs if s < T)
                blocks = [(b - a + align - 1) // align for a, b in segs]          # whole blocks of `align` frames, as even as array_split
                assert max(blocks) - min(blocks) <= 1


def test_clip_window_equals_the_reference_samplers_sliding_memo():
    """Replay of CocoVideoDataset.ref_img_sampling(method='test_with_fix_stride') + FGFA's memo update (coco_video_dataset.py:204-215,
    fgfa.py:234-257): the window the one-rank run holds at frame t is clip_window(t)."""
    from lowlightenvironmentvideoobjectdetection_amd.apis import clip_window
    for T, left, right, stride in ((40, 3, 5, 1), (17, 14, 15, 1), (50, 2, 2, 3), (9, 4, 4, 2), (1, 2, 2, 1)):
        memo = None
        for t in range(T):
            if t == 0:
                memo = [0] * (left + 1) + [min(round(i * stride), T - 1) for i in range(1, right + 1)]
            elif t % stride == 0:
                memo = (memo + [min(round(t + right * stride), T - 1)])[1:]
            assert memo == clip_window(t, T, left, right, stride), (T, left, right, stride, t)
            if t % stride == 0:
                assert memo[left] == t


def test_feed_items_look_like_the_reference_pipelines():
    from lowlightenvironmentvideoobjectdetection_amd.apis import ClipTestFeed
    feed = ClipTestFeed(_frames(12), dict(img_shape=(2, 2, 3)), dict(num_ref_imgs=4, frame_range=[-2, 2], stride=1, method='test_with_fix_stride'))
    items = list(feed)
    t, d = items[0]
    assert t == 0 and d['img'][0].shape == (1, 3, 2, 2) and d['ref_img'][0].shape == (1, 5, 3, 2, 2)
    assert d['img_metas'][0][0]['frame_id'] == 0 and d['img_metas'][0][0]['num_left_ref_imgs'] == 2 and d['img_metas'][0][0]['frame_stride'] == 1
    assert _ids(d['ref_img'][0]) == [0, 0, 0, 1, 2]
    assert _ids(items[5][1]['ref_img'][0]) == [7] and _ids(items[11][1]['ref_img'][0]) == [11]
    seg = ClipTestFeed(_frames(12), dict(), dict(num_ref_imgs=4, frame_range=[-2, 2], stride=1, method='test_with_fix_stride'), start=6, stop=9)
    assert seg.halo() == (4, 11) and _ids(next(iter(seg))[1]['ref_img'][0]) == [4, 5, 6, 7, 8]
    ad = ClipTestFeed(_frames(10), dict(), dict(num_ref_imgs=4, frame_range=[-2, 2], method='test_with_adaptive_stride'), start=5)
    first = next(iter(ad))[1]
    assert _ids(first['ref_img'][0]) == [0, 3, 6, 9] and 'frame_stride' not in first['img_metas'][0][0]
    with pytest.raises(Exception):
        ClipTestFeed(_frames(12), dict(), dict(frame_range=[-2, 2], stride=3, method='test_with_fix_stride'), start=4)
    with pytest.raises(Exception):
        ClipTestFeed(_frames(12), dict(), dict(method='uniform'))


@pytest.mark.parametrize('case', sorted(CASES))
@pytest.mark.parametrize('T', [37, 64])
def test_sequential_ranks_reproduce_the_one_rank_trace(case, T):
    """The same comparison without process groups (every rank's segment run in this process), for many world sizes."""
    sys.path.insert(0, ROOT)
    from lowlightenvironmentvideoobjectdetection_amd import apis, dist_utils
    cls, sampler = CASES[case]
    want = _run(case, T, 0, 1)['bbox_results']
    assert len(want) == T
    for world in (2, 3, 5, 8, 16):
        got = []
        for rank in range(world):
            model = cls()
            start, stop = dist_utils.shard_clip(T, rank, world, apis.shard_alignment(model, sampler))
            feed = apis.make_feed(model, _frames(T), dict(), sampler, start, stop)
            got += apis.single_gpu_test(model, feed)['bbox_results'] if stop > start else []
        assert got == want, (case, T, world)


# ------------------------------------------------------------------------------------------------ real process groups (gloo)
def _worker(rank, world, port, T, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import test_clip_shard as me
    from lowlightenvironmentvideoobjectdetection_amd import apis
    out = {}
    for case in sorted(me.CASES):
        cls, sampler = me.CASES[case]
        res = apis.sharded_clip_test(cls(), me._frames(T), dict(img_shape=(2, 2, 3)), sampler)      # rank / world from the process group
        out[case] = None if res is None else res['bbox_results']
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_frame_sharded_clip_over_gloo_equals_one_rank(world):
    T = 61
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1500) + 17 * world
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for case in sorted(CASES):
        want = _run(case, T, 0, 1)['bbox_results']
        assert res[0][case] == want, case                      # rank 0 holds the whole clip's results in frame order
        assert all(res[r][case] is None for r in range(1, world))
