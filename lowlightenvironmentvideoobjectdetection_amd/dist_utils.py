"""Multi-GPU plumbing: one process per GPU, RCCL through torch.distributed ('nccl' backend name on ROCm).

Inference shards with no data-path collective, two ways: a dataset of several videos by WHOLE video (shard_videos: per-video state
lives on one device, as the reference's DistributedVideoSampler does), and a single long clip BY FRAME into contiguous segments with
a halo that rebuilds the state of the one-rank run at each boundary (shard_clip: FastDVDnet 2 + 2 frames, DFF restart on a key frame,
FGFA / LLVOD window warm-up). The only exchange is the final gather of results to rank 0 (reference: pickle files + barrier,
mmtracking/mmtrack/apis/test.py:125-173).
Training averages gradients (mmdetection/mmdet/core/utils/dist_utils.py:10-51)."""
from collections import OrderedDict

import torch
import torch.distributed as dist


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_videos(video_lengths, rank, world):
    """Frame indices of rank's share: contiguous chunks of WHOLE videos, as DistributedVideoSampler does
    (mmtracking/mmtrack/datasets/samplers/distributed_video_sampler.py:24-45: np.array_split over videos)."""
    n = len(video_lengths)
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    stop = start + base + (1 if rank < extra else 0)
    first = sum(video_lengths[:start])
    count = sum(video_lengths[start:stop])
    return list(range(first, first + count))


def shard_clip(num_frames, rank, world, align=1):
    """Frames [start, stop) of ONE clip that `rank` owns: `world` contiguous segments whose boundaries are multiples of `align`
    (the key-frame interval / sampling stride, so that a segment starts where the one-rank run itself rebuilds its state), sizes as
    even as np.array_split makes them (the rule DistributedVideoSampler applies to videos, distributed_video_sampler.py:35-41,
    applied to blocks of `align` frames). A rank may get an empty segment when the clip has fewer blocks than ranks."""
    assert num_frames >= 0 and world >= 1 and 0 <= rank < world and align >= 1
    blocks = (num_frames + align - 1) // align
    base, extra = divmod(blocks, world)
    b0 = rank * base + min(rank, extra)
    b1 = b0 + base + (1 if rank < extra else 0)
    return min(b0 * align, num_frames), min(b1 * align, num_frames)


def collect_result_dicts(result_part):
    """multi_gpu_test's collect step for the dict-of-lists the video detectors return (apis/test.py:100-122,125-173): the parts are
    concatenated key by key in rank order (= frame order for contiguous segments). One all_gather_object; rank 0 gets the dict."""
    rank, world = get_dist_info()
    if world == 1:
        return {k: list(v) for k, v in result_part.items()}
    parts = [None] * world
    dist.all_gather_object(parts, {k: list(v) for k, v in result_part.items()})
    if rank != 0:
        return None
    out = {}
    for p in parts:
        for k, v in p.items():
            out.setdefault(k, []).extend(v)
    return out


def collect_results(result_part, size):
    """multi_gpu_test's collect step (mmtracking/mmtrack/apis/test.py:125-173) as one all_gather_object."""
    rank, world = get_dist_info()
    if world == 1:
        return result_part[:size]
    parts = [None] * world
    dist.all_gather_object(parts, result_part)
    if rank != 0:
        return None
    out = []
    for p in parts:
        out.extend(p)
    return out[:size]


def allreduce_grads(params, coalesce=True, bucket_size_mb=-1):
    """mmdetection/mmdet/core/utils/dist_utils.py:32-51: average gradients over ranks (flatten -> all_reduce -> /world)."""
    grads = [p.grad.data for p in params if p.requires_grad and p.grad is not None]
    world = dist.get_world_size()
    if not grads:
        return
    if coalesce:
        if bucket_size_mb > 0:
            limit = bucket_size_mb * 1024 * 1024
            buckets, cur, sz = [], [], 0
            for g in grads:
                cur.append(g)
                sz += g.numel() * g.element_size()
                if sz >= limit:
                    buckets.append(cur)
                    cur, sz = [], 0
            if cur:
                buckets.append(cur)
        else:
            by_type = OrderedDict()
            for g in grads:
                by_type.setdefault(g.type(), []).append(g)
            buckets = list(by_type.values())
        for bucket in buckets:
            flat = torch.cat([g.reshape(-1) for g in bucket])
            dist.all_reduce(flat)
            flat.div_(world)
            off = 0
            for g in bucket:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
    else:
        for g in grads:
            dist.all_reduce(g.div_(world))


def reduce_mean(tensor):
    """dist_utils.py:63-69"""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor
